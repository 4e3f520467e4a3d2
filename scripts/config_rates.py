#!/usr/bin/env python3
"""Chain-steps/s of the other BASELINE.json configurations on one GPU (not bench lines: parity-test cases timed for the record):
full-material torus at maxdepth 8 and 12 (cfg 3), veach-door LMC (cfg 4) and H2MC (cfg 5), torus H2MC.  (GPU)"""
import importlib, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
p = importlib.import_module("langevin-mcmc_amd")
S = os.path.join(ROOT, "scenes")
cases = [
    ("torus full materials, maxdepth 8, LMC", os.path.join(S, "torus", "lmc.xml"), dict(force_diffuse=0, max_depth=8), 1 << 20, 48, 40),
    ("torus full materials, maxdepth 12, LMC (cfg 3)", os.path.join(S, "torus", "lmc.xml"), dict(force_diffuse=0, max_depth=12), 1 << 19, 48, 40),
    ("torus full materials, maxdepth 8, H2MC", os.path.join(S, "torus", "h2mc.xml"), dict(force_diffuse=0, max_depth=8), 1 << 18, 8, 8),
    ("veach-door, scene's maxdepth, LMC (cfg 4)", os.path.join(S, "veachdoor", "lmc.xml"), dict(force_diffuse=0), 1 << 20, 48, 40),
    ("veach-door, scene's maxdepth, H2MC (cfg 5)", os.path.join(S, "veachdoor", "h2mc.xml"), dict(force_diffuse=0), 1 << 18, 8, 8),
]
for name, xml, kw, n, warm, steps in cases:
    if not os.path.exists(xml):
        continue
    try:
        ren = p.Renderer(xml, seed_offset=0, device=0, use_gradient=1, **kw)
        ren.init_chains(8 * n, n, 65536, warm + steps + 8, 0, 0, n)
        ren.step(warm)
        ren.sync()
        t0 = time.time()
        ren.step(steps)
        ren.sync()
        dt = time.time() - t0
        st = ren.stats()
        # kernel split of a few more steps, side launches serialised (not part of the rate above)
        ren.set_option("timing", 1)
        ren.set_option("overlap", 0)
        ren.step(4)
        _, nl = ren.step_timing()
        small_ms, large_ms, _ = ren.kernel_timing()
        print(json.dumps({"config": name, "chains": n, "steps": steps, "after_warmup_steps": warm, "chain_steps_per_s": n * steps / dt, "ms_per_step": dt * 1e3 / steps,
                          "accept_rate": st["accepted"] / max(st["steps"], 1), "large_step_frac": st["largeSteps"] / max(st["steps"], 1),
                          "serial_ms": {"lean": small_ms / max(nl, 1), "large_and_generic": large_ms / max(nl, 1)}, "cache_ready_mask": st["cacheReadyMask"]}), flush=True)
        ren.close()
    except Exception as e:  # a configuration that cannot run is reported, not hidden
        print(json.dumps({"config": name, "error": str(e)}), flush=True)
