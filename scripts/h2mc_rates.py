#!/usr/bin/env python3
"""H2MC chain-steps/s on the two shipped scenes (A/B of builds through LMC_LIB).  (GPU)"""
import importlib, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
p = importlib.import_module("langevin-mcmc_amd")
S = os.path.join(ROOT, "scenes")
for name, xml, kw in (("torus H2MC", os.path.join(S, "torus", "h2mc.xml"), dict(force_diffuse=0, max_depth=8)),
                      ("veach-door H2MC", os.path.join(S, "veachdoor", "h2mc.xml"), dict(force_diffuse=0))):
    n, warm, steps = 1 << 18, 6, 8
    ren = p.Renderer(xml, seed_offset=0, device=0, use_gradient=1, **kw)
    ren.init_chains(8 * n, n, 65536, warm + steps + 8, 0, 0, n)
    ren.step(warm)
    ren.sync()
    t0 = time.time()
    ren.step(steps)
    ren.sync()
    dt = time.time() - t0
    st = ren.stats()
    print(json.dumps({"lib": os.environ.get("LMC_LIB", "tree"), "config": name, "chain_steps_per_s": n * steps / dt, "ms_per_step": dt * 1e3 / steps,
                      "accept_rate": st["accepted"] / max(st["steps"], 1)}), flush=True)
    ren.close()
