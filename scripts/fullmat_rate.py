#!/usr/bin/env python3
"""Full-material torus (BASELINE configs[2] materials), maxdepth 8: rate and kernel split (A/B of builds through LMC_LIB).  (GPU)"""
import importlib, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
p = importlib.import_module("langevin-mcmc_amd")
n, warm, steps = 1 << 20, 48, 32
ren = p.Renderer(os.path.join(ROOT, "scenes", "torus", "lmc.xml"), seed_offset=0, device=0, use_gradient=1, force_diffuse=0, max_depth=8)
ren.init_chains(8 * n, n, 65536, warm + steps + 8, 0, 0, n)
ren.step(warm)
ren.sync()
t0 = time.time()
ren.step(steps)
ren.sync()
dt = time.time() - t0
st = ren.stats()
ren.set_option("timing", 1)
ren.set_option("overlap", 0)
ren.step(4)
_, nl = ren.step_timing()
small_ms, large_ms, _ = ren.kernel_timing()
print(json.dumps({"lib": os.environ.get("LMC_LIB", "tree"), "chain_steps_per_s": n * steps / dt, "ms_per_step": dt * 1e3 / steps, "accept_rate": st["accepted"] / st["steps"],
                  "serial_ms": {"lean": small_ms / nl, "large_and_generic": large_ms / nl}}))
