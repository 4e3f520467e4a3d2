#!/usr/bin/env python3
"""One of the bench's workloads by itself, for profiling under rocprofv3 (kernel trace / PMC passes): warm-up, then N steps.
usage: run_one_config.py torus6|torus12|door|door_h2mc [steps] [log2 chains]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
p = importlib.import_module("langevin-mcmc_amd")
S = os.path.join(ROOT, "scenes")
which = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
cfg = {"torus6": (os.path.join(S, "torus", "lmc.xml"), dict(force_diffuse=1, max_depth=6), 20, 40),
       "torus12": (os.path.join(S, "torus", "lmc.xml"), dict(force_diffuse=0, max_depth=12), 20, 40),
       "door": (os.path.join(S, "veachdoor", "lmc.xml"), {}, 20, 40),
       "door_h2mc": (os.path.join(S, "veachdoor", "h2mc.xml"), {}, 18, 6)}[which]
n = 1 << (int(sys.argv[3]) if len(sys.argv) > 3 else cfg[2])
ren = p.Renderer(cfg[0], seed_offset=0, use_gradient=1, **cfg[1])
ren.init_chains(8 * n, n, 65536, 256, 0)
ren.step(cfg[3])
ren.sync()
t0 = time.time()
ren.step(steps)
ren.sync()
dt = time.time() - t0
st = ren.stats()
print(json.dumps({"config": which, "chains": n, "steps": steps, "chain_steps_per_s": n * steps / dt, "ms_per_step": dt * 1e3 / steps, "large_step_frac": st["largeSteps"] / st["steps"]}))
ren.close()
