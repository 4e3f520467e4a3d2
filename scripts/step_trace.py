#!/usr/bin/env python3
"""Per-step wall and kernel times of a fresh chain population (the driver's bench window is steps 5..25): where the start-up
goes (cache fill with the generic gradient kernel, kd-tree / grid builds on the host, then the lean kernel).
usage: python scripts/step_trace.py [steps] > out.jsonl   (GPU)"""
import importlib, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_checks as gc

p = importlib.import_module("langevin-mcmc_amd")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
chains = 1 << 20
ren = p.Renderer(gc.TORUS, force_diffuse=1, max_depth=6, seed_offset=0, device=0, use_gradient=1)
ren.init_chains(8 * chains, chains, 65536, 256, 0, 0, chains)
ren.set_option("timing", 1)
ren.sync()
prev = ren.stats()
for s in range(steps):
    t0 = time.time()
    ren.step(1)
    ren.sync()
    wall = (time.time() - t0) * 1e3
    kernel_ms, launches = ren.step_timing()
    small_ms, large_ms, lean = ren.kernel_timing()
    st = ren.stats()
    d = {k: st[k] - prev[k] for k in ("steps", "largeSteps", "gradCalls", "cacheQueries", "cacheHits") if k in st}
    prev = st
    print(json.dumps({"step": s, "wall_ms": round(wall, 3), "kernels_ms": round(kernel_ms, 3), "lean_ms": round(small_ms, 3), "large_and_generic_ms": round(large_ms, 3), **d}))
