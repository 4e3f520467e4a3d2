"""Per-step timeline of the headline workload (torus, 2^20 chains, bench.py's defaults) over the first steps after MLTInit: which
launch bounds a step while the gradient caches fill -- the window the round-end driver measures (bench.py --steps 20 --warmup 5)."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
p = importlib.import_module("langevin-mcmc_amd")
n = 1 << 20
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ren = p.Renderer(os.path.join(ROOT, "scenes", "torus", "lmc.xml"), force_diffuse=1, max_depth=6, width=1024, height=768, seed_offset=0, use_gradient=1)
ren.set_option("timing", 1)
ren.init_chains(8 * n, n, 4096, 256, 0)
ren.step_timing()
prev = ren.stats()
for s in range(nsteps):
    t0 = time.perf_counter()
    ren.step(1); ren.sync()
    wall = (time.perf_counter() - t0) * 1e3
    ms, k = ren.step_timing()
    sp = ren.kernel_timing_split()
    st = ren.stats()
    d = {k_: st[k_] - prev[k_] for k_ in ("largeSteps", "gradCalls", "cacheQueries", "accepted")}
    prev = st
    print(json.dumps({"step": s, "wall_ms": round(wall, 3), "gpu_ms": round(ms, 3), "lean_ms": round(sp["lean_ms"], 3), "large_ms": round(sp["large_ms"], 3),
                      "generic_ms": round(sp["generic_ms"], 3), "ready": st["cacheReadyMask"], **d}), flush=True)
ren.close()
