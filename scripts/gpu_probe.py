"""Ad-hoc GPU diagnostics: prints the GPU-vs-oracle comparison figures (used while developing; the asserted
versions live in tests/test_gpu_*.py)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from tests import _orc, gpu_checks as gc

what = sys.argv[1:] or ["rng", "trace", "grad", "chain"]
L = gc.oracle_lib()
p = gc.pkg()
t0 = time.time()
if "rng" in what:
    print("RNG", gc.check_rng(L), flush=True)
if "trace" in what or "grad" in what:
    orc = _orc.Oracle(L, gc.TORUS, 1, 6, 160, 120, 0, gc.pathref())
    ren = p.Renderer(gc.TORUS, force_diffuse=1, max_depth=6, width=160, height=120, seed_offset=0)
    print("scene", ren.num_tris, "tris", ren.num_nodes, "nodes depth", ren.bvh_depth, flush=True)
    assert np.array_equal(orc.scene_params(), ren.scene_params())
if "trace" in what:
    t = time.time()
    print("TRACE", gc.check_trace(L, orc, ren, n=100000), time.time() - t, flush=True)
if "grad" in what and gc.pathref():
    orc.init(40000, 1024, 8)
    inp = gc.collect_grad_inputs(orc, 1024)
    for k, v in gc.check_grad(orc, inp, ren.scene_params()).items():
        print("GRAD", k, v, flush=True)
if "chain" in what:
    for ug in ([0, 1] if gc.pathref() else [0]):
        t = time.time()
        r = gc.run_pair(160, 120, 40000, 256, 8, 400, 40, use_gradient=ug)
        print("CHAIN use_gradient=%d" % ug, json.dumps(r, default=float), "sec", time.time() - t, flush=True)
if "bench" in what:
    ren = p.Renderer(gc.TORUS, force_diffuse=1, max_depth=6, seed_offset=0, use_gradient=1)
    n = 1 << 18
    t = time.time()
    print("init", ren.init_chains(1 << 21, n, 16384, 1 << 20), time.time() - t, flush=True)
    for it in range(4):
        t = time.time()
        ren.step(8)
        ren.sync()
        dt = time.time() - t
        ms, nl = ren.step_timing()
        print("steps/s", 8 * n / dt, "kernel ms/launch", ms / max(nl, 1), ren.stats(), flush=True)
print("total sec", time.time() - t0)
