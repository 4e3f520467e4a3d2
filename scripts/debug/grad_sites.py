"""Diagnostic: for the states where the product's gradient differs from the reference's derivative program, find which of
chad's pass-through assignments drop a non-zero adjoint and which of them (switched to accumulation) explain the difference."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))
from tests import _orc, gpu_checks as gc
from tests._orc import P
import chad_instrument

scene = sys.argv[1] if len(sys.argv) > 1 else gc.TORUS
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
MAXSHOW = int(os.environ.get("SHOW", "6"))
L = gc.oracle_lib()
H = ctypes.CDLL(gc.host_pathfunc_lib())
o = _orc.Oracle(L, scene, 0, 8, 160, 120, 0, gc.pathref())
o.init(80000, N, 8)
sp = o.scene_params()
libs = {}
np.set_printoptions(precision=4, suppress=True, linewidth=220)
shown = 0
lens = np.zeros(2, np.float32)
agg = {}
for i in range(N):
    c, l, prim, vert = o.serialize_init_state(i)
    if c + l > int(os.environ.get("MAXCL", "6")):
        continue
    r = o.ref_eval(c, l, prim, vert)
    if r is None: continue
    ll, g = r
    if not np.isfinite(ll) or not np.isfinite(g).all(): continue
    if l == 0 and vert[3 + 59 * (c - 2) + 46 + 35] >= 256: continue
    g2 = np.zeros(16, np.float32); ll2 = np.zeros(1, np.float32)
    H.lmc_test_pathfunc_host(c, l, P(prim), P(sp), P(vert), P(ll2), P(g2))
    dim = 2 * (c + l - 1)
    ours = g2[:dim]
    err = np.linalg.norm(g - ours) / max(np.linalg.norm(g), 1e-2)
    if err <= float(os.environ.get("TOL", "1e-2")): continue
    if (c, l) not in libs:
        libs[(c, l)] = chad_instrument.build(c, l)
    lib, sites, name = libs[(c, l)]
    fn = getattr(lib, name)
    ns = len(sites)
    tog = (ctypes.c_int * (ns + 1)).in_dll(lib, "lmc_toggle")
    old = (ctypes.c_float * (ns + 1)).in_dll(lib, "lmc_old")
    hit = (ctypes.c_int * (ns + 1)).in_dll(lib, "lmc_hit")
    def run(S):
        for s in range(ns): tog[s] = 1 if s in S else 0; hit[s] = 0; old[s] = 0
        out = np.zeros(dim, np.float32)
        fn(P(lens), P(prim), P(sp), P(vert), P(out))
        return out
    gr = run(set())
    active = [s for s in range(ns) if hit[s] > 0 and old[s] != 0.0]
    olds = {s: old[s] for s in active}
    gt = run(set(range(ns)))
    # greedy: toggle sites that bring the instrumented program closer to ours
    S = set(); cur = gr; best = np.linalg.norm(cur - ours)
    improved = True
    while improved:
        improved = False
        for s in active:
            if s in S: continue
            t = run(S | {s}); e = np.linalg.norm(t - ours)
            if e < best * 0.7:
                S.add(s); best = e; cur = t; improved = True
    bk = [int(vert[3 + 59 * k + 48]) for k in range(max(c - 2, 0))]
    print("state %d (c=%d,l=%d) err %.3g  cam-vertex BSDF kinds %s | active dropping sites %d | explains: %s -> residual %.3g (true-grad dist %.3g)" % (
        i, c, l, err, bk, len(active), sorted(S), best / max(np.linalg.norm(g), 1e-2), np.linalg.norm(gt - ours) / max(np.linalg.norm(g), 1e-2)))
    for s in S:
        agg.setdefault((c, l, s), 0); agg[(c, l, s)] += 1
    if shown < MAXSHOW:
        shown += 1
        print("   ref ", g); print("   ours", ours); print("   true", gt)
        for s in sorted(S):
            d = sites[s]
            print("   site %d line %d: _acc%d = _acc%d  dropped %.4g; _t%d defs: %s" % (s, d["line"], d["x"], d["out"], olds[s], d["x"], d["defs"][:3]))
print(sorted(agg.items(), key=lambda kv: -kv[1]))
