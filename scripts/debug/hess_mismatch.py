"""Diagnostic: product Hessian program (host twin) vs the reference's H2MC derivative programs, per technique."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import _orc, gpu_checks as gc
from tests._orc import P
from collections import Counter
scene = sys.argv[1] if len(sys.argv) > 1 else gc.TORUS
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
MAXCL = int(os.environ.get("MAXCL", "7"))
L = gc.oracle_lib()
ref = ctypes.CDLL(gc.pathref())
mine = ctypes.CDLL(gc.host_pathfunc_lib())
orc = _orc.Oracle(L, scene, 0, 8, 160, 120, 0, gc.pathref())
orc.init(40000, N, 8)
sp = orc.scene_params()
lens = np.zeros(2, np.float32)
tot = Counter(); ok = Counter(); worst = []
for i in range(N):
    r = orc.serialize_init_state(i)
    if r is None: continue
    c, l, prim, vert = r
    if c + l > MAXCL: continue
    name = "evaluate_path_bidir_%d_%d_static_derv" % (c, l)
    if not hasattr(ref, name): continue
    dim = 2 * max(c + l - 1, 2)
    g1, h1, g2, h2, ll = np.zeros(16, np.float32), np.zeros(256, np.float32), np.zeros(16, np.float32), np.zeros(256, np.float32), np.zeros(1, np.float32)
    getattr(ref, name)(P(lens), P(prim), P(sp), P(vert), P(g1), P(h1))
    mine.lmc_test_pathfunc_hess_host(c, l, P(prim), P(sp), P(vert), P(ll), P(g2), P(h2))
    H1, H2 = h1[: dim * dim].reshape(dim, dim), h2[: dim * dim].reshape(dim, dim)
    if not (np.isfinite(H1).all() and np.isfinite(g1).all()): continue
    tot[(c, l)] += 1
    eg = np.linalg.norm(g1[:dim] - g2[:dim]) / max(np.linalg.norm(g1[:dim]), 1e-2)
    eh = np.linalg.norm(H1 - H2) / max(np.linalg.norm(H1), 1e-1)
    if eg < 1e-2 and eh < 1e-2: ok[(c, l)] += 1
    else: worst.append((eh, eg, i, c, l))
for k in sorted(tot): print(k, ok[k], "/", tot[k])
print("total", sum(ok.values()), "/", sum(tot.values()))
worst.sort(reverse=True)
print(worst[:10])
