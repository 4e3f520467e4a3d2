"""Diagnostic: which full-material states disagree between the product's path program (host twin) and the reference's
derivative programs; prints technique, per-vertex BSDF kinds / useAbs flags and the per-component error."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import _orc, gpu_checks as gc
from tests._orc import P

scene = sys.argv[1] if len(sys.argv) > 1 else gc.TORUS
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
L = gc.oracle_lib()
H = ctypes.CDLL(gc.host_pathfunc_lib())
o = _orc.Oracle(L, scene, 0, 8, 160, 120, 0, gc.pathref())
o.init(80000, N, 8)
sp = o.scene_params()
n = ok = 0
bad = []
for i in range(N):
    c, l, prim, vert = o.serialize_init_state(i)
    r = o.ref_eval(c, l, prim, vert)
    if r is None:
        continue
    ll, g = r
    if not np.isfinite(ll) or not np.isfinite(g).all():
        continue
    if l == 0 and vert[3 + 59 * (c - 2) + 46 + 35] >= 256:
        continue
    ll2 = np.zeros(1, np.float32); g2 = np.zeros(16, np.float32)
    H.lmc_test_pathfunc_host(c, l, P(prim), P(sp), P(vert), P(ll2), P(g2))
    dim = 2 * (c + l - 1)
    n += 1
    err = np.linalg.norm(g - g2[:dim]) / max(np.linalg.norm(g), 1e-2)
    if err <= 1e-2:
        ok += 1
    else:
        bad.append((i, c, l, err, g.copy(), g2[:dim].copy()))
print("states", n, "ok", ok, "bad", len(bad))
from collections import Counter
print(Counter((b[1], b[2]) for b in bad))
np.set_printoptions(precision=4, suppress=True, linewidth=200)
for b in bad[:int(os.environ.get("SHOW", "12"))]:
    print("state %d (c=%d,l=%d) err %.3g" % b[:4]); print("  ref ", b[4]); print("  ours", b[5])
