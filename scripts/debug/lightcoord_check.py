"""Diagnostic (CPU): uselightcoordinatesampling on the door scene -- identity log(ssScore) == the reference's forward program on the
oracle's Serialize output, and the product's path program (host twin) vs the reference's programs, on states that hit the area light."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from tests import _orc, gpu_checks as gc
from tests._orc import P
L = gc.oracle_lib()
door = os.path.join(gc.ROOT, 'scenes', 'veachdoor', 'lmc.xml')
o = _orc.Oracle(L, door, 0, 8, 160, 90, 0, gc.pathref())
assert L.orc_set_option(o.h, b"uselightcoordinatesampling", 1.0) == 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
o.init(8 * N, N, 64)
sp = o.scene_params()
assert sp[0] == 1.0
s = o.summary(1)
H = ctypes.CDLL(gc.host_pathfunc_lib())
n = bad_id = bad_ll = bad_g = 0
for i in range(N):
    c, l, prim, vert = o.serialize_init_state(i)
    if l != 0 or c < 4 or vert[3 + 59 * (c - 2) + 46] != 1.0:
        continue
    r = o.ref_eval(c, l, prim, vert)
    if r is None or not np.isfinite(r[0]):
        continue
    ll, g = r
    n += 1
    bad_id += abs(ll - np.log(s[i, 4])) > 3e-3
    ll2 = np.zeros(1, np.float32); g2 = np.zeros(16, np.float32)
    H.lmc_test_pathfunc_host(c, l, P(prim), P(sp), P(vert), P(ll2), P(g2))
    dim = 2 * (c + l - 1)
    bad_ll += abs(ll - ll2[0]) > 2e-3
    e = np.linalg.norm(g - g2[:dim]) / max(np.linalg.norm(g), 1e-2)
    bad_g += e > 1e-2
    if n <= 3 or (e > 1e-2 and bad_g <= 3):
        print(c, l, "ref ll %.5f scalar %.5f ours %.5f" % (ll, np.log(s[i, 4]), ll2[0]), "grad err %.3g" % e)
print("area-light hit states", n, "| identity failures", bad_id, "| product value failures", bad_ll, "| product gradient failures", bad_g)
