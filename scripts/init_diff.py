"""Parity probe: MLTInit contributions of the GPU and of the oracle side by side (same streams); prints where they differ.
usage: init_diff.py SCENE.xml force_diffuse max_depth num_init init_threads"""
import ctypes, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import _orc, gpu_checks as gc
from tests._orc import P
xml, fd, md, ninit, nth = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
p = gc.pkg()
L = gc.oracle_lib()
orc = _orc.Oracle(L, xml, fd, md, 160, 90, 0, "")
ren = p.Renderer(xml, force_diffuse=fd, max_depth=md, width=160, height=90, seed_offset=0, use_gradient=0)
orc.init(ninit, 64, nth)
ren.init_chains(ninit, 64, nth, 10)
def dump(fn, h):
    cap = 4 * ninit
    s, cl, ls = np.zeros(cap, np.int64), np.zeros(cap, np.int32), np.zeros(cap, np.float32)
    fn.restype = ctypes.c_longlong
    fn.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    n = fn(h, cap, P(s), P(cl), P(ls))
    return s[:n], cl[:n], ls[:n]
so, co, lo = dump(L.orc_init_contribs, orc.h)
sg, cg, lg = dump(p.lib().lmc_init_contribs, ren.h)
print("contribs oracle", len(so), "gpu", len(sg))
# per-sample comparison
from collections import defaultdict
do, dg = defaultdict(list), defaultdict(list)
for a, b, c in zip(so, co, lo): do[int(a)].append((int(b), float(c)))
for a, b, c in zip(sg, cg, lg): dg[int(a)].append((int(b), float(c)))
bad = 0
for k in sorted(set(do) | set(dg)):
    a, b = do.get(k, []), dg.get(k, [])
    same = len(a) == len(b) and all(x[0] == y[0] and abs(x[1] - y[1]) <= 1e-5 * abs(x[1]) for x, y in zip(a, b))
    if not same:
        bad += 1
        if bad <= 12:
            print("sample", k, "oracle", [(x[0] >> 4, x[0] & 15, "%.6g" % x[1]) for x in a], "| gpu", [(x[0] >> 4, x[0] & 15, "%.6g" % x[1]) for x in b])
print("differing samples:", bad, "of", ninit)
