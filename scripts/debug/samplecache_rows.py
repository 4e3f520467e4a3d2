"""GPU box: cache rows (dim 6) of oracle and device after the cache became ready, samplecache on."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import gpu_checks as gc, _orc
from tests._orc import P
np.set_printoptions(precision=6, linewidth=220)
p = gc.pkg(); L = gc.oracle_lib()
opts = {"largestepprob": 0.3, "largestepscale": 1.0, "largestepmultiplexed": 1, "samplecache": 1}
orc = _orc.Oracle(L, gc.TORUS, 1, 6, 128, 96, 0, gc.pathref())
ren = p.Renderer(gc.TORUS, force_diffuse=1, max_depth=6, width=128, height=96, seed_offset=0, use_gradient=1)
for k, v in opts.items():
    L.orc_set_option(orc.h, k.encode(), float(v)); ren.set_option(k, v)
orc.init(200000, 16384, 64); ren.init_chains(200000, 16384, 64, 120)
orc.setup_chains(120, 0)
orc.step(30); ren.step(30)
print(orc.stats()["cacheReadyMask"], ren.stats()["cacheReadyMask"])
dim = 6
po, wo, io = np.zeros((3000, dim), np.float32), np.zeros(3000, np.float32), np.zeros((3000, 8), np.float32)
no = L.orc_cache_rows(orc.h, dim, P(po), P(wo), P(io))
pg, wg, xg = np.zeros((3000, dim), np.float32), np.zeros(3000, np.float32), np.zeros((3000, 313), np.float32)
lib = p.lib(); lib.lmc_cache_rows.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
ng = lib.lmc_cache_rows(ren.h, dim, P(pg), P(wg), P(xg))
print("rows", no, ng)
print("pss equal rows", (np.abs(po - pg).max(axis=1) < 1e-6).sum(), "weight equal", (np.abs(wo - wg) <= 1e-6 * np.abs(wo)).sum())
xi = xg.view(np.int32)
cg, lg = xi[:, 304], xi[:, 305]
print("cl equal", ((cg == io[:, 0]) & (lg == io[:, 1])).sum(), "ls equal", (np.abs(xg[:, 304 + 7] - io[:, 2]) <= 1e-5 * np.abs(io[:, 2])).sum())
print("time equal", (np.abs(xg[:, 0] - io[:, 4]) < 1e-6).sum(), "screen equal", ((np.abs(xg[:, 1] - io[:, 5]) < 1e-6) & (np.abs(xg[:, 2] - io[:, 6]) < 1e-6)).sum())
print("camCount", np.bincount(xi[:, 12].clip(0, 20))[:8], "oracle cam verts", np.bincount(io[:, 7].astype(int))[:8])
print("path camDepth/lgtDepth vs contrib", ((xi[:, 10] == cg) & (xi[:, 11] == lg)).sum())
bad = np.nonzero(np.abs(xg[:, 1] - io[:, 5]) >= 1e-6)[0][:5]
for r in bad: print(r, "oracle", io[r], "gpu head", xg[r, :3], xi[r, 7:14], "contrib", cg[r], lg[r], xg[r, 311:313], "pss", po[r], pg[r])
d = np.abs(po - pg).max(axis=1)
badrows = np.nonzero(d >= 1e-6)[0]
print("in-place differing rows", len(badrows), "first", badrows[:20])
import collections
so_ = {tuple(np.round(r, 5)) for r in po}; sg_ = {tuple(np.round(r, 5)) for r in pg}
print("set overlap (5 decimals)", len(so_ & sg_), len(so_), len(sg_))
# nearest gpu row for a few differing oracle rows
for r in badrows[5:15]:
    dist = np.abs(pg - po[r]).max(axis=1); q = int(dist.argmin())
    print(r, "nearest gpu row", q, "dist", dist[q], "w", wo[r], wg[q], "cl", io[r, :2], cg[q], lg[q])
