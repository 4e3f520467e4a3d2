#!/usr/bin/env python3
"""Design data for the existence test in front of the global-cache query (dsmall.h PrepareGaussianLean, dchain.h DCacheDim): on the resident chains
of the headline workload, per cache dimension,
  * how many queries reach the candidate scan of the dilated 4-D grid (cells whose 81-neighbourhood holds a cache row), how long the scan is per
    query and per WAVE (64 consecutive chains of one technique: a wave scans as long as its longest lane),
  * what further dilated occupancy bitmaps over OTHER coordinate groups of the same point would let through (a cache row within the query radius is
    within the radius in every coordinate, hence in the 3^m neighbourhood of the query's cell in every projection),
  * how many queries have a row within the radius at all.
usage (GPU box): python scripts/query_filter_study.py [log2 chains] [steps]  -> one JSON line per dimension"""
import ctypes
import importlib
import itertools
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
p = importlib.import_module("langevin-mcmc_amd")
from tests import gpu_checks as gc  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 18
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 48
N = 1 << lg
ren = p.Renderer(gc.TORUS, force_diffuse=1, max_depth=6, seed_offset=0, use_gradient=1)
ren.init_chains(8 * N, N, 65536, 256, 0)
ren.step(steps)
st = ren.stats()
summ = ren.summary(0)
lib = p.lib()
lib.lmc_cache_rows.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
P = p.P


def grid_g(dim):  # dchain.h CacheGridG
    g = int(np.float32(1.0) / (np.sqrt(np.float32(dim)) * np.float32(0.01)))
    return max(1, min(64, g))


def cells(x, g):
    return np.clip((x * np.float32(g)).astype(np.int64), 0, g - 1)


def dilated(rows, coords, g):
    """count of rows in the 3^m neighbourhood of every cell of the projection on `coords` (no wrap: the query distance does not wrap either)"""
    m = len(coords)
    own = np.zeros((g,) * m, np.int32)
    np.add.at(own, tuple(cells(rows[:, c], g) for c in coords), 1)
    out = np.zeros_like(own)
    for off in itertools.product((-1, 0, 1), repeat=m):
        src = tuple(slice(max(0, -o), g - max(0, o)) for o in off)
        dst = tuple(slice(max(0, o), g - max(0, -o)) for o in off)
        out[dst] += own[src]
    return out


valid = summ[:, 0] > 0
cdep, ldep = summ[:, 1].astype(int), summ[:, 2].astype(int)
dims = 2 * np.maximum(cdep + ldep - 1, 2)
print(json.dumps({"chains": N, "steps": steps, "cacheReadyMask": st["cacheReadyMask"], "valid": int(valid.sum())}))
for dim in (6, 8, 10, 12):
    rows = np.zeros((3000, dim), np.float32)
    w = np.zeros(3000, np.float32)
    n = lib.lmc_cache_rows(ren.h, dim, P(rows), P(w), None)
    if n < 3000:
        print(json.dumps({"dim": dim, "rows": int(n), "note": "cache not full"}))
        continue
    sel = valid & (dims == dim)
    # wave order: the relocated slots are grouped by technique; inside a technique the order is the chains'
    order = np.lexsort((np.arange(len(sel)), ldep, cdep))
    order = order[sel[order]]
    Q = summ[order, 16:16 + min(dim, 16)]
    g = grid_g(dim)
    radius_sq = np.float32(dim) * np.float32(0.01) * np.float32(0.01)
    nq = len(Q)
    out = {"dim": dim, "queries": int(nq), "gridG": g, "cells_4d": g ** 4}
    if nq == 0:
        print(json.dumps(out))
        continue
    def wave_stats(c):
        nw = len(c) // 64
        if nw == 0:
            return None
        m = c[:nw * 64].reshape(nw, 64)
        return {"lanes_scanning": round(float((m > 0).mean()), 4), "mean_scan_per_query": round(float(c.mean()), 3), "mean_wave_scan": round(float(m.max(1).mean()), 3)}

    # which four coordinates should the grid use?  The test is exact for ANY choice (a row within the radius is within a cell of the query in every
    # coordinate); the choice decides how many candidates a query scans.  Heuristic available at build time: the coordinates in which the cache rows
    # themselves collide least (sum of squared 1-D cell occupancies).
    coll = [float((np.bincount(cells(rows[:, c], g), minlength=g).astype(np.float64) ** 2).sum()) / 3000.0 ** 2 for c in range(dim)]
    heur = tuple(sorted(np.argsort(coll)[:4].tolist()))
    out["collision_1d_per_coordinate"] = [round(x, 4) for x in coll]
    sets = {"first4 (in force)": (0, 1, 2, 3), "last4": tuple(range(dim - 4, dim)), "least_colliding4": heur}
    rng = np.random.default_rng(1)
    allsets = list(itertools.combinations(range(dim), 4))
    for k in rng.choice(len(allsets), size=min(12, len(allsets)), replace=False):
        sets["random " + str(allsets[k])] = allsets[k]
    res = {}
    for name, co in sets.items():
        d = dilated(rows, co, g)
        c = d[tuple(cells(Q[:, cc], g) for cc in co)]
        res[name] = dict(coords=list(co), **wave_stats(c))
    out["grids"] = res
    best = min(res, key=lambda k: res[k]["mean_wave_scan"])
    out["best"] = {best: res[best]}
    # exact matches
    match = 0
    for i0 in range(0, min(nq, 200000), 4096):
        d2 = ((Q[i0:i0 + 4096, None, :] - rows[None, :, :Q.shape[1]]) ** 2).sum(-1)
        match += int((d2 < radius_sq).any(1).sum())
    out["queries_with_a_row_in_radius"] = match / float(min(nq, 200000))
    print(json.dumps(out))
ren.close()
