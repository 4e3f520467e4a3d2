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
    # the grid in force: coordinates 0..3
    groups = {"0-3": (0, 1, 2, 3)}
    if dim >= 8:
        groups["4-7"] = (4, 5, 6, 7)
    if dim >= 12:
        groups["8-11"] = (8, 9, 10, 11)
    if dim == 10:
        groups["6-9"] = (6, 7, 8, 9)
    if dim == 6:
        groups["2-5"] = (2, 3, 4, 5)
    cnt = {}
    for name, co in groups.items():
        d = dilated(rows, co, g)
        cnt[name] = d[tuple(cells(Q[:, c], g) for c in co)]
        out["pass_" + name] = float((cnt[name] > 0).mean())
    base = cnt["0-3"]
    allpass = np.ones(nq, bool)
    for name in groups:
        allpass &= cnt[name] > 0
    out["pass_all_groups"] = float(allpass.mean())
    # exact matches (first 16 coordinates are all the summary carries: exact for dim <= 16)
    match = np.zeros(nq, bool)
    idx = np.nonzero(allpass)[0]
    for i0 in range(0, len(idx), 4096):
        ii = idx[i0:i0 + 4096]
        d2 = ((Q[ii, None, :] - rows[None, :, :Q.shape[1]]) ** 2).sum(-1)
        match[ii] = (d2 < radius_sq).any(1)
    out["queries_with_a_row_in_radius"] = float(match.mean())

    def wave_stats(c):
        nw = len(c) // 64
        if nw == 0:
            return None
        m = c[:nw * 64].reshape(nw, 64)
        return {"mean_scan_per_query": float(c.mean()), "mean_wave_scan(max over lanes)": float(m.max(1).mean()), "waves_with_no_scan": float((m.max(1) == 0).mean()),
                "lanes_scanning": float((m > 0).mean())}

    out["scan_today"] = wave_stats(base)
    out["scan_with_all_bitmaps"] = wave_stats(np.where(allpass, base, 0))
    # a finer split of the scan: the candidate list of the cell restricted to rows that also pass the other groups cannot be had without per-row
    # work; what CAN be had cheaply is the shortest of the groups' lists
    shortest = np.where(allpass, np.min(np.stack([cnt[k] for k in groups]), 0), 0)
    out["scan_shortest_group_list"] = wave_stats(shortest)
    out["bitmap_bytes_per_group"] = g ** 4 // 8
    print(json.dumps(out))
ren.close()
