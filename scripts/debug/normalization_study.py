#!/usr/bin/env python3
"""is the constant +1.5 % of our images against the render the reference ships (scenes/torus/lmc_timeuse_44.689152s.exr)
the shipped render's OWN normaliser?  An MLT image is  histogram x normalization  (every mutation deposits `normalization` of luminance,
mlt.cpp:103-112), and the reference estimates normalization ONCE, from numinitsamples = 300 000 bidirectional samples spread over
NumSystemCores() init streams seeded RNG(threadIndex + seedOffset) (mlt.h:41-154, lmc.xml:9).  This script replays exactly that estimate on the
CPU oracle (test infrastructure) for a range of core counts and seed offsets and compares it with the estimate from 64 x as many samples.
CPU only.  usage: python scripts/debug/normalization_study.py [scene: torus|door] [big samples]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import _orc, gpu_checks as gc  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "torus"
big = int(sys.argv[2]) if len(sys.argv) > 2 else 300000 * 64
xml = gc.TORUS if scene == "torus" else os.path.join(ROOT, "scenes", "veachdoor", "lmc.xml")
L = gc.oracle_lib()


def norm(num_init, threads, seed):
    o = _orc.Oracle(L, xml, 0, 0, 64, 48, seed, "")  # the film size does not enter the normalisation: screen positions are in [0,1)^2
    n, nc = o.init(num_init, 128, threads)
    o.close()
    return n, nc


ref, nc = norm(big, 64, 0)
print(json.dumps({"scene": scene, "reference_estimate": {"samples": big, "threads": 64, "normalization": ref, "contributions": nc}}))
sys.stdout.flush()
rows = []
for threads in (8, 16, 24, 32, 40, 48, 64, 96, 128):
    n, nc = norm(300000, threads, 0)
    rows.append((threads, 0, n))
    print(json.dumps({"init_threads": threads, "seedoffset": 0, "normalization_300k": n, "ratio_to_reference": n / ref}))
    sys.stdout.flush()
r = []
for seed in range(1000, 1000 + 48 * 200, 200):  # disjoint stream sets: RNG(threadIndex + seedOffset)
    n, nc = norm(300000, 32, seed)
    r.append(n / ref)
r = np.array(r)
print(json.dumps({"spread_of_a_300k_estimate": {"estimates": len(r), "init_threads": 32, "mean_ratio": float(r.mean()), "std_of_ratio": float(r.std(ddof=1)),
                                                "min": float(r.min()), "max": float(r.max()),
                                                "fraction_of_estimates_at_or_below_0.985": float((r <= 0.985).mean())}}))
