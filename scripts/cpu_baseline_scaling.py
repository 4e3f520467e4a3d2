#!/usr/bin/env python3
"""Thread scaling of the CPU baseline (the oracle under the reference's scheduling, orc_run_async) on this host: chain-steps/s and per-thread
rate for a few thread counts, both builds (parity: -O2 no contraction; fast: -O3 -ffast-math x86-64-v3).  VERDICT r3 item 7: why 15 k steps/s
per thread on the GPU box's 256 hardware threads against 135 k per core in the reference's own figures.  usage: cpu_baseline_scaling.py [seconds=3]"""
import json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import _orc, gpu_checks as gc

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
cores = os.cpu_count() or 1
for name, so in (("parity", gc.ORACLE_SO), ("fast", os.path.join(ROOT, "oracle", "liblmc_oracle_fast.so"))):
    if not os.path.exists(so):
        continue
    L = _orc.load(so)
    for threads in sorted({1, 8, 32, min(64, cores), min(128, cores), cores}):
        if threads > cores:
            continue
        for chains_per_thread in (1, 4):
            orc = _orc.Oracle(L, gc.TORUS, 1, 6, 0, 0, 0, gc.pathref())
            n = threads * chains_per_thread
            orc.init(300000, n, min(cores, 64))
            orc.setup_chains(1 << 30, 0)
            rate, done = orc.run_async(threads, secs)
            orc.close()
            print(json.dumps({"build": name, "threads": threads, "chains": n, "steps_per_s": rate, "per_thread": rate / threads, "host_threads": cores}), flush=True)
