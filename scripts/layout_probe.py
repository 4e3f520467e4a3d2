#!/usr/bin/env python3
"""HBM bandwidth of the step kernels' state-streaming pattern, by layout (kernels.hip k_layout_probe).  (GPU)"""
import ctypes, importlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
p = importlib.import_module("langevin-mcmc_amd")
L = p.lib()
L.lmc_layout_probe.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_double)]
N = 1 << 20
for words in (160, 320):
    for batch in (4, 12):
        for mode in (0, 1):
            ms = ctypes.c_double()
            assert L.lmc_layout_probe(N, words, mode, batch, 10, ctypes.byref(ms)) == 0
            gb = N * words * 4 * 1.5 / 1e9
            print(json.dumps({"layout": "[word][chain]" if mode == 0 else "[tile64][word][lane]", "words_per_chain": words, "loads_in_flight": batch,
                              "ms": round(ms.value, 4), "GB_moved": round(gb, 3), "TB_per_s": round(gb / ms.value, 3)}))
