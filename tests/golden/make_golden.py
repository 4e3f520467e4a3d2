"""Generates the committed golden vectors from the reference's own code compiled in place
(oracle/_ref/*.so, built by `make -f oracle/Makefile.ref` from /root/reference).  Run in the build
container only; the outputs (small JSON / npz files next to this script) are data, not source."""
import ctypes
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests._orc import P  # noqa: E402


def main():
    drv = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "librefdrv.so"))
    seeds = [0, 1, 7, 127, (1 << 20) - 1]
    out = {"seeds": {}}
    for s in seeds:
        u32 = np.zeros(256, np.uint32)
        drv.ref_pcg_u32(ctypes.c_uint64(s), 256, P(u32))
        uni = np.zeros(256, np.float32)
        drv.ref_pcg_uniform(ctypes.c_uint64(s), 256, P(uni))
        nor = np.zeros(255, np.float32)
        drv.ref_pcg_normal(ctypes.c_uint64(s), 255, ctypes.c_float(0.0), ctypes.c_float(1.0), P(nor))
        out["seeds"][str(s)] = {"u32": u32.tolist(), "uniform_bits": uni.view(np.uint32).tolist(), "normal_bits": nor.view(np.uint32).tolist()}
    json.dump(out, open(os.path.join(HERE, "pcg_streams.json"), "w"))
    x = np.exp(np.linspace(np.log(1e-12), np.log(1e12), 512)).astype(np.float32)
    y = np.zeros_like(x)
    drv.ref_fastlog(len(x), P(x), P(y))
    json.dump({"x_bits": x.view(np.uint32).tolist(), "y_bits": y.view(np.uint32).tolist()}, open(os.path.join(HERE, "fastlog.json"), "w"))
    # nanoflann (reference-modified radiusSearch) on a clustered 6-D cloud
    from tests.test_oracle_pins import kd_case, kd_run

    pts, q, radius = kd_case(6)
    pts, q = pts[:3000], q[:200]
    n, idx, dist = kd_run(drv, "ref_kd_query", 6, pts, q, radius)
    np.savez_compressed(os.path.join(HERE, "kdtree_dim6.npz"), pts=pts, q=q, radius=radius, n=n, idx=idx, dist=dist)
    print("wrote pcg_streams.json, fastlog.json, kdtree_dim6.npz")


if __name__ == "__main__":
    main()
