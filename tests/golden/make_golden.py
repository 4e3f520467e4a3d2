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
    # (input, output) vectors of the reference's generated forward + derivative programs (oracle/_ref/libpathref.so)
    # on paths sampled from the torus scene by the oracle
    from tests import _orc
    from tests import gpu_checks as gc

    L = gc.oracle_lib()
    o = _orc.Oracle(L, gc.TORUS, 1, 6, 160, 120, 0, gc.pathref())
    o.init(60000, 768, 8)
    rec = {"c": [], "l": [], "primary": [], "vert": [], "loglum": [], "grad": []}
    per = {}
    for i in range(768):
        c, l, prim, vert = o.serialize_init_state(i)
        if per.get((c, l), 0) >= 24:
            continue
        per[(c, l)] = per.get((c, l), 0) + 1
        ll, g = o.ref_eval(c, l, prim, vert)
        gg = np.zeros(16, np.float32)
        gg[: len(g)] = g
        rec["c"].append(c), rec["l"].append(l), rec["primary"].append(prim), rec["vert"].append(vert[:600]), rec["loglum"].append(ll), rec["grad"].append(gg)
    # the survey's hand-built known answer: (c,l) = (2,1), point light, one Lambertian triangle
    f = np.float32
    ka_primary = np.zeros(17, f)
    ka_primary[:5] = [0.3, 0.55, 0.45, 0.2, 0.7]
    ka_scene = np.zeros(38, f)
    M = np.array([[1, 0, 0, -0.5], [0, 1, 0, -0.5], [0, 0, 1, 1], [0, 0, 0, 1]], f)
    ka_scene[1:17] = M.T.reshape(-1)
    ka_scene[24:28] = [0, 0, 0, 1]
    ka_scene[28:32] = [0, 0, 0, 1]
    ka_scene[32], ka_scene[33], ka_scene[34:37], ka_scene[37] = 4096, 64, [0, 0, 5], 20000
    ka_vert = np.zeros(600, f)
    tri = np.zeros(46, f)
    tri[2:5], tri[5:8], tri[8:11] = [-10, -10, 5], [20, 0, 0], [0, 20, 0]
    tri[11:20] = [0, 0, -1] * 3
    tri[20:38] = tri[2:20]
    tri[38], tri[45] = 1, 0.005
    ka_vert[3:49] = tri
    ka_vert[49:56] = [0, 1, 2, 1, 10, 10, 10]
    ka_vert[105:109] = [0, 0.8, 0.6, 0.4]
    ka_vert[115] = 1
    np.savez_compressed(os.path.join(HERE, "grad_vectors.npz"), scene=o.scene_params(), ka_primary=ka_primary, ka_scene=ka_scene, ka_vert=ka_vert,
                        **{k: np.array(v) for k, v in rec.items()})
    print("wrote grad_vectors.npz:", len(rec["c"]), "vectors", per)


if __name__ == "__main__":
    main()
