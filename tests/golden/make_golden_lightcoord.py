"""Generates tests/golden/derv_vectors_lightcoord.npz: states of the veach-door scene whose camera path ends on the area light, with
`uselightcoordinatesampling` on (scene[0] = 1): inputs as the oracle serialises them and the outputs of the reference's generated
forward / MALA-gradient / H2MC programs (oracle/_ref), which carry the doLightCoordinateSampling branch (path.cpp:2979-3025).
Run in the build container only; the output is data."""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests import _orc  # noqa: E402
from tests import gpu_checks as gc  # noqa: E402
from tests._orc import P  # noqa: E402


def main():
    L = gc.oracle_lib()
    ref = ctypes.CDLL(gc.pathref())
    lens = np.zeros(2, np.float32)
    door = os.path.join(ROOT, "scenes", "veachdoor", "lmc.xml")
    o = _orc.Oracle(L, door, 0, 8, 160, 90, 0, gc.pathref())
    assert L.orc_set_option(o.h, b"uselightcoordinatesampling", 1.0) == 0
    n = 1 << 17
    o.init(8 * n, n, 64)
    sp = o.scene_params()
    s = o.summary(1)
    rec = {k: [] for k in ("c", "l", "primary", "vert", "loglum", "mala_grad", "h2_grad", "h2_hess", "scalar_ss")}
    for i in range(n):
        c, l, prim, vert = o.serialize_init_state(i)
        if l != 0 or c < 4 or vert[3 + 59 * (c - 2) + 46] != 1.0:
            continue
        ev = o.ref_eval(c, l, prim, vert)
        if ev is None or not np.isfinite(ev[0]) or not np.isfinite(ev[1]).all():
            continue
        g1, h1 = np.zeros(16, np.float32), np.zeros(256, np.float32)
        getattr(ref, "evaluate_path_bidir_%d_%d_static_derv" % (c, l))(P(lens), P(prim), P(sp), P(vert), P(g1), P(h1))
        mg = np.zeros(16, np.float32)
        mg[: len(ev[1])] = ev[1]
        rec["c"].append(c), rec["l"].append(l), rec["primary"].append(prim.copy()), rec["vert"].append(vert[:600].copy()), rec["loglum"].append(ev[0])
        rec["mala_grad"].append(mg), rec["h2_grad"].append(g1), rec["h2_hess"].append(h1), rec["scalar_ss"].append(s[i, 4])
    o.close()
    np.savez_compressed(os.path.join(HERE, "derv_vectors_lightcoord.npz"), scene=sp, **{k: np.array(v) for k, v in rec.items()})
    print("wrote derv_vectors_lightcoord.npz:", len(rec["c"]), "vectors", sorted(set(zip(rec["c"], rec["l"]))))


if __name__ == "__main__":
    main()
