"""Generates tests/golden/derv_vectors_full.npz: (input, output) vectors of the reference's generated MALA-gradient and
H2MC gradient + Hessian programs (all 42 (c,l) pairs are built into oracle/_ref/libpathref.so from /root/reference by
oracle/Makefile.ref) on full-material states of BOTH shipped scenes, including the 14- and 16-dimensional states
(c + l = 8, 9) that no test rebuilds the 76 k-line programs for.  Run in the build container only; the output is data."""
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

PER_PAIR = 6


def main():
    L = gc.oracle_lib()
    ref = ctypes.CDLL(gc.pathref())
    lens = np.zeros(2, np.float32)
    rec = {k: [] for k in ("scene_id", "c", "l", "primary", "vert", "loglum", "mala_grad", "h2_grad", "h2_hess")}
    scenes = []
    for sid, xml in enumerate((gc.TORUS, os.path.join(ROOT, "scenes", "veachdoor", "lmc.xml"))):
        o = _orc.Oracle(L, xml, 0, 8, 160, 120, 0, gc.pathref())
        o.init(80000, 4096, 8)
        scenes.append(o.scene_params())
        per = {}
        for i in range(4096):
            r = o.serialize_init_state(i)
            if r is None:
                continue
            c, l, prim, vert = r
            if per.get((c, l), 0) >= PER_PAIR:
                continue
            ev = o.ref_eval(c, l, prim, vert)
            if ev is None or not np.isfinite(ev[0]) or not np.isfinite(ev[1]).all():
                continue
            if l == 0 and vert[3 + 59 * (c - 2) + 46 + 35] >= 256:  # wrapped env texel: the AD program's radiance is negative there
                continue
            g1, h1 = np.zeros(16, np.float32), np.zeros(256, np.float32)
            getattr(ref, "evaluate_path_bidir_%d_%d_static_derv" % (c, l))(P(lens), P(prim), P(scenes[sid]), P(vert), P(g1), P(h1))
            if not (np.isfinite(g1).all() and np.isfinite(h1).all()):
                continue
            per[(c, l)] = per.get((c, l), 0) + 1
            mg = np.zeros(16, np.float32)
            mg[: len(ev[1])] = ev[1]
            rec["scene_id"].append(sid), rec["c"].append(c), rec["l"].append(l), rec["primary"].append(prim.copy()), rec["vert"].append(vert[:600].copy())
            rec["loglum"].append(ev[0]), rec["mala_grad"].append(mg), rec["h2_grad"].append(g1), rec["h2_hess"].append(h1)
        o.close()
        print("scene", sid, dict(sorted(per.items())))
    np.savez_compressed(os.path.join(HERE, "derv_vectors_full.npz"), scenes=np.array(scenes), **{k: np.array(v) for k, v in rec.items()})
    print("wrote derv_vectors_full.npz:", len(rec["c"]), "vectors")


if __name__ == "__main__":
    main()
