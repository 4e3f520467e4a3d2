"""GPU box: init states of scenes/torus/lmc_arealight.xml (force_diffuse=1), oracle vs device, uselightcoordinatesampling off/on."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import gpu_checks as gc, _orc
L = gc.oracle_lib(); p = gc.pkg()
AREA = os.path.join(gc.ROOT, "scenes", "torus", "lmc_arealight.xml")
np.set_printoptions(precision=6, linewidth=200)
for flag in (0, 1):
    orc = _orc.Oracle(L, AREA, 1, 6, 160, 120, 0, "")
    ren = p.Renderer(AREA, force_diffuse=1, max_depth=6, width=160, height=120, seed_offset=0, use_gradient=1)
    L.orc_set_option(orc.h, b"uselightcoordinatesampling", float(flag)); ren.set_option("uselightcoordinatesampling", flag)
    print("flag", flag, orc.init(1 << 17, 1 << 13, 4096), ren.init_chains(1 << 17, 1 << 13, 4096, 10))
    si, gi = orc.summary(1), ren.summary(1)
    print(" cl equal", np.array_equal(si[:, 1:3], gi[:, 1:3]))
    for col, nm in ((3, "ls"), (4, "ss")):
        rel = np.abs(si[:, col] - gi[:, col]) / np.maximum(np.abs(si[:, col]), 1e-30)
        bad = np.nonzero(rel > 1e-4)[0]
        print(" ", nm, "max rel", rel.max(), "bad", len(bad))
        for i in bad[:8]:
            print("   chain", i, "c,l", si[i, 1:3], "oracle", si[i, 3:5], "gpu", gi[i, 3:5])
            print("    pss o", si[i, 16:16 + 14]); print("    pss g", gi[i, 16:16 + 14])
    d = np.abs(si[:, 16:] - gi[:, 16:])
    print("  pss max diff", d.max(), "rows >1e-5:", (d.max(axis=1) > 1e-5).sum())
    orc.close(); ren.close()
