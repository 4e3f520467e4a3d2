"""GPU box: per-sample MLTInit contributions of scenes/torus/lmc_arealight.xml (force_diffuse=1), oracle vs device, flag off/on."""
import ctypes, os, sys
from collections import defaultdict
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import gpu_checks as gc, _orc
from tests._orc import P
L = gc.oracle_lib(); p = gc.pkg()
AREA = os.path.join(gc.ROOT, "scenes", "torus", sys.argv[1] if len(sys.argv) > 1 else "lmc_arealight.xml")
ninit = 1 << 19
for flag in (0, 1):
    orc = _orc.Oracle(L, AREA, 1, 6, 160, 120, 0, "")
    ren = p.Renderer(AREA, force_diffuse=1, max_depth=6, width=160, height=120, seed_offset=0, use_gradient=0)
    L.orc_set_option(orc.h, b"uselightcoordinatesampling", float(flag)); ren.set_option("uselightcoordinatesampling", flag)
    orc.init(ninit, 64, 4096); ren.init_chains(ninit, 64, 4096, 10)
    def dump(fn, h):
        cap = 8 * ninit
        s, cl, ls = np.zeros(cap, np.int64), np.zeros(cap, np.int32), np.zeros(cap, np.float32)
        fn.restype = ctypes.c_longlong
        fn.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        n = fn(h, cap, P(s), P(cl), P(ls))
        d = defaultdict(list)
        for a, b, c in zip(s[:n], cl[:n], ls[:n]):
            d[int(a)].append((int(b), float(c)))
        return d, n
    do, no = dump(L.orc_init_contribs, orc.h); dg, ng = dump(p.lib().lmc_init_contribs, ren.h)
    print("flag", flag, "contribs", no, ng)
    shown = 0
    for k in range(ninit):
        a, b = do.get(k, []), dg.get(k, [])
        if [x[0] for x in a] != [x[0] for x in b] or any(abs(x[1] - y[1]) > 1e-3 * abs(x[1]) for x, y in zip(a, b)):
            if shown < 25:
                print(" sample", k, "oracle", a, "gpu", b)
            shown += 1
    print(" differing samples", shown)
    orc.close(); ren.close()
