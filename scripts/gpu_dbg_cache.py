import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import _orc, gpu_checks as gc
L = gc.oracle_lib(); p = gc.pkg()
orc = _orc.Oracle(L, gc.TORUS, 1, 6, 128, 96, 0, gc.pathref())
ren = p.Renderer(gc.TORUS, force_diffuse=1, max_depth=6, width=128, height=96, seed_offset=0, use_gradient=1)
for o in (("largestepprob", 0.3),):
    L.orc_set_option(orc.h, o[0].encode(), o[1]); ren.set_option(*o)
orc.init(200000, 8192, 64); ren.init_chains(200000, 8192, 64, 120)
orc.setup_chains(120, 0)
for s in range(24):
    orc.step(1); ren.step(1)
    a, b = orc.stats(), ren.stats()
    print(s, {k: (a[k], b[k]) for k in ("largeSteps", "accepted", "gradCalls", "cacheQueries", "cacheHits", "cacheReadyMask")}, flush=True)
