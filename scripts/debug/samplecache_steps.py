"""GPU box: lock-step oracle vs device with samplecache; per step the number of chains whose state agrees, and the first disagreements."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import gpu_checks as gc, _orc
np.set_printoptions(precision=6, linewidth=220)
p = gc.pkg(); L = gc.oracle_lib()
opts = {"largestepprob": 0.3, "largestepscale": 1.0, "largestepmultiplexed": 1, "samplecache": 1}
orc = _orc.Oracle(L, gc.TORUS, 1, 6, 128, 96, 0, gc.pathref())
ren = p.Renderer(gc.TORUS, force_diffuse=1, max_depth=6, width=128, height=96, seed_offset=0, use_gradient=1)
for k, v in opts.items():
    L.orc_set_option(orc.h, k.encode(), float(v)); ren.set_option(k, v)
orc.init(200000, 16384, 64); ren.init_chains(200000, 16384, 64, 120)
orc.setup_chains(120, 0)
prev_ok = None
for step in range(45):
    orc.step(1); ren.step(1)
    co, cg = orc.summary(0), ren.summary(0)
    same = (co[:, 0] == cg[:, 0]) & (co[:, 1] == cg[:, 1]) & (co[:, 2] == cg[:, 2]) & (np.abs(co[:, 3] - cg[:, 3]) <= 1e-3 * np.abs(co[:, 3]) + 1e-12)
    so, sg = orc.stats(), ren.stats()
    print(step, "match", same.mean(), "ready", so["cacheReadyMask"], sg["cacheReadyMask"], "large", so["largeSteps"], sg["largeSteps"], "acc", so["accepted"], sg["accepted"], flush=True)
    if prev_ok is not None:
        newbad = np.nonzero(prev_ok & ~same)[0]
        for i in newbad[:4]:
            print("   chain", i, "oracle", co[i, :5], "pss", co[i, 16:22]); print("   chain", i, "gpu   ", cg[i, :5], "pss", cg[i, 16:22])
    prev_ok = same
