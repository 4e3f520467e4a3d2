import numpy as np, sys
sys.path.insert(0, '/root/repo')
from tests import test_gpu_relocate as t, gpu_checks as gc
opts = {"largestepprob": 0.3, "largestepscale": 1.0}
import os
def run(rel):
    os.environ["LMC_RELOCATE"] = "1" if rel else "0"
    p = gc.pkg()
    ren = p.Renderer(gc.TORUS, force_diffuse=1, max_depth=6, width=256, height=192, seed_offset=0, use_gradient=1)
    for k, v in opts.items(): ren.set_option(k, v)
    ren.init_chains(1 << 20, 1 << 16, 4096, 200)   # 200 mutations per chain: the chains finish inside the run
    out = []
    for upto in (60, 150, 230):
        ren.step(upto - (out[-1][0] if out else 0)); out.append((upto, ren.summary(0).copy(), ren.stats()))
    f = ren.film().copy(); ren.close(); return out, f
a, fa = run(False); b, fb = run(True)
for (u, s0, st0), (_, s1, st1) in zip(a, b):
    t._same_states(s0, s1)
    for k in ("steps", "largeSteps", "accepted", "resets", "cacheQueries", "cacheHits", "gradCalls", "cacheReadyMask"): assert st0[k] == st1[k], (u, k, st0[k], st1[k])
    print("step", u, "states equal; steps", st0["steps"], "accepted", st0["accepted"], "resets", st0["resets"], "cache mask", st0["cacheReadyMask"])
la, lb = gc.lum(fa), gc.lum(fb)
print("film rel diff", float(np.linalg.norm(la - lb) / np.linalg.norm(la)))
