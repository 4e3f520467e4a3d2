import importlib, os, sys, numpy as np
sys.path.insert(0, "/root/repo")
p = importlib.import_module("langevin-mcmc_amd")
L = lambda x: x @ np.array([0.212671, 0.715160, 0.072169])
regions = {"floor": (5, 50, 75, 125), "left face": (100, 120, 60, 100), "front face": (175, 225, 62, 112), "torus": (140, 170, 65, 100), "top": (110, 190, 22, 37)}
def means(img): return {k: float(L(img[y0:y1, x0:x1]).mean()) for k, (x0, x1, y0, y1) in regions.items()}
prev_pt = prev_m = None
for md in (2, 3, 4, 5, 6):
    ren = p.Renderer("/root/repo/scenes/torus/lmc.xml", width=256, height=192, max_depth=md)
    ren.set_option("largestepprob", 1.0)
    pt = means(ren.path_trace(256) / 256)
    if md >= 3:
        direct = ren.direct_lighting(128) / 128
        chains = 1 << 14; mspp = 256; per = mspp * 256 * 192 // chains
        ren.init_chains(64 * chains, chains, 65536, per, per % chains)
        ren.step(per + 1)
        m = means(direct + ren.film() / mspp)
    else:
        m = means(ren.direct_lighting(256) / 256)
    if prev_pt:
        print("length", md, {k: "pt %.4f mlt %.4f" % (pt[k] - prev_pt[k], m[k] - prev_m[k]) for k in regions})
    prev_pt, prev_m = pt, m
    ren.close()
