import importlib, os, sys, numpy as np
sys.path.insert(0, "/root/repo")
p = importlib.import_module("langevin-mcmc_amd")
L = lambda x: x @ np.array([0.212671, 0.715160, 0.072169])
regions = {"floor": (5, 50, 75, 125), "left face": (100, 120, 60, 100), "front face": (175, 225, 62, 112), "torus": (140, 170, 65, 100), "top": (110, 190, 22, 37), "whole": (0, 256, 0, 192)}
def means(img): return {k: float(L(img[y0:y1, x0:x1]).mean()) for k, (x0, x1, y0, y1) in regions.items()}
ren = p.Renderer("/root/repo/scenes/torus/lmc.xml", width=256, height=192)
mc = means(ren.bidir_mc(256))
res = {}
for seed in (0, 1):
    for chains in (1 << 12, 1 << 16):
        ren.set_option("seedchains", seed)
        mspp = 256; per = mspp * 256 * 192 // chains
        ren.film()  # noop
        import ctypes
        p.lib().lmc_film_clear(ren.h)
        ren.init_chains(max(300000, 16 * chains), chains, 65536, per, per % chains)
        ren.step(per + 1)
        res[(seed, chains)] = means(ren.film() / mspp)
for k in regions:
    print("%-11s MC %.4f | " % (k, mc[k]) + "  ".join("seed%d/%dch(%d steps) %.4f" % (s, c, 256 * 256 * 192 // c, v[k]) for (s, c), v in res.items()))
