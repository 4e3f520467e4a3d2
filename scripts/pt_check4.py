import importlib, os, sys, numpy as np
sys.path.insert(0, "/root/repo")
p = importlib.import_module("langevin-mcmc_amd")
L = lambda x: x @ np.array([0.212671, 0.715160, 0.072169])
regions = {"floor": (5, 50, 75, 125), "left face": (100, 120, 60, 100), "front face": (175, 225, 62, 112), "torus": (140, 170, 65, 100), "top": (110, 190, 22, 37), "whole": (0, 256, 0, 192)}
def means(img): return {k: float(L(img[y0:y1, x0:x1]).mean()) for k, (x0, x1, y0, y1) in regions.items()}
ren = p.Renderer("/root/repo/scenes/torus/lmc.xml", width=256, height=192)
pt = means(ren.path_trace(128) / 128 - ren.direct_lighting(128) / 128)
mc = means(ren.bidir_mc(128))
chains = 1 << 14; mspp = 256; per = mspp * 256 * 192 // chains
ren.set_option("largestepprob", 1.0)
ren.init_chains(64 * chains, chains, 65536, per, per % chains)
ren.step(per + 1)
ml = means(ren.film() / mspp)
for k in regions: print("%-12s PT(>=3) %.4f   bidir MC %.4f   MLT(large only) %.4f" % (k, pt[k], mc[k], ml[k]))
