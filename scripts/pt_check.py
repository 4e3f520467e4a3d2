import importlib, os, sys, numpy as np
sys.path.insert(0, "/root/repo")
p = importlib.import_module("langevin-mcmc_amd")
ren = p.Renderer("/root/repo/scenes/torus/lmc.xml", width=256, height=192)
for kv in sys.argv[1:]:
    k, v = kv.split("="); ren.set_option(k, float(v)); print("set", k, v)
spp = 128
pt = ren.path_trace(spp) / spp
direct = ren.direct_lighting(128) / 128
chains = 1 << 16
mspp = 512
per = mspp * 256 * 192 // chains
ren.init_chains(8 * chains, chains, 65536, per, per % chains)
ren.step(per + 1)
mlt = direct + ren.film() / mspp
np.savez_compressed("/root/repo/gpurun_out/pt_check.npz", pt=pt, mlt=mlt)
L = lambda x: x @ np.array([0.212671, 0.715160, 0.072169])
def box(name, x0, x1, y0, y1):
    a, b = L(pt[y0:y1, x0:x1]).mean(), L(mlt[y0:y1, x0:x1]).mean()
    print("%-20s pt %.4f  direct+mlt %.4f  ratio %.3f" % (name, a, b, b / a))
box("floor far left", 5, 50, 75, 125); box("cube left face", 100, 120, 60, 100); box("cube top face", 110, 190, 22, 37)
box("cube front face", 175, 225, 62, 112); box("torus ring", 140, 170, 65, 100); box("shadow right", 220, 252, 120, 155); box("whole", 0, 256, 0, 192)
