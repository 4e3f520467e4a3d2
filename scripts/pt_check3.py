import importlib, os, sys, numpy as np
sys.path.insert(0, "/root/repo")
p = importlib.import_module("langevin-mcmc_amd")
L = lambda x: x @ np.array([0.212671, 0.715160, 0.072169])
for fd in (1, 0):
    for md in (3, 4, 8):
        ren = p.Renderer("/root/repo/scenes/torus/lmc.xml", width=128, height=96, max_depth=md, force_diffuse=fd)
        pt = L(ren.path_trace(64) / 64).mean()
        d = L(ren.direct_lighting(64) / 64).mean()
        norm, nc = ren.init_chains(4000000, 1024, 65536, 10, 0)
        print("force_diffuse", fd, "maxdepth", md, "PT(>=3) %.5f" % (pt - d), "MLTInit normalization %.5f" % norm, "ratio %.4f" % (norm / (pt - d)))
        ren.close()
