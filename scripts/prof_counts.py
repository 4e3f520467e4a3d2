import importlib, sys, ctypes
sys.path.insert(0, "/root/repo")
p = importlib.import_module("langevin-mcmc_amd")
from tests import gpu_checks as gc
N = 1 << 20
ren = p.Renderer(gc.TORUS, force_diffuse=1, max_depth=6, seed_offset=0, device=0, use_gradient=1)
ren.init_chains(8 * N, N, 65536, 256, 0, 0, N)
ren.step(48); ren.sync()
a = (ctypes.c_float * 12)(); p.lib().lmc_prof_read(a); a0 = list(a); ren.step_timing(); l0 = ren.kernel_timing()[2]
ren.step(16); ren.sync()
p.lib().lmc_prof_read(a); a1 = list(a); ren.step_timing(); l1 = ren.kernel_timing()[2]
lean = l1 - l0; waves = lean / 64.0
d = [x - y for x, y in zip(a1, a0)]
names = ["rays(lane)", "node it (wave)", "node it (lane)", "leaf phases (wave)", "tri it (wave)", "tri it (lane)", "kd it (wave)", "kd it (lane)", "kdpt it (wave)", "kdpt (lane)", "camvert it (wave)", "camvert (lane)"]
for n, v in zip(names, d):
    print("%-20s per wave-step %.1f   per lane-step %.2f" % (n, v / waves, v / lean))
