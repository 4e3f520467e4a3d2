"""GPU box: run-to-run spread of the block means (seed_offset 0 / 1000 / 2000) for the default and the multiplexed large step."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_checks as gc
p = gc.pkg()
W, H = 128, 96
n, steps = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16384, 3000)
def blocks(a): return a.reshape(4, H // 4, 4, W // 4).mean(axis=(1, 3))
ren = p.Renderer(gc.TORUS, force_diffuse=1, max_depth=6, width=W, height=H, seed_offset=0, use_gradient=1)
gt = blocks(gc.lum(ren.bidir_mc(8192))); ren.close()
np.set_printoptions(precision=3, linewidth=200)
for name, opts in (("default", {}), ("mux", {"largestepmultiplexed": 1})):
    rs = []
    for seed in (0, 1000):
        ren = p.Renderer(gc.TORUS, force_diffuse=1, max_depth=6, width=W, height=H, seed_offset=seed, use_gradient=1)
        for k, v in opts.items(): ren.set_option(k, v)
        ren.init_chains(8 * n, n, 4096, steps, 0); ren.step(steps)
        rs.append(blocks(gc.lum(ren.film()) / (n * steps) * (W * H)) / gt); ren.close()
    rs = np.array(rs)
    print(name, "block ratio to ground truth, mean over 3 seeds:\n", rs.mean(axis=0), "\n spread (max - min over seeds):\n", rs.max(axis=0) - rs.min(axis=0), flush=True)
