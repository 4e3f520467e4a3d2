"""GPU box: the image of the Lambertian torus rendered with the default large step, with `largestepmultiplexed`, and with
`largestepmultiplexed` + `samplecache`: all three are Metropolis-Hastings samplers of the same target, so the images must agree."""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_checks as gc
p = gc.pkg()
W, H = 128, 96
n, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, int(sys.argv[2]) if len(sys.argv) > 2 else 3000
imgs = {}
for name, opts in (("default", {}), ("mux", {"largestepmultiplexed": 1}), ("mux+cache", {"largestepmultiplexed": 1, "samplecache": 1})):
    ren = p.Renderer(gc.TORUS, force_diffuse=1, max_depth=6, width=W, height=H, seed_offset=0, use_gradient=1)
    for k, v in opts.items():
        ren.set_option(k, v)
    norm, _ = ren.init_chains(8 * n, n, 4096, steps, 0)
    ren.step(steps)
    st = ren.stats()
    lum = gc.lum(ren.film()) / (n * steps) * (W * H)
    imgs[name] = lum
    print(name, "mean", float(lum.mean()), "accept", st["accepted"] / st["steps"], "large", st["largeSteps"] / st["steps"], "ready", st["cacheReadyMask"], flush=True)
    ren.close()
ren = p.Renderer(gc.TORUS, force_diffuse=1, max_depth=6, width=W, height=H, seed_offset=0, use_gradient=1)
gt = gc.lum(ren.bidir_mc(int(sys.argv[3]) if len(sys.argv) > 3 else 8192))
ren.close()
print("ground truth (plain bidirectional MC, path length >= 3) mean", float(gt.mean()))
def blocks(a): return a.reshape(4, H // 4, 4, W // 4).mean(axis=(1, 3))
for name in ("default", "mux", "mux+cache"):
    r = blocks(imgs[name]) / blocks(gt)
    print(name, "vs ground truth: mean ratio", float(imgs[name].mean() / gt.mean()), "block ratios min/max", float(r.min()), float(r.max()))
    print(np.round(r, 3))
ref = imgs["default"]
def blocks(a): return a.reshape(4, H // 4, 4, W // 4).mean(axis=(1, 3))
for name in ("mux", "mux+cache"):
    r = blocks(imgs[name]) / blocks(ref)
    rel = np.mean((imgs[name] - ref) ** 2 / (ref ** 2 + 0.01))
    print(name, "mean ratio", float(imgs[name].mean() / ref.mean()), "block ratios min/max", float(r.min()), float(r.max()), "relMSE", float(rel))
    print(np.round(r, 3))
