#!/usr/bin/env python3
"""relMSE of the plain-MC truth estimator (direct pre-pass + lmc_bidir_mc) against the reference's shipped torus render, by sample count. (GPU)"""
import importlib, json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_checks as gc

p = importlib.import_module("langevin-mcmc_amd")
lum = lambda x: x.astype(np.float64) @ np.array([0.212671, 0.715160, 0.072169])
ref = np.load(os.path.join(ROOT, "tests", "golden", "torus_ref_images_256x192.npz"))
lr, lr2 = lum(ref["lmc"]), lum(ref["h2mc"])
def relmse(a, b, trim=0.0):
    e = np.sort(((a - b) ** 2 / (b ** 2 + 0.01)).ravel())
    return float(e[: int(len(e) * (1 - trim))].mean())
print(json.dumps({"ref_lmc_vs_h2mc": relmse(lr2, lr), "trimmed": relmse(lr2, lr, 0.005)}))
ren = p.Renderer(gc.TORUS, force_diffuse=0, max_depth=8, width=256, height=192, seed_offset=0, use_gradient=0)
d = lum(ren.direct_lighting(256) / 256)
for spp in (2048, 8192, 32768):
    t0 = time.time()
    img = d + lum(ren.bidir_mc(spp))
    print(json.dumps({"spp": spp, "seconds": time.time() - t0, "relMSE": relmse(img, lr), "relMSE_trim0.5%": relmse(img, lr, 0.005), "mean_ratio": float(img.mean() / lr.mean())}), flush=True)
