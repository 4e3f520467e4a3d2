#!/usr/bin/env python3
"""where does the constant +1.5 % of our torus images against the render the reference ships come from?  Hypothesis
(scripts/debug/normalization_study.py, CPU): from the shipped render's own normaliser -- the reference estimates `normalization` once from
numinitsamples = 300 000 samples on NumSystemCores() init streams (mlt.h:41-154), and on the torus that estimate has a standard deviation of 5.8 %;
with 32 streams and seedoffset 0 it is 0.982 of the converged value.  Test on the GPU: the same render with (A) the reference's own init
configuration, (B) a converged normaliser, (C) 300 000 samples on 64 streams (0.9994 of converged).  An MLT image is histogram x normalization,
so the indirect image scales with it; the direct pre-pass does not.
usage (GPU box): python scripts/debug/mean_offset_study.py [mutations per chain / 1000]"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_checks as gc  # noqa: E402

p = importlib.import_module("langevin-mcmc_amd")
kmut = int(sys.argv[1]) if len(sys.argv) > 1 else 47
ref = np.load(os.path.join(ROOT, "tests", "golden", "torus_ref_images_256x192.npz"))
lum = lambda x: x @ np.array([0.212671, 0.715160, 0.072169])
lr, lr2 = lum(ref["lmc"]), lum(ref["h2mc"])
W, H, dspp, chains = 256, 192, 256, 2048
per = kmut * 1000
spp = per * chains / (W * H)
ren = p.Renderer(gc.TORUS, width=W, height=H, seed_offset=0)
direct = lum(ren.direct_lighting(dspp)) / dspp
ren.close()
print(json.dumps({"direct_share_of_the_shipped_image_mean": float(direct.mean() / lr.mean()), "shipped_lmc_over_shipped_h2mc_mean": float(lr.mean() / lr2.mean())}))
norms = {}
for name, ninit, threads in (("A: the reference's init (300 000 samples, 32 streams)", 300000, 32), ("B: converged (2^23 samples, 65536 streams)", 1 << 23, 65536),
                             ("C: 300 000 samples, 64 streams", 300000, 64), ("D: 300 000 samples, 16 streams", 300000, 16)):
    ren = p.Renderer(gc.TORUS, width=W, height=H, seed_offset=0)
    norm, nc = ren.init_chains(ninit, chains, threads, per, 0)
    done = 0
    while done < per + 1:
        ren.step(min(4096, per + 1 - done))
        done += 4096
    ind = lum(ren.film()) / spp
    ren.close()
    img = direct + ind
    norms[name[0]] = norm
    print(json.dumps({"config": name, "normalization": norm, "normalization_over_converged": None if "B" not in norms else norm / norms["B"], "chains": chains, "mutations_per_chain": per,
                      "image_mean_over_shipped_lmc": float(img.mean() / lr.mean()), "indirect_mean_over_(shipped_minus_direct)": float(ind.mean() / (lr.mean() - direct.mean())),
                      "floor_region": float(img[75:125, 5:50].mean() / lr[75:125, 5:50].mean())}), flush=True)
print(json.dumps({"normalization_A_over_B": norms["A"] / norms["B"], "C_over_B": norms["C"] / norms["B"], "D_over_B": norms["D"] / norms["B"]}))
