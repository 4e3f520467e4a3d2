#!/usr/bin/env python3
"""Full render of the shipped torus scene (its own materials, maxdepth 8, spp 245, directspp 256) on the GPU, compared with the
reference authors' own render of the same scene file (tests/golden/torus_lmc_ref_256x192.npz = the shipped
scenes/torus/lmc_timeuse_44.689152s.exr box-downsampled 4x; made by tests/golden/make_golden_images.py).
Prints relMSE figures; used for DESIGN.md §5 and by tests/test_gpu_parity.py::test_full_render_matches_reference_image."""
import importlib, json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def down(img, f):
    h, w, _ = img.shape
    return img.reshape(h // f, f, w // f, f, 3).mean(axis=(1, 3))


def rel_mse(a, b):
    """mean over pixels of |a-b|^2 / (b^2 + 1e-2), on luminance"""
    la = a @ np.array([0.212671, 0.715160, 0.072169]); lb = b @ np.array([0.212671, 0.715160, 0.072169])
    return float(np.mean((la - lb) ** 2 / (lb ** 2 + 1e-2)))


def render(spp=245, chains=1 << 18, direct_spp=256, seed=0, init_mult=32):
    p = importlib.import_module("langevin-mcmc_amd")
    scene = os.path.join(ROOT, "scenes", "torus", "lmc.xml")
    ren = p.Renderer(scene, seed_offset=seed)
    W, H = ren.width, ren.height
    t0 = time.time()
    direct = ren.direct_lighting(direct_spp)
    t_direct = time.time() - t0
    total = spp * W * H
    per = total // chains
    ren.init_chains(max(int(ren.get_option("numinitsamples")), init_mult * chains), chains, 65536, per, per % chains)  # MLTInit needs #contribs >= #chains (mlt.h:101-105)
    t0 = time.time()
    ren.step(per + 1)
    ren.sync()
    t_mlt = time.time() - t0
    img = direct / direct_spp + ren.film() / spp
    st = ren.stats()
    ren.close()
    return img, dict(t_direct=t_direct, t_mlt=t_mlt, mutations=st["steps"], accept=st["accepted"] / max(st["steps"], 1))


if __name__ == "__main__":
    z = np.load(os.path.join(ROOT, "tests", "golden", "torus_ref_images_256x192.npz"))
    ref_lmc, ref_h2mc = z["lmc"], z["h2mc"]
    img, info = render(chains=int(os.environ.get("CHAINS", str(1 << 18))))
    d = down(img, 4)
    out = dict(info)
    out["relmse_gpu_vs_ref_lmc"] = rel_mse(d, ref_lmc)
    out["relmse_gpu_vs_ref_h2mc"] = rel_mse(d, ref_h2mc)
    out["relmse_ref_lmc_vs_ref_h2mc"] = rel_mse(ref_lmc, ref_h2mc)
    L = lambda x: x @ np.array([0.212671, 0.715160, 0.072169])
    reg = {"floor": (5, 50, 75, 125), "left face": (100, 120, 60, 100), "front face": (175, 225, 62, 112), "torus": (140, 170, 65, 100), "top": (110, 190, 22, 37)}
    out["region_ratio_gpu_over_ref"] = {k: float(L(d[y0:y1, x0:x1]).mean() / L(ref_lmc[y0:y1, x0:x1]).mean()) for k, (x0, x1, y0, y1) in reg.items()}
    lg, lr = L(d).ravel(), L(ref_lmc).ravel()
    out["energy_frac_above_0.5"] = dict(gpu=float(lg[lg > 0.5].sum() / lg.sum()), ref=float(lr[lr > 0.5].sum() / lr.sum()))
    out["mean_lum"] = dict(gpu=float(d.mean()), ref_lmc=float(ref_lmc.mean()), ref_h2mc=float(ref_h2mc.mean()))
    print(json.dumps(out))
    if len(sys.argv) > 1:
        p = importlib.import_module("langevin-mcmc_amd")
        p.write_exr(sys.argv[1], img)
