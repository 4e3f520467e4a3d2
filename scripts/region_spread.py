"""Seed-to-seed spread of the region ratios used by test_full_render_matches_reference_image (same configuration as the test).
Purpose: derive the test's tolerances from measured run-to-run variation instead of from one run."""
import importlib, json, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
p = importlib.import_module("langevin-mcmc_amd")
ref = np.load(os.path.join(ROOT, "tests", "golden", "torus_ref_images_256x192.npz"))["lmc"]
lum = lambda x: x @ np.array([0.212671, 0.715160, 0.072169])
reg = {"floor": (5, 50, 75, 125), "left face": (100, 120, 60, 100), "front face": (175, 225, 62, 112), "torus": (140, 170, 65, 100), "top face": (110, 190, 22, 37)}
rows = []
for seedchains in (1, 0):
    for so in (0, 1 << 20, 2 << 20, 3 << 20):  # chain seeds are chainId + seedOffset: offsets closer than the stream count reuse the same streams
        ren = p.Renderer(os.path.join(ROOT, "scenes", "torus", "lmc.xml"), width=512, height=384, seed_offset=so)
        ren.set_option("seedchains", seedchains)
        dspp, spp, chains = 64, 160, 1 << 16
        direct = ren.direct_lighting(dspp)
        per = spp * 512 * 384 // chains
        ren.init_chains(32 * chains, chains, 65536, per, per % chains)
        ren.step(per + 1)
        img = direct / dspp + ren.film() / spp
        ren.close()
        d = img.reshape(192, 2, 256, 2, 3).mean(axis=(1, 3))
        lg, lr = lum(d), lum(ref)
        err = np.sort(((lg - lr) ** 2 / (lr ** 2 + 1e-2)).ravel())
        row = dict(seedchains=seedchains, seed_offset=so, mean=float(lg.mean() / lr.mean()), trimmed_relmse=float(err[: int(0.995 * err.size)].mean()))
        row.update({k: float(lg[y0:y1, x0:x1].mean() / lr[y0:y1, x0:x1].mean()) for k, (x0, x1, y0, y1) in reg.items()})
        rows.append(row)
        print(json.dumps(row), flush=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "region_spread.json"), "w"), indent=1)
