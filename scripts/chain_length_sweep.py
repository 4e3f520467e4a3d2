"""Chain-length sweep of the end-to-end image against the reference's shipped render.

Renders scenes/torus/lmc.xml with the reference's own start-up semantics (every chain begins with a forced
large step, mlt.h:121) at a fixed mutation budget (spp x W x H) split over different numbers of chains, and reports probe-region
luminance ratios against tests/golden/torus_ref_images_256x192.npz (= scenes/torus/lmc_timeuse_44.689152s.exr box-downsampled).
The reference itself runs 128 chains x 1.5 M steps.

usage: chain_length_sweep.py OUT.json WIDTH HEIGHT SPP chains:seeds [chains:seeds ...]"""
import importlib, json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
p = importlib.import_module("langevin-mcmc_amd")
ref = np.load(os.path.join(ROOT, "tests", "golden", "torus_ref_images_256x192.npz"))["lmc"]
lum = lambda x: x @ np.array([0.212671, 0.715160, 0.072169])
REG = {"floor": (5, 50, 75, 125), "left face": (100, 120, 60, 100), "front face": (175, 225, 62, 112), "torus": (140, 170, 65, 100), "top face": (110, 190, 22, 37)}


def compare(img):
    h, w = img.shape[:2]
    d = img.reshape(192, h // 192, 256, w // 256, 3).mean(axis=(1, 3))
    lg, lr = lum(d), lum(ref)
    err = np.sort(((lg - lr) ** 2 / (lr ** 2 + 1e-2)).ravel())
    row = dict(mean=float(lg.mean() / lr.mean()), relmse=float(err.mean()), trimmed_relmse=float(err[: int(0.995 * err.size)].mean()),
               bright_energy_frac=float(lg[lg > 0.5].sum() / lg.sum()), ref_bright_energy_frac=float(lr[lr > 0.5].sum() / lr.sum()))
    row.update({k: float(lg[y0:y1, x0:x1].mean() / lr[y0:y1, x0:x1].mean()) for k, (x0, x1, y0, y1) in REG.items()})
    return row


def main():
    out, W, H, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    rows = []
    direct_cache = {}
    for spec in sys.argv[5:]:
        chains, seeds = (int(x) for x in spec.split(":"))
        for s in range(seeds):
            so = s << 20  # chain seeds are chainId + seedOffset: offsets closer than the stream count reuse the same streams
            ren = p.Renderer(os.path.join(ROOT, "scenes", "torus", "lmc.xml"), width=W, height=H, seed_offset=so)
            dspp = 64
            if so not in direct_cache:
                direct_cache[so] = ren.direct_lighting(dspp) / dspp
            total = spp * W * H
            per = total // chains
            ninit = max(300000, 32 * chains)
            t0 = time.time()
            norm, nc = ren.init_chains(ninit, chains, min(65536, ninit // 4), per, per % chains)
            t1 = time.time()
            done = 0
            while done < per + 1:
                n = min(4096, per + 1 - done)
                ren.step(n)
                done += n
            ren.sync()
            t2 = time.time()
            img = direct_cache[so] + ren.film() / spp
            st = ren.stats()
            ren.close()
            row = dict(width=W, height=H, spp=spp, chains=chains, steps_per_chain=per, seed_offset=so, normalization=norm, init_s=t1 - t0, loop_s=t2 - t1,
                       us_per_step=(t2 - t1) / (per + 1) * 1e6, mutations_per_s=st["steps"] / (t2 - t1), accept=st["accepted"] / st["steps"],
                       large_frac=st["largeSteps"] / st["steps"], resets=st["resets"], cache_mask=st["cacheReadyMask"])
            row.update(compare(img))
            rows.append(row)
            print(json.dumps(row), flush=True)
            json.dump(rows, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
