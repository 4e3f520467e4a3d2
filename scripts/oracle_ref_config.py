"""CPU oracle run of the reference's own chain configuration (numchains = 128, dptoptions.h:27; scenes/torus/lmc.xml) with the
reference's scheduling (one chain per work item, orc_run_async), compared with the shipped render like chain_length_sweep.py.
usage: oracle_ref_config.py OUT.json WIDTH HEIGHT SPP CHAINS SEEDS [THREADS]"""
import json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import _orc, gpu_checks as gc
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def main():
    out, W, H, spp, chains, seeds = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    threads = int(sys.argv[7]) if len(sys.argv) > 7 else (os.cpu_count() or 1)
    ref = np.load(os.path.join(ROOT, "tests", "golden", "torus_ref_images_256x192.npz"))["lmc"]
    lum = lambda x: x @ np.array([0.212671, 0.715160, 0.072169])
    REG = {"floor": (5, 50, 75, 125), "left face": (100, 120, 60, 100), "front face": (175, 225, 62, 112), "torus": (140, 170, 65, 100), "top face": (110, 190, 22, 37)}
    L = gc.oracle_lib()
    rows = []
    for s in range(seeds):
        so = s << 20
        orc = _orc.Oracle(L, gc.TORUS, 0, 8, W, H, so, gc.pathref())
        dspp = 16
        direct = orc.direct(dspp) / dspp
        norm, nc = orc.init(max(300000, 32 * chains), chains, 32)
        per = spp * W * H // chains
        orc.setup_chains(per, per % chains)
        t0 = time.time()
        rate, done = orc.run_async(threads)
        dt = time.time() - t0
        img = direct + orc.film() / spp
        st = orc.stats()
        orc.close()
        d = img.reshape(192, H // 192, 256, W // 256, 3).mean(axis=(1, 3))
        lg, lr = lum(d), lum(ref)
        err = np.sort(((lg - lr) ** 2 / (lr ** 2 + 1e-2)).ravel())
        row = dict(side="cpu-oracle", width=W, height=H, spp=spp, chains=chains, steps_per_chain=per, seed_offset=so, threads=threads, normalization=norm,
                   loop_s=dt, mutations_per_s=rate, accept=st["accepted"] / st["steps"], large_frac=st["largeSteps"] / st["steps"], resets=st["resets"],
                   cache_mask=st["cacheReadyMask"], mean=float(lg.mean() / lr.mean()), relmse=float(err.mean()),
                   trimmed_relmse=float(err[: int(0.995 * err.size)].mean()), bright_energy_frac=float(lg[lg > 0.5].sum() / lg.sum()),
                   ref_bright_energy_frac=float(lr[lr > 0.5].sum() / lr.sum()))
        row.update({k: float(lg[y0:y1, x0:x1].mean() / lr[y0:y1, x0:x1].mean()) for k, (x0, x1, y0, y1) in REG.items()})
        rows.append(row)
        print(json.dumps(row), flush=True)
        json.dump(rows, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
