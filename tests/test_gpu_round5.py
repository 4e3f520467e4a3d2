"""GPU tier, round 5: pins that do not go through "the same text under two compilers", the explanation of the image-mean offset, hygiene."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import gpu_checks as gc
from tests._orc import P

pytestmark = pytest.mark.gpu


def test_device_transcendentals_against_float64():
    """device/dtrans.h is compiled by hipcc for the product and by g++ for the oracle, so "GPU == oracle" says
    nothing about lexpf / llogf / lpowf themselves.  Here the DEVICE's results (lmc_trans_probe) are held against an independent float64 numpy
    evaluation on 6 x 2^20 arguments: exp and log within 1.05 ulp over the whole float range, pow within 1.5 ulp where the result exceeds 1e-10
    (the Phong lobe's cut-off, phong.cpp:44) and within 5 ulp down to the subnormal range -- the accuracy contract the glossy BSDFs rely on in
    place of libm's powf / expf / logf (<= 1 ulp-ish each, microfacet.h:17,173, phong.cpp:42,109)."""
    lib = gc.pkg().lib()

    def ulps(got, ref64):
        ulp = np.spacing(np.abs(ref64.astype(np.float32))).astype(np.float64)
        return np.abs(got.astype(np.float64) - ref64) / ulp

    worst = {}
    for mode, x, y in gc.trans_cases(seed=5):
        o = np.zeros(len(x), np.float32)
        assert lib.lmc_trans_probe(len(x), mode, P(x), P(y), P(o)) == 0
        x64, y64 = x.astype(np.float64), y.astype(np.float64)
        with np.errstate(all="ignore"):
            ref = np.exp(x64) if mode == 0 else np.log(x64) if mode == 1 else np.power(x64, y64)
        normal = (np.abs(ref) > 1.2e-38) & (np.abs(ref) < 3.4e38)
        e = ulps(o[normal], ref[normal])
        worst[mode] = max(worst.get(mode, 0.0), float(e.max()))
        if mode < 2:
            assert e.max() <= 1.05, (mode, e.max())
        else:
            big = np.abs(ref[normal]) > 1e-10
            assert e[big].max() <= 1.5 and e.max() <= 5.0, (e[big].max(), e.max())
        # results beyond the float range land on the same side as the float64 value
        over, under = ref > 3.5e38, np.abs(ref) < 1e-46
        assert np.all(np.isinf(o[over])) and np.all(o[under] == 0)
    print("worst ulp error on the device: exp %.3f log %.3f pow %.3f" % (worst[0], worst[1], worst[2]))


def test_image_mean_offset_is_the_shipped_renders_own_normaliser():
    """our torus images sit a constant +1.5 % above the render the reference ships, at every sample count.  An MLT image is
    histogram x `normalization`, and the reference estimates `normalization` ONCE from numinitsamples = 300 000 samples on NumSystemCores() init
    streams (mlt.h:41-154, lmc.xml:9): on this scene that estimate has a standard deviation of 5.8 % (profiles/r05_k_normalization_of_300k_init_samples_torus.jsonl,
    48 disjoint stream sets; veach-door: 1.0 %), and with 32 streams and seedoffset 0 -- the configuration of a 32-core machine, which is what the
    render's file name and the authors' README imply -- it is 0.982 .. 0.984 of the converged value.  Asserted on the GPU: (a) that ratio;
    (b) rendered with the reference's OWN init configuration the image mean is within 1 % of the shipped render, while the converged normaliser
    gives the familiar +1 .. 2 % (the direct pre-pass, 60 % of the image's energy, does not depend on the normaliser: the indirect part moves by
    the full ratio); (c) the two renders differ by exactly the normaliser ratio, i.e. nothing else in the pipeline depends on the init sample count."""
    p = gc.pkg()
    ref = np.load(os.path.join(gc.ROOT, "tests", "golden", "torus_ref_images_256x192.npz"))["lmc"]
    lum = lambda x: x @ np.array([0.212671, 0.715160, 0.072169])
    lr = lum(ref)
    W, H, dspp, chains, per = 256, 192, 256, 2048, 47000
    spp = per * chains / (W * H)
    ren = p.Renderer(gc.TORUS, width=W, height=H, seed_offset=0)
    direct = lum(ren.direct_lighting(dspp)) / dspp
    ren.close()
    out = {}
    for name, ninit, threads in (("reference_init", 300000, 32), ("converged", 1 << 23, 65536)):
        ren = p.Renderer(gc.TORUS, width=W, height=H, seed_offset=0)
        norm, _ = ren.init_chains(ninit, chains, threads, per, 0)
        done = 0
        while done < per + 1:
            ren.step(min(4096, per + 1 - done))
            done += 4096
        ind = lum(ren.film()) / spp
        ren.close()
        err = (direct + ind - lr) ** 2 / (lr ** 2 + 1e-2)
        out[name] = {"normalization": norm, "indirect_mean": float(ind.mean()), "image_mean_over_shipped": float((direct + ind).mean() / lr.mean()),
                     "relMSE": float(err.mean()), "spp": spp}
    print(json.dumps(out))
    ratio = out["reference_init"]["normalization"] / out["converged"]["normalization"]
    assert 0.975 < ratio < 0.99  # measured 0.9844 (oracle, CPU: 0.9820)
    assert abs(out["reference_init"]["image_mean_over_shipped"] - 1) < 0.01  # measured 1.0056
    # SURVEY.md 8(d)'s bar itself, untrimmed, at the reference's sample count (1958 spp here = half of the 245 x 16 samples behind a pixel of the 4 x
    # down-sampled fixture): relMSE <= 2 x the relMSE between the two renders the reference ships (0.00547) -- with the reference's init, and with the
    # converged normaliser too (the offset costs 1e-4)
    assert out["reference_init"]["relMSE"] <= 0.0109 and out["converged"]["relMSE"] <= 0.0109, out
    assert 1.004 < out["converged"]["image_mean_over_shipped"] < 1.025  # measured 1.0121 (plain Monte Carlo truth estimator of bench.py: 1.015)
    # the indirect images of the two runs differ by the normaliser ratio and by nothing else (two independent 47 k-mutation renders: 1 % noise)
    assert abs(out["reference_init"]["indirect_mean"] / out["converged"]["indirect_mean"] / ratio - 1) < 0.012


def test_work_skipping_switches_are_not_in_the_shipped_library():
    """LMC_EXP_NOSPLAT / NOQUERY / NOGRAD / ... skip work inside the step kernels.  They are compiled into
    -DLMC_EXP_SWITCHES builds only (scripts/build_exp.sh); the shipped library refuses to create a context while one is set, and bench.py refuses
    before it gets that far."""
    code = "import importlib,sys; sys.path.insert(0, %r); p = importlib.import_module('langevin-mcmc_amd'); from tests import gpu_checks as gc; p.Renderer(gc.TORUS, force_diffuse=1, max_depth=6)" % gc.ROOT
    for var in ("LMC_EXP_NOSPLAT", "LMC_EXP_NOQUERY", "LMC_EXP_NOHESS"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **{var: "1"}), cwd=gc.ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert r.returncode != 0 and "no work-skipping measurement switches" in r.stderr, (var, r.stderr[-500:])
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LMC_EXP_NOSPLAT="0"), cwd=gc.ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-500:]
