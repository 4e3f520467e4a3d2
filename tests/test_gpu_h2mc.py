"""`-m gpu` tier, H2MC (BASELINE.json configs[4]): the second-order path program on the device against the reference's generated
gradient + Hessian programs, the H2MC chain loop against the CPU oracle, and the end-to-end image of the shipped veach-door
h2mc.xml against the render the reference ships."""
import ctypes
import os

import numpy as np
import pytest

from tests import _orc
from tests import gpu_checks as gc
from tests._orc import P

pytestmark = pytest.mark.gpu
DOOR_H2 = os.path.join(gc.ROOT, "scenes", "veachdoor", "h2mc.xml")


@pytest.fixture(scope="module")
def L():
    return gc.oracle_lib()


def _hess_batch(c, l, prim, sp, vert):
    lib = gc.pkg().lib()
    lib.lmc_hess_batch.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 6
    n = len(prim)
    dim = 2 * max(c + l - 1, 2)
    ps, vs = np.ascontiguousarray(prim.T, np.float32), np.ascontiguousarray(vert.T, np.float32)
    ll, g, h = np.zeros(n, np.float32), np.zeros((dim, n), np.float32), np.zeros((dim * dim, n), np.float32)
    r = lib.lmc_hess_batch(c, l, n, P(ps), P(sp), P(vs), P(ll), P(g), P(h))
    assert r == 0, lib.lmc_last_error()
    return ll, g.T.copy(), h.T.reshape(n, dim, dim).copy()


def test_hessian_kernel_matches_reference_programs(L):
    """lmc_hess_batch (HIP) vs evaluate_path_bidir_<c>_<l>_static_derv of the reference (oracle/_ref) on Lambertian states of the
    torus: gradient and Hessian within 1e-2 relative for every state; plus the dlsym'd plugin symbol on single paths."""
    if not gc.pathref():
        pytest.skip("oracle/_ref not built")
    ref = ctypes.CDLL(gc.pathref())
    if not hasattr(ref, "evaluate_path_bidir_3_1_static_derv"):
        pytest.skip("oracle/_ref built without the H2MC programs")
    orc = _orc.Oracle(L, gc.TORUS, 1, 6, 160, 120, 0, gc.pathref())
    orc.init(30000, 512, 8)
    sp = orc.scene_params()
    inp = gc.collect_grad_inputs(orc, 512)
    lens = np.zeros(2, np.float32)
    lib = gc.pkg().lib()
    checked = 0
    for (c, l), (prim, vert) in sorted(inp.items()):
        if c + l > 7:
            continue
        dim = 2 * max(c + l - 1, 2)
        ll, g, h = _hess_batch(c, l, prim, sp, vert)
        f = getattr(ref, "evaluate_path_bidir_%d_%d_static_derv" % (c, l))
        for i in range(min(len(prim), 64)):
            g1, h1 = np.zeros(16, np.float32), np.zeros(256, np.float32)
            f(P(lens), P(prim[i]), P(sp), P(vert[i]), P(g1), P(h1))
            H1 = h1[: dim * dim].reshape(dim, dim)
            if not (np.isfinite(H1).all() and np.isfinite(g1).all()):
                continue
            assert np.linalg.norm(g1[:dim] - g[i]) <= 1e-2 * max(np.linalg.norm(g1[:dim]), 1e-2), (c, l, i)
            assert np.linalg.norm(H1 - h[i]) <= 1e-2 * max(np.linalg.norm(H1), 1e-1), (c, l, i)
            checked += 1
        # the plugin symbol the reference would dlsym from pathlibbidir.so: 6 pointer arguments, one path per call
        d = getattr(lib, "evaluate_path_bidir_%d_%d_static_derv" % (c, l))
        g2, h2 = np.zeros(16, np.float32), np.zeros(256, np.float32)
        d(P(lens), P(prim[0]), P(sp), P(vert[0]), P(g2), P(h2))
        assert np.allclose(g2[:dim], g[0], rtol=1e-5, atol=1e-6) and np.allclose(h2[: dim * dim].reshape(dim, dim), h[0], rtol=1e-5, atol=1e-5)
    orc.close()
    assert checked > 150


def test_hessian_kernel_full_materials_torus_and_door(L):
    """lmc_hess_batch (HIP) on FULL-MATERIAL states (Phong, rough dielectric, textures; area light on the door) of both shipped
    scenes against the reference's H2MC programs (all 42 are built into oracle/_ref): >= 95 % of the states of every technique and
    >= 99 % overall within 1e-2 relative (gradient L2, Hessian Frobenius).  cfg 5 (veach-door H2MC) runs exactly this material set."""
    if not gc.pathref():
        pytest.skip("oracle/_ref not built")
    ref = ctypes.CDLL(gc.pathref())
    if not hasattr(ref, "evaluate_path_bidir_9_0_static_derv"):
        pytest.skip("oracle/_ref built without all H2MC programs")
    lens = np.zeros(2, np.float32)
    for xml in (gc.TORUS, os.path.join(gc.ROOT, "scenes", "veachdoor", "lmc.xml")):
        orc = _orc.Oracle(L, xml, 0, 8, 160, 120, 0, gc.pathref())
        orc.init(40000, 1024, 8)
        sp = orc.scene_params()
        inp = gc.collect_grad_inputs(orc, 1024)
        tot_all = ok_all = 0
        dims = set()
        for (c, l), (prim, vert) in sorted(inp.items()):
            dim = 2 * max(c + l - 1, 2)
            ll, g, h = _hess_batch(c, l, prim, sp, vert)
            f = getattr(ref, "evaluate_path_bidir_%d_%d_static_derv" % (c, l))
            tot = ok = 0
            for i in range(min(len(prim), 48)):
                g1, h1 = np.zeros(16, np.float32), np.zeros(256, np.float32)
                pv, vv = np.zeros(17, np.float32), np.zeros(1000, np.float32)
                pv[: prim.shape[1]], vv[: vert.shape[1]] = prim[i], vert[i]
                f(P(lens), P(pv), P(sp), P(vv), P(g1), P(h1))
                H1 = h1[: dim * dim].reshape(dim, dim)
                if not (np.isfinite(H1).all() and np.isfinite(g1).all()):
                    continue
                tot += 1
                eg = np.linalg.norm(g1[:dim] - g[i]) / max(np.linalg.norm(g1[:dim]), 1e-2)
                eh = np.linalg.norm(H1 - h[i]) / max(np.linalg.norm(H1), 1e-1)
                ok += (eg < 1e-2) and (eh < 1e-2)
            assert ok >= 0.95 * tot - 1, (xml, c, l, ok, tot)
            tot_all, ok_all = tot_all + tot, ok_all + ok
            dims.add(dim)
        orc.close()
        assert tot_all > 400 and ok_all >= 0.99 * tot_all, (xml, ok_all, tot_all)
        assert max(dims) >= 14


def test_golden_derivative_vectors_through_the_c_abi():
    """tests/golden/derv_vectors_full.npz (the reference's MALA-gradient and H2MC gradient + Hessian programs on full-material
    states of both scenes, c + l up to 9) through lmc_grad_batch / lmc_hess_batch: needs neither /root/reference nor oracle/_ref."""
    z = np.load(os.path.join(gc.ROOT, "tests", "golden", "derv_vectors_full.npz"))
    p = gc.pkg()
    n = len(z["c"])
    bad = []
    for sid in (0, 1):
        sp = z["scenes"][sid].copy()
        for c, l in sorted({(int(a), int(b)) for a, b, s in zip(z["c"], z["l"], z["scene_id"]) if s == sid}):
            idx = [i for i in range(n) if z["scene_id"][i] == sid and z["c"][i] == c and z["l"][i] == l]
            dim = 2 * (c + l - 1)
            V = 238 + 59 * (c + l - 3)
            prim, vert = z["primary"][idx][:, : dim + 1].copy(), z["vert"][idx][:, :V].copy()
            ll, g = p.grad_batch(c, l, prim.T.copy(), sp, vert.T.copy())
            ll2, g2, h2 = _hess_batch(c, l, prim, sp, vert)
            for k, i in enumerate(idx):
                assert abs(ll[k] - z["loglum"][i]) < 5e-3
                rg = z["mala_grad"][i][:dim]
                e1 = np.linalg.norm(rg - g[:, k]) / max(np.linalg.norm(rg), 1e-2)
                H1 = z["h2_hess"][i][: dim * dim].reshape(dim, dim)
                e2 = np.linalg.norm(z["h2_grad"][i][:dim] - g2[k]) / max(np.linalg.norm(z["h2_grad"][i][:dim]), 1e-2)
                e3 = np.linalg.norm(H1 - h2[k]) / max(np.linalg.norm(H1), 1e-1)
                if max(e1, e2, e3) > 1e-2:
                    bad.append((sid, c, l, i, float(e1), float(e2), float(e3)))
    assert n >= 250 and len(bad) <= n // 100, bad


def test_h2mc_chain_parity_diffuse():
    """MLTInit + 30 lock-step H2MC mutations of 256 chains, Lambertian torus: same PCG streams, same Jacobi solver on both sides.
    The Hessian is ill-conditioned input to an eigen-solve, so a last-bit difference of the device libm can move an acceptance
    test: 1 % on the accept count, film 5 %."""
    r = gc.run_pair(160, 120, 40000, 256, 8, 400, 30, use_gradient=1, opts={"h2mc": 1, "largestepprob": 0.2, "perturbstddev": 0.01}, oracle_grad="reference")
    assert r["contribs_gpu"] == r["contribs_oracle"] and r["init_cl_match"] == 1.0
    so, sg = r["stats_oracle"], r["stats_gpu"]
    assert sg["steps"] == so["steps"] == 256 * 30
    assert abs(sg["largeSteps"] - so["largeSteps"]) <= 0.005 * so["largeSteps"] + 2  # the oracle's Hessians now come from the reference's programs (1e-2 agreement, not bit equality)
    assert abs(sg["accepted"] - so["accepted"]) <= 0.01 * so["accepted"] + 2
    assert abs(sg["gradCalls"] - so["gradCalls"]) <= 0.01 * so["gradCalls"] + 2 and sg["gradCalls"] > 256 * 10
    assert r["film_rel_l2"] < 0.3  # 0.16 measured: a handful of the 256 chains part ways within 30 steps (see the docstring)
    assert r["final_state_match"] > 0.85
    assert r["nonfinite_gpu"] == 0 and abs(r["energy_gpu"] - 1.0) < 1e-4


def test_h2mc_chain_parity_full_materials():
    r = gc.run_pair(160, 120, 20000, 256, 20000, 400, 30, use_gradient=1, max_depth=8, force_diffuse=0,
                    opts={"h2mc": 1, "largestepprob": 0.2, "perturbstddev": 0.01}, oracle_grad="reference")
    so, sg = r["stats_oracle"], r["stats_gpu"]
    assert sg["steps"] == so["steps"] == 256 * 30
    assert abs(sg["accepted"] - so["accepted"]) <= 0.03 * so["accepted"]
    assert abs(sg["gradCalls"] - so["gradCalls"]) <= 0.03 * so["gradCalls"]
    assert r["film_rel_l2"] < 0.5
    assert r["final_state_match"] > 0.75
    assert r["nonfinite_gpu"] == 0 and abs(r["energy_gpu"] - 1.0) < 1e-4


def test_h2mc_door_render_matches_reference_image():
    """scenes/veachdoor/h2mc.xml as shipped (largestepprob 0.2, sigma 0.01) at 320x180 with 2048 chains x 900 mutations (32 spp, half
    the shipped budget: every mutation costs two Hessians) against the reference authors' H2MC render: image mean 5 %, 3x4 region
    grid 25 % (32 spp of a scene whose two shipped renders differ by relMSE 0.03 at 65 / 105 spp).  (H2MC chains mix faster than
    LMC ones; the door scene has no strongly peaked glass transport in most regions.)"""
    p = gc.pkg()
    ref = np.load(os.path.join(gc.ROOT, "tests", "golden", "veachdoor_ref_images_320x180.npz"))["h2mc"]
    lum = lambda x: x @ np.array([0.212671, 0.715160, 0.072169])
    lr = lum(ref)
    W, H, dspp, spp, chains = 320, 180, 32, 32, 2048
    ren = p.Renderer(DOOR_H2, width=W, height=H, seed_offset=0)
    assert ren.get_option("h2mc") == 1
    direct = ren.direct_lighting(dspp)
    per = spp * W * H // chains
    ren.init_chains(300000, chains, 8192, per, per % chains)
    ren.step(per + 1)
    lg = lum(direct / dspp + ren.film() / spp)
    st = ren.stats()
    ren.close()
    assert st["gradCalls"] > 0.2 * st["steps"]  # Hessian evaluations: proposals of the 70 % H2MC steps that survive the re-trace, plus fresh current states
    assert np.isfinite(lg).all()
    assert abs(lg.mean() / lr.mean() - 1) < 0.05
    for gy in range(3):
        for gx in range(4):
            a, b = lg[gy * 60:(gy + 1) * 60, gx * 80:(gx + 1) * 80].mean(), lr[gy * 60:(gy + 1) * 60, gx * 80:(gx + 1) * 80].mean()
            assert abs(a / b - 1) < 0.25, (gy, gx, a / b)
