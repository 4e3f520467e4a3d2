"""CPU tier: the C-ABI library loads and exports every symbol include/lmc_abi.h declares (no compute calls),
the scene front end, the oracle's self-consistency, and the product's path program (host instantiation)
against the reference's generated programs."""
import ctypes
import importlib
import os
import re
import subprocess

import numpy as np
import pytest

from tests import _orc
from tests import gpu_checks as gc
from tests._orc import P

ROOT = gc.ROOT


def _product_lib():
    p = gc.pkg()
    if not os.path.exists(p.LIB_PATH):
        pytest.skip("liblmc_hip.so not built (run `python __graft_entry__.py`)")
    return ctypes.CDLL(p.LIB_PATH)


def test_abi_exports_every_declared_symbol():
    lib = _product_lib()
    hdr = open(os.path.join(ROOT, "include", "lmc_abi.h")).read()
    names = set(re.findall(r"\b(lmc_[a-z_0-9]+)\s*\(", hdr))
    names.discard("lmc_ctx")
    assert len(names) >= 18
    for n in sorted(names):
        assert hasattr(lib, n), "missing export " + n
    # the reference's plugin symbols: 42 forward + 42 derivative programs (path.cpp:3955-3959)
    cnt = 0
    for c in range(1, 10):
        for l in range(0, 9):
            if 3 <= c + l <= 9:
                for suffix in ("static", "static_derv"):
                    assert hasattr(lib, "evaluate_path_bidir_mala_%d_%d_%s" % (c, l, suffix))
                    cnt += 1
    assert cnt == 84


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: on a box without a HIP device lmc_create must fail with a clear message."""
    try:
        import torch

        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    p = gc.pkg()
    if not os.path.exists(p.LIB_PATH):
        pytest.skip("liblmc_hip.so not built")
    with pytest.raises(RuntimeError, match="HIP|device"):
        p.Renderer(gc.TORUS, force_diffuse=1, max_depth=6)


def test_bench_refuses_a_job_it_cannot_place():
    """`bench.py --gpus N` must never measure a smaller job under that name (VERDICT r3 missing item 2): without a launcher it starts one
    process per GPU itself and exits 2 when fewer than N devices are visible (so does --in-process); under a launcher WORLD_SIZE must equal --gpus."""
    import subprocess
    import sys

    p = gc.pkg()
    if not os.path.exists(p.LIB_PATH):
        pytest.skip("liblmc_hip.so not built")
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LMC_BENCH_OVERSUBSCRIBE", "LMC_BENCH_FORCE_DIST", "LMC_BENCH_DRY_RUN", "LMC_BENCH_BOOT")}
    if p.device_count() < 2:
        for extra in ([], ["--in-process"]):
            r = subprocess.run([sys.executable, bench, "--gpus", "2", "--chains", "4096", "--steps", "2", "--warmup", "1"] + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
            assert r.returncode == 2 and "HIP device(s) visible" in r.stderr and r.stdout.strip() == "", (r.returncode, r.stderr[-400:])
    r = subprocess.run([sys.executable, bench, "--gpus", "8"], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 2 and "launcher started 2 rank(s)" in r.stderr and r.stdout.strip() == "", (r.returncode, r.stderr[-400:])
    # a number measured with a work-skipping measurement switch set must not come out of bench.py as a benchmark line
    r = subprocess.run([sys.executable, bench, "--steps", "2"], env=dict(env, LMC_EXP_NOSPLAT="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 2 and "work-skipping" in r.stderr and r.stdout.strip() == "", (r.returncode, r.stderr[-400:])


def test_bench_starts_one_process_per_gpu_and_hands_the_rccl_id_over():
    """plain `python bench.py --gpus N` is an RCCL job of N processes.  The launch path without a GPU (LMC_BENCH_DRY_RUN: the
    workers stop after the hand-off): N workers with RANK / LOCAL_RANK / WORLD_SIZE, rank 0's 128 bytes reach every rank unchanged, once per job
    (the line's two workloads are two jobs with their own communicators), over the private directory of the spawned form and over the launcher's
    gloo rendezvous of the torch.distributed.run form.  The communicator itself, its collectives and the barriers live in the library
    (lmc_comm_init / lmc_film_allreduce / lmc_comm_allreduce_f64) and need GPUs: tests/test_gpu_cli.py."""
    import json
    import subprocess
    import sys

    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LMC_BENCH_OVERSUBSCRIBE", "LMC_BENCH_FORCE_DIST", "LMC_BENCH_BOOT")}
    env["LMC_BENCH_DRY_RUN"] = "1"
    r = subprocess.run([sys.executable, bench, "--gpus", "4"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["dry_run"] and d["n_gpus"] == 4 and d["boot"] == "file" and d["ids_equal"] and d["ids_distinct_per_job"]
    assert [x["rank"] for x in d["ranks"]] == [0, 1, 2, 3] and [x["local"] for x in d["ranks"]] == [0, 1, 2, 3] and all(x["world"] == 4 and x["lens"] == [128, 128] for x in d["ranks"])
    # weak scaling (the driver's form): --chains per GPU, contiguous global id ranges; --scaling strong: --chains is the job's total
    assert all(x["scaling"] == "weak" and x["chains_per_gpu"] == 1 << 20 and x["chain_range"] == [x["rank"] << 20, (x["rank"] + 1) << 20] and x["chains_total"] == 4 << 20 for x in d["ranks"])
    r = subprocess.run([sys.executable, bench, "--gpus", "4", "--scaling", "strong"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert all(x["scaling"] == "strong" and x["chains_per_gpu"] == 1 << 18 and x["chain_range"] == [x["rank"] << 18, (x["rank"] + 1) << 18] and x["chains_total"] == 1 << 20 for x in d["ranks"])
    r = subprocess.run([sys.executable, bench, "--gpus", "3", "--scaling", "strong"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "multiple of --gpus" in r.stderr
    # a rank that dies takes the job down with its exit code instead of leaving the others in a collective
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--bogus-flag"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533", bench, "--gpus", "2"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["dry_run"] and d["n_gpus"] == 2 and d["boot"].startswith("torch.distributed") and d["ids_equal"] and d["ids_distinct_per_job"]


def test_scene_front_end(oracle):
    o = _orc.Oracle(oracle, gc.TORUS, 1, 6, 0, 0, 0, "")
    assert (o.width, o.height, o.num_tris, o.max_depth, o.num_lights) == (1024, 768, 23614, 6, 1)
    sp = o.scene_params()
    assert sp[0] == 0 and sp[32] == 1024 * 768 and abs(sp[33] - 1642.72) < 0.01  # pixel count, camera dist
    assert np.allclose(sp[18:21], [-24.173, -38.184, 30.0076], atol=1e-4)  # camera origin (lookat)
    assert abs(np.linalg.norm(sp[24:28]) - 1) < 1e-5  # unit quaternion
    o.close()
    with pytest.raises(RuntimeError):
        _orc.Oracle(oracle, os.path.join(ROOT, "scenes", "torus", "nope.xml"))
    # shipped materials (phong + bitmap texture, rough dielectric): loads and runs
    f = _orc.Oracle(oracle, gc.TORUS, 0, 6, 64, 48, 0, "")
    norm, nc = f.init(4000, 32, 4)
    assert norm > 0 and nc > 32
    f.close()


def test_oracle_energy_and_determinism(oracle):
    a = _orc.Oracle(oracle, gc.TORUS, 1, 6, 64, 48, 0, "")
    norm, nc = a.init(8000, 128, 4)
    a.setup_chains(200, 0)
    a.step(25)
    st = a.stats()
    f = a.film()
    assert np.isfinite(f).all()
    # every step deposits luminance `normalization` in total (mlt.cpp:103-112, mutation_*.h toSplat)
    assert gc.lum(f).sum() == pytest.approx(norm * st["weightSum"], rel=2e-5)
    assert 0.3 < st["accepted"] / st["steps"] < 0.98
    b = _orc.Oracle(oracle, gc.TORUS, 1, 6, 64, 48, 0, "")
    b.init(8000, 128, 4)
    b.setup_chains(200, 0)
    b.step(25)
    assert np.array_equal(b.film(), f)
    a.close(), b.close()


def test_oracle_scalar_path_matches_reference_forward_program(oracle, pathref_path):
    """log(ssScore) of the oracle's scalar sampler == the reference's generated forward program on the oracle's
    Serialize output (SURVEY.md §8c identity).  Env-hit paths whose texel column wrapped (col >= W/2, i.e. atan2 < 0)
    are excluded: there the reference's AD program itself disagrees with its scalar code (frozen texel, envlight.cpp:348-377)."""
    o = _orc.Oracle(oracle, gc.TORUS, 1, 6, 160, 120, 0, pathref_path)
    o.init(60000, 512, 8)
    s = o.summary(1)
    checked = 0
    for i in range(512):
        c, l, prim, vert = o.serialize_init_state(i)
        if l == 0:
            col = vert[3 + 59 * (c - 2) + 46 + 35]
            if col >= 256:
                continue
        ll, g = o.ref_eval(c, l, prim, vert)
        assert abs(ll - np.log(s[i, 4])) < 2e-4, (i, c, l)
        assert np.isfinite(g).all()
        checked += 1
    assert checked > 300
    o.close()


def _host_pathfunc():
    return ctypes.CDLL(gc.host_pathfunc_lib())


def test_product_path_program_matches_reference_programs(oracle, pathref_path):
    """The product's generic path program (device/pathfunc.h, instantiated on the host by a test helper) against the
    reference's own generated forward + derivative programs, technique by technique.  Tolerances: logLum 5e-4 abs,
    gradient 1e-2 relative L2 (the generated code truncates constants to 6 decimals)."""
    H = _host_pathfunc()
    o = _orc.Oracle(oracle, gc.TORUS, 1, 6, 160, 120, 0, pathref_path)
    o.init(60000, 768, 8)
    sp = o.scene_params()
    seen = set()
    for i in range(768):
        c, l, prim, vert = o.serialize_init_state(i)
        ll, g = o.ref_eval(c, l, prim, vert)
        ll2 = np.zeros(1, np.float32)
        g2 = np.zeros(16, np.float32)
        H.lmc_test_pathfunc_host(c, l, P(prim), P(sp), P(vert), P(ll2), P(g2))
        dim = 2 * (c + l - 1)
        if not np.isfinite(ll):  # wrapped env texel (see the identity test): the AD program's radiance goes negative, log -> NaN on both sides
            assert not np.isfinite(ll2[0])
            continue
        assert abs(ll - ll2[0]) < 5e-4
        assert np.linalg.norm(g - g2[:dim]) <= 1e-2 * max(np.linalg.norm(g), 1e-2)
        seen.add((c, l))
    assert {(3, 1), (4, 0), (4, 1), (5, 0)} <= seen
    o.close()


def test_full_material_identity_and_path_program(oracle, pathref_path):
    """Shipped torus materials (Phong incl. the checker bitmap, rough dielectric), maxdepth 8:
    (1) log(ssScore) of the oracle's scalar sampler == the reference's forward programs on the oracle's Serialize output
        (3e-3: fastpow texture gamma, 6-decimal constants, up to 8 vertices);
    (2) the product's path program == the reference's forward programs (2e-3) and the gradients agree with the reference's
        derivative programs within 1e-2 on >= 99.5 % of the states (round 3: the pass-through sites of chad's reverse emitter --
        fabs operands, CoordinateSystem's doubled output, the two-sided normal -- are all reproduced, pathfunc.h; measured
        859 / 859 on this set, the slack is for ill-conditioned states only)."""
    H = _host_pathfunc()
    o = _orc.Oracle(oracle, gc.TORUS, 0, 8, 160, 120, 0, pathref_path)
    o.init(60000, 768, 8)
    s = o.summary(1)
    sp = o.scene_params()
    n = ok = n_ill = 0
    kinds = set()
    for i in range(768):
        c, l, prim, vert = o.serialize_init_state(i)
        ll, g = o.ref_eval(c, l, prim, vert)
        if not np.isfinite(ll) or not np.isfinite(g).all():
            continue
        if l == 0 and vert[3 + 59 * (c - 2) + 46 + 35] >= 256:  # wrapped env texel, see the identity test above
            continue
        ll2 = np.zeros(1, np.float32)
        g2 = np.zeros(16, np.float32)
        H.lmc_test_pathfunc_host(c, l, P(prim), P(sp), P(vert), P(ll2), P(g2))
        dev = max(abs(ll - np.log(s[i, 4])) / 3e-3, abs(ll - ll2[0]) / 2e-3)
        if dev >= 1.0:
            # a state may exceed the bars only if it is ill-conditioned: the reference's OWN program must move by more than the deviation when its
            # primary sample moves by 1e-6 (round 6, after the switch to dtrig.h re-drew the init states: state 510, a (8,1) path through five rough
            # dielectric interfaces, d logLum / d pss ~ 1e5 -- reference -3.2273, oracle -3.2144, product -3.2158 with dtrig.h AND with libm)
            prng = np.random.default_rng(i)
            spread = max(abs(o.ref_eval(c, l, prim + (prng.uniform(-1, 1, len(prim)) * 1e-6).astype(np.float32), vert)[0] - ll) for _ in range(4))
            assert spread > max(abs(ll - np.log(s[i, 4])), abs(ll - ll2[0])), (i, c, l, float(ll), float(np.log(s[i, 4])), float(ll2[0]), float(spread))
            n_ill += 1
            continue
        dim = 2 * (c + l - 1)
        n += 1
        ok += np.linalg.norm(g - g2[:dim]) <= 1e-2 * max(np.linalg.norm(g), 1e-2)
        for k in range(c - 2):
            kinds.add(int(vert[3 + 59 * k + 48]))
    assert n > 400 and n - ok <= n // 200 and n_ill <= 3, (ok, n, n_ill)
    assert kinds == {0, 1, 2}
    o.close()


def test_golden_full_material_derivative_vectors():
    """Committed vectors of the reference's MALA-gradient and H2MC gradient + Hessian programs on full-material states of the
    torus AND the veach-door scene, every technique the two scenes produce up to c + l = 9 (dims 14 / 16 included)
    (tests/golden/derv_vectors_full.npz, made by tests/golden/make_golden_derv.py from oracle/_ref): holds without
    /root/reference.  1e-2 relative (L2 / Frobenius) for >= 99 % of the vectors, logLum 5e-3."""
    H = _host_pathfunc()
    z = np.load(os.path.join(ROOT, "tests", "golden", "derv_vectors_full.npz"))
    n = len(z["c"])
    assert n >= 250 and {int(c) + int(l) for c, l in zip(z["c"], z["l"])} >= {4, 5, 6, 7, 8, 9}
    bad_g, bad_h = [], []
    for i in range(n):
        c, l = int(z["c"][i]), int(z["l"][i])
        dim = 2 * (c + l - 1)
        sp = z["scenes"][int(z["scene_id"][i])].copy()
        prim, vert = z["primary"][i].copy(), z["vert"][i].copy()
        ll, g = np.zeros(1, np.float32), np.zeros(16, np.float32)
        H.lmc_test_pathfunc_host(c, l, P(prim), P(sp), P(vert), P(ll), P(g))
        assert abs(ll[0] - z["loglum"][i]) < 5e-3, (i, c, l)
        rg = z["mala_grad"][i][:dim]
        if np.linalg.norm(rg - g[:dim]) > 1e-2 * max(np.linalg.norm(rg), 1e-2):
            bad_g.append((i, c, l))
        g2, h2 = np.zeros(16, np.float32), np.zeros(256, np.float32)
        H.lmc_test_pathfunc_hess_host(c, l, P(prim), P(sp), P(vert), P(ll), P(g2), P(h2))
        H1, H2 = z["h2_hess"][i][: dim * dim].reshape(dim, dim), h2[: dim * dim].reshape(dim, dim)
        eg = np.linalg.norm(z["h2_grad"][i][:dim] - g2[:dim]) / max(np.linalg.norm(z["h2_grad"][i][:dim]), 1e-2)
        eh = np.linalg.norm(H1 - H2) / max(np.linalg.norm(H1), 1e-1)
        if eg > 1e-2 or eh > 1e-2:
            bad_h.append((i, c, l, float(eg), float(eh)))
    assert len(bad_g) <= n // 100, bad_g
    assert len(bad_h) <= n // 100, bad_h


def test_golden_gradient_vectors(oracle):
    """Committed (input, output) vectors of the reference's generated programs (tests/golden/grad_vectors.npz, made by
    tests/golden/make_golden.py): holds without /root/reference and without oracle/_ref."""
    H = _host_pathfunc()
    z = np.load(os.path.join(ROOT, "tests", "golden", "grad_vectors.npz"))
    n = len(z["c"])
    assert n >= 64
    for i in range(n):
        c, l = int(z["c"][i]), int(z["l"][i])
        ll2 = np.zeros(1, np.float32)
        g2 = np.zeros(16, np.float32)
        prim, vert = z["primary"][i].copy(), z["vert"][i].copy()
        H.lmc_test_pathfunc_host(c, l, P(prim), P(z["scene"].copy()), P(vert), P(ll2), P(g2))
        dim = 2 * (c + l - 1)
        if not np.isfinite(z["loglum"][i]):
            assert not np.isfinite(ll2[0])
            continue
        assert abs(z["loglum"][i] - ll2[0]) < 5e-4
        assert np.linalg.norm(z["grad"][i][:dim] - g2[:dim]) <= 1e-2 * max(np.linalg.norm(z["grad"][i][:dim]), 1e-2)
    # the survey's hand-built known answer for (c,l) = (2,1)  
    k = z["ka_primary"], z["ka_scene"], z["ka_vert"]
    ll2 = np.zeros(1, np.float32)
    g2 = np.zeros(16, np.float32)
    H.lmc_test_pathfunc_host(2, 1, P(k[0].copy()), P(k[1].copy()), P(k[2].copy()), P(ll2), P(g2))
    assert abs(ll2[0] - (-2.53792)) < 1e-4
    assert np.allclose(g2[:4], [0.513756, 1.54127, 0, 0], atol=2e-4)


def test_cache_query_existence_test_is_exact():
    """The lean kernel runs the kd-tree search only when a grid over the first three coordinates finds a cache point within the
    query radius (dchain.h, dsmall.h).  Exactness in BOTH directions against the reference's radius search (restated nanoflann,
    the oracle): "found something" <=> "the search returns at least one match", on clustered clouds with queries ON, NEAR,
    just inside the radius of, and far from the points, for the smallest and the largest radius."""
    lib = _product_lib()
    L = gc.oracle_lib()
    rng = np.random.default_rng(4)
    for dim in (6, 8, 12):
        pts = rng.random((3000, dim)).astype(np.float32)
        pts[:600] = (pts[rng.integers(600, 3000, 600)] + rng.normal(0, 0.01, (600, dim))).astype(np.float32)  # clusters
        pts = np.clip(pts, 0, np.nextafter(np.float32(1), np.float32(0)))
        r2 = np.float32(dim * 0.01 * 0.01)
        q_near = (pts[rng.integers(0, 3000, 4000)] + rng.normal(0, 0.012, (4000, dim))).astype(np.float32)
        q_edge = pts[rng.integers(0, 3000, 2000)].copy()
        q_edge[:, :3] += (rng.choice([-1, 1], (2000, 3)) * np.sqrt(r2 / 3) * rng.uniform(0.98, 1.02, (2000, 1))).astype(np.float32)
        q_far = rng.random((4000, dim)).astype(np.float32)
        q = np.clip(np.concatenate([q_near, q_edge, q_far]), 0, np.nextafter(np.float32(1), np.float32(0))).astype(np.float32)
        n = np.zeros(len(q), np.int32)
        idx = np.zeros((len(q), 5), np.int32)
        dist = np.zeros((len(q), 5), np.float32)
        L.orc_kd_query(dim, 3000, P(pts), len(q), P(q), ctypes.c_float(r2), 5, P(n), P(idx), P(dist))
        found = np.zeros(len(q), np.int32)
        assert lib.lmc_cache_filter_probe(dim, 3000, P(pts), len(q), P(q), P(found)) == 0
        assert 500 < (n > 0).sum() < len(q) - 500  # both answers occur
        assert np.array_equal(found == 1, n > 0), dim


def test_bvh_builders_agree_and_stack_bound_holds(tmp_path):
    """tests/helpers/bvh_stats.cpp: Morton tree, binned-SAH tree and the four-wide tree the renderer uploads
    (accel.cpp CollapseToBvh4) return the same (triangle id, t) for every ray of a camera-rays-plus-bounces workload on the
    torus scene (the closest hit is defined tree-independently: smallest t, ties to the lower id), and the wide traversal
    never holds more pending entries than the bound the host sizes the stack with."""
    import json, subprocess

    build = os.path.join(ROOT, "langevin-mcmc_amd", "csrc", "_build")
    objs = [os.path.join(build, o) for o in ("accel.o", "scene.o", "imageio.o", "jpeg.o")]
    if not all(os.path.exists(o) for o in objs):
        pytest.skip("product objects not built")
    exe = str(tmp_path / "bvh_stats")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off", "-x", "hip", os.path.join(ROOT, "tests", "helpers", "bvh_stats.cpp"),
                    "-x", "none"] + objs + ["-lz", "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe, os.path.join(ROOT, "scenes", "torus", "lmc.xml"), "30000"], capture_output=True, text=True)
    rows = [json.loads(l) for l in r.stdout.splitlines()]
    assert r.returncode == 0, r.stdout
    assert rows[-1] == {"mismatches": 0}
    wide = rows[2]
    assert wide["tree"] == "sah_4wide" and wide["max_stack"] <= wide["stack_bound"] <= 32
    # the 64-byte nodes with 8-bit child boxes (build option LMC_BVH_QUANT, dscene.h BvhNode4Q): walked with the device's arithmetic they return
    # the same (triangle id, t) -- counted in `mismatches` above -- at the price of a few more visits
    quant = rows[3]
    assert quant["tree"] == "sah_4wide_quantised" and quant["max_stack"] <= wide["stack_bound"]
    assert quant["node_visits_per_ray"] < 1.25 * wide["node_visits_per_ray"]
    assert wide["node_visits_per_ray"] < 0.65 * rows[1]["node_visits_per_ray"]


def test_deterministic_transcendentals_accuracy_and_conventions():
    """device/dtrans.h (lexpf / llogf / lpowf: float-only, bit-reproducible across compilers) against float64: exp and log within
    1 ulp over the whole float range, pow within 1.5 ulp where the result exceeds 1e-10 (the Phong lobe's own cut-off, phong.cpp:44)
    and within 5 ulp down to the subnormal range; libm's conventions for the special arguments the BSDF code can produce."""
    lib = ctypes.CDLL(gc.host_trans_lib())

    def ulps(got, ref64):
        ulp = np.spacing(np.abs(ref64.astype(np.float32))).astype(np.float64)
        return np.abs(got.astype(np.float64) - ref64) / ulp

    for mode, x, y in gc.trans_cases():
        o = np.zeros(len(x), np.float32)
        lib.lmc_test_trans_host(len(x), mode, P(x), P(y), P(o))
        x64, y64 = x.astype(np.float64), y.astype(np.float64)
        with np.errstate(all="ignore"):
            ref = np.exp(x64) if mode == 0 else np.log(x64) if mode == 1 else np.power(x64, y64)
        normal = (np.abs(ref) > 1.2e-38) & (np.abs(ref) < 3.4e38)
        e = ulps(o[normal], ref[normal])
        if mode < 2:
            assert e.max() <= 1.05, (mode, e.max())  # 1.008 measured (exp, near the underflow threshold)
        else:
            big = np.abs(ref[normal]) > 1e-10
            assert e[big].max() <= 1.5 and e.max() <= 5.0, (e[big].max(), e.max())
        assert (o[normal] == ref[normal].astype(np.float32)).mean() > 0.85  # mostly correctly rounded
    a = np.array([-2, -2, -2, 0, 0, 1, np.inf, 2, -0.9, 0.5, np.nan, 3], np.float32)
    b = np.array([3, 2, 0.5, 2, -1, 5, 2, 0, 100, -2000, 1, np.nan], np.float32)
    o = np.zeros(len(a), np.float32)
    lib.lmc_test_trans_host(len(a), 2, P(a), P(b), P(o))
    with np.errstate(all="ignore"):
        ref = np.power(a.astype(np.float64), b.astype(np.float64)).astype(np.float32)
    assert np.array_equal(np.isnan(o), np.isnan(ref)) and np.allclose(o[~np.isnan(o)], ref[~np.isnan(ref)], rtol=2e-7)
    sp = np.array([0, -1, np.inf, np.nan, 1e-45], np.float32)
    o = np.zeros(len(sp), np.float32)
    lib.lmc_test_trans_host(len(sp), 1, P(sp), P(sp), P(o))
    assert o[0] == -np.inf and np.isnan(o[1]) and o[2] == np.inf and np.isnan(o[3]) and abs(o[4] - np.log(1.4e-45)) < 1e-3
    ex = np.array([89, -104, np.nan, 0, -90], np.float32)
    lib.lmc_test_trans_host(len(ex), 0, P(ex), P(ex), P(o))
    assert o[0] == np.inf and o[1] == 0 and np.isnan(o[2]) and o[3] == 1 and 0 < o[4] < 1e-38


def test_deterministic_trigonometry_accuracy_and_conventions():
    """device/dtrig.h (dsinf / dcosf / dacosf / datan2f: float only, explicit fma, bit-reproducible across compilers) against float64
    on the arguments the sampling code produces and on wider ranges: sin / cos within 1.5 ulp (every float up to 2.9e9 was swept when the
    routines were written: 1.49), acos within 1 ulp (0.89 over all of [-1, 1]), atan2 within 1 ulp (0.68 over 6.4e8 pairs); libm's conventions
    for the special arguments."""
    lib = ctypes.CDLL(gc.host_trans_lib())

    def ulps(got, ref64):
        ulp = np.spacing(np.abs(ref64.astype(np.float32))).astype(np.float64)
        return np.abs(got.astype(np.float64) - ref64) / ulp

    bars = {3: 1.5, 4: 1.5, 5: 1.0, 6: 1.0}
    for mode, x, y in gc.trig_cases():
        if mode not in bars:
            continue
        o = np.zeros(len(x), np.float32)
        lib.lmc_test_trans_host(len(x), mode, P(x), P(y), P(o))
        x64, y64 = x.astype(np.float64), y.astype(np.float64)
        with np.errstate(all="ignore"):
            ref = np.sin(x64) if mode == 3 else np.cos(x64) if mode == 4 else np.arccos(x64) if mode == 5 else np.arctan2(x64, y64)
        ok = np.isfinite(ref) & (np.abs(ref) > 1.2e-38)
        e = ulps(o[ok], ref[ok])
        assert e.max() <= bars[mode], (mode, float(e.max()), float(x[ok][np.argmax(e)]))
        assert (o[ok] == ref[ok].astype(np.float32)).mean() > 0.8  # mostly correctly rounded
    # conventions: atan2's signed zeros / quadrants / infinities, acos at +-1 and outside, sin / cos of 0 and of non-finite arguments
    ay = np.array([0.0, 0.0, -0.0, 1.0, -1.0, 1.0, 1.0, -1.0, 1e-30, -1e-30, 1.0, 1.0, np.inf, 1.0, 1.0, np.inf, -np.inf, np.nan], np.float32)
    ax = np.array([1.0, -1.0, -1.0, 0.0, 0.0, -0.0, 1.0, -1.0, -1.0, -1.0, 1e30, -1e30, 1.0, np.inf, -np.inf, np.inf, -np.inf, 1.0], np.float32)
    o = np.zeros(len(ax), np.float32)
    lib.lmc_test_trans_host(len(ax), 6, P(ay), P(ax), P(o))
    ref = np.arctan2(ay.astype(np.float64), ax.astype(np.float64)).astype(np.float32)
    assert np.array_equal(np.isnan(o), np.isnan(ref)) and np.array_equal(o[~np.isnan(o)], ref[~np.isnan(ref)]) and np.array_equal(np.signbit(o[:3]), np.signbit(ref[:3]))
    sp = np.array([1.0, -1.0, 0.0, 1.5, -1.5, np.nan, 1e-30], np.float32)
    o = np.zeros(len(sp), np.float32)
    lib.lmc_test_trans_host(len(sp), 5, P(sp), P(sp), P(o))
    assert o[0] == 0 and o[1] == np.float32(np.pi) and o[2] == np.float32(np.pi / 2) and np.isnan(o[3:6]).all() and o[6] == np.float32(np.pi / 2)
    z = np.array([0.0, -0.0, np.inf, np.nan], np.float32)
    o = np.zeros(4, np.float32)
    lib.lmc_test_trans_host(4, 3, P(z), P(z), P(o))
    assert o[0] == 0 and o[1] == 0 and np.signbit(o[1]) and np.isnan(o[2:]).all()
    lib.lmc_test_trans_host(4, 4, P(z), P(z), P(o))
    assert o[0] == 1 and o[1] == 1 and np.isnan(o[2:]).all()


def test_restated_glibc_logf_is_the_hosts_logf_on_the_whole_polar_domain():
    """drng.h GlibcLogf restates glibc's logf (the `std::log(r2)` of libstdc++'s normal_distribution<float>, which the oracle calls through
    std::normal_distribution's own arithmetic in oracle/rng.h) so that the device's normal variates are the reference's bit for bit
    (the device libm's logf agreed on 84.6 % of them).  Pinned against the real thing: EVERY positive float up to 1.0 --
    1 065 353 215 arguments, the whole domain of the polar method's `r2`, subnormals included -- must give the bits of this host's logf; plus
    2^22 arguments over the rest of the float range and the special values."""
    lib = ctypes.CDLL(gc.host_trans_lib())
    lib.lmc_test_logf_exhaustive.restype = ctypes.c_ulonglong
    lib.lmc_test_logf_exhaustive.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_int, ctypes.c_void_p]
    bad_at = ctypes.c_float(0)
    n_bad = lib.lmc_test_logf_exhaustive(1, 0x3F800000, max(1, min(16, len(os.sched_getaffinity(0)))), ctypes.byref(bad_at))
    assert n_bad == 0, (n_bad, bad_at.value)
    rng = np.random.default_rng(5)
    x = np.exp(rng.uniform(0, np.log(3e38), 1 << 22)).astype(np.float32)
    a, b = np.zeros(len(x), np.float32), np.zeros(len(x), np.float32)
    lib.lmc_test_trans_host(len(x), 7, P(x), P(x), P(a))
    lib.lmc_test_trans_host(len(x), 8, P(x), P(x), P(b))
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    sp = np.array([0.0, -0.0, np.inf, -1.0, np.nan, 1.0], np.float32)
    a, b = np.zeros(len(sp), np.float32), np.zeros(len(sp), np.float32)
    with np.errstate(all="ignore"):
        lib.lmc_test_trans_host(len(sp), 7, P(sp), P(sp), P(a))
        lib.lmc_test_trans_host(len(sp), 8, P(sp), P(sp), P(b))
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])


def test_golden_light_coordinate_vectors():
    """`uselightcoordinatesampling`: committed vectors of the reference's programs with scene[0] = 1 on veach-door states
    whose camera path ends on the area light (tests/golden/derv_vectors_lightcoord.npz, techniques (4..9, 0)): the product's path
    program takes the doLightCoordinateSampling branch (pathfunc.h BSDFSamplingT; path.cpp:2979-3025) -- value 2e-3, MALA gradient
    and H2MC gradient + Hessian 1e-2.  Also recorded there: the scalar sampler's ssScore of the same states.  The reference's scalar
    GeneratePathBidir multiplies by SamplePdf() where its own derivative program divides (path.cpp:1359 vs :3013), so
    logLum - log(ssScore) is not 0 but the same constant, 2 log(light area), on every state -- a quirk the oracle keeps."""
    H = _host_pathfunc()
    z = np.load(os.path.join(ROOT, "tests", "golden", "derv_vectors_lightcoord.npz"))
    n = len(z["c"])
    assert n >= 10 and z["scene"][0] == 1.0
    sp = z["scene"].copy()
    off = []
    for i in range(n):
        c, l = int(z["c"][i]), int(z["l"][i])
        dim = 2 * (c + l - 1)
        prim, vert = z["primary"][i].copy(), z["vert"][i].copy()
        ll, g = np.zeros(1, np.float32), np.zeros(16, np.float32)
        H.lmc_test_pathfunc_host(c, l, P(prim), P(sp), P(vert), P(ll), P(g))
        assert abs(ll[0] - z["loglum"][i]) < 2e-3, (i, c)
        rg = z["mala_grad"][i][:dim]
        assert np.linalg.norm(rg - g[:dim]) <= 1e-2 * max(np.linalg.norm(rg), 1e-2), (i, c)
        g2, h2 = np.zeros(16, np.float32), np.zeros(256, np.float32)
        H.lmc_test_pathfunc_hess_host(c, l, P(prim), P(sp), P(vert), P(ll), P(g2), P(h2))
        H1, H2 = z["h2_hess"][i][: dim * dim].reshape(dim, dim), h2[: dim * dim].reshape(dim, dim)
        assert np.linalg.norm(z["h2_grad"][i][:dim] - g2[:dim]) <= 1e-2 * max(np.linalg.norm(z["h2_grad"][i][:dim]), 1e-2)
        assert np.linalg.norm(H1 - H2) <= 1e-2 * max(np.linalg.norm(H1), 1e-1), (i, c)
        # with the flag off the same inputs take the BSDF-sampling branch: another function
        sp0 = sp.copy()
        sp0[0] = 0.0
        ll0 = np.zeros(1, np.float32)
        H.lmc_test_pathfunc_host(c, l, P(prim), P(sp0), P(vert), P(ll0), P(g))
        assert not (abs(ll0[0] - ll[0]) < 1e-3)
        off.append(float(z["loglum"][i] - np.log(z["scalar_ss"][i])))
    assert max(off) - min(off) < 5e-3 and min(off) > 10.0, (min(off), max(off))  # 2 log(area of the door scene's emitter) = 18.23


def test_oracle_subpath_generator_agrees_with_bidir_generator_in_expectation():
    """The multiplexed large step's generator (GenerateSubpath, path.cpp:1451-1658: one technique, no Russian roulette) and MLTInit's
    (GeneratePathBidir, :1240-1449) estimate the same MIS-weighted integral per technique (c, l): mean lsScore of 10^5 GenerateSubpath
    samples == what the technique contributes per GeneratePathBidir sample, within 5 sigma, for every technique of the Lambertian
    torus (environment light: only l = 0 and l = 1 carry energy; the others must be empty on both sides).  Pins the restatement
    of GenerateSubpath to the generator that the reference images already validate.  (On scenes/torus/lmc_arealight.xml the two
    DISAGREE beyond length 4, by the reference's design: GeneratePathBidir ends a camera path at the first emitter it hits --
    "Assume lights have zero reflectance", path.cpp:1372 -- GenerateSubpath only looks for the emitter at the last vertex, and that
    scene's emitter is its floor.)"""
    import ctypes
    from tests import gpu_checks as gc, _orc
    from tests._orc import P

    L = gc.oracle_lib()
    orc = _orc.Oracle(L, gc.TORUS, 1, 6, 128, 96, 0, "")
    ninit = 200000
    orc.init(ninit, 64, 64)
    cap = 8 * ninit
    s, cl, ls = np.zeros(cap, np.int64), np.zeros(cap, np.int32), np.zeros(cap, np.float32)
    L.orc_init_contribs.restype = ctypes.c_longlong
    L.orc_init_contribs.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    n = L.orc_init_contribs(orc.h, cap, P(s), P(cl), P(ls))
    L.orc_subpath_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    N = 100000
    checked = 0
    for length in range(3, 7):
        for l in range(0, length + 1):
            c = length - l + 1
            bid = ls[:n][cl[:n] == c * 16 + l].astype(np.float64)
            mean_b = bid.sum() / ninit
            se_b = np.sqrt(max((bid ** 2).sum() / ninit - mean_b ** 2, 0) / ninit)
            sm, sq, cnt = ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
            assert L.orc_subpath_probe(orc.h, c, l, N, 1234 + c * 16 + l, ctypes.byref(sm), ctypes.byref(sq), ctypes.byref(cnt)) == 0
            mean_s = sm.value / N
            se_s = np.sqrt(max(sq.value / N - mean_s ** 2, 0) / N)
            if l >= 2:
                assert len(bid) == 0 and cnt.value == 0, (c, l)
                continue
            assert cnt.value > 500 and abs(mean_s - mean_b) <= 5 * np.hypot(se_b, se_s), (c, l, mean_b, se_b, mean_s, se_s)
            checked += 1
    assert checked == 8
    orc.close()


def test_oracle_large_step_cache_runs_and_keeps_the_image_mean():
    """CPU side of SURVEY §8(f) item 4: the oracle's multiplexed large step and LargeStepCache (`largestepmultiplexed`, `samplecache`;
    mutation_large.h:45-58,87-102, mutation_large_cache.h:22-141) through the cache phase on the Lambertian torus, 8192 chains x
    100 lock-step mutations.  No reference binary can be run here, so what is asserted is what any correct Metropolis-Hastings
    proposal must keep: the estimate of the image mean (film / splat weight) agrees with the plain large step's within 5 %, the dim-6
    cache becomes ready, and once it is the cache-sampling run really takes other trajectories than the multiplexed one."""
    from tests import gpu_checks as gc, _orc

    L = gc.oracle_lib()
    if not gc.pathref():
        pytest.skip("oracle/_ref not built")
    res = {}
    for mode in ((0, 0), (1, 0), (1, 1)):
        orc = _orc.Oracle(L, gc.TORUS, 1, 6, 128, 96, 0, gc.pathref())
        for k, v in (("largestepmultiplexed", mode[0]), ("largestepprob", 0.3), ("largestepscale", 1.0), ("samplecache", mode[1])):
            assert L.orc_set_option(orc.h, k.encode(), float(v)) == 0
        orc.init(100000, 8192, 64)
        orc.setup_chains(120, 0)
        orc.step(100)
        st = orc.stats()
        res[mode] = (orc.film().sum() / st["weightSum"], st)
        orc.close()
    m0 = res[(0, 0)][0]
    for mode in ((1, 0), (1, 1)):
        assert abs(res[mode][0] / m0 - 1) < 0.05, (mode, res[mode][0], m0)
        assert res[mode][1]["cacheReadyMask"] & 64
    assert res[(1, 1)][1]["accepted"] != res[(1, 0)][1]["accepted"]
    # single-technique proposals fail more often than the all-techniques large step: more large steps are needed to leave the invalid start
    assert res[(1, 0)][1]["largeSteps"] > res[(0, 0)][1]["largeSteps"]
