"""GPU-vs-oracle parity checks, shared by the `-m gpu` tests and __graft_entry__.smoke().
Everything numeric on the product side goes through the C ABI (include/lmc_abi.h) of liblmc_hip.so;
the CPU oracle (oracle/) is only the checker."""
import ctypes
import importlib
import os

import numpy as np

from tests import _orc
from tests._orc import P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TORUS = os.path.join(ROOT, "scenes", "torus", "lmc.xml")
ORACLE_SO = os.path.join(ROOT, "oracle", "liblmc_oracle.so")
PATHREF_SO = os.path.join(ROOT, "oracle", "_ref", "libpathref.so")


def pkg():
    return importlib.import_module("langevin-mcmc_amd")


def oracle_lib():
    if not os.path.exists(ORACLE_SO):
        import __graft_entry__ as ge

        ge.build_oracle()
    return _orc.load(ORACLE_SO)


def pathref():
    return PATHREF_SO if os.path.exists(PATHREF_SO) else ""


def tick_seed(user_draw=5):
    MULT, INC, M = 6364136223846793005, 1442695040888963407, 1 << 64
    inv = pow(MULT, -1, M)
    s = 0xABCDEF12 << 32
    for _ in range(66 + user_draw):
        s = ((s - INC) * inv) % M
    return ((s - INC) * inv - INC) % M


# ------------------------------------------------------------------------------------------------ RNG
def rng_probe(seeds, mode, n, mean=0.0, stddev=1.0):
    lib = pkg().lib()
    seeds = np.array(seeds, np.uint64)
    out = np.zeros((len(seeds), n + 66), np.uint32)
    r = lib.lmc_rng_probe(len(seeds), P(seeds), mode, n, ctypes.c_float(mean), ctypes.c_float(stddev), P(out))
    if r != 0:
        raise RuntimeError(lib.lmc_last_error().decode())
    return out


def check_rng(L):
    seeds = [0, 1, 7, 127, (1 << 20) - 1, tick_seed()]
    n = 1024
    raw = rng_probe(seeds, 0, n)
    uni = rng_probe(seeds, 1, n)
    nor = rng_probe(seeds, 2, n - 1, 0.0, 0.01)
    mix = rng_probe(seeds, 3, 9 * 64)
    # the chain kernels never read a stream's extension table before its first tick: they synthesise the entries from the seed (drng.h).  Same
    # streams, same state and table afterwards -- including the seed whose stream ticks inside the window (the table is then materialised)
    for mode, ref, nn in ((0, raw, n), (1, uni, n), (2, nor, n - 1), (3, mix, 9 * 64)):
        assert np.array_equal(rng_probe(seeds, mode | 8, nn, 0.0, 0.01 if mode == 2 else 1.0), ref), "synthesised extension table: mode %d differs" % mode
    res = {}
    for k, s in enumerate(seeds):
        a = np.zeros(n, np.uint32)
        L.orc_pcg_u32(s, n, P(a))
        assert np.array_equal(raw[k, :n], a), "raw PCG stream differs for seed %d" % s
        d = np.zeros(66, np.uint32)
        L.orc_pcg_dump(s, n, P(d))
        assert np.array_equal(raw[k, n:], d), "PCG state/table after %d draws differs for seed %d" % (n, s)
        u = np.zeros(n, np.float32)
        L.orc_pcg_uniform(s, n, P(u))
        assert np.array_equal(uni[k, :n], u.view(np.uint32)), "uniform01 differs for seed %d" % s
        g = np.zeros(n - 1, np.float32)
        L.orc_pcg_normal(s, n - 1, ctypes.c_float(0.0), ctypes.c_float(0.01), P(g))
        gg = nor[k, : n - 1].view(np.float32)
        # bit-equal since round 6: the polar method's logf is glibc's, restated on the device (drng.h GlibcLogf); sqrtf and the division are IEEE on both sides
        assert np.array_equal(gg.view(np.uint32), g.view(np.uint32)), "normal draws differ for seed %d (%d of %d)" % (s, int((gg != g).sum()), len(g))
        res.setdefault("normal_exact_frac", []).append(float((gg == g).mean()))
        m = np.zeros(9 * 64, np.float32)
        L.orc_pcg_mixed(s, 64, 7, P(m))
        assert np.array_equal(mix[k, : 9 * 64], m.view(np.uint32)), "mixed uniform / normal stream differs for seed %d" % s
    res["normal_exact_frac"] = float(np.mean(res["normal_exact_frac"]))
    return res


# ------------------------------------------------------------------------------------------------ rays
def random_rays(rng, n, center, radius):
    org = center + rng.normal(0, 1, (n, 3)) * radius * 0.8
    tgt = center + rng.normal(0, 1, (n, 3)) * radius * 0.3
    d = tgt - org
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = org
    rays[:, 3:6] = d
    rays[:, 6] = 5e-4
    rays[:, 7] = np.inf
    # a third of the rays get a finite tfar (shadow-ray style)
    k = n // 3
    rays[:k, 7] = rng.uniform(0.5, 2.0, k).astype(np.float32) * radius
    return rays


def check_trace(L, orc, ren, n=200000, brute_n=2000, seed=1):
    rng = np.random.default_rng(seed)
    rays = random_rays(rng, n, np.array([0.0, 0.0, 4.0]), 12.0)
    prim, t = ren.trace(rays)
    oprim = np.zeros(n, np.int32)
    ot = np.zeros(n, np.float32)
    L.orc_trace(orc.h, n, P(rays), P(oprim), P(ot))
    hit_frac = float((oprim >= 0).mean())
    mism = int((prim != oprim).sum())
    tm = int(((t != ot) & (prim == oprim)).sum())
    occ = ren.occluded(rays)
    oocc = np.zeros(n, np.int32)
    L.orc_occluded(orc.h, n, P(rays), P(oocc))
    occ_m = int((occ != oocc).sum())
    bp = np.zeros(brute_n, np.int32)
    bt = np.zeros(brute_n, np.float32)
    L.orc_trace_brute(orc.h, brute_n, P(rays[:brute_n]), P(bp), P(bt))
    brute_m = int((prim[:brute_n] != bp).sum())
    return dict(n=n, hit_frac=hit_frac, prim_mismatch=mism, t_mismatch=tm, occ_mismatch=occ_m, brute_mismatch=brute_m)


# ------------------------------------------------------------------------------------------------ gradient
def collect_grad_inputs(orc, max_states):
    """(c,l) -> (primary [n, 2L+1], vert [n, V]) from the oracle's MLT init states."""
    by = {}
    for i in range(max_states):
        r = orc.serialize_init_state(i)
        if r is None:
            continue
        c, l, prim, vert = r
        L_ = max(c + l - 1, 2)
        V = 238 + 59 * (c + l - 3)
        by.setdefault((c, l), ([], []))
        by[(c, l)][0].append(prim[: 2 * L_ + 1].copy())
        by[(c, l)][1].append(vert[:V].copy())
    return {k: (np.array(v[0], np.float32), np.array(v[1], np.float32)) for k, v in by.items()}


def check_grad(orc, inputs, scene38):
    """GPU path program vs the reference's generated programs (oracle/_ref) on identical inputs."""
    p = pkg()
    out = {}
    for (c, l), (prim, vert) in sorted(inputs.items()):
        ll, g = p.grad_batch(c, l, prim.T.copy(), scene38, vert.T.copy())
        e_ll, e_g, nan_ref = [], [], 0
        for i in range(len(prim)):
            r = orc.ref_eval(c, l, prim[i], vert[i])
            if r is None:
                continue
            rll, rg = r
            if not np.isfinite(rg).all() or not np.isfinite(rll):
                nan_ref += 1
                continue
            e_ll.append(abs(rll - ll[i]))
            e_g.append(np.linalg.norm(rg - g[:, i]) / max(np.linalg.norm(rg), 1e-2))
        out[(c, l)] = dict(n=len(prim), max_dloglum=float(np.max(e_ll)) if e_ll else 0.0, max_rel_dgrad=float(np.max(e_g)) if e_g else 0.0,
                           p99_rel_dgrad=float(np.percentile(e_g, 99)) if e_g else 0.0, ref_nonfinite=nan_ref)
    return out


# ------------------------------------------------------------------------------------------------ chains
def lum(film):
    return film.astype(np.float64) @ np.array([0.212671, 0.715160, 0.072169])


def host_pathfunc_lib():
    """Test helper: the product's path program compiled for the host, exporting the plugin symbol names, so that the
    oracle can draw its gradients from the same program the GPU runs (chain-loop parity independent of the AD)."""
    import subprocess

    so = os.path.join(ROOT, "tests", "helpers", "libpathfunc_host.so")
    src = os.path.join(ROOT, "tests", "helpers", "pathfunc_host.cpp")
    hdr = os.path.join(ROOT, "langevin-mcmc_amd", "csrc", "device", "pathfunc.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", src, "-o", so], cwd=ROOT)
    return so


def host_trans_lib():
    """Test helper: device/dtrans.h, dtrig.h and drng.h's GlibcLogf compiled for the host (same flags as the oracle: no contraction; -mfma so that the
    EXPLICIT fused multiply-adds are the instruction -- without it they are libm calls with the same result)."""
    import subprocess

    so = os.path.join(ROOT, "tests", "helpers", "libdtrans_host.so")
    src = os.path.join(ROOT, "tests", "helpers", "dtrans_host.cpp")
    dev = os.path.join(ROOT, "langevin-mcmc_amd", "csrc", "device")
    hdrs = [os.path.join(dev, h) for h in ("dtrans.h", "dtrig.h", "drng.h", "dmath.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma", "-pthread", src, "-o", so], cwd=ROOT)
    return so


def rccl_stub_lib():
    """Test helper: the six RCCL entry points over host shared memory (tests/helpers/rccl_stub.cpp), for multi-PROCESS jobs on one GPU (LMC_RCCL_LIB)"""
    import subprocess

    so = os.path.join(ROOT, "tests", "helpers", "librccl_stub.so")
    src = os.path.join(ROOT, "tests", "helpers", "rccl_stub.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", so, "-lrt"], cwd=ROOT)
    return so


def trans_cases(seed=1, n=1 << 20):
    """(mode, x, y) argument sets of the exp / log / pow checks: the BSDFs' ranges and the full float range"""
    rng = np.random.default_rng(seed)
    f = np.float32
    return [
        (0, rng.uniform(-104, 89, n).astype(f), np.zeros(n, f)),
        (0, rng.uniform(-30, 0, n).astype(f), np.zeros(n, f)),  # Beckmann exponents
        (1, np.exp(rng.uniform(np.log(1e-38), np.log(3e38), n)).astype(f), np.zeros(n, f)),
        (1, rng.uniform(1e-6, 1.0, n).astype(f), np.zeros(n, f)),  # -log(1 - u)
        (2, rng.uniform(0, 1, n).astype(f), rng.choice([20.0, 100.0, 200.0, 1 / 21.0, 1 / 101.0, 1 / 201.0, 37.5], n).astype(f)),  # Phong lobes
        (2, np.exp(rng.uniform(-20, 20, n)).astype(f), rng.uniform(-4, 4, n).astype(f)),
    ]


def trig_cases(seed=2, n=1 << 20):
    """(mode, x, y) of the sin / cos / acos / atan2 / logf checks (host helper modes 3 .. 7): the arguments the sampling code produces -- angles in
    [-pi, 2 pi], direction cosines, quotients of direction components, r2 of the polar method in (0, 1] -- and wider ranges"""
    rng = np.random.default_rng(seed)
    f = np.float32
    z = np.zeros(n, f)
    u = rng.uniform(0, 1, n).astype(f)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(f)
    return [
        (3, (f(2 * np.pi) * u).astype(f), z), (4, (f(2 * np.pi) * u).astype(f), z), (3, (f(np.pi) * u).astype(f), z), (4, (f(np.pi) * u).astype(f), z),
        (3, rng.uniform(-400, 400, n).astype(f), z), (4, rng.uniform(-400, 400, n).astype(f), z), (3, rng.uniform(-3e9, 3e9, n).astype(f), z), (4, rng.uniform(-1e6, 1e6, n).astype(f), z),
        (3, (rng.uniform(-1, 1, n) * np.exp(rng.uniform(-60, 0, n))).astype(f), z),
        (5, d[:, 2].copy(), z), (5, rng.uniform(-1, 1, n).astype(f), z), (5, (1 - np.exp(rng.uniform(-17, 0, n))).astype(f), z), (5, (-1 + np.exp(rng.uniform(-17, 0, n))).astype(f), z),
        (6, d[:, 1].copy(), d[:, 0].copy()), (6, rng.uniform(-1, 1, n).astype(f), rng.uniform(-1, 1, n).astype(f)),
        (6, (rng.uniform(-1, 1, n) * np.exp(rng.uniform(-40, 40, n))).astype(f), (rng.uniform(-1, 1, n) * np.exp(rng.uniform(-40, 40, n))).astype(f)),
        (7, (1.0 - u).astype(f) + f(1e-38), z), (7, np.exp(rng.uniform(np.log(1e-38), 0, n)).astype(f), z), (7, np.exp(rng.uniform(np.log(1e-44), np.log(3e38), n)).astype(f), z),
    ]


def run_pair(width, height, num_init, n_chains, init_threads, per_chain, steps, use_gradient, max_depth=6, scene=TORUS, mala=True, opts=None,
             force_diffuse=1, oracle_grad="reference"):
    """Runs the same configuration on the oracle and on the GPU; returns a dict of comparison figures.
    oracle_grad: "reference" = the reference's generated derivative programs (oracle/_ref), "product" = the product's
    path program compiled for the host (tests/helpers)."""
    p = pkg()
    L = oracle_lib()
    glib = (pathref() if oracle_grad == "reference" else host_pathfunc_lib()) if use_gradient else ""
    orc = _orc.Oracle(L, scene, force_diffuse, max_depth, width, height, 0, glib)
    ren = p.Renderer(scene, force_diffuse=force_diffuse, max_depth=max_depth, width=width, height=height, seed_offset=0, use_gradient=use_gradient)
    for k, v in (opts or {}).items():
        L.orc_set_option(orc.h, k.encode(), float(v))
        ren.set_option(k, v)
    if not mala:
        L.orc_set_option(orc.h, b"mala", 0.0)
        ren.set_option("mala", 0)
    on, oc = orc.init(num_init, n_chains, init_threads)
    gn, gc = ren.init_chains(num_init, n_chains, init_threads, per_chain)
    si, gi = orc.summary(1), ren.summary(1)
    res = dict(norm_oracle=on, norm_gpu=gn, contribs_oracle=oc, contribs_gpu=gc)
    same_cl = (si[:, 1] == gi[:, 1]) & (si[:, 2] == gi[:, 2])
    res["init_cl_match"] = float(same_cl.mean())
    ok = same_cl & (si[:, 3] > 0)
    res["init_ls_relerr_max"] = float(np.max(np.abs(si[ok, 3] - gi[ok, 3]) / si[ok, 3])) if ok.any() else 0.0
    res["init_pss_maxdiff"] = float(np.max(np.abs(si[ok, 16:] - gi[ok, 16:]))) if ok.any() else 0.0
    orc.setup_chains(per_chain, 0)
    orc.step(steps)
    ren.step(steps)
    so, sg = orc.stats(), ren.stats()
    res["stats_oracle"], res["stats_gpu"] = so, sg
    fo, fg = orc.film(), ren.film()
    lo, lg = lum(fo), lum(fg)
    res["film_sum_oracle"], res["film_sum_gpu"] = float(lo.sum()), float(lg.sum())
    res["film_rel_l2"] = float(np.linalg.norm(lo - lg) / max(np.linalg.norm(lo), 1e-30))
    res["energy_gpu"] = float(lg.sum() / (gn * sg["weightSum"])) if sg["weightSum"] > 0 else 0.0
    res["energy_oracle"] = float(lo.sum() / (on * so["weightSum"])) if so["weightSum"] > 0 else 0.0
    co, cg = orc.summary(0), ren.summary(0)
    same = (co[:, 0] == cg[:, 0]) & (co[:, 1] == cg[:, 1]) & (co[:, 2] == cg[:, 2])
    close = same & (np.abs(co[:, 3] - cg[:, 3]) <= 1e-3 * np.abs(co[:, 3]) + 1e-12)
    res["final_state_match"] = float(close.mean())
    # technique histograms of the final states (c * 16 + l -> fraction): a comparison that survives re-seeded chains
    for name, cs in (("hist_oracle", co), ("hist_gpu", cg)):
        keys, cnt = np.unique((cs[:, 1] * 16 + cs[:, 2]).astype(int), return_counts=True)
        res[name] = {int(k): float(v) / len(cs) for k, v in zip(keys, cnt)}
    res["nonfinite_gpu"] = int((~np.isfinite(fg)).sum())
    orc.close()
    ren.close()
    return res


def smoke():
    L = oracle_lib()
    r = check_rng(L)
    res = run_pair(64, 48, 4000, 64, 4, 100, 8, use_gradient=1 if pathref() else 0)
    assert res["nonfinite_gpu"] == 0
    assert res["init_cl_match"] > 0.95, res
    assert abs(res["norm_gpu"] - res["norm_oracle"]) <= 1e-3 * res["norm_oracle"], res
    assert abs(res["energy_gpu"] - 1.0) < 1e-3, res
    assert res["final_state_match"] > 0.8, res
    print("smoke ok:", {k: v for k, v in res.items() if not k.startswith("stats")}, r)
