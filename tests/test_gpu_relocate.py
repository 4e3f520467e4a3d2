"""Chain relocation (langevin-mcmc_amd/csrc/device/relocate.hip): the chains are kept physically grouped by technique so that a wave
of the step kernels retraces one (c,l).  The reference has no counterpart (its chains are objects a thread walks, mlt.cpp:60-196), so
the contract checked here is transparency: with relocation on, EVERY chain follows the trajectory it follows with relocation off --
same RNG stream, same states, same counters -- and the film differs by the order of its float atomics only."""
import os

import numpy as np
import pytest

from tests import gpu_checks as gc

pytestmark = pytest.mark.gpu


def _run(relocate, mala, n_chains, steps, opts, checkpoints=(), max_depth=6, outlier_test=False):
    os.environ["LMC_RELOCATE"] = "1" if relocate else "0"
    os.environ["LMC_EXP_OUTLIER_TEST"] = "1" if outlier_test else "0"
    os.environ["LMC_RESORT_EVERY"] = "4" if relocate else "0"  # the periodic full re-sort by (technique, screen Morton code) rides along: every 4th step here
    try:
        p = gc.pkg()
        ren = p.Renderer(gc.TORUS, force_diffuse=1, max_depth=max_depth, width=128, height=96, seed_offset=0, use_gradient=1 if gc.pathref() else 0)
        for k, v in opts.items():
            ren.set_option(k, v)
        if not mala:
            ren.set_option("mala", 0)
        ren.init_chains(200000, n_chains, 64, 10 ** 6)
    finally:
        del os.environ["LMC_RELOCATE"], os.environ["LMC_EXP_OUTLIER_TEST"], os.environ["LMC_RESORT_EVERY"]
    out = []
    done = 0
    for upto in list(checkpoints) + [steps]:
        ren.step(upto - done)
        done = upto
        out.append((ren.summary(0).copy(), ren.stats(), ren.relocation_stats()))
    film = ren.film().copy()
    ren.close()
    return out, film


def _same_states(a, b):
    # rows are reported in chain order whatever slot a chain lives in.  Of an invalid state only technique and lsScore mean anything
    valid = a[:, 0] == 1
    assert (a[:, 0] == b[:, 0]).all()
    assert np.array_equal(a[valid], b[valid])
    assert np.array_equal(a[~valid][:, 1:4], b[~valid][:, 1:4])
    assert np.array_equal(a[:, 9], b[:, 9])  # sampleIdx


def test_relocation_is_transparent_plain_mlt():
    """Plain MLT (mala = 0): no gradient cache, relocation runs from the first step on.  4096 chains x 40 steps, states compared at three points."""
    opts = {"largestepprob": 0.3, "largestepscale": 1.0}
    off, film0 = _run(False, False, 4096, 40, opts, checkpoints=(1, 7))
    on, film1 = _run(True, False, 4096, 40, opts, checkpoints=(1, 7))
    for (s0, st0, r0), (s1, st1, r1) in zip(off, on):
        assert r0 is None and r1 is not None and r1["relocations"] > 0
        _same_states(s0, s1)
        for k in ("steps", "largeSteps", "accepted", "resets"):
            assert st0[k] == st1[k], k
        assert st1["weightSum"] == pytest.approx(st0["weightSum"], rel=1e-6)
    l0, l1 = gc.lum(film0), gc.lum(film1)
    assert np.linalg.norm(l0 - l1) <= 1e-5 * np.linalg.norm(l0)
    # and the chains ARE grouped: at most a few technique changes along the slots per technique, instead of one at most slot boundaries
    r = on[-1][2]
    assert r["breaks"] < 0.35 * r["slots"], r  # ~30 % of the chains changed technique in the step just run and have not been placed yet


def test_relocation_is_transparent_through_the_cache_fill_phase_and_after():
    """MALA with the gradient cache (the set-up of test_cache_phase_parity): relocation runs from the first step on, through the whole cache-fill phase (beside the MALA gradient pipeline and the early apply of the pushes); the chains carry
    stored Gaussians, moment vectors and cache look-ups -- all of which must travel with them."""
    opts = {"largestepprob": 0.5, "largestepscale": 1.0}  # maxdepth 4: two cache dims (6, 8), both full after ~25 steps of 16384 chains
    off, film0 = _run(False, True, 16384, 72, opts, checkpoints=(8, 40), max_depth=4)
    on, film1 = _run(True, True, 16384, 72, opts, checkpoints=(8, 40), max_depth=4)
    assert off[-1][1]["cacheReadyMask"] != 0, "test set-up: the cache never filled"
    assert on[-1][2]["relocations"] > 0, "test set-up: relocation never started"
    # the first checkpoint lies INSIDE the fill phase (the contract is "relocated from the first step on", beside the gradient pipeline
    # and the early apply of the cache pushes): a cache is still filling, gradients are being evaluated, and eight relocations have already run
    assert off[0][1]["cacheReadyMask"] != off[-1][1]["cacheReadyMask"] and off[0][1]["gradCalls"] > 0 and on[0][2]["relocations"] >= 8
    for (s0, st0, r0), (s1, st1, r1) in zip(off, on):
        _same_states(s0, s1)
        for k in ("steps", "largeSteps", "accepted", "resets", "cacheQueries", "cacheHits", "gradCalls", "cacheReadyMask"):
            assert st0[k] == st1[k], k
    l0, l1 = gc.lum(film0), gc.lum(film1)
    assert np.linalg.norm(l0 - l1) <= 1e-5 * np.linalg.norm(l0)


def test_relocation_is_transparent_h2mc():
    """H2MC renders: only chains without a stored Gaussian move (the dense Gaussian lives in the pipeline's per-slot buffers)."""
    opts = {"h2mc": 1, "largestepprob": 0.2, "perturbstddev": 0.01}
    off, film0 = _run(False, True, 4096, 30, opts, checkpoints=(2, 9))
    on, film1 = _run(True, True, 4096, 30, opts, checkpoints=(2, 9))
    assert on[-1][2]["relocations"] > 0 and on[-1][2]["moved"] > 0
    for (s0, st0, r0), (s1, st1, r1) in zip(off, on):
        _same_states(s0, s1)
        for k in ("steps", "largeSteps", "accepted", "resets"):
            assert st0[k] == st1[k], k
    l0, l1 = gc.lum(film0), gc.lum(film1)
    assert np.linalg.norm(l0 - l1) <= 1e-5 * np.linalg.norm(l0)


@pytest.mark.parametrize("mala", [False, True])
def test_relocated_chains_reset_to_the_init_state_of_their_own_id(mala):
    """The outlier reset (mlt.cpp:147-169) walks CHAIN ids from the chain's own (`chainId = (chainId + sampleIdx + cnt) % numChains`); a relocated
    chain lives in another slot and carries its id in `chainId`.  With the reference's counts (1000 / 10000 adjacent rejections) no test reaches
    the reset, so `LMC_EXP_OUTLIER_TEST=1` lowers them to 2 / 6 on BOTH sides: thousands of resets, and still every chain in the same state."""
    opts = {"largestepprob": 0.1, "largestepscale": 1.0}
    off, film0 = _run(False, mala, 8192, 40, opts, checkpoints=(10,), outlier_test=True)
    on, film1 = _run(True, mala, 8192, 40, opts, checkpoints=(10,), outlier_test=True)
    assert off[-1][1]["resets"] > 1000, "test set-up: too few resets"
    for (s0, st0, r0), (s1, st1, r1) in zip(off, on):
        _same_states(s0, s1)
        for k in ("steps", "largeSteps", "accepted", "resets"):
            assert st0[k] == st1[k], k
    l0, l1 = gc.lum(film0), gc.lum(film1)
    assert np.linalg.norm(l0 - l1) <= 1e-5 * np.linalg.norm(l0)


def test_benchmarked_size_through_the_cache_ready_transition():
    """the size bench.py measures -- 2^20 chains of the torus workload (Lambertian, maxdepth 6, 1024x768 film), 30 steps of
    a fresh population, i.e. through the whole cache-fill phase and the step in which dims 10 and 12 become ready -- with chain relocation AND the
    periodic full re-sort on, against the same run with neither: every chain of a 4096-chain sample (plus the first and last 64 ids) in the SAME state,
    exact counters, the energy identity, 64-bit-safe indexing (the state arrays hold 2^20 x ~3000 words: word offsets beyond 2^31)."""
    p = gc.pkg()
    n = 1 << 20
    runs = []
    for relocate in (False, True):
        os.environ["LMC_RELOCATE"] = "1" if relocate else "0"
        os.environ["LMC_RESORT_EVERY"] = "8" if relocate else "0"
        try:
            ren = p.Renderer(gc.TORUS, force_diffuse=1, max_depth=6, seed_offset=0, use_gradient=1 if gc.pathref() else 0)
            norm, nc = ren.init_chains(8 * n, n, 65536, 256)
        finally:
            del os.environ["LMC_RELOCATE"], os.environ["LMC_RESORT_EVERY"]
        assert nc >= n
        ren.step(30)
        st, rs, summ, film = ren.stats(), ren.relocation_stats(), ren.summary(0), ren.film()
        ren.close()
        assert st["steps"] == 30 * n and np.isfinite(film).all() and (film >= 0).all()
        assert gc.lum(film).sum() == pytest.approx(norm * st["weightSum"], rel=2e-4)
        runs.append((st, rs, summ, film, norm))
    (st0, rs0, s0, f0, n0), (st1, rs1, s1, f1, n1) = runs
    assert n0 == n1 and rs0 is None and rs1 is not None and rs1["relocations"] >= 29 and rs1["skipped"] == 0
    if gc.pathref():
        assert st0["cacheReadyMask"] != 0 and st0["gradCalls"] > 0, "test set-up: the run never left the cache-fill phase"
    for k in ("steps", "largeSteps", "accepted", "resets", "cacheQueries", "cacheHits", "gradCalls", "cacheReadyMask"):
        assert st0[k] == st1[k], (k, st0[k], st1[k])
    sample = np.r_[np.arange(64), np.random.default_rng(0).choice(n, 4096, replace=False), np.arange(n - 64, n)]
    _same_states(s0[sample], s1[sample])
    _same_states(s0, s1)  # ... and, since the rows are here anyway, every chain
    l0, l1 = gc.lum(f0), gc.lum(f1)
    assert np.linalg.norm(l0 - l1) <= 1e-5 * np.linalg.norm(l0)
    # the chains ARE grouped: few technique changes along the slots right after a step whose last full re-sort was at most 8 steps ago
    assert rs1["breaks"] < 0.2 * rs1["slots"], rs1


def test_full_resort_really_sorts_the_slots():
    """The transparency tests above would also pass if the full re-sort moved nothing.  With a re-sort after EVERY step (LMC_RESORT_EVERY=1, first after step 0)
    the slots must be in technique order right after a step -- at most one technique change per populated technique -- while the per-step relocation alone
    (round 4's quantile rule) leaves hundreds to thousands of them after the same steps; and the order inside a technique must follow the screen Morton
    code (not observable through the ABI: rows are reported by chain id; its effect is the lane utilisation of profiles/r06_n_pmc_lanes_*)."""
    p = gc.pkg()
    n = 1 << 16
    res = {}
    for every in (0, 1):
        ren = p.Renderer(gc.TORUS, force_diffuse=1, max_depth=6, width=256, height=192, seed_offset=0, use_gradient=0)
        ren.set_option("mala", 0)
        ren.set_option("resort_every", every)  # the option form of LMC_RESORT_EVERY / LMC_RESORT_FIRST (include/lmc_abi.h)
        ren.set_option("resort_first", 0)
        ren.init_chains(8 * n, n, 4096, 10 ** 6)
        ren.step(12)
        rs, summ = ren.relocation_stats(), ren.summary(0)
        # rows come in chain order; the slot of a chain is not exposed, but `breaks` is counted along the slots
        res[every] = (rs, summ)
        ren.close()
    techniques = len({(int(r[1]), int(r[2])) for r in res[1][1] if r[0] == 1})
    assert res[1][0]["breaks"] <= techniques + 2, (res[1][0], techniques)
    assert res[0][0]["breaks"] > 20 * techniques, (res[0][0], techniques)
    # both runs hold the same chains in the same states (the sort is transparent) ...
    _same_states(res[0][1], res[1][1])
