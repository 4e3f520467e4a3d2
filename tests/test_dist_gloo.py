"""world_size-2 gloo test of the multi-process path (CPU): chain sharding + film all-reduce.
The per-rank renderer is the CPU oracle here (no GPU in this tier); the sharding helpers are the product's."""
import importlib
import os
import socket

import numpy as np
import pytest

from tests import _orc
from tests import gpu_checks as gc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, per_chain, steps, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = importlib.import_module("langevin-mcmc_amd.sharding")
    L = gc.oracle_lib()
    orc = _orc.Oracle(L, gc.TORUS, 1, 6, 64, 48, 0, "")
    norm, _ = orc.init(6000, total, 4)
    b, e = sh.chain_range(rank, world, total // world)
    L.orc_setup_chains_range(orc.h, per_chain, 0, b, e)
    orc.num_chains = e - b
    orc.step(steps)
    film, n = sh.allreduce_film(orc.film(), norm, dist)
    if rank == 0:
        q.put((film, n, orc.stats()["steps"]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    import torch.multiprocessing as mp

    total, per_chain, steps = 64, 100, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, per_chain, steps, q)) for r in range(2)]
    for p in procs:
        p.start()
    film2, n2, steps0 = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    L = gc.oracle_lib()
    orc = _orc.Oracle(L, gc.TORUS, 1, 6, 64, 48, 0, "")
    norm, _ = orc.init(6000, total, 4)
    orc.setup_chains(per_chain, 0)
    orc.step(steps)
    film1 = orc.film()
    assert n2 == pytest.approx(norm, rel=0, abs=0)
    assert steps0 == steps * total // 2
    # identical chains, identical splats; only the float add order across ranks differs
    assert np.allclose(film2, film1, rtol=1e-5, atol=1e-7)
    assert abs(film2.sum() - film1.sum()) <= 1e-5 * abs(film1.sum())


def test_chain_range_helpers():
    sh = importlib.import_module("langevin-mcmc_amd.sharding")
    assert sh.chain_range(3, 8, 1 << 20) == (3 << 20, 4 << 20)
    parts = sh.split_total(10, 4)
    assert parts == [(0, 3), (3, 6), (6, 8), (8, 10)]
    with pytest.raises(ValueError):
        sh.chain_range(2, 2, 4)


# ------------------------------------------------------------------------------------------------ sharded MLTInit, exchange logic
def _init_worker(rank, world, port, ninit, V, chains, q):
    """One rank of the sharded MLTInit, on CPU: its 'device phases' are played by the oracle (it runs the whole init and the rank
    keeps the slice its streams produced), the EXCHANGE and the PLAN are the product's: padded blocks laid out by lmc_shard_layout,
    all-gathered over gloo exactly like the ranks of an RCCL job all-gather them, put back into stream order and walked by
    lmc_shard_plan_probe (= host/shardplan.cpp, the code lmc_chains_init runs between its device phases)."""
    import ctypes

    import torch
    import torch.distributed as dist
    from tests._orc import P

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = ctypes.CDLL(gc.pkg().LIB_PATH)
    L = gc.oracle_lib()
    orc = _orc.Oracle(L, gc.TORUS, 1, 6, 64, 48, 0, "")
    norm, ncontrib = orc.init(ninit, chains, V)
    cap = 8 * ninit
    smp, cl, ls = np.zeros(cap, np.int64), np.zeros(cap, np.int32), np.zeros(cap, np.float32)
    L.orc_init_contribs.restype = ctypes.c_longlong
    L.orc_init_contribs.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    n = L.orc_init_contribs(orc.h, cap, P(smp), P(cl), P(ls))
    smp, cl, ls = smp[:n], cl[:n].astype(np.uint8), ls[:n]
    counts = np.bincount(smp, minlength=ninit).astype(np.uint8)
    lay = (ctypes.c_longlong * 5)()
    assert lib.lmc_shard_layout(world, rank, V, ctypes.c_longlong(ninit), lay) == 0
    t0, t1, g0, g1, max_samples = list(lay)
    # exchange 1: contributions per sample, blocks padded to the largest rank
    send = np.zeros(max_samples, np.uint8)
    send[: g1 - g0] = counts[g0:g1]
    got = [torch.zeros(max_samples, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(got, torch.from_numpy(send))
    padded_counts = np.concatenate([g.numpy() for g in got])
    rank_first = (ctypes.c_ulonglong * (world + 1))()
    assert lib.lmc_shard_counts_probe(world, V, ctypes.c_longlong(ninit), P(padded_counts), ctypes.c_longlong(max_samples), rank_first) == 0
    rf = list(rank_first)
    assert rf[world] == n
    max_contribs = max(rf[r + 1] - rf[r] for r in range(world))
    # exchange 2: technique and score of every contribution of the local samples
    o0, o1 = rf[rank], rf[rank + 1]
    scl, sls = np.zeros(max_contribs, np.uint8), np.zeros(max_contribs, np.float32)
    scl[: o1 - o0], sls[: o1 - o0] = cl[o0:o1], ls[o0:o1]
    gcl = [torch.zeros(max_contribs, dtype=torch.uint8) for _ in range(world)]
    gls = [torch.zeros(max_contribs, dtype=torch.float32) for _ in range(world)]
    dist.all_gather(gcl, torch.from_numpy(scl))
    dist.all_gather(gls, torch.from_numpy(sls))
    pcl, pls = np.concatenate([g.numpy() for g in gcl]), np.concatenate([g.numpy() for g in gls])
    seed_sample, seed_cl, seed_ls = np.zeros(chains, np.int64), np.zeros(chains, np.uint8), np.zeros(chains, np.float32)
    owned = (ctypes.c_int * (world + 1))()
    nrm = ctypes.c_float()
    r_ = lib.lmc_shard_plan_probe(world, V, ctypes.c_longlong(ninit), chains, P(padded_counts), ctypes.c_longlong(max_samples), P(pcl), P(pls),
                                  ctypes.c_longlong(max_contribs), P(seed_sample), P(seed_cl), P(seed_ls), owned, ctypes.byref(nrm))
    assert r_ == 0
    init = orc.summary(1)
    q.put((rank, nrm.value, norm, seed_sample, seed_cl, seed_ls, list(owned), init[:, 1] * 16 + init[:, 2], init[:, 3], (g0, g1)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_init_exchange_and_plan():
    """the exchange logic of the sharded MLTInit with world_size 2 over gloo.  Both ranks must arrive at the
    one-rank result: the oracle's normalization (bit equal), the oracle's init state for every chain (technique and lsScore of the
    seeding contribution), and a consistent ownership split (rank r's samples seed exactly the chains [owned[r], owned[r + 1]))."""
    import torch.multiprocessing as mp

    ninit, V, chains, world = 12000, 37, 96, 2  # 37 streams: uneven split (18 / 19), streams with and without the extra sample
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_init_worker, args=(r, world, port, ninit, V, chains, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, nrm, norm, ss, scl, sls, owned, init_cl, init_ls, (g0, g1) in res:
        assert nrm == norm  # the same sequential float sum as the unsharded init
        assert np.array_equal(scl.astype(np.float32), init_cl) and np.array_equal(sls, init_ls)
        assert owned[0] == 0 and owned[world] == chains and all(owned[r] <= owned[r + 1] for r in range(world))
        mine = ss[owned[rank]:owned[rank + 1]]
        assert ((mine >= g0) & (mine < g1)).all()  # the checkpoints this rank would send are of its own samples
        assert np.all(np.diff(ss) >= 0)
    assert np.array_equal(res[0][3], res[1][3]) and res[0][6] == res[1][6]
