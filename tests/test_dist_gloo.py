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
