#!/usr/bin/env python3
"""TEST HELPER: one rank PROCESS of a multi-rank job through the library's RCCL code path (lmc_comm_init -> collective lmc_chains_init -> lmc_chains_step with
the per-step all-gather of the cache pushes -> lmc_film_allreduce), the way bench.py's spawned ranks and an integrator's MPI ranks drive it.  On the one-GPU test
tier the communicator is tests/helpers/rccl_stub.cpp (LMC_RCCL_LIB): all ranks share device 0.
usage: rank_worker.py <rank> <world> <dir> <chains> <steps> <init samples> <init streams> [device]"""
import importlib, json, os, sys, time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
rank, world, d, n, steps, ninit, streams = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
device = int(sys.argv[8]) if len(sys.argv) > 8 else 0
p = importlib.import_module("langevin-mcmc_amd")
sharding = importlib.import_module("langevin-mcmc_amd.sharding")
scene = os.path.join(ROOT, "scenes", "torus", "lmc.xml")
ren = p.Renderer(scene, force_diffuse=1, max_depth=6, width=96, height=72, seed_offset=0, device=device, use_gradient=1)
idf = os.path.join(d, "id.bin")
if rank == 0:
    open(idf + ".tmp", "wb").write(p.comm_unique_id())
    os.rename(idf + ".tmp", idf)
t0 = time.time()
while not os.path.exists(idf):
    if time.time() - t0 > 120:
        sys.exit("rank %d: no communicator id" % rank)
    time.sleep(0.01)
ren.comm_init(world, rank, open(idf, "rb").read())
b, e = sharding.group_ranges(n, world)[rank]
norm, nc = ren.init_chains(ninit, n, streams, steps, 0, b, e)
init = ren.summary(1)
ren.step(steps)
st, fin = ren.stats(), ren.summary(0)
own_film = ren.film()
ren.film_allreduce()
film = ren.film()
mx = ren.comm_allreduce([float(rank)], "max")[0]
ren.comm_barrier()
np.savez(os.path.join(d, "rank%d.npz" % rank), norm=norm, nc=nc, init=init, fin=fin, film=film, own_film=own_film, stats=json.dumps(st), range=np.array([b, e]), max_rank=mx)
ren.close()
