#!/usr/bin/env python3
"""Benchmark of the LMC hot path on MI355X: MALA chain-steps/s on the torus scene (BASELINE.json configs[1]:
torus geometry, every BSDF diffuse, maxdepth 6, 2^20 persistent chains per GPU, sunsky environment light).

One "step" = one lock-step pass of the chain loop body (mlt.cpp:91-170) over every resident chain.
Multi-GPU: one process per GPU (torch.distributed / RCCL), chains sharded by contiguous global chain-id range
(weak scaling: 2^20 chains per GPU), one all-reduce of the film + the normalisation scalar at the end.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects."""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_STEP = 2240  # algorithmic HBM bytes per chain-step, cfg 2 (L = 6, dim 12)
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--chains", type=int, default=1 << 20, help="chains per GPU (default 2^20)")
    ap.add_argument("--init-samples", type=int, default=0, help="MLT init samples (default 8 * total chains)")
    ap.add_argument("--init-threads", type=int, default=65536)
    ap.add_argument("--samples-per-chain", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-chains", type=int, default=32768)
    ap.add_argument("--cpu-steps", type=int, default=48)
    return ap.parse_args()


def cpu_baseline(args):
    """The CPU oracle (port of the reference's chain loop; gradients from the reference's own generated programs when
    oracle/_ref is present) on a bounded sample of the same workload, all host cores."""
    from tests import _orc, gpu_checks as gc

    L = gc.oracle_lib()
    cores = os.cpu_count() or 1
    cores = max(1, min(cores, args.cpu_chains // 64))  # at least 64 chains per worker thread between barriers
    orc = _orc.Oracle(L, gc.TORUS, 1, 6, 0, 0, 0, gc.pathref())
    n = args.cpu_chains
    orc.init(max(8 * n, 20000), n, 64)
    orc.setup_chains(args.samples_per_chain, 0)
    import ctypes

    done = ctypes.c_longlong()
    # untimed warm-up so that the timed part sits in the same regime as the GPU measurement
    L.orc_bench_steps(orc.h, min(args.warmup, 16), cores, ctypes.byref(done))
    t0 = time.time()
    rate = L.orc_bench_steps(orc.h, args.cpu_steps, cores, ctypes.byref(done))
    dt = time.time() - t0
    orc.close()
    return {
        "value": rate,
        "unit": "chain-steps/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d chains x %d steps (%.1f s), torus diffuse maxdepth 6, full-size film, gradients via %s"
        % (n, args.cpu_steps, dt, "reference derivative programs (oracle/_ref)" if gc.pathref() else "none (isotropic)"),
    }


def pmc_traffic():
    """HBM bytes per launch of the step kernel from the committed rocprofv3 --pmc summary, if any."""
    p = os.path.join(ROOT, "profiles", "pmc_step_kernel.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get("hbm_bytes_per_launch")
        except Exception:
            return None
    return None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist_

        torch.cuda.set_device(local)
        dist_.init_process_group("nccl")  # RCCL
        dist = dist_
    import numpy as np

    p = importlib.import_module("langevin-mcmc_amd")
    from tests import gpu_checks as gc

    per_gpu = args.chains
    total = per_gpu * world
    init_samples = args.init_samples or 8 * total
    ren = p.Renderer(gc.TORUS, force_diffuse=1, max_depth=6, seed_offset=0, device=local, use_gradient=1)
    t_init = time.time()
    norm, ncontrib = ren.init_chains(init_samples, total, args.init_threads, args.samples_per_chain, 0, rank * per_gpu, (rank + 1) * per_gpu)
    t_init = time.time() - t_init

    def barrier():
        ren.sync()
        if dist is not None:
            import torch

            dist.barrier()
            torch.cuda.synchronize()

    ren.set_option("timing", 1)  # per-step HIP events on the launch stream (lmc_step_timing / lmc_kernel_timing)
    ren.step(args.warmup)
    ren.step_timing()  # discard warm-up launches
    lean_before = ren.kernel_timing()[2]
    barrier()
    t0 = time.time()
    ren.step(args.steps)
    film = None
    if dist is not None:
        import torch

        # the single data-path collective: film buffer + normalisation scalar over RCCL/xGMI
        film_t = torch.from_numpy(ren.film()).cuda()
        norm_t = torch.tensor([norm], device="cuda")
        dist.all_reduce(film_t)
        dist.all_reduce(norm_t)
        film = film_t
    barrier()
    dt = time.time() - t0
    if dist is not None:
        import torch

        t = torch.tensor([dt], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kernel_ms, launches = ren.step_timing()
    small_ms, large_ms, lean_after = ren.kernel_timing()
    stats = ren.stats()
    if rank == 0:
        value = args.steps * total / dt
        # dominant kernel: k_step_small (plain small steps); its own HIP-event bracket on the launch stream
        avg_launch_s = (small_ms / max(launches, 1)) * 1e-3
        lean_steps_per_launch = (lean_after - lean_before) / max(launches, 1)
        achieved = ALGO_BYTES_PER_STEP * lean_steps_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        out = {
            "metric": "MALA chain-steps/sec, torus scene",
            "value": value,
            "unit": "chain-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic chains on the shipped torus geometry + sunsky env map (random-seeded PCG streams)",
            "config": {
                "workload": "torus scene, %d persistent chains per GPU, Lambertian-only BSDF, max path length 6 (BASELINE.json configs[1])" % per_gpu,
                "chains_per_gpu": per_gpu,
                "init_samples": init_samples,
                "samples_per_chain": args.samples_per_chain,
                "film": [ren.width, ren.height],
                "parallelism": "chains sharded x%d" % world,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pmc_traffic(),
                "kernel": "k_step_small<true>",
                "avg_launch_ms": avg_launch_s * 1e3,
                "chain_steps_per_launch": lean_steps_per_launch,
                "algorithmic_bytes_per_step": ALGO_BYTES_PER_STEP,
            },
            "step_ms": {"all_launches": kernel_ms / max(launches, 1), "k_step_small": small_ms / max(launches, 1),
                        "large_and_generic": large_ms / max(launches, 1)},
            "init_seconds": t_init,
            "normalization": norm,
            "accept_rate": stats["accepted"] / max(stats["steps"], 1),
            "large_step_frac": stats["largeSteps"] / max(stats["steps"], 1),
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(args)
            except Exception as e:  # the bench line must still come out
                out["cpu_baseline"] = {"value": None, "unit": "chain-steps/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %s" % e}
        print(json.dumps(out))
    ren.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
