#!/usr/bin/env python3
"""Benchmark of the LMC hot path on MI355X: MALA chain-steps/s on the torus scene (BASELINE.json configs[1]:
torus geometry, every BSDF diffuse, maxdepth 6, 2^20 persistent chains per GPU, sunsky environment light).

One "step" = one lock-step pass of the chain loop body (mlt.cpp:91-170) over every resident chain.
Multi-GPU: chains sharded by contiguous global chain-id range (weak scaling: 2^20 chains per GPU), ONE PROCESS PER GPU, the exchanges are RCCL
collectives inside the library (sharded MLTInit: ncclAllGather; cache pushes while the gradient caches fill: ncclAllGather per step; one
ncclAllReduce of the per-GPU films at the end, inside the timed region).  Launch forms, same sharding, same trajectories:
  * `python bench.py --gpus N`: this script starts the N rank processes itself (main_spawn; the 128-byte RCCL id travels through a private
    directory, barriers and max-over-ranks timing through the job's communicator -- no torch);
  * `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` (the driver's N > 1 command): the launcher starts the ranks, the id
    travels over its gloo rendezvous (WORLD_SIZE must equal --gpus); everything else as above;
  * `python bench.py --gpus N --in-process`: explicit fallback, ONE process drives N contexts with device copies instead of RCCL (lmc_group_*;
    the reference merges its per-thread films in-process too, mlt.cpp:203-207); with LMC_BENCH_OVERSUBSCRIBE=1 the N contexts may share devices
    (bring-up, reported as `oversubscribed`).
Every form refuses (exit code 2) to run with fewer than N visible devices: never a silently smaller job.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (every N) and, at N = 1, `cpu_baseline`, `equal_time_rmse`, `configs`."""
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
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall-time bound of the CPU baseline sample")
    ap.add_argument("--no-rmse", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE.json workloads (the `configs` array of the JSON line)")
    ap.add_argument("--rmse-seconds", type=float, default=2.0, help="GPU wall time of the equal-time RMSE leg")
    ap.add_argument("--rmse-gt-spp", type=int, default=8192)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default, the driver's form): --chains per GPU, the job grows with N; strong: --chains is the job's TOTAL, split evenly over the N GPUs (the other reading of north_star's '>= 6x at 8 GPUs')")
    ap.add_argument("--in-process", action="store_true", help="--gpus N > 1 without a launcher: ONE process drives N contexts (device copies instead of RCCL); the default is one process per GPU over RCCL")
    args = ap.parse_args()
    if args.scaling == "strong":  # everything below works with chains PER GPU
        if args.chains % args.gpus:
            ap.error("--scaling strong: --chains (the job's total) must be a multiple of --gpus")
        args.chains //= args.gpus
    return args


def host_cpus():
    """CPUs this process can actually use: the smaller of the hardware-thread count, the affinity mask and the cgroup CPU quota (cpu.max).
    The GPU box's container shows 256 hardware threads but is granted 16 CPUs of time (cpu.max = 1600000 100000): rounds 1-3 ran the
    baseline on "256 cores", i.e. 256 threads sharing 16 CPUs (scripts/cpu_baseline_scaling.py: flat from 32 threads on)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:  # noqa: BLE001
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, per = open(path).read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(round(int(q) / int(per)))))
        except Exception:  # noqa: BLE001
            pass
    try:
        q, per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, int(round(q / per))))
    except Exception:  # noqa: BLE001
        pass
    return n


def cpu_baseline(args):
    """The CPU oracle (port of the reference's chain loop; gradients from the reference's own generated programs when
    oracle/_ref is present) with the reference's own scheduling -- one chain per work item on a pool of host threads
    (parallel.cpp:82-142), chains never wait for each other, cache pushes under a mutex (mlt.cpp:120-127) -- on a bounded
    sample of the same workload (torus, Lambertian-only, maxdepth 6, full-size film).
    Two builds of the same sources, half the time budget each: `value` is the parity build (-O2, no contraction: the arithmetic the
    GPU tests compare with) on every hardware thread, 4 chains per thread; `reference_style` is the build the reference's own Tupfile
    describes (-Ofast class flags, x86-64-v3) on the reference's shape of job: 128 chains (scenes/torus/lmc.xml) on at most the physical cores."""
    from tests import _orc, gpu_checks as gc

    cores = host_cpus()
    hw = os.cpu_count() or 1
    threads_all = min(hw, 2 * cores)  # two workers per granted CPU: fills the quota whether or not the workers land on SMT siblings

    def run(lib_path, n_chains, threads, seconds):
        L = _orc.load(lib_path)
        orc = _orc.Oracle(L, gc.TORUS, 1, 6, 0, 0, 0, gc.pathref())
        orc.init(300000, n_chains, min(cores, 64))
        orc.setup_chains(1 << 30, 0)  # far more mutations than the time limit allows: every worker runs until the deadline
        t0 = time.time()
        rate, done = orc.run_async(threads, seconds)
        dt = time.time() - t0
        orc.close()
        return rate, done, dt

    n = 4 * threads_all  # a few chains per worker, like the reference's 128 chains on 32 cores
    rate, done, dt = run(gc.ORACLE_SO, n, threads_all, args.cpu_seconds / 2)
    out = {
        "value": rate,
        "unit": "chain-steps/s",
        "cores": cores,
        "kind": "port",
        "threads": threads_all,
        "host": "%d hardware threads visible, %d CPUs granted (affinity / cgroup cpu.max)" % (hw, cores),
        "sample": "%d chains on %d threads (%d CPUs granted) for %.1f s = %d mutations; torus diffuse maxdepth 6, 1024x768 film, one chain per work item (the reference's "
        "scheduling), gradients via %s" % (n, threads_all, cores, dt, done, "the reference's derivative programs (oracle/_ref)" if gc.pathref() else "none (isotropic)"),
        "reference_authors": {"value": 4.31e6, "cores": 32, "note": "derived from the shipped render's file name: 245 spp x 1024x768 in 44.69 s, full-material torus (BASELINE.md)"},
    }
    fast = os.path.join(ROOT, "oracle", "liblmc_oracle_fast.so")
    if os.path.exists(fast):
        try:
            threads = max(1, min(128, threads_all))  # the reference's shape of job: 128 chains, a worker per granted hardware thread
            r2, d2, t2 = run(fast, 128, threads, args.cpu_seconds / 2)
            out["reference_style"] = {"value": r2, "unit": "chain-steps/s", "cores": cores, "threads": threads, "per_core": r2 / cores, "kind": "port",
                                      "sample": "128 chains on %d threads (%d CPUs granted) for %.1f s = %d mutations; the same sources built -O3 -ffast-math -march=x86-64-v3 (the reference builds with -Ofast, src/Tupfile:17)" % (threads, cores, t2, d2),
                                      "note": "the reference's authors report 135 k mutations/s per core on the full-material scene (BASELINE.md); never a parity oracle"}
        except Exception as e:  # noqa: BLE001
            out["reference_style"] = {"failed": str(e)[:200]}
    return out


def lum(img):
    import numpy as np

    return img.astype(np.float64) @ np.array([0.212671, 0.715160, 0.072169])


def equal_time_rmse(args, p, gc, gpu_rate, cpu_rate, cores):
    """BASELINE.json's second metric.  Same workload as the throughput figure at a 256x192 film: the GPU renders for about
    args.rmse_seconds, the CPU oracle (reference scheduling, all host cores) gets the same wall time, both are compared with a
    plain Monte Carlo estimate of the same quantity (lmc_bidir_mc: bidirectional samples of path length >= 3, no Markov chain).
    The Markov chains only carry paths of length >= 3 (mlt.h:76); the direct pre-pass is the same estimator on both sides and
    is left out, like it is left out of the reference's timer (mlt.cpp:56-57,200)."""
    import numpy as np
    from tests import _orc

    W, H = 256, 192
    chains = 1 << 16
    spp = max(64, int(gpu_rate * 0.25 * args.rmse_seconds / (W * H)))  # 2^16 chains run at about a quarter of the 2^20-chain rate
    per = spp * W * H // chains
    # Ground truth: at least 16 x the samples of the leg under test, as TWO independent half-budget estimates (different RNG streams)
    # whose difference measures the truth's own noise: the truth is their mean, its noise half the RMS of their difference (at
    # 8192 spp against a 5700-spp leg the reported rel_rmse sat on the truth's noise floor)
    gt_half = max(args.rmse_gt_spp, 16 * spp) // 2
    t0 = time.time()
    halves = []
    for so in (0, 1000003):
        r2 = p.Renderer(gc.TORUS, force_diffuse=1, max_depth=6, width=W, height=H, seed_offset=so, use_gradient=1)
        halves.append(lum(r2.bidir_mc(gt_half)))
        r2.close()
    t_gt = time.time() - t0
    gt = 0.5 * (halves[0] + halves[1])
    gt_noise = float(np.sqrt(np.mean((0.5 * (halves[0] - halves[1])) ** 2)) / gt.mean())
    ren = p.Renderer(gc.TORUS, force_diffuse=1, max_depth=6, width=W, height=H, seed_offset=0, use_gradient=1)
    ren.init_chains(8 * chains, chains, 65536, per, per % chains)
    ren.sync()
    t0 = time.time()
    ren.step(per + 1)
    ren.sync()
    t_gpu = time.time() - t0
    img_gpu = lum(ren.film()) / spp
    ren.close()
    L = gc.oracle_lib()
    orc = _orc.Oracle(L, gc.TORUS, 1, 6, W, H, 0, gc.pathref())
    n = 4 * cores
    spp_cpu = max(1, int(cpu_rate * t_gpu / (W * H)))
    per_c = spp_cpu * W * H // n
    orc.init(300000, n, min(cores, 64))
    orc.setup_chains(per_c, per_c % n)
    t0 = time.time()
    orc.run_async(cores, 0.0)
    t_cpu = time.time() - t0
    img_cpu = lum(orc.film()) / spp_cpu
    orc.close()
    rel = lambda a: float(np.sqrt(np.mean((a - gt) ** 2)) / gt.mean())
    net = lambda a: float(np.sqrt(max(rel(a) ** 2 - gt_noise ** 2, 0.0)))  # the leg's own error with the truth's (independent) noise taken out
    return {
        "truth_crosscheck": truth_crosscheck(p, gc),
        "film": [W, H],
        "metric": "relative RMSE of the luminance of the indirect (path length >= 3) image vs a plain-MC bidirectional estimate",
        "ground_truth": {"estimator": "lmc_bidir_mc, mean of two independent halves (seed offsets 0 and 1000003)", "spp": 2 * gt_half, "spp_over_gpu_leg": 2 * gt_half / spp, "seconds": t_gt,
                         "own_noise_rel_rmse": gt_noise, "note": "own_noise = half the RMS difference of the two halves, relative to the mean luminance: the floor under every rel_rmse below"},
        "gpu": {"seconds": t_gpu, "spp": spp, "chains": chains, "mutations_per_chain": per, "rel_rmse": rel(img_gpu), "rel_rmse_net_of_truth_noise": net(img_gpu), "mean_ratio": float(img_gpu.mean() / gt.mean())},
        "cpu": {"seconds": t_cpu, "spp": spp_cpu, "chains": n, "threads": cores, "cpus_granted": host_cpus(), "mutations_per_chain": per_c, "rel_rmse": rel(img_cpu), "rel_rmse_net_of_truth_noise": net(img_cpu),
                "mean_ratio": float(img_cpu.mean() / gt.mean())},
    }


def truth_crosscheck(p, gc):
    """the ground-truth ESTIMATOR of the RMSE leg against something this repository did not compute -- the render the
    reference ships for scenes/torus/lmc.xml (lmc_timeuse_44.689152s.exr, 245 spp LMC on the reference's own CPU build; committed as a 4 x
    box-downsampled fixture, tests/golden/torus_ref_images_256x192.npz).  The shipped scene as is (full materials, maxdepth 8) through
    the same two estimators the leg uses as truth: direct pre-pass + plain-MC bidirectional samples.  relMSE = mean((a - b)^2 / (b^2 + 0.01))
    on luminance; the reference's own LMC and H2MC renders of this scene differ by 0.005, SURVEY's bar is twice that."""
    import numpy as np

    try:
        ref = np.load(os.path.join(ROOT, "tests", "golden", "torus_ref_images_256x192.npz"))
        lr, lr2 = lum(ref["lmc"]), lum(ref["h2mc"])
        W, H = 256, 192
        ren = p.Renderer(gc.TORUS, force_diffuse=0, max_depth=8, width=W, height=H, seed_offset=0, use_gradient=0)
        dspp, spp = 256, 32768  # at 8192 spp the estimator's own noise (relMSE 0.018) exceeds SURVEY's bar; 32768: 0.0066 (profiles/r04_o_truth_crosscheck.jsonl)
        t0 = time.time()
        img = lum(ren.direct_lighting(dspp) / dspp + ren.bidir_mc(spp))
        dt = time.time() - t0
        ren.close()
        def relmse(a, b, trim=0.0):
            e = np.sort(((a - b) ** 2 / (b ** 2 + 0.01)).ravel())
            return float(e[: int(len(e) * (1 - trim))].mean())

        return {"truth": "direct pre-pass %d spp + lmc_bidir_mc %d spp, shipped torus lmc.xml at 256x192" % (dspp, spp), "seconds": dt,
                "against": "the reference's shipped render lmc_timeuse_44.689152s.exr (245 spp), box-downsampled 4x",
                "relMSE": relmse(img, lr), "relMSE_trimmed_0.5pct": relmse(img, lr, 0.005), "mean_ratio": float(img.mean() / lr.mean()),
                "relMSE_between_the_reference's_own_lmc_and_h2mc_renders": relmse(lr2, lr), "bar": "2 x that (SURVEY.md 8d)",
                "passes_bar": bool(relmse(img, lr) <= 2 * relmse(lr2, lr)),
                "convergence": "the figure is the plain-MC estimator's own noise: 0.057 / 0.018 / 0.0066 at 2048 / 8192 / 32768 spp, mean ratio 1.015 throughout (profiles/r04_o_truth_crosscheck.jsonl)",
                "mean_offset": "the shipped render's own normaliser: the reference estimates `normalization` once from 300 000 init samples (mlt.h:41-154), an estimate with "
                               "a standard deviation of 5.8 % on this scene; with the reference's init configuration (32 streams, seedoffset 0) it is 0.984 of the converged value, "
                               "and a GPU render with that init is within 0.6 % of the shipped image (profiles/r05_k_*, r05_l_*, tests/test_gpu_round5.py)"}
    except Exception as e:  # noqa: BLE001
        return {"failed": str(e)[:300]}


def algorithmic_bytes(L, h2mc=False):
    """SURVEY.md §8(d)'s per-chain-step figure for maximum path length L (dim = 2L): read + write of {RNG 264 B; path record
    52 (L + 1) + 64 B; SubpathContrib 44 B; proposal Gaussian; adaptive vectors; one pending splat 20 B} + 48 B of splat traffic.
    Gaussian: diagonal (3 dim + 1) floats, plus v1, v2, g, M = 4 dim floats (LMC); dense mean + covL + invCov + logDet =
    (dim + 2 dim^2 + 1) floats and no adaptive vectors (H2MC).  L = 6 gives the headline's 2240 B."""
    dim = 2 * L
    path = 52 * (L + 1) + 64
    gauss = (dim + 2 * dim * dim + 1) * 4 if h2mc else (3 * dim + 1) * 4 + 4 * dim * 4
    return 2 * (264 + path + 44 + gauss + 20) + 48


def other_configs(args, p, gc):
    """The other BASELINE.json workloads, driver-observed: each runs after the headline's timed region on its
    own chain population (fresh Renderer), `warm` untimed steps to get past the start-up (cache fill), then `steps` timed steps
    bracketed by stream syncs.  `frac` = that workload's algorithmic bytes (algorithmic_bytes(L) with its own L) x the chain-steps
    its dominant kernel ran per launch / that kernel's HIP-event bracket / 8 TB/s."""
    door = os.path.join(ROOT, "scenes", "veachdoor")
    cfgs = [
        dict(name="torus, full BSDF set (Phong, rough dielectric, bitmap texture), max path length 12, LMC (BASELINE.json configs[2])",
             xml=gc.TORUS, kw=dict(force_diffuse=0, max_depth=12), chains=args.chains, L=12, h2mc=False, warm=40, steps=40, lanes_key="torus12"),
        dict(name="veach-door, shipped lmc.xml (area light, textures, max path length 8), LMC (BASELINE.json configs[3], one GPU's shard)",
             xml=os.path.join(door, "lmc.xml"), kw={}, chains=args.chains, L=8, h2mc=False, warm=40, steps=40, lanes_key="door"),
        dict(name="veach-door, shipped h2mc.xml, H2MC (BASELINE.json configs[4], one GPU's shard)",
             xml=os.path.join(door, "h2mc.xml"), kw={}, chains=args.chains, L=8, h2mc=True, warm=40, steps=40),
    ]
    out = []
    for c in cfgs:
        try:
            ren = p.Renderer(c["xml"], seed_offset=0, use_gradient=1, **c["kw"])
            t0 = time.time()
            ren.init_chains(8 * c["chains"], c["chains"], args.init_threads, args.samples_per_chain, 0)
            t_init = time.time() - t0
            ren.set_option("timing", 1)
            ren.step(c["warm"])
            ren.step_timing()
            k0 = ren.kernel_timing_split()
            s0 = ren.stats()
            ren.sync()
            t0 = time.time()
            ren.step(c["steps"])
            ren.sync()
            dt = time.time() - t0
            _, launches = ren.step_timing()
            k1 = ren.kernel_timing_split()
            s1 = ren.stats()
            # path length of the resident states (the stream a step moves scales with it): mean over the valid chains
            summ = ren.summary(0)
            valid = summ[:, 0] > 0
            mean_L = float((summ[valid, 1] + summ[valid, 2] - 1).mean()) if valid.any() else float(c["L"])
            ren.close()
            steps_total = s1["steps"] - s0["steps"]
            large_steps = s1["largeSteps"] - s0["largeSteps"]
            lean_steps = k1["lean_steps"] - k0["lean_steps"]
            ker = {"k_step_small (plain small steps)": (k1["lean_ms"], lean_steps), "k_step<large>": (k1["large_ms"], large_steps),
                   ("H2MC small-step pipeline (k_h2_begin .. k_h2_finish: all small steps)" if c["h2mc"] else "k_step_small_grad (cache-filling small steps)"): (k1["generic_ms"], steps_total - large_steps - lean_steps)}
            dom = max(ker, key=lambda k: ker[k][0])
            ab = algorithmic_bytes(c["L"], c["h2mc"])
            ab_mean = algorithmic_bytes(mean_L, c["h2mc"])  # at the step-weighted mean path length instead of the workload's maximum
            dom_ms, dom_steps = ker[dom][0] / max(launches, 1), ker[dom][1] / max(launches, 1)
            ach = ab * dom_steps / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
            out.append({
                "workload": c["name"], "chains": c["chains"], "value": steps_total / dt, "unit": "chain-steps/s", "ms_per_step": dt * 1e3 / c["steps"],
                "steps": c["steps"], "warmup": c["warm"], "init_seconds": t_init,
                "kernel_ms_per_step": {k: v[0] / max(launches, 1) for k, v in ker.items()},
                "roofline": {"bound": "hbm", "kernel": dom, "avg_launch_ms": dom_ms, "chain_steps_per_launch": dom_steps, "algorithmic_bytes_per_step": ab,
                             "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                             "mean_path_length": mean_L, "algorithmic_bytes_per_step_at_mean_length": ab_mean, "frac_at_mean_length": ach / HBM_PEAK_GBS * ab_mean / ab,
                             "concurrent_launches": "the three step launches share the GPU inside each bracket"},
                "accept_rate": (s1["accepted"] - s0["accepted"]) / max(steps_total, 1), "large_step_frac": large_steps / max(steps_total, 1),
            })
            lw = lanes_by_workload().get(c.get("lanes_key"))
            if lw:
                out[-1]["lanes_active"] = {"kernels": {k: x["lanes_active"] for k, x in lw["kernels"].items()},
                                           "source": "REPLAYED from profiles/lanes_by_workload.json (%s): lanes per issued vector instruction, not counted during this run" % lw["source"]}
            if c["h2mc"]:
                # the pipeline's dominant launch (k_h2_hess) is ARITHMETIC-bound: the HBM fraction above says nothing about it.  Its governing figure -- the share
                # of a SIMD's cycles in which the vector ALU issues -- needs counters, i.e. its own rocprofv3 --pmc passes: replayed from the committed summary
                vr = os.path.join(ROOT, "profiles", "h2mc_valu_roofline.json")
                if os.path.exists(vr):
                    v = json.load(open(vr))
                    hk = v["kernels"].get("k_h2_hess", {})
                    out[-1]["roofline"]["governing"] = {"bound": "valu", "kernel": "k_h2_hess (two launches per step, the largest part of the pipeline)", "valu_busy": hk.get("valu_busy"),
                                                         "lanes_active": hk.get("lanes_active"), "all_pipeline_kernels": {k: {"valu_busy": x["valu_busy"], "lanes_active": x["lanes_active"]} for k, x in v["kernels"].items()},
                                                         "source": "REPLAYED from profiles/h2mc_valu_roofline.json (%s; %s): not counted during this run" % (v.get("scene"), v.get("source"))}
        except Exception as e:  # noqa: BLE001 -- the headline line must still come out
            out.append({"workload": c["name"], "failed": str(e)[:300]})
    return out


def lanes_by_workload():
    """profiles/lanes_by_workload.json (scripts/lanes_summary.py): lanes active per issued vector instruction of the step launches, by workload"""
    f = os.path.join(ROOT, "profiles", "lanes_by_workload.json")
    return json.load(open(f)).get("workloads", {}) if os.path.exists(f) else {}


def kernel_source_sha():
    """Fingerprint of the sources the dominant kernel is compiled from: its translation unit and every header it includes, followed
    recursively (so that a change to a header only another kernel includes -- the H2MC launches', say -- leaves it alone)."""
    import hashlib, re

    dev = os.path.join(ROOT, "langevin-mcmc_amd", "csrc", "device")
    seen, todo = [], ["step_small_plain.hip"]
    while todo:
        f = todo.pop()
        if f in seen or not os.path.exists(os.path.join(dev, f)):
            continue
        seen.append(f)
        todo += re.findall(r'^\s*#\s*include\s+"([^"/]+)"', open(os.path.join(dev, f)).read(), flags=re.M)
    h = hashlib.sha256()
    for f in sorted(seen):
        h.update(open(os.path.join(dev, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic():
    """HBM bytes per launch of the step kernel from the committed rocprofv3 --pmc summary (scripts/pmc_to_json.py) -- only if
    that summary was measured on the kernel sources of this tree; otherwise null (a stale counter figure is worse than none)."""
    p = os.path.join(ROOT, "profiles", "pmc_step_kernel.json")
    if os.path.exists(p):
        try:
            d = json.load(open(p))
            return d.get("hbm_bytes_per_launch") if d.get("kernel_source_sha16") == kernel_source_sha() else None
        except Exception:
            return None
    return None


def pmc_traffic_meta():
    """what the committed counter figure was measured on: chain-steps per launch of that run (the wasted-traffic ratio is
    traffic / (algorithmic bytes x THESE steps), not x the steps of the window timed here) and the profile it comes from"""
    pth = os.path.join(ROOT, "profiles", "pmc_step_kernel.json")
    try:
        d = json.load(open(pth))
        if d.get("kernel_source_sha16") != kernel_source_sha():
            return None
        spl = d.get("chain_steps_per_launch")
        return {"chain_steps_per_launch": spl, "traffic_over_algorithmic": (d["hbm_bytes_per_launch"] / (ALGO_BYTES_PER_STEP * spl)) if spl else None, "source": d.get("source")}
    except Exception:
        return None


def die(msg, code=2):
    sys.stderr.write("bench.py: " + msg + "\n")
    sys.exit(code)


def inprocess_job(args, p, gc, name, xml, kw, devices, warm, steps, algo_bytes):
    """One workload on len(devices) GPUs driven from THIS process (`--in-process`): one context per device, joined into an in-process job (sharded
    MLTInit, contiguous chain ranges, per-step exchange of the cache pushes while the gradient caches fill, one film merge at the end); a host thread
    per member queues its launches.  Timed like the single-GPU line: `warm` untimed steps, then `steps` steps + the film merge bracketed by syncs
    of every context."""
    n = len(devices)
    per_gpu, total = args.chains, args.chains * n
    rens = [p.Renderer(xml, seed_offset=0, device=d, use_gradient=1, **kw) for d in devices]
    grp = p.Group(rens)
    t0 = time.time()
    norm, ncontrib = grp.init_chains(8 * total, total, args.init_threads, args.samples_per_chain, 0)
    t_init = time.time() - t0
    for r in rens:
        r.set_option("timing", 1)
    grp.step(warm)
    for r in rens:
        r.step_timing()
        r.host_issue_timing()
        r.sync()
    s0 = [r.stats() for r in rens]
    k0 = [r.kernel_timing_split() for r in rens]
    t0 = time.time()
    grp.step(steps)
    reduce_ms = grp.film_reduce()
    for r in rens:
        r.sync()
    dt = time.time() - t0
    per_rank, fracs, dom_ms_r, dom_names, issue = [], [], [], [], []
    s1 = [r.stats() for r in rens]
    for r, kk0, ss0, ss1 in zip(rens, k0, s0, s1):
        ms, launches = r.step_timing()
        launches = max(launches, 1)
        per_rank.append(ms / launches)
        dom, dom_ms, dom_steps, ach = dominant_kernel(kk0, r.kernel_timing_split(), ss0, ss1, launches, algo_bytes, "k_step_small (plain small steps)")
        dom_ms_r.append(dom_ms), dom_names.append(dom), fracs.append(ach / HBM_PEAK_GBS)
        ims, isteps = r.host_issue_timing()
        issue.append(ims / max(isteps, 1))
    steps_total = sum(b["steps"] - a["steps"] for a, b in zip(s0, s1))
    film_sum = float(rens[0].film().sum())
    info = grp.info()
    for r in rens:
        r.close()
    return {
        "workload": name, "n_gpus": n, "devices": list(devices), "chains_per_gpu": per_gpu, "value": steps_total / dt, "unit": "chain-steps/s",
        "ms_per_step": dt * 1e3 / steps, "steps": steps, "warmup": warm, "init_seconds": t_init, "normalization": norm,
        "per_rank_step_ms": per_rank, "per_rank_step_ms_spread": (max(per_rank) - min(per_rank)) if per_rank else 0.0,
        "host_issue_ms_per_step_per_rank": issue, "group": info,
        "film_merge_ms": reduce_ms, "film_sum": film_sum,
        "accept_rate": sum(b["accepted"] - a["accepted"] for a, b in zip(s0, s1)) / max(steps_total, 1),
        "roofline": {"bound": "hbm", "kernel": dom_names[0] + " (rank 0's dominant launch; each rank's own HIP-event bracket)", "frac": sum(fracs) / len(fracs), "achieved": sum(fracs) / len(fracs) * HBM_PEAK_GBS,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac_per_rank": fracs, "frac_min": min(fracs), "frac_max": max(fracs), "avg_launch_ms_per_rank": dom_ms_r,
                     "algorithmic_bytes_per_step": algo_bytes, "traffic": None},
    }


def main_inprocess(args):
    """`python bench.py --gpus N --in-process`: N contexts in this process, device copies instead of RCCL (the explicit fallback; the default for
    N > 1 is one process per GPU over RCCL, main_spawn)."""
    p = importlib.import_module("langevin-mcmc_amd")
    from tests import gpu_checks as gc

    have = p.device_count()
    devices = list(range(args.gpus))
    oversub = False
    if have < args.gpus:
        if have >= 1 and os.environ.get("LMC_BENCH_OVERSUBSCRIBE"):  # bring-up / test aid ONLY: the N ranks share the visible devices; reported as such
            devices = [k % have for k in range(args.gpus)]
            oversub = True
        else:
            die("--gpus %d but only %d HIP device(s) visible to this process: refusing to measure a smaller job under that name" % (args.gpus, have))
    door = os.path.join(ROOT, "scenes", "veachdoor", "lmc.xml")
    head = inprocess_job(args, p, gc, "torus scene, %d persistent chains per GPU, Lambertian-only BSDF, max path length 6 (BASELINE.json configs[1])" % args.chains,
                         gc.TORUS, dict(force_diffuse=1, max_depth=6), devices, args.warmup, args.steps, ALGO_BYTES_PER_STEP)
    out = {
        "metric": "MALA chain-steps/sec, torus scene", "value": head["value"], "unit": "chain-steps/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32",
        "data": "synthetic chains on the shipped torus geometry + sunsky env map (random-seeded PCG streams)",
        "config": {"workload": head["workload"], "chains_per_gpu": args.chains, "init_samples": 8 * args.chains * args.gpus, "samples_per_chain": args.samples_per_chain,
                   "parallelism": "chains sharded x%d, one process, one context per device, one host thread per context" % args.gpus,
                   "collective": "in-process job (--in-process): device-to-device copies for the sharded MLTInit and the cache pushes, a reduce-scatter + all-gather of peer copies for the film merge (lmc_group_*); NOT RCCL",
                   "rccl_ranks": 0, "devices": devices, "oversubscribed": oversub},
        "roofline": head["roofline"],
        "multi_gpu": {k: head[k] for k in ("per_rank_step_ms", "per_rank_step_ms_spread", "host_issue_ms_per_step_per_rank", "group", "film_merge_ms", "init_seconds", "accept_rate", "film_sum")},
        "lmc_env": lmc_env(),
    }
    if not args.no_configs:
        try:  # north_star names both scenes: the veach-door LMC workload as a second multi-GPU line
            out["configs"] = [inprocess_job(args, p, gc, "veach-door, shipped lmc.xml, LMC (BASELINE.json configs[3]), chains sharded over the GPUs", door, {}, devices, 40, 40, algorithmic_bytes(8))]
        except Exception as e:  # noqa: BLE001 -- the headline line must still come out
            out["configs"] = [{"workload": "veach-door lmc.xml", "failed": str(e)[:300]}]
    print(json.dumps(out))


# ------------------------------------------------------------------------------------------------ one process per GPU
class FileBoot:
    """Side channel of a job spawned by this script (`python bench.py --gpus N` without a launcher): a private directory the parent made.
    Only two things ever cross it: the 128-byte RCCL id of a job (rank 0 writes it, atomically; the others wait for the file) and, in the
    dry run, the ranks' reports.  Everything else -- barriers, max-over-ranks timing, the per-rank figures -- goes through the job's own
    communicator inside the library (lmc_comm_barrier / lmc_comm_allreduce_f64)."""

    kind = "file"

    def __init__(self, rank, world, directory):
        self.rank, self.world, self.dir, self.seq = rank, world, directory, 0

    def share_id(self, make_id, timeout=300.0):
        path = os.path.join(self.dir, "id_%d" % self.seq)
        self.seq += 1
        if self.rank == 0:
            data = bytes(make_id())
            with open(path + ".tmp", "wb") as f:
                f.write(data)
            os.rename(path + ".tmp", path)  # atomic: a reader sees all 128 bytes or no file
            return data
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > timeout:
                die("rank %d: no RCCL id from rank 0 after %.0f s (%s)" % (self.rank, timeout, path), 3)
            time.sleep(0.005)
        return open(path, "rb").read()

    def gather_reports(self, rep, timeout=120.0):
        """dry run only: every rank's report on rank 0 (files in the private directory)"""
        mine = os.path.join(self.dir, "report_%d.json" % self.rank)
        with open(mine + ".tmp", "w") as f:
            json.dump(rep, f)
        os.rename(mine + ".tmp", mine)
        if self.rank != 0:
            return None
        reps, t0 = [], time.time()
        for k in range(self.world):
            pth = os.path.join(self.dir, "report_%d.json" % k)
            while not os.path.exists(pth):
                if time.time() - t0 > timeout:
                    die("dry run: no report from rank %d" % k, 3)
                time.sleep(0.01)
            reps.append(json.load(open(pth)))
        return reps

    def close(self):
        pass


class TorchBoot:
    """Side channel under `python -m torch.distributed.run` (the driver's N > 1 command): the launcher's own rendezvous, through a gloo
    process group -- used for the id hand-off only, like FileBoot."""

    kind = "torch.distributed (gloo)"

    def __init__(self, rank, world):
        import torch.distributed as dist

        self.rank, self.world, self.dist = rank, world, dist
        dist.init_process_group("gloo")

    def share_id(self, make_id, timeout=300.0):
        obj = [bytes(make_id()) if self.rank == 0 else None]
        self.dist.broadcast_object_list(obj, 0)
        return obj[0]

    def gather_reports(self, rep, timeout=120.0):
        out = [None] * self.world
        self.dist.all_gather_object(out, rep)
        return out if self.rank == 0 else None

    def close(self):
        self.dist.destroy_process_group()


def rank_job(args, p, gc, boot, rank, world, local, name, xml, kw, warm, steps, algo_bytes, kernel_name, standalone_ok):
    """One workload on this rank's GPU as rank `rank` of a `world`-rank job: scene, communicator (world > 1: RCCL inside the library, its id over
    `boot`), sharded MLTInit, `warm` untimed steps, then EXACTLY `steps` steps + the film all-reduce bracketed by a barrier + stream sync on
    both sides; the time is the MAX over the ranks.  Returns the job's figures on every rank (per-rank lists gathered through the communicator)."""
    import numpy as np  # noqa: F401

    sharding = importlib.import_module("langevin-mcmc_amd.sharding")
    per_gpu, total = args.chains, args.chains * world
    init_samples = args.init_samples or 8 * total
    ren = p.Renderer(xml, seed_offset=0, device=local, use_gradient=1, **kw)
    multi = world > 1 or bool(os.environ.get("LMC_BENCH_FORCE_DIST"))
    if multi:
        ren.comm_init(world, rank, boot.share_id(p.comm_unique_id))  # a failure here is fatal (exit 1 with the library's message): no fallback transport

    def barrier():
        if multi:
            ren.comm_barrier()  # own stream drained, then a one-word all-reduce over the job's communicator
        else:
            ren.sync()

    t_init = time.time()
    chain_begin, chain_end = sharding.chain_range(rank, world, per_gpu)  # contiguous global chain ids, weak scaling: per_gpu chains on every rank
    norm, ncontrib = ren.init_chains(init_samples, total, args.init_threads, args.samples_per_chain, 0, chain_begin, chain_end)
    t_init = time.time() - t_init
    ren.set_option("timing", 1)  # per-step HIP events on the launch streams (lmc_step_timing / lmc_kernel_timing)
    ren.step(warm)
    ren.step_timing()  # discard warm-up launches
    ren.host_issue_timing()
    k0 = ren.kernel_timing_split()
    s0 = ren.stats()
    barrier()
    t0 = time.time()
    ren.step(steps)
    if multi:
        ren.film_allreduce()  # the data-path collective: the device films summed in place (RCCL over xGMI, on the step stream, no host staging)
    barrier()
    dt = time.time() - t0
    if multi:
        dt = ren.comm_allreduce([dt], "max")[0]
    kernel_ms, launches = ren.step_timing()
    k1 = ren.kernel_timing_split()
    issue_ms, issue_steps = ren.host_issue_timing()
    s1 = ren.stats()
    launches = max(launches, 1)
    lean_ms = k1["lean_ms"] / launches
    large_ms, generic_ms = k1["large_ms"] / launches, k1["generic_ms"] / launches
    steps_rank = s1["steps"] - s0["steps"]
    dom, dom_ms, dom_steps, achieved = dominant_kernel(k0, k1, s0, s1, launches, algo_bytes, kernel_name)  # of THIS rank
    standalone = None
    extra = max(0, 48 - (warm + steps))  # a short window (the driver's 5 + 20) ends inside the start-up: run past it, untimed, first
    if standalone_ok and not multi and warm + steps + extra + 20 <= args.samples_per_chain:
        # outside the timed region: the dominant kernel with the GPU to itself (the large-step launch normally runs beside it on another stream)
        if extra:
            ren.step(extra)
            ren.step_timing()
        ren.set_option("overlap", 0)
        ren.step(4)
        ren.step_timing()
        lean0 = ren.kernel_timing()[2]
        ren.step(16)
        _, n_sa = ren.step_timing()
        sa_small_ms, _, lean1 = ren.kernel_timing()
        ren.set_option("overlap", 1)
        standalone = (sa_small_ms / max(n_sa, 1), (lean1 - lean0) / max(n_sa, 1))
    out = {
        "workload": name, "value": steps * total / dt, "unit": "chain-steps/s", "ms_per_step": dt * 1e3 / steps, "steps": steps, "warmup": warm,
        "bvh_nodes": "quantised 64 B" if ren.get_option("bvh_quantised") else "exact 128 B", "bvh_thick_flat_share": ren.get_option("bvh_thick_flat_share"),
        "value_from_step_counter": steps_rank / dt, "init_seconds": t_init, "normalization": norm, "init_samples": init_samples, "film": [ren.width, ren.height],
        "accept_rate": (s1["accepted"] - s0["accepted"]) / max(steps_rank, 1), "large_step_frac": (s1["largeSteps"] - s0["largeSteps"]) / max(steps_rank, 1),
        "cache_queries_per_step": (s1["cacheQueries"] - s0["cacheQueries"]) / max(steps_rank, 1), "cache_hits_per_query": (s1["cacheHits"] - s0["cacheHits"]) / max(s1["cacheQueries"] - s0["cacheQueries"], 1),
        "step_ms": {"all_launches": kernel_ms / launches, "k_step_small": lean_ms, "large_and_generic": large_ms + generic_ms},
        "host_issue_ms_per_step": issue_ms / max(issue_steps, 1),
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "kernel": dom, "avg_launch_ms": dom_ms,
                     "chain_steps_per_launch": dom_steps, "algorithmic_bytes_per_step": algo_bytes,
                     "concurrent_launches": "the step's other launches run beside this kernel on their own streams inside the bracket"},
    }
    if standalone is not None and standalone[0] > 0:
        sa_ach = algo_bytes * standalone[1] / (standalone[0] * 1e-3) / 1e9
        out["roofline"]["standalone"] = {"avg_launch_ms": standalone[0], "chain_steps_per_launch": standalone[1], "achieved": sa_ach, "frac": sa_ach / HBM_PEAK_GBS,
                                         "note": "same kernel, 16 launches after the timed region (and, for a window shorter than 48 steps, after enough further untimed steps to be past the cache fill) with the side launches serialised"}
    try:  # how the resident chains are laid out (device/relocate.hip): grouped by technique unless LMC_RELOCATE=0
        rs = ren.relocation_stats()
        out["chain_relocation"] = ({"on": True, "relocations": rs["relocations"], "chains_moved_by_the_last": rs["moved"],
                                    "technique_breaks_along_the_slots": rs["breaks"], "slots": rs["slots"]} if rs else {"on": False})
    except Exception as e:  # noqa: BLE001
        out["chain_relocation"] = {"failed": str(e)}
    if multi:
        # every rank's own figures, in rank order, on every rank (sums of one-hot vectors over the job's communicator)
        g = lambda v: ren.comm_gather(v, rank, world)
        fr, ms_, st_, iss = g(out["roofline"]["frac"]), g(dom_ms), g(out["step_ms"]["all_launches"]), g(out["host_issue_ms_per_step"])
        out["roofline"].update({"frac_per_rank": fr, "frac_min": min(fr), "frac_max": max(fr), "avg_launch_ms_per_rank": ms_, "frac": sum(fr) / len(fr),
                                "achieved": sum(fr) / len(fr) * HBM_PEAK_GBS, "note": "frac / achieved: mean over the ranks of each rank's own dominant-kernel bracket; rank 0's kernel named"})
        out["multi_gpu"] = {"rccl_ranks": world, "per_rank_step_ms": st_, "per_rank_step_ms_spread": max(st_) - min(st_), "host_issue_ms_per_step_per_rank": iss,
                            "film_sum": float(ren.film().sum()) if rank == 0 else None, "boot": boot.kind if boot else None,
                            "collective": "in-library RCCL (librccl, one communicator per job, created from a 128-byte id): MLTInit sharded by init stream (ncclAllGather x 4), "
                                          "per-step ncclAllGather of the cache pushes while the gradient caches fill, ONE ncclAllReduce of the device film + the splat-weight scalar "
                                          "inside the timed region (lmc_film_allreduce); barriers / max-over-ranks timing: ncclAllReduce of host scalars (lmc_comm_allreduce_f64)"}
    ren.close()
    return out


def dominant_kernel(k0, k1, s0, s1, launches, algo_bytes, lean_name):
    """the step launch with the longest HIP-event bracket over the timed launches and its roofline figures:
    (name, avg launch ms, chain-steps per launch, achieved GB/s of algorithmic bytes)"""
    launches = max(launches, 1)
    steps_rank = s1["steps"] - s0["steps"]
    large_steps = (s1["largeSteps"] - s0["largeSteps"]) / launches
    lean_ms, lean_steps = k1["lean_ms"] / launches, (k1["lean_steps"] - k0["lean_steps"]) / launches
    ker = {lean_name: (lean_ms, lean_steps), "k_step<large>": (k1["large_ms"] / launches, large_steps),
           "generic small-step launches (cache-filling gradient pipeline / H2MC pipeline)": (k1["generic_ms"] / launches, steps_rank / launches - large_steps - lean_steps)}
    dom = max(ker, key=lambda k: ker[k][0])
    dom_ms, dom_steps = ker[dom]
    return dom, dom_ms, dom_steps, (algo_bytes * dom_steps / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0)


def lmc_env():
    """every LMC_* switch in the environment: a number measured with one set must say so (INTEGRATION.md has the table of switches)"""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("LMC_")}


WORK_SKIPPING = ("LMC_EXP_NOSPLAT", "LMC_EXP_NOQUERY", "LMC_EXP_NOGRAD", "LMC_EXP_NOSTATS", "LMC_EXP_NOHESS", "LMC_EXP_NOEIGEN", "LMC_EXP_NOHESSLAUNCH", "LMC_EXP_QUERY_STOP")


def main_rank(args):
    """one rank of the job: the whole of a one-GPU run; rank r of `--gpus N` when spawned by main_spawn or by torch.distributed.run"""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not os.environ.get("LMC_BENCH_FORCE_DIST"):
        die("--gpus %d but the launcher started %d rank(s) (WORLD_SIZE): launch with --nproc-per-node %d, or without a launcher (bench.py then starts one process per GPU itself)" % (args.gpus, world, args.gpus))
    skipping = [k for k in WORK_SKIPPING if os.environ.get(k, "0") not in ("", "0")]
    if skipping and not os.environ.get("LMC_BENCH_ALLOW_EXP"):
        die("work-skipping measurement switches set (%s): such a run is an ablation, not a benchmark -- run it through scripts/pmc_ab.sh / scripts/ab_bench.sh (LMC_BENCH_ALLOW_EXP=1), which label it" % ", ".join(skipping))
    dry = bool(os.environ.get("LMC_BENCH_DRY_RUN"))
    boot = None
    if "LMC_BENCH_BOOT" in os.environ:
        boot = FileBoot(rank, world, os.environ["LMC_BENCH_BOOT"])
    elif world > 1 or os.environ.get("LMC_BENCH_FORCE_DIST"):
        boot = TorchBoot(rank, world)
    if dry:  # CPU test of the launch path (tests/test_host.py): the id hand-off of two consecutive jobs, no GPU work
        import hashlib

        ids = [boot.share_id(lambda: os.urandom(128)) for _ in range(2)]
        sharding = importlib.import_module("langevin-mcmc_amd.sharding")
        rep = {"rank": rank, "world": world, "local": local, "ids": [hashlib.sha256(i).hexdigest() for i in ids], "lens": [len(i) for i in ids],
               "scaling": args.scaling, "chains_per_gpu": args.chains, "chain_range": list(sharding.chain_range(rank, world, args.chains)), "chains_total": args.chains * world}
        reps = boot.gather_reports(rep)
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "boot": boot.kind, "ranks": reps, "ids_equal": all(r["ids"] == reps[0]["ids"] for r in reps),
                              "ids_distinct_per_job": reps[0]["ids"][0] != reps[0]["ids"][1]}))
        boot.close()
        return
    p = importlib.import_module("langevin-mcmc_amd")
    from tests import gpu_checks as gc

    oversubscribed = bool(world > p.device_count() and os.environ.get("LMC_BENCH_OVERSUBSCRIBE") and os.environ.get("LMC_RCCL_LIB"))
    if p.device_count() <= local:
        if p.device_count() >= 1 and os.environ.get("LMC_BENCH_OVERSUBSCRIBE") and os.environ.get("LMC_RCCL_LIB"):  # test aid (main_spawn): ranks share the devices
            local, oversubscribed = local % p.device_count(), True
        else:
            die("rank %d: local device %d is not visible (%d HIP device(s))" % (rank, local, p.device_count()))
    multi = world > 1 or bool(os.environ.get("LMC_BENCH_FORCE_DIST"))
    head_name = "torus scene, %d persistent chains per GPU, Lambertian-only BSDF, max path length 6 (BASELINE.json configs[1])" % args.chains
    head = rank_job(args, p, gc, boot, rank, world, local, head_name, gc.TORUS, dict(force_diffuse=1, max_depth=6), args.warmup, args.steps, ALGO_BYTES_PER_STEP,
                    "k_step_small<true, false, false, true, true> (rocprofv3's name: LDS stack, Lambertian, no region profiling, no light sub-paths, quantised nodes; plain small steps of a scene lit by its environment map alone)", True)
    door = None
    if multi and not args.no_configs:  # north_star names both scenes at 1 / 2 / 4 / 8 GPUs: the veach-door LMC workload as the job's second line
        try:
            door = rank_job(args, p, gc, boot, rank, world, local, "veach-door, shipped lmc.xml (area light, textures, max path length 8), LMC (BASELINE.json configs[3]), chains sharded over the GPUs",
                            os.path.join(ROOT, "scenes", "veachdoor", "lmc.xml"), {}, 40, 40, algorithmic_bytes(8), "k_step_small<glossy> (plain small steps)", False)
        except Exception as e:  # noqa: BLE001 -- the headline line must still come out
            door = {"workload": "veach-door lmc.xml", "failed": str(e)[:300]}
    if rank == 0:
        value = head["value"]
        roof = head["roofline"]
        if roof["kernel"].startswith("k_step_small"):  # the counter figure belongs to the lean kernel
            roof["traffic"] = pmc_traffic()
            roof["traffic_measured_at"] = pmc_traffic_meta()
            roof["traffic_source"] = ("REPLAYED from the committed rocprofv3 --pmc summary profiles/pmc_step_kernel.json (gated on the fingerprint of the kernel's sources); "
                                      "not counted during this run: counters need their own rocprofv3 passes (scripts/pmc_passes.sh)")
        else:
            roof["traffic"] = None
        lw = lanes_by_workload().get("torus6")
        if lw:
            roof["lanes_active"] = {"kernels": {k: x["lanes_active"] for k, x in lw["kernels"].items()},
                                    "source": "REPLAYED from profiles/lanes_by_workload.json (%s): lanes per issued vector instruction, not counted during this run" % lw["source"]}
        out = {
            "metric": "MALA chain-steps/sec, torus scene",
            "value": value,
            "unit": "chain-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic chains on the shipped torus geometry + sunsky env map (random-seeded PCG streams)",
            "config": {
                "workload": head_name,
                "chains_per_gpu": args.chains,
                "init_samples": head["init_samples"],
                "samples_per_chain": args.samples_per_chain,
                "timed_steps_of_the_population": "steps %d..%d of a fresh population.  What those are: steps 0-4 nearly every chain takes a large step; the gradient caches of dims 6 / 8 "
                                                 "are ready after step 5, those of dims 10 / 12 after step 22 -- until then ~15 k chains per step evaluate gradients (the fill pipeline "
                                                 "beside the hot launch); from step 23 on >= 99.99 %% of the proposals are the isotropic ones of mutation_mala.h:131-164 (cache misses), and "
                                                 "from 10 %% of a chain's samples on (step %d here) its large-step probability is 0.2 instead of 0.05 (mlt.cpp:96-97).  The driver's "
                                                 "--steps 20 --warmup 5 window is the fill phase; --steps 64 --warmup 40 is the steady state" % (args.warmup, args.warmup + args.steps - 1, int(0.1 * args.samples_per_chain) + 1),
                "film": head["film"],
                "bvh_nodes": head["bvh_nodes"],
                "parallelism": "chains sharded x%d, one process per GPU" % world,
                "collective": head["multi_gpu"]["collective"] if multi else "none (one GPU)",
                "rccl_ranks": world if multi else 0,
                **({"oversubscribed": True} if oversubscribed else {}),
                **({"rccl_library": os.environ["LMC_RCCL_LIB"] + " (NOT RCCL: a stand-in over host shared memory, tests only)"} if os.environ.get("LMC_RCCL_LIB") else {}),
                "launch": ("spawned by bench.py, id over a private directory" if boot and boot.kind == "file" else "torch.distributed.run, id over its gloo rendezvous") if multi else "single process",
            },
            "roofline": roof,
            "step_ms": head["step_ms"],
            "host_issue_ms_per_step": head["host_issue_ms_per_step"],
            "init_seconds": head["init_seconds"],
            "normalization": head["normalization"],
            "value_from_step_counter": head["value_from_step_counter"],
            "accept_rate": head["accept_rate"],
            "large_step_frac": head["large_step_frac"],
            "cache_queries_per_step": head["cache_queries_per_step"],
            "cache_hits_per_query": head["cache_hits_per_query"],
            "chain_relocation": head.get("chain_relocation"),
            "lmc_env": lmc_env(),
        }
        if multi:
            out["multi_gpu"] = head["multi_gpu"]
            if door is not None:
                out["configs"] = [door]
        if not args.no_configs and not multi:
            out["configs"] = other_configs(args, p, gc)
        if not args.no_cpu_baseline and not multi:
            try:
                out["cpu_baseline"] = cpu_baseline(args)
            except Exception as e:  # the bench line must still come out
                out["cpu_baseline"] = {"value": None, "unit": "chain-steps/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %s" % e}
            if not args.no_rmse and out["cpu_baseline"].get("value"):
                try:
                    out["equal_time_rmse"] = equal_time_rmse(args, p, gc, value, out["cpu_baseline"]["value"], out["cpu_baseline"].get("threads", out["cpu_baseline"]["cores"]))
                except Exception as e:
                    out["equal_time_rmse"] = {"failed": str(e)}
        print(json.dumps(out))
        sys.stdout.flush()
    if boot:
        boot.close()


def main_spawn(args):
    """`python bench.py --gpus N` without a launcher: this process starts N workers, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE in their
    environment, a private directory as the side channel of the RCCL id), relays rank 0's JSON line and waits for all of them.  It touches no
    GPU itself beyond counting the devices."""
    import shutil
    import subprocess
    import tempfile

    dry = bool(os.environ.get("LMC_BENCH_DRY_RUN"))
    if not dry:
        p = importlib.import_module("langevin-mcmc_amd")
        have = p.device_count()
        # bring-up / test aid ONLY: with a stand-in for RCCL that accepts several ranks per device (LMC_RCCL_LIB, tests/helpers/rccl_stub.cpp) AND
        # LMC_BENCH_OVERSUBSCRIBE=1 the N rank processes share the visible devices; the line says so (`oversubscribed`, `rccl_library`)
        stub_ok = have >= 1 and os.environ.get("LMC_BENCH_OVERSUBSCRIBE") and os.environ.get("LMC_RCCL_LIB")
        if have < args.gpus and not stub_ok:
            die("--gpus %d but only %d HIP device(s) visible to this process: refusing to measure a smaller job under that name "
                "(RCCL cannot place two ranks on one device; for bring-up on fewer devices: --in-process with LMC_BENCH_OVERSUBSCRIBE=1)" % (args.gpus, have))
    boot_dir = tempfile.mkdtemp(prefix="lmc_bench_")
    procs = []
    try:
        for k in range(args.gpus):
            env = dict(os.environ, RANK=str(k), LOCAL_RANK=str(k), WORLD_SIZE=str(args.gpus), LMC_BENCH_BOOT=boot_dir)
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL's peer-to-peer transport needs on this host driver
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=subprocess.PIPE if k == 0 else subprocess.DEVNULL))
        line, code = None, 0
        out0 = procs[0].stdout
        pending = set(range(args.gpus))
        import threading

        buf = []
        th = threading.Thread(target=lambda: buf.extend(out0.read().decode().splitlines()), daemon=True)
        th.start()
        while pending:
            for k in list(pending):
                rc = procs[k].poll()
                if rc is None:
                    continue
                pending.discard(k)
                if rc != 0:  # one rank failed: the others would wait in a collective for ever
                    code = code or rc
                    sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other ranks\n" % (k, rc))
                    for q in pending:
                        procs[q].terminate()
            time.sleep(0.05)
        th.join(timeout=10)
        for l in buf:
            if l.startswith("{"):
                line = l
        if code:
            sys.exit(code)
        if line is None:
            die("rank 0 printed no JSON line", 3)
        print(line)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
        shutil.rmtree(boot_dir, ignore_errors=True)


def main():
    args = parse()
    if args.gpus < 1:
        die("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("LMC_BENCH_FORCE_SPAWN")):  # the switch: the spawned form with ONE rank (tests on a one-GPU box)
        return main_inprocess(args) if args.in_process else main_spawn(args)
    return main_rank(args)


if __name__ == "__main__":
    main()
