#!/usr/bin/env python3
"""VERDICT r4 next item 3b, measured before built: would software-pipelining two halves of the chain population through the H2MC pipeline (half A's
ALU-bound Hessian launch beside half B's latency-bound phases) pay?  Two contexts on ONE device, each with half of the chains, joined as an
in-process group (lmc_group_*: every member has its own streams, a host thread each, nothing orders one member's launches against the other's) ARE
that schedule: the same 2^lg chains, the same trajectories (a group equals one context chain for chain), two independent pipelines on the GPU.
usage (GPU box): python scripts/debug/h2mc_two_halves.py [door|torus] [log2 chains] [steps] [warmup]"""
import importlib, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
p = importlib.import_module("langevin-mcmc_amd")
which = sys.argv[1] if len(sys.argv) > 1 else "door"
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 20
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 24
warm = int(sys.argv[4]) if len(sys.argv) > 4 else 30
xml = os.path.join(ROOT, "scenes", "veachdoor" if which == "door" else "torus", "h2mc.xml")
kw = {} if which == "door" else dict(max_depth=8)
n = 1 << lg
for parts in (1, 2, 4):
    rens = [p.Renderer(xml, seed_offset=0, device=0, use_gradient=1, **kw) for _ in range(parts)]
    grp = p.Group(rens)
    grp.init_chains(8 * n, n, 65536, warm + steps + 8, 0)
    grp.step(warm)
    for r in rens:
        r.sync()
    t0 = time.time()
    grp.step(steps)
    for r in rens:
        r.sync()
    dt = time.time() - t0
    st = [r.stats() for r in rens]
    print(json.dumps({"scene": which, "chains": n, "contexts_on_one_device": parts, "chain_steps_per_s": n * steps / dt, "ms_per_step": dt * 1e3 / steps,
                      "accepted": sum(s["accepted"] for s in st), "steps": sum(s["steps"] for s in st)}), flush=True)
    for r in rens:
        r.close()
