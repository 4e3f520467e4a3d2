#!/usr/bin/env python3
"""Upper bound of what a finer slot order buys the step launches: at steady state the resident chains are fully
re-sorted by a fine key worked out on the host (lmc_set_option "exp_resort", relocate.hip k_reloc_finekey) and the lean / large-step launches
are timed ALONE (side launches serialised) step by step before and after -- the decay shows how long a placement stays good.
usage: python scripts/finesort_probe.py [workload: torus|fullmat|door] [log2 chains] [samples per chain] [warm-up steps] [modes, comma separated] > out.jsonl   (GPU)"""
import importlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_checks as gc

p = importlib.import_module("langevin-mcmc_amd")
work = sys.argv[1] if len(sys.argv) > 1 else "torus"
chains = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
if work == "torus":
    ren = p.Renderer(gc.TORUS, force_diffuse=1, max_depth=6, seed_offset=0, device=0, use_gradient=1)
elif work == "fullmat":
    ren = p.Renderer(gc.TORUS, force_diffuse=0, max_depth=12, seed_offset=0, device=0, use_gradient=1)
else:
    ren = p.Renderer(os.path.join(ROOT, "scenes", "veachdoor", "lmc.xml"), seed_offset=0, device=0, use_gradient=1)
spc = int(sys.argv[3]) if len(sys.argv) > 3 else 1024  # samples per chain: the large-step probability is scaled x4 from 10 % of them on (mlt.cpp:96-97)
warm = int(sys.argv[4]) if len(sys.argv) > 4 else 60
modes = [int(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else [0, 1, 2, 3, 0]
ren.init_chains(8 * chains, chains, 65536, spc, 0, 0, chains)
ren.step(warm)
ren.sync()
ren.set_option("timing", 1)
ren.set_option("overlap", 0)


def one(tag, k):
    ren.step(1)
    ren.sync()
    ren.step_timing()
    t = ren.kernel_timing_split()
    return {"tag": tag, "k": k, "lean_ms": round(t["lean_ms"], 4), "large_ms": round(t["large_ms"], 4), "lean_steps": t["lean_steps"]}


prev = None
for mode in modes:
    if mode:
        ren.set_option("exp_resort", mode)
        ren.step(1)  # the sort runs at the end of this step
        ren.sync()
        ren.step_timing()
    rows = [one("mode%d" % mode, k) for k in range(16)]
    for r in rows:
        d = r["lean_steps"] - (prev if prev is not None else r["lean_steps"])
        prev = r["lean_steps"]
        r["lean_chain_steps"] = d
        print(json.dumps(r), flush=True)
    print(json.dumps({"summary": "mode%d" % mode, "lean_ms_first4": round(sum(r["lean_ms"] for r in rows[:4]) / 4, 4), "lean_ms_last4": round(sum(r["lean_ms"] for r in rows[-4:]) / 4, 4),
                      "large_ms_first4": round(sum(r["large_ms"] for r in rows[:4]) / 4, 4), "large_ms_last4": round(sum(r["large_ms"] for r in rows[-4:]) / 4, 4), "reloc": ren.relocation_stats()}), flush=True)
