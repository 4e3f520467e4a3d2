#!/usr/bin/env python3
"""Lanes active per issued vector instruction of the step launches, by bench workload: from PMC summaries of
`scripts/run_one_config.py <workload>` (scripts/session.sh ... H2PMC=short pmc_cmd=...) this writes profiles/lanes_by_workload.json, which bench.py replays next to
each workload's line (labelled as replayed: counters need their own rocprofv3 passes).  lanes_active = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU), counters
summed over all launches of the run (warm-up included).
usage: python scripts/lanes_summary.py torus6=<pmc.json> torus12=<pmc.json> door=<pmc.json>   (CPU)"""
import json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"definition": "lanes_active = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) over all launches of `scripts/run_one_config.py <workload>` at 2^20 chains", "workloads": {}}
for arg in sys.argv[1:]:
    name, path = arg.split("=", 1)
    d = json.load(open(path))
    w = {"source": os.path.relpath(os.path.abspath(path), ROOT), "kernels": {}}
    for k, v in d.items():
        if not ("k_step" in k) or "small_grad" in k or not v.get("SQ_ACTIVE_INST_VALU") or v.get("SQ_INSTS_VALU", 0) < 1e6:
            continue
        short = "lean small steps (k_step_small)" if "k_step_small<" in k else "large steps (k_step<large>)" if "k_step<" in k else k.split("(")[0]
        w["kernels"][short] = {"lanes_active": round(v["SQ_THREAD_CYCLES_VALU"] / (64 * v["SQ_ACTIVE_INST_VALU"]), 4), "valu_instructions_per_launch": v["SQ_INSTS_VALU"], "launches": v.get("launches"), "name": k.split("(")[0]}
    out["workloads"][name] = w
json.dump(out, open(os.path.join(ROOT, "profiles", "lanes_by_workload.json"), "w"), indent=1)
print(json.dumps({n: {k: x["lanes_active"] for k, x in w["kernels"].items()} for n, w in out["workloads"].items()}))
