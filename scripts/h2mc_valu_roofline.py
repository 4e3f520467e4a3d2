#!/usr/bin/env python3
"""The H2MC pipeline's governing fraction: its dominant launch k_h2_hess is arithmetic-bound, an HBM fraction says nothing about it.
From a PMC summary of the H2MC workload (scripts/session.sh ... h2mc_pmc=door 20 -> pmc_<n>.json) this writes profiles/h2mc_valu_roofline.json:
per pipeline kernel the share of a SIMD's cycles in which its vector ALU issues (SQ_INSTS_VALU x 4 cycles / (SQ_BUSY_CYCLES / 32 shader engines x 1024
SIMDs): the same arithmetic as DESIGN.md §4 uses for the lean kernel) and the lanes active per issued instruction (SQ_THREAD_CYCLES_VALU / 64
SQ_ACTIVE_INST_VALU).  bench.py replays the file next to the H2MC workload's line, labelled as replayed.
usage: python scripts/h2mc_valu_roofline.py <pmc.json> <scene label> [out.json]   (CPU)"""
import json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(sys.argv[1]))
out = {"source": os.path.relpath(os.path.abspath(sys.argv[1]), ROOT), "scene": sys.argv[2], "kernels": {},
       "definition": "valu_busy = SQ_INSTS_VALU x 4 / (SQ_BUSY_CYCLES / 32 x 1024); lanes_active = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); counters summed over all launches of the run"}
for k, v in d.items():
    if "k_h2_" not in k or not v.get("SQ_BUSY_CYCLES"):
        continue
    name = k.split("(")[0].split("::")[-1].strip()
    out["kernels"][name] = {"valu_busy": round(v["SQ_INSTS_VALU"] * 4 / (v["SQ_BUSY_CYCLES"] / 32 * 1024), 4), "lanes_active": round(v["SQ_THREAD_CYCLES_VALU"] / (64 * v["SQ_ACTIVE_INST_VALU"]), 4),
                            "valu_instructions_per_launch": v["SQ_INSTS_VALU"], "launches": v.get("launches")}
dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "h2mc_valu_roofline.json")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out["kernels"]))
