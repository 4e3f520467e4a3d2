#!/usr/bin/env python3
"""Condense the rocprofv3 --pmc csv output of scripts/pmc_passes.sh: per kernel, the average counter value per launch
(over the launches of the last third of the run, i.e. steady state) and the launch count."""
import csv, glob, json, os, sys, collections

out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
prefix = "calib_*" if len(sys.argv) > 2 else "pass*"
for f in glob.glob(os.path.join(out, prefix, "**", "*counter_collection.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    # one row per (dispatch, counter)
    per = collections.defaultdict(dict)
    for r in rows:
        per[(r["Dispatch_Id"], r["Kernel_Name"])][r["Counter_Name"]] = float(r["Counter_Value"])
    byk = collections.defaultdict(list)
    for (d, k), v in per.items():
        byk[k].append((int(d), v))
    for k, lst in byk.items():
        lst.sort()
        tail = lst[len(lst) * 2 // 3:]
        for name in tail[0][1]:
            res[k][name] = [sum(v.get(name, 0.0) for _, v in tail) / len(tail), len(lst)]
summary = {}
for k, d in res.items():
    short = k.split("(")[0]
    summary[short] = {n: v[0] for n, v in d.items()}
    summary[short]["launches"] = max(v[1] for v in d.values())
print(json.dumps(summary, indent=1))
