#!/usr/bin/env python3
"""Per-kernel durations from a rocprofv3 --kernel-trace csv: for every kernel the mean over its LAST n launches (steady state), and the
timeline of the last step (or of step number `step`, counted in k_build_lists launches).  usage: kernel_trace_summary.py <dir or csv> [n=8] [step]"""
import csv, glob, os, sys, collections

src = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
f = src if src.endswith(".csv") else sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda k: k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("lmcd::", "")
by = collections.defaultdict(list)
for r in rows:
    by[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
print("%-52s %6s %10s %10s" % ("kernel", "calls", "mean ms", "last-n ms"))
for k, v in sorted(by.items(), key=lambda kv: -sum(e - s for s, e in kv[1][-n:])):
    d = [(e - s) / 1e6 for s, e in v]
    if sum(d) < 0.05:
        continue
    print("%-52s %6d %10.3f %10.3f" % (k[:52], len(d), sum(d) / len(d), sum(d[-n:]) / len(d[-n:])))
# timeline of the last step: everything from the last k_build_lists-but-one on
marks = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]).startswith("k_build_lists")]
if len(sys.argv) > 3 and len(marks) > int(sys.argv[3]):
    marks = marks[:int(sys.argv[3]) + 1]
if len(marks) >= 2:
    a, b = marks[-2], marks[-1]
    t0 = int(rows[a]["End_Timestamp"])
    print("step %d:" % (len(marks) - 1))
    for r in rows[a + 1:b + 1]:
        print("  +%8.3f ms  %-48s %8.3f ms" % ((int(r["Start_Timestamp"]) - t0) / 1e6, short(r["Kernel_Name"])[:48], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
