#!/usr/bin/env python3
"""Per-step durations from a rocprofv3 --kernel-trace csv: a step = from one k_build_lists launch to the next; for every step its length,
the hot launch (k_step_small) and the sum / span of the side launches.  usage: step_durations.py <dir or csv> [first=1] [last=30]"""
import csv, glob, os, sys

src = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
last = int(sys.argv[3]) if len(sys.argv) > 3 else 30
f = src if src.endswith(".csv") else sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda k: k.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("lmcd::", "")
marks = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]).startswith("k_build_lists")]
print("%4s %8s %8s %8s %8s  %s" % ("step", "ms", "hot", "large", "side", "side launches (ms)"))
for k in range(max(first, 1), min(last, len(marks) - 1) + 1):
    a, b = marks[k - 1], marks[k]
    t0, t1 = int(rows[a]["End_Timestamp"]), int(rows[b]["End_Timestamp"])
    hot = large = 0.0
    side = []
    s0, s1 = None, None
    for r in rows[a + 1:b]:
        n, d = short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        if n.startswith("k_step_small<"):
            hot = d
        elif n.startswith("k_step<"):
            large = d
        elif n.startswith(("k_mala_", "k_h2_", "k_step_small_grad")):
            side.append("%s %.2f" % (n.split("<")[0].replace("k_mala_", "").replace("k_h2_", ""), d))
            s0 = int(r["Start_Timestamp"]) if s0 is None else s0
            s1 = int(r["End_Timestamp"])
    print("%4d %8.3f %8.3f %8.3f %8.3f  %s" % (k, (t1 - t0) / 1e6, hot, large, (s1 - s0) / 1e6 if s0 else 0.0, ", ".join(side)))
