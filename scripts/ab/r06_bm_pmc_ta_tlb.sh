#!/bin/bash
# one-off (two counters per group: larger groups of these blocks are refused -- "exceeds the capabilities of the hardware" -- and the refused run hangs): the vector memory path's own counters over the default bench (texture addresser busy, L1 stalls by cause, UTCL1 translation hits / misses), one --pmc group per run
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out/r06_bm; mkdir -p $O
CMD="python $PWD/bench.py --no-cpu-baseline --no-rmse --no-configs --steps 32 --warmup 40"
G=("TA_BUSY_avr TA_BUSY_max"
   "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum"
   "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
   "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
   "TA_TOTAL_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum")
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "${G[@]}"; do
  i=$((i+1))
  timeout -s KILL 100 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pass$i -- $CMD > $O/pass$i.log 2>&1
done
cd ${GRAFT_REPO_ROOT:-/root/repo}
python - "$O" <<'PY'
import csv, glob, json, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(O + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if "k_step" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
out = {k: {c: acc[k][c] / cnt[k][c] for c in acc[k]} for k in acc}
json.dump(out, open(O + "/pmc_ta_tlb.json", "w"), indent=1)
for k, v in out.items():
    print(k); print("  ", {c: round(x, 1) for c, x in sorted(v.items())})
PY
rm -rf $O/pass*/
