#!/bin/bash
# A/B: predicated leaf-triangle loads (LMC_LEAF_PRED=1) vs. the duplicate-address loads; step kernels + the closest-hit probe
OUT=gpurun_out/r03_j; mkdir -p $OUT; export TMPDIR=/tmp
V="LMC_LIB=$PWD/langevin-mcmc_amd/csrc/_build/$1/liblmc_hip.so"
bash scripts/ab_bench.sh $OUT/ab_$1.jsonl -- - "$V"
for v in "LMC_X=1" "$V"; do
  tag=$(echo $v | md5sum | cut -c1-6)
  env $v rocprofv3 --kernel-trace --stats -d $OUT/trace_$tag -o t -- python scripts/trace_tcc_probe.py scenes/torus/lmc.xml 20 6 > $OUT/trace_$tag.log 2>&1
  echo "$v" >> $OUT/trace_summary.txt
  find $OUT/trace_$tag -name "*kernel_stats.csv" | head -1 | xargs grep -h "k_trace" >> $OUT/trace_summary.txt
done
