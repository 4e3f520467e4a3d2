# usage (GPU box): scripts/h2mc_quick.sh <tag> [tests]  -- H2MC rates of the two shipped scenes at 2^20 chains, twice, + a kernel trace of one door step
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
if [ "${2:-}" = tests ]; then timeout 900 python -m pytest tests/test_gpu_h2mc.py tests/test_gpu_relocate.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt; fi
for rep in 1 2; do for sc in door torus; do
  timeout 300 python scripts/h2mc_rates.py $sc 20 24 8 2>>$O/err.txt | tee -a $O/rates.jsonl
done; done
( cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python scripts/h2mc_rates.py door 20 24 8 > $O/trace.log 2>&1 )
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python scripts/kernel_trace_summary.py $f > $O/timeline.txt 2>&1
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/trace
sed -n '/^step/,$p' $O/timeline.txt | head -60
