# usage (GPU box): scripts/window_trace.sh <tag> [step numbers...]  -- kernel trace of the first 30 steps of the headline workload, issued without a
# per-step wait (as bench.py's timed region does); prints the timelines of the steps asked for (default 12 and 27)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O; shift
cat > /tmp/wt.py <<'PY'
import importlib, os, sys
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_checks as gc
p = importlib.import_module("langevin-mcmc_amd")
chains = 1 << 20
ren = p.Renderer(gc.TORUS, force_diffuse=1, max_depth=6, seed_offset=0, device=0, use_gradient=1)
ren.init_chains(8 * chains, chains, 65536, 256, 0, 0, chains)
ren.sync()
ren.step(30)
ren.sync()
PY
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python /tmp/wt.py > $O/trace.log 2>&1 )
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
for st in ${@:-12 27}; do python scripts/kernel_trace_summary.py $f 8 $st | sed -n '/^step/,$p' > $O/timeline_step_$st.txt; done
python scripts/kernel_trace_summary.py $f 8 | sed -n '1,/^step/p' > $O/kernels.txt
rm -rf $O/trace
cat $O/timeline_step_*.txt
