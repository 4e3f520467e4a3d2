# A/B: per-dim zero-filled grid scratch (no fills inside the step that builds a cache's existence grid) against the shared scratch (LMC_GRID_SCRATCH_SHARED=1)
mkdir -p gpurun_out/bo
scripts/ab_bench.sh gpurun_out/bo/window.jsonl -s 20 -w 5 -- "LMC_GRID_SCRATCH_SHARED=1" "-" "LMC_GRID_SCRATCH_SHARED=1" "-" 2>/dev/null | cut -c1-160
LMC_GRID_SCRATCH_SHARED=1 timeout 300 python scripts/step_trace.py 26 > gpurun_out/bo/step_trace_shared.jsonl 2>/dev/null
timeout 300 python scripts/step_trace.py 26 > gpurun_out/bo/step_trace_own.jsonl 2>/dev/null
python - <<'PY'
import json
a=[json.loads(l) for l in open('gpurun_out/bo/step_trace_shared.jsonl')]; b=[json.loads(l) for l in open('gpurun_out/bo/step_trace_own.jsonl')]
print('step 5', a[5]['wall_ms'], b[5]['wall_ms'], 'step 22', a[22]['wall_ms'], b[22]['wall_ms'], 'sum 5..24', round(sum(x['wall_ms'] for x in a[5:25]),2), round(sum(x['wall_ms'] for x in b[5:25]),2))
PY
timeout 800 python -m pytest tests -m gpu -q -x > gpurun_out/bo/pytest_gpu.txt 2>&1; grep -E "passed|failed" gpurun_out/bo/pytest_gpu.txt
