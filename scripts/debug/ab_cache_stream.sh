# A/B of the cache-ready transition beside the step (host/context.cpp CacheApplyFinish; LMC_NO_CACHE_STREAM=1 = the former order: behind the hot launch)
mkdir -p gpurun_out/bh
LMC_NO_CACHE_STREAM=1 timeout 300 python scripts/step_trace.py 30 > gpurun_out/bh/step_trace_behind.jsonl 2>/dev/null
timeout 300 python scripts/step_trace.py 30 > gpurun_out/bh/step_trace_beside.jsonl 2>/dev/null
python - <<'PY'
import json
a=[json.loads(l) for l in open('gpurun_out/bh/step_trace_behind.jsonl')]; b=[json.loads(l) for l in open('gpurun_out/bh/step_trace_beside.jsonl')]
for x,y in zip(a,b):
    if abs(x['wall_ms']-y['wall_ms'])>0.15: print('step',x['step'],'behind',x['wall_ms'],'beside',y['wall_ms'])
print('sum steps 5..24: behind %.2f ms, beside %.2f ms' % (sum(x['wall_ms'] for x in a[5:25]), sum(y['wall_ms'] for y in b[5:25])))
PY
scripts/ab_bench.sh gpurun_out/bh/window.jsonl -s 20 -w 5 -- "LMC_NO_CACHE_STREAM=1" "-" "LMC_NO_CACHE_STREAM=1" "-" 2>/dev/null | cut -c1-200
timeout 800 python -m pytest tests -m gpu -q -x > gpurun_out/bh/pytest_gpu.txt 2>&1; grep -E "passed|failed" gpurun_out/bh/pytest_gpu.txt
