#!/bin/bash
# quick A/B: steady-state step time split (no CPU baseline); extra args go to bench.py
python bench.py --no-cpu-baseline --steps 64 --warmup 40 "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value %.1fM  step %.3f ms  small %.3f  large %.3f  accept %.6f' % (d['value']/1e6, d['ms_per_step'], d['step_ms']['k_step_small'], d['step_ms']['large_and_generic'], d['accept_rate']))"
