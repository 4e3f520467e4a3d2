#!/bin/bash
# full GPU suite + the round's measurements of record
OUT=gpurun_out/r03_final; mkdir -p $OUT
timeout 3000 python -m pytest tests/ -q -m gpu 2>&1 | tail -15 > $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1
bash scripts/final_measure.sh $OUT/measure > $OUT/final_measure.log 2>&1
