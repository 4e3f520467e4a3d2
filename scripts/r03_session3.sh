#!/bin/bash
OUT=gpurun_out/r03_c; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_h2mc.py::test_h2mc_chain_parity_diffuse "tests/test_gpu_parity.py::test_cfg1_twin_four_chains_thousand_mutations" "tests/test_gpu_parity.py::test_point_light_scene_chain_parity" -q 2>&1 | grep -E "^E  |assert|Error|passed|failed" | head -60 > $OUT/failing_tests.txt
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $OUT/gputests_split.txt
LMC_LEAN_SPLIT=0 timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -8 > $OUT/gputests_fused.txt
timeout 1500 bash scripts/ab_bench.sh $OUT/ab_lean_split.jsonl -- "LMC_LEAN_SPLIT=0" "LMC_LEAN_SPLIT=1" 2> $OUT/ab_lean_split.err
timeout 600 bash scripts/ab_bench.sh $OUT/ab_lean_split_driver_window.jsonl -s 20 -w 5 -- "LMC_LEAN_SPLIT=0" "LMC_LEAN_SPLIT=1" 2>> $OUT/ab_lean_split.err
