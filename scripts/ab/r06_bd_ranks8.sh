#!/bin/bash
# one-off: the rank code path with 4 and 8 rank PROCESSES on the one GPU over the RCCL stand-in, both launch forms of bench.py --gpus N (spawned ranks and
# torch.distributed.run), weak and strong; the numbers mean nothing (ranks share one GPU through host memory), the lines' shape and exit codes do
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_bd; mkdir -p $O
STUB=$(python -c "import sys; sys.path.insert(0,'tests'); import gpu_checks as gc; print(gc.rccl_stub_lib())")
export LMC_RCCL_LIB=$STUB LMC_BENCH_OVERSUBSCRIBE=1
ARGS="--chains 65536 --steps 20 --warmup 5 --no-configs --no-cpu-baseline --no-rmse"
for n in 4 8; do
  for sc in weak strong; do
    echo "== spawned ranks N=$n $sc"; timeout 600 python bench.py --gpus $n $ARGS --scaling $sc 2> $O/spawn_${n}_$sc.err | tail -1 | tee -a $O/lines.jsonl | cut -c1-300; echo "rc=$?"
  done
  echo "== torch.distributed.run N=$n"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $n $ARGS 2> $O/torchrun_$n.err | tail -1 | tee -a $O/lines.jsonl | cut -c1-300
done
