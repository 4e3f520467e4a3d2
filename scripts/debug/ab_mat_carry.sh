P=$PWD/langevin-mcmc_amd/csrc/_ab
mkdir -p gpurun_out/ab_mat
LMC_LIB=$P/full2/liblmc_hip.so timeout 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_door.py tests/test_gpu_round5.py -m gpu -q -x > gpurun_out/ab_mat/pytest_full2.txt 2>&1; tail -n 2 gpurun_out/ab_mat/pytest_full2.txt
scripts/ab_bench.sh gpurun_out/ab_mat/headline.jsonl -- "LMC_LIB=$P/base/liblmc_hip.so" "LMC_LIB=$P/mat2/liblmc_hip.so" "LMC_LIB=$P/full1/liblmc_hip.so" "LMC_LIB=$P/full2/liblmc_hip.so" 2>/dev/null | cut -c1-260
scripts/ab_configs.sh gpurun_out/ab_mat/configs.jsonl -- "LMC_LIB=$P/base/liblmc_hip.so" "LMC_LIB=$P/full2/liblmc_hip.so" 2>/dev/null | cut -c1-1200
