# A/B: the step's proposal normals drawn through DrawNormals (drng.h; lean TU built with -DLMC_NORMALS_BATCH=1) against one nd(rng) call per dimension
P=$PWD/langevin-mcmc_amd/csrc/_ab
mkdir -p gpurun_out/bm
scripts/ab_bench.sh gpurun_out/bm/steady.jsonl -- "-" "LMC_LIB=$P/nbatch/liblmc_hip.so" "-" "LMC_LIB=$P/nbatch/liblmc_hip.so" 2>/dev/null | cut -c1-330
LMC_LIB=$P/nbatch/liblmc_hip.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_relocate.py -m gpu -q -x > gpurun_out/bm/pytest_nbatch.txt 2>&1; grep -E "passed|failed" gpurun_out/bm/pytest_nbatch.txt
scripts/ab_configs.sh gpurun_out/bm/configs.jsonl -- "-" "LMC_LIB=$P/nbatch/liblmc_hip.so" 2>/dev/null | cut -c1-600
