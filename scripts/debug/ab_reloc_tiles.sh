# A/B: chains of a technique sub-ordered by the screen tile of their camera vertex at relocation (relocate.hip LMC_RELOC_TILES)
P=$PWD/langevin-mcmc_amd/csrc/_ab
mkdir -p gpurun_out/bj
scripts/ab_bench.sh gpurun_out/bj/steady.jsonl -- "-" "LMC_LIB=$P/tiles/liblmc_hip.so" "-" "LMC_LIB=$P/tiles/liblmc_hip.so" 2>/dev/null | cut -c1-330
scripts/ab_bench.sh gpurun_out/bj/window.jsonl -s 20 -w 5 -- "-" "LMC_LIB=$P/tiles/liblmc_hip.so" 2>/dev/null | cut -c1-200
LMC_LIB=$P/tiles/liblmc_hip.so timeout 600 python -m pytest tests/test_gpu_relocate.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/bj/pytest_tiles.txt 2>&1; grep -E "passed|failed" gpurun_out/bj/pytest_tiles.txt
