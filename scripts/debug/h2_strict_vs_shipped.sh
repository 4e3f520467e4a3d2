cd /root/repo
for v in strict shipped; do
  if [ $v = strict ]; then export LMC_LIB=/root/repo/langevin-mcmc_amd/csrc/_ab/h2strict/liblmc_hip.so; else unset LMC_LIB; fi
  LMC_H2_REPORT=1 python -m pytest tests/test_gpu_h2mc.py -q -s -k "chain_parity_diffuse or chain_parity_full" -p no:cacheprovider 2>&1 | grep -E "H2REPORT|passed|failed" | sed "s/^/$v /"
done
