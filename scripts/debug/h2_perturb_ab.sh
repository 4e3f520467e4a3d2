cd $GRAFT_REPO_ROOT
O=gpurun_out/r05x_h2perturb; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_h2mc.py tests/test_gpu_relocate.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for v in generic streamed generic streamed; do
  for sc in door torus; do
    LMC_H2_PERTURB=$v timeout 300 python scripts/h2mc_rates.py $sc 20 24 8 2>>$O/err.txt | sed "s/^{/{\"perturb\": \"$v\", /" | tee -a $O/rates.jsonl
  done
done
