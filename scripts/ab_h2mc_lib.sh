cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
for v in tree $2 tree $2; do for sc in door torus; do
  if [ $v = tree ]; then L=""; else L=$PWD/langevin-mcmc_amd/csrc/_ab/$v/liblmc_hip.so; fi
  LMC_LIB=$L timeout 300 python scripts/h2mc_rates.py $sc 20 24 8 2>>$O/err.txt | sed "s/^{/{\"build\": \"$v\", /" | tee -a $O/rates.jsonl
done; done
