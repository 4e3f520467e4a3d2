# usage (GPU box): scripts/debug/h2_parts_ab.sh <tag>  -- the H2MC pipeline in 1 .. 4 parts of the chain population (LMC_H2_PARTS), both scenes, 2^20 chains
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
for parts in 2 1 3 4 2; do for sc in door torus; do
  LMC_H2_PARTS=$parts timeout 300 python scripts/h2mc_rates.py $sc 20 24 8 2>>$O/err.txt | sed "s/^{/{\"parts\": $parts, /" | tee -a $O/rates.jsonl
done; done
