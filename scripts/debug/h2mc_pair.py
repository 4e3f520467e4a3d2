"""GPU box: the full-material H2MC chain comparison of tests/test_gpu_h2mc.py, all figures printed (A/B of builds through LMC_LIB)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import gpu_checks as gc
for chains in (256, 2048):
    r = gc.run_pair(160, 120, 20000, chains, 20000, 400, 30, use_gradient=1, max_depth=8, force_diffuse=0,
                    opts={"h2mc": 1, "largestepprob": 0.2, "perturbstddev": 0.01}, oracle_grad="reference")
    print(os.environ.get("LMC_LIB", "tree")[-30:], chains, json.dumps({k: r[k] for k in ("stats_oracle", "stats_gpu", "film_rel_l2", "final_state_match", "energy_gpu")}), flush=True)
