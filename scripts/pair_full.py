import sys, json
sys.path.insert(0, "/root/repo")
from tests import gpu_checks as gc
for ug in (0, 1):
    r = gc.run_pair(160, 120, 40000, 256, 8, 400, 40, use_gradient=ug, max_depth=8, force_diffuse=0, oracle_grad="product")
    print(ug, json.dumps({k: v for k, v in r.items()}))
