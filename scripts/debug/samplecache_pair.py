"""GPU box: oracle vs device with largestepmultiplexed + samplecache through the cache phase; prints every comparison of gc.run_pair."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import gpu_checks as gc
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for opts in ({"largestepprob": 0.3, "largestepscale": 1.0, "largestepmultiplexed": 1, "samplecache": 1}, {"largestepprob": 0.3, "largestepscale": 1.0, "largestepmultiplexed": 1}):
    r = gc.run_pair(128, 96, 200000, 16384, 64, 120, steps, use_gradient=1, opts=opts)
    print(json.dumps({k: v for k, v in r.items() if not hasattr(v, "shape")}, default=float), flush=True)
