"""Scene-only L2 behaviour: the closest-hit kernel (k_trace) on 2^20 incoherent rays per launch, run under
`rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum FETCH_SIZE` -- the only memory k_trace touches besides its 40 B per ray of input /
output is the scene (BVH nodes, leaf triangles), so its TCC hit rate IS the scene's, separated from the chain-state streaming
that shares the L2 inside the step kernel.  Generation 0: random rays through the scene's bounding region; generation 1: rays
that start on the surfaces generation 0 hit, in random directions (the segments a chain traces).
usage: python scripts/trace_tcc_probe.py [scene.xml] [log2 rays] [launches per generation]"""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import gpu_checks as gc

xml = sys.argv[1] if len(sys.argv) > 1 else gc.TORUS
n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
p = importlib.import_module("langevin-mcmc_amd")
ren = p.Renderer(xml, force_diffuse=1, max_depth=6, seed_offset=0)
rng = np.random.default_rng(5)
center, radius = (np.array([0.0, 0.0, 4.0]), 12.0) if "torus" in xml else (np.array([-71.39, 71.49, 205.3]), 150.0)
rays = gc.random_rays(rng, n, center, radius)
rays[:, 7] = np.inf
out = {"scene": os.path.basename(os.path.dirname(xml)), "rays_per_launch": n, "bvh4_nodes": ren.num_nodes, "node_bytes": ren.num_nodes * 128, "leaf_tri_bytes": ren.num_tris * 48}
for gen in (0, 1):
    t0 = time.time()
    for _ in range(reps):
        prim, t = ren.trace(rays)
    out["gen%d" % gen] = {"hit_frac": float((prim >= 0).mean()), "host_seconds_per_launch_incl_copies": (time.time() - t0) / reps}
    hit = prim >= 0
    pos = rays[:, 0:3] + rays[:, 3:6] * np.where(hit, t, 0.0)[:, None]
    d = rng.normal(0, 1, (n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    nxt = rays.copy()
    nxt[hit, 0:3] = pos[hit]
    nxt[:, 3:6] = d
    rays = nxt
print(json.dumps(out))
ren.close()
