#!/usr/bin/env python3
"""Oracle vs device, chain by chain, over the <dpt> options that change the chain loop -- with the oracle drawing its gradients from the product's path program built
for the host, so that what is compared is the loop and not two gradient implementations.  One JSON line per case.   usage: python scripts/debug/option_matrix_parity.py  (GPU)"""
import json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_checks as gc

AREA = os.path.join(gc.ROOT, "scenes", "torus", "lmc_arealight.xml")
POINT = os.path.join(gc.ROOT, "scenes", "torus", "lmc_pointlight.xml")
DOOR = os.path.join(gc.ROOT, "scenes", "veachdoor", "lmc.xml")
CASES = {
    "mux_lambertian": dict(args=(160, 120, 40000, 256, 8, 400, 100), kw=dict(use_gradient=1, opts={"largestepmultiplexed": 1})),
    "mux_arealight_lightcoord": dict(args=(160, 120, 1 << 17, 2048, 4096, 400, 40), kw=dict(use_gradient=1, max_depth=6, scene=AREA, force_diffuse=1, opts={"largestepmultiplexed": 1, "uselightcoordinatesampling": 1})),
    "mux_full_materials": dict(args=(160, 120, 20000, 2048, 20000, 400, 40), kw=dict(use_gradient=1, max_depth=8, force_diffuse=0, opts={"largestepmultiplexed": 1})),
    "lightcoord_arealight": dict(args=(160, 120, 1 << 17, 1 << 13, 4096, 400, 40), kw=dict(use_gradient=1, max_depth=6, scene=AREA, force_diffuse=1, opts={"uselightcoordinatesampling": 1})),
    "pointlight_full_materials": dict(args=(160, 120, 20000, 1024, 20000, 400, 40), kw=dict(use_gradient=1, max_depth=8, scene=POINT, force_diffuse=0)),
    "samplecache_lambertian": dict(args=(96, 72, 1 << 17, 1 << 14, 2048, 400, 60), kw=dict(use_gradient=1, max_depth=4, opts={"largestepprob": 0.5, "largestepscale": 1.0, "largestepmultiplexed": 1, "samplecache": 1})),
    "door_lightcoord": dict(args=(160, 90, 40000, 2048, 40000, 400, 40), kw=dict(use_gradient=1, max_depth=8, scene=DOOR, force_diffuse=0, opts={"uselightcoordinatesampling": 1})),
    "plain_mlt_full_materials": dict(args=(160, 120, 20000, 1024, 20000, 400, 60), kw=dict(use_gradient=0, max_depth=8, force_diffuse=0, mala=False)),
}
keep = ("contribs_gpu", "contribs_oracle", "norm_gpu", "norm_oracle", "init_cl_match", "init_ls_relerr_max", "init_pss_maxdiff", "stats_oracle", "stats_gpu", "film_rel_l2", "final_state_match", "energy_gpu", "energy_oracle", "nonfinite_gpu")
only = sys.argv[1:]
for name, c in CASES.items():
    if only and name not in only:
        continue
    try:
        r = gc.run_pair(*c["args"], oracle_grad="product", **c["kw"])
        print(json.dumps({"case": name, **{k: r[k] for k in keep if k in r}}), flush=True)
    except Exception as e:  # noqa: BLE001
        print(json.dumps({"case": name, "error": repr(e)[:300]}), flush=True)
