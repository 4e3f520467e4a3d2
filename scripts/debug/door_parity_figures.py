#!/usr/bin/env python3
"""The figures behind the veach-door parity bars (tests/test_gpu_door.py): oracle vs device on the shipped door scene, LMC without / with gradients
and H2MC -- after round 6's deterministic trigonometry (device/dtrig.h) and restated glibc logf (drng.h) the chains can be compared one by one.
usage: python scripts/debug/door_parity_figures.py > out.jsonl   (GPU)"""
import json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_checks as gc

DOOR = os.path.join(gc.ROOT, "scenes", "veachdoor", "lmc.xml")
DOOR_H2 = os.path.join(gc.ROOT, "scenes", "veachdoor", "h2mc.xml")
keep = ("contribs_gpu", "contribs_oracle", "norm_gpu", "norm_oracle", "init_cl_match", "init_ls_relerr_max", "init_pss_maxdiff", "stats_oracle", "stats_gpu", "film_rel_l2", "final_state_match",
        "film_sum_gpu", "film_sum_oracle", "energy_gpu", "energy_oracle", "nonfinite_gpu")
import importlib

if len(sys.argv) > 1 and sys.argv[1] == "torus":  # the torus tests' configurations (tests/test_gpu_parity.py test_chain_loop_parity, test_full_material_scene_chain_parity)
    for name, kw in (("torus_lambert_nograd", dict(use_gradient=0)), ("torus_lambert_grad_reference", dict(use_gradient=1)), ("torus_lambert_grad_product", dict(use_gradient=1, oracle_grad="product"))):
        r = gc.run_pair(160, 120, 40000, 256, 8, 400, 40, **kw)
        print(json.dumps({"case": name, **{k: r[k] for k in keep if k in r}}), flush=True)
    for name, kw in (("torus_fullmat_nograd", dict(use_gradient=0)), ("torus_fullmat_grad_reference", dict(use_gradient=1, oracle_grad="reference")), ("torus_fullmat_grad_product", dict(use_gradient=1, oracle_grad="product"))):
        r = gc.run_pair(160, 120, 20000, 256, 20000, 400, 40, max_depth=8, force_diffuse=0, **kw)
        print(json.dumps({"case": name, **{k: r[k] for k in keep if k in r}}), flush=True)
    sys.exit(0)
cases = (("lmc_nograd", DOOR, 0, "reference"), ("lmc_grad_reference", DOOR, 1, "reference"), ("lmc_grad_product", DOOR, 1, "product"), ("h2mc_product", DOOR_H2, 1, "product"))
if len(sys.argv) > 1 and sys.argv[1] == "h2only":  # e.g. with LMC_LIB=<the strict build of the Hessian / eigen-solve units> (scripts/build_h2strict.sh)
    cases = cases[3:] + (("h2mc_torus_product", os.path.join(gc.ROOT, "scenes", "torus", "h2mc.xml"), 1, "product"),)
for name, scene, grad, og in cases:
    try:
        r = gc.run_pair(160, 90, 40000, 2048, 40000, 400, 60, use_gradient=grad, max_depth=8, scene=scene, force_diffuse=0, oracle_grad=og)
        print(json.dumps({"case": name, **{k: r[k] for k in keep if k in r}}), flush=True)
    except Exception as e:
        print(json.dumps({"case": name, "error": repr(e)}), flush=True)
