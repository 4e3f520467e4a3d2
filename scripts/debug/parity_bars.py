#!/usr/bin/env python3
"""The measured figures behind the statistical bars of the chain-parity tests: runs the same gc.run_pair calls
and prints film_rel_l2 / final_state_match / counter differences, so that the bars can be set at 2 x what is measured.  (GPU + oracle/_ref)"""
import json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_checks as gc

cases = {
    "h2mc_diffuse": dict(a=(160, 120, 40000, 256, 8, 400, 30), k=dict(use_gradient=1, opts={"h2mc": 1, "largestepprob": 0.2, "perturbstddev": 0.01}, oracle_grad="reference")),
    "h2mc_full": dict(a=(160, 120, 20000, 256, 20000, 400, 30), k=dict(use_gradient=1, max_depth=8, force_diffuse=0, opts={"h2mc": 1, "largestepprob": 0.2, "perturbstddev": 0.01}, oracle_grad="reference")),
    "h2mc_full_2048": dict(a=(160, 120, 40000, 2048, 40000, 400, 30), k=dict(use_gradient=1, max_depth=8, force_diffuse=0, opts={"h2mc": 1, "largestepprob": 0.2, "perturbstddev": 0.01}, oracle_grad="reference")),
    "h2mc_full_2048_3steps": dict(a=(160, 120, 40000, 2048, 40000, 400, 3), k=dict(use_gradient=1, max_depth=8, force_diffuse=0, opts={"h2mc": 1, "largestepprob": 0.2, "perturbstddev": 0.01}, oracle_grad="reference")),
    "h2mc_full_2048_6steps": dict(a=(160, 120, 40000, 2048, 40000, 400, 6), k=dict(use_gradient=1, max_depth=8, force_diffuse=0, opts={"h2mc": 1, "largestepprob": 0.2, "perturbstddev": 0.01}, oracle_grad="reference")),
    "h2mc_diffuse_2048_6steps": dict(a=(160, 120, 40000, 2048, 8, 400, 6), k=dict(use_gradient=1, opts={"h2mc": 1, "largestepprob": 0.2, "perturbstddev": 0.01}, oracle_grad="reference")),
    "lmc_full_g0": dict(a=(160, 120, 20000, 256, 20000, 400, 40), k=dict(use_gradient=0, max_depth=8, force_diffuse=0, oracle_grad="reference")),
    "lmc_full_g1": dict(a=(160, 120, 20000, 256, 20000, 400, 40), k=dict(use_gradient=1, max_depth=8, force_diffuse=0, oracle_grad="reference")),
    "lmc_depth12": dict(a=(160, 120, 20000, 256, 20000, 400, 40), k=dict(use_gradient=1, max_depth=12, force_diffuse=0, oracle_grad="reference")),
}
for name in (sys.argv[1:] or list(cases)):
    c = cases[name]
    r = gc.run_pair(*c["a"], **c["k"])
    so, sg = r["stats_oracle"], r["stats_gpu"]
    print(json.dumps({"case": name, "film_rel_l2": r["film_rel_l2"], "final_state_match": r["final_state_match"], "init_cl_match": r.get("init_cl_match"),
                      "accepted": [sg["accepted"], so["accepted"]], "largeSteps": [sg["largeSteps"], so["largeSteps"]], "gradCalls": [sg["gradCalls"], so["gradCalls"]],
                      "contribs": [r.get("contribs_gpu"), r.get("contribs_oracle")]}), flush=True)
