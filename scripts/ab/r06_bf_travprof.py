#!/usr/bin/env python3
"""one-off (experiment build -DLMC_PROF_TRAV of step_small_plain, LMC_LIB=.../_ab/travprof/liblmc_hip.so, LMC_PROF=1): the closest-hit walk's own
counters per wave-step -- cycles in the inner-node loops / the leaf loops, wave-level loop iterations, calls, lane-level node visits"""
import ctypes, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_checks as gc
p = importlib.import_module("langevin-mcmc_amd")
chains = 1 << 20
ren = p.Renderer(gc.TORUS, force_diffuse=1, max_depth=6, seed_offset=0, device=0, use_gradient=1)
ren.init_chains(8 * chains, chains, 65536, 256, 0, 0, chains)
ren.step(40)
out = (ctypes.c_ulonglong * 16)()
assert p.lib().lmc_prof_read(ren.h, out) == 0
ren.step(32)
assert p.lib().lmc_prof_read(ren.h, out) == 0
PR = ["prologue", "gauss_current", "offsets", "vertex_load", "traverse", "shade", "loop_exit", "shadow_ray", "gauss_proposal", "splat", "accept", "queue_next", "isotropic_offsets", "buffered_reset", "gauss_stage"]
d = dict(zip(PR, out[:15])); w = out[15]
res = {"waves": w, "traverse_region_cycles_per_wave": d["traverse"] / w, "shadow_region_cycles_per_wave": d["shadow_ray"] / w,
       "walk_hint_cycles_per_wave": d["isotropic_offsets"] / w, "walk_inner_loop_cycles_per_wave": d["buffered_reset"] / w, "walk_leaf_loop_cycles_per_wave": d["gauss_stage"] / w,
       "inner_wave_iterations_per_wave": d["gauss_current"] / w, "leaf_wave_iterations_per_wave": d["offsets"] / w, "closest_hit_calls_per_wave": d["vertex_load"] / w,
       "lane_node_visits_per_wave": d["splat"] / w}
res["cycles_per_inner_iteration"] = res["walk_inner_loop_cycles_per_wave"] / max(res["inner_wave_iterations_per_wave"], 1e-9)
res["cycles_per_leaf_iteration"] = res["walk_leaf_loop_cycles_per_wave"] / max(res["leaf_wave_iterations_per_wave"], 1e-9)
res["inner_iterations_per_call"] = res["inner_wave_iterations_per_wave"] / max(res["closest_hit_calls_per_wave"], 1e-9)
res["lane_visits_per_call_per_lane"] = res["lane_node_visits_per_wave"] / max(res["closest_hit_calls_per_wave"], 1e-9) / 64
print(json.dumps(res, indent=1))
