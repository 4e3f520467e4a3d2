#!/usr/bin/env python3
"""Where a wave of the lean small-step kernel spends its cycles (LMC_PROF=1 instantiation, dsmall.h WaveProf): steady state at
2^20 chains.  usage: LMC_PROF=1 python scripts/lean_region_profile.py [full|full12|door] > out.json   (GPU; `full` = the torus scene's own materials, maxdepth 8;
`full12` = BASELINE configs[2]: the same at maxdepth 12; `door` = the shipped veach-door lmc.xml)"""
import ctypes, importlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_checks as gc

assert os.environ.get("LMC_PROF") == "1", "run with LMC_PROF=1"
p = importlib.import_module("langevin-mcmc_amd")
chains = 1 << 20
which = sys.argv[1] if len(sys.argv) > 1 else ""
full = which in ("full", "full12", "door")
if which == "door":
    ren = p.Renderer(os.path.join(ROOT, "scenes", "veachdoor", "lmc.xml"), seed_offset=0, device=0, use_gradient=1)
else:
    ren = p.Renderer(gc.TORUS, force_diffuse=0 if full else 1, max_depth={"full": 8, "full12": 12}.get(which, 6), seed_offset=0, device=0, use_gradient=1)
ren.init_chains(8 * chains, chains, 65536, 256, 0, 0, chains)
ren.step(56 if full else 40)
out = (ctypes.c_ulonglong * 16)()
assert p.lib().lmc_prof_read(ren.h, out) == 0  # discard warm-up
ren.set_option("timing", 1)
ren.step_timing()
ren.step(32)
kernel_ms, launches = ren.step_timing()
small_ms, large_ms, lean = ren.kernel_timing()
assert p.lib().lmc_prof_read(ren.h, out) == 0
names = ["prologue", "gauss_current", "offsets", "vertex_load", "traverse", "shade", "loop_exit", "shadow_ray", "gauss_proposal", "splat", "accept", "queue_next", "isotropic_offsets", "buffered_reset", "gauss_stage"]
tot = sum(out[:15])
print(json.dumps({"waves": out[15], "lean_ms_per_launch": small_ms / launches, "cycles_per_wave": tot / max(out[15], 1),
                  "share": {n: round(out[k] / tot, 4) for k, n in enumerate(names)}}, indent=1))
