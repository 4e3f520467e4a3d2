#!/usr/bin/env python3
"""H2MC chain-steps/s on the two shipped scenes (A/B of builds through LMC_LIB).  (GPU)
usage: h2mc_rates.py [scene=both|torus|door] [log2_chains=18] [steps=8] [warmup=6]"""
import importlib, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
p = importlib.import_module("langevin-mcmc_amd")
S = os.path.join(ROOT, "scenes")
which = sys.argv[1] if len(sys.argv) > 1 else "both"
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 18
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
warm = int(sys.argv[4]) if len(sys.argv) > 4 else 6
for key, name, xml, kw in (("torus", "torus H2MC", os.path.join(S, "torus", "h2mc.xml"), dict(force_diffuse=0, max_depth=8)),
                           ("door", "veach-door H2MC", os.path.join(S, "veachdoor", "h2mc.xml"), dict(force_diffuse=0))):
    if which not in ("both", key):
        continue
    n = 1 << lg
    ren = p.Renderer(xml, seed_offset=0, device=0, use_gradient=1, **kw)
    ren.init_chains(8 * n, n, 65536, warm + steps + 8, 0, 0, n)
    ren.step(warm)
    ren.sync()
    t0 = time.time()
    ren.step(steps)
    ren.sync()
    dt = time.time() - t0
    st = ren.stats()
    print(json.dumps({"lib": os.environ.get("LMC_LIB", "tree"), "config": name, "chains": n, "chain_steps_per_s": n * steps / dt, "ms_per_step": dt * 1e3 / steps,
                      "accept_rate": st["accepted"] / max(st["steps"], 1), "grad_calls": st.get("gradCalls")}), flush=True)
    ren.close()
