#!/usr/bin/env python3
"""Lock-step comparison of a render with and without chain relocation (relocate.hip): steps both one step at a time and reports the first
step / chains whose summaries differ.  usage: relocate_pair.py [max_depth] [force_diffuse] [chains] [steps] [mala]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
p = importlib.import_module("langevin-mcmc_amd")
scene = os.path.join(ROOT, "scenes", "torus", "lmc.xml")
md = int(sys.argv[1]) if len(sys.argv) > 1 else 12
fd = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 256
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
mala = int(sys.argv[5]) if len(sys.argv) > 5 else 1
rens = []
for rel in (0, 1):
    os.environ["LMC_RELOCATE"] = str(rel)
    r = p.Renderer(scene, force_diffuse=fd, max_depth=md, width=160, height=120, seed_offset=0, use_gradient=1)
    if not mala:
        r.set_option("mala", 0)
    r.init_chains(20000, n, 20000, 400)
    rens.append(r)
names = ["valid", "c", "l", "ls", "ss", "scoreSum", "time", "gauss", "buffered", "sampleIdx", "sx", "sy", "r", "g", "b", "nSplat"]
for it in range(steps):
    for r in rens:
        r.step(1)
    a, b = rens[0].summary(0), rens[1].summary(0)
    valid = a[:, 0] == 1
    bad = np.where(np.any(a != b, axis=1) & (valid | (b[:, 0] == 1)))[0]
    st = [r.stats() for r in rens]
    print("step", it, "differing valid rows:", len(bad), "accepted", st[0]["accepted"], st[1]["accepted"], "reloc", rens[1].relocation_stats())
    for i in bad[:4]:
        cols = np.where(a[i] != b[i])[0]
        print("   chain", i, {(names[c] if c < 16 else "pss%d" % (c - 16)): (float(a[i, c]), float(b[i, c])) for c in cols[:8]})
    if len(bad) and it > 3:
        break
