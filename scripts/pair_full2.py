import sys, json, numpy as np
sys.path.insert(0, "/root/repo")
from tests import gpu_checks as gc, _orc
p = gc.pkg(); L = gc.oracle_lib()
orc = _orc.Oracle(L, gc.TORUS, 0, 8, 160, 120, 0, "")
ren = p.Renderer(gc.TORUS, force_diffuse=0, max_depth=8, width=160, height=120, seed_offset=0, use_gradient=0)
NI = 20000
on, oc = orc.init(NI, 4096, NI)
gn, gcn = ren.init_chains(NI, 4096, NI, 100)
print("norm", on, gn, "contribs", oc, gcn)
si, gi = orc.summary(1), ren.summary(1)
same = (si[:, 1] == gi[:, 1]) & (si[:, 2] == gi[:, 2])
rel = np.abs(si[:, 3] - gi[:, 3]) / np.maximum(si[:, 3], 1e-30)
print("cl match", same.mean(), "max rel ls", rel[same].max(), "n rel>1e-4", (rel[same] > 1e-4).sum())
bad = np.where(~same | (rel > 1e-3))[0]
for i in bad[:12]:
    print(i, "orc c,l,ls", si[i, 1:4], "gpu", gi[i, 1:4], "pss0", si[i, 16:20], gi[i, 16:20])
