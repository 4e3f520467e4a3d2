import sys, re, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "helpers"))
import chad_instrument
c, l = int(sys.argv[1]), int(sys.argv[2])
lib, sites, name = chad_instrument.build(c, l)
src = open("/tmp/chad_instr_mala_%d_%d.c" % (c, l)).read().split("\n")
for s in map(int, sys.argv[3:]):
    d = sites[s]
    print("=== site", s, d)
    x = "_t%d" % d["x"]; out = "_t%d" % d["out"]
    for i, ln in enumerate(src):
        if re.search(r"\b%s\b" % x, ln) or re.search(r"\b%s =" % out, ln):
            if "float " in ln[:12] : continue
            print("%6d: %s" % (i + 1, ln.strip()[:170]))
