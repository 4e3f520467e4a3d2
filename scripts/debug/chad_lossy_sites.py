"""Diagnostic: list the 'lossy' adjoint assignments of one of the reference's generated derivative programs.
chad's reverse emitter (chad.cpp:230-331) writes `_accX = _accR;` for the operand X a conditional passes through; whatever the
sweep had accumulated into _accX before that line (from uses of X later in program order) is dropped.  This script scans the
reverse section and reports every such assignment whose target was already written, with the forward definition of X.
usage: chad_lossy_sites.py /root/reference/src/bin/evaluate_path_bidir_mala_3_1_static_derv.ispc [depth]"""
import re
import sys

path = sys.argv[1]
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lines = open(path).read().split("\n")
rev = next(i for i, l in enumerate(lines) if "Reverse accumulation" in l)
fwd = {}
for i, l in enumerate(lines[:rev]):
    m = re.match(r"\s*(_t\d+) = (.*);", l)
    if m:
        fwd.setdefault(m.group(1), []).append((i + 1, m.group(2)))


def show(name, d, ind="    "):
    for ln, rhs in fwd.get(name, []):
        print("%s%s = %s   (fwd line %d)" % (ind, name, rhs[:160], ln))
        if d > 0:
            for sub in sorted(set(re.findall(r"_t\d+", rhs))):
                show(sub, d - 1, ind + "    ")


touched = {}
n = 0
for i, l in enumerate(lines[rev:], rev + 1):
    m = re.match(r"\s*_acc(\d+) (\+=|-=|=) (.*);", l)
    if not m:
        continue
    x, op, rhs = m.groups()
    if op == "=" and rhs.startswith("_acc"):
        if x in touched:
            n += 1
            print("LOSSY line %d: %s   (first written at line %d: %s)" % (i, l.strip(), touched[x][0], touched[x][1]))
            show("_t" + x, depth)
            src = rhs.strip()
            print("    passes adjoint of output %s" % src)
    touched.setdefault(x, (i, l.strip()))
print("lossy sites:", n)
