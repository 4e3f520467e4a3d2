"""TEST HELPER: compiles an INSTRUMENTED copy of one of the reference's generated derivative programs, read from where it lies
under /root/reference, into a temporary directory (nothing of it is kept in the repo).  Every pass-through assignment `_accX = _accY;` of the reverse sweep becomes a numbered site that
 (a) logs the adjoint it drops (old value of _accX) and (b) can be switched to `+=` through lmc_toggle[site].
usage as a module: lib, sites = build(c, l)"""
import ctypes, os, re, subprocess, sys

REF = "/root/reference/src/bin"


def build_forward_double(c, l, outdir="/tmp"):
    """The reference's VERBATIM forward program compiled with -Dfloat=double: an accurate function to difference."""
    name = "evaluate_path_bidir_mala_%d_%d_static" % (c, l)
    out = os.path.join(outdir, "chad_fwd_double_%d_%d.so" % (c, l))
    subprocess.check_call(["gcc", "-O1", "-fPIC", "-w", "-shared", "-fno-math-errno", "-Dfloat=double", os.path.join(REF, name + ".c"), "-o", out, "-lm"])
    return ctypes.CDLL(out), name


def build(c, l, kind="mala", outdir="/tmp"):
    name = "evaluate_path_bidir_%s%d_%d_static_derv" % ("mala_" if kind == "mala" else "", c, l)
    src = open(os.path.join(REF, name + ".ispc")).read()
    src = re.sub(r"\buniform ", "", src)
    src = re.sub(r"\bcif\b", "if", src)
    src = re.sub(r"^export ", "", src, flags=re.M)
    src = re.sub(r"foreach \(index = 0 \.\.\. (\d+)\)", r"for (int index=0; index<\1; index++)", src)
    lines = src.split("\n")
    rev = next(i for i, ln in enumerate(lines) if "Reverse accumulation" in ln)
    sites = []
    fwd = {}
    for i, ln in enumerate(lines[:rev]):
        m = re.match(r"\s*(_t\d+) = (.*);", ln)
        if m:
            fwd.setdefault(m.group(1), []).append(m.group(2))
    for i in range(rev, len(lines)):
        m = re.match(r"(\s*)_acc(\d+) = _acc(\d+);", lines[i])
        if m:
            s = len(sites)
            sites.append(dict(site=s, line=i + 1, x=int(m.group(2)), out=int(m.group(3)), defs=fwd.get("_t" + m.group(2), [])))
            lines[i] = "%s{ lmc_site(%d, _acc%s, _acc%s); if (lmc_toggle[%d]) _acc%s += _acc%s; else _acc%s = _acc%s; }" % (
                m.group(1), s, m.group(2), m.group(3), s, m.group(2), m.group(3), m.group(2), m.group(3))
    pre = """#include <math.h>
int lmc_toggle[%d]; float lmc_old[%d]; float lmc_new[%d]; int lmc_hit[%d];
static inline void lmc_site(int s, float o, float n) { lmc_old[s] = o; lmc_new[s] = n; lmc_hit[s]++; }
""" % ((len(sites) + 1,) * 4)
    out = os.path.join(outdir, "chad_instr_%s_%d_%d" % (kind, c, l))
    open(out + ".c", "w").write(pre + "\n".join(lines))
    subprocess.check_call(["gcc", "-O1", "-fPIC", "-w", "-shared", "-fno-math-errno", out + ".c", "-o", out + ".so", "-lm"])
    lib = ctypes.CDLL(out + ".so")
    return lib, sites, name


if __name__ == "__main__":
    lib, sites, name = build(int(sys.argv[1]), int(sys.argv[2]))
    print(name, len(sites), "pass-through sites")
