#!/usr/bin/env python3
"""Where a kernel waits for ONE load at a time (CPU, no GPU needed).  hipcc sinks a load to its first use; when that use sits behind arithmetic (a
node's child indices behind the slab tests, a triangle's p0 behind the divisor test, a material's fields one by one) the load becomes a dependent
round trip of its own although the line it reads was fetched a few instructions earlier.  This lists, per source line, the global loads of one
kernel and for each the number of instructions / further loads between its issue and the next `s_waitcnt vmcnt`: a load with (few, 0) is waited
for on its own.  The last load of a batch also shows up that way -- read the groups, not the single entries.  Round 5 found the node / leaf /
hit-record / material / texture sites this way (dscene.h LMC_PIN; profiles/r05_ae_*, r05_af_*, r05_ag_*).

usage: scripts/isa_load_waits.py <device source .hip> <mangled-name regex of the kernel> [max distance = 8] [extra hipcc flags...]
  e.g. scripts/isa_load_waits.py langevin-mcmc_amd/csrc/device/step_small_plain.hip 12k_step_smallILb1ELb0ELb0ELb1ELb1E
Also prints the kernel's slow integer / division instructions by source line with --slow."""
import collections, os, re, subprocess, sys, tempfile

args = [a for a in sys.argv[1:] if a != "--slow"]
slow = "--slow" in sys.argv
src, pat = args[0], args[1]
maxd = int(args[2]) if len(args) > 2 and args[2].isdigit() else 8
extra = [a for a in args[2:] if not a.isdigit()]
out = os.path.join(tempfile.mkdtemp(prefix="isa_"), "k.s")
cmd = ["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics", "-fno-slp-vectorize", "-Wno-unused-result", "-gline-tables-only", "-S",
       "--cuda-device-only", "-o", out, src] + extra
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
txt = open(out).read().split("\n")
files = {}
for l in txt:
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
    if m:
        files[int(m.group(1))] = os.path.basename(m.group(2))
a = next((n for n, l in enumerate(txt) if re.match(r"^_Z\S*" + pat + r"\S*:", l)), None)
if a is None:
    sys.exit("no kernel matches " + pat + "; candidates:\n" + "\n".join(sorted({l.split(":")[0] for l in txt if re.match(r"^_Z\S*k_\S*:$", l)})[:40]))
b = next(n for n in range(a, len(txt)) if txt[n].startswith(".Lfunc_end"))
cur, ins = None, []
for l in txt[a:b]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    ins.append(("L" if t.endswith(":") else "I", t.split(";")[0].strip(), cur))
print("%s: %d instructions, %d global loads, %d scratch accesses" % (txt[a][:60], sum(1 for k, _, _ in ins if k == "I"),
      sum(1 for k, t, _ in ins if k == "I" and t.startswith("global_load")), sum(1 for k, t, _ in ins if k == "I" and t.startswith("scratch_"))))
by = collections.defaultdict(list)
for i, (k, t, loc) in enumerate(ins):
    if k == "I" and t.startswith("global_load"):
        d = nl = 0
        for kk, tt, _ in ins[i + 1:i + 500]:
            if kk == "L":
                continue
            if tt.startswith("s_waitcnt") and "vmcnt" in tt:
                break
            if tt.startswith("global_load") or tt.startswith("scratch_load"):
                nl += 1
            d += 1
        by[loc].append((t.split()[0].replace("global_load_", ""), d, nl))
for loc, v in sorted(by.items(), key=lambda x: (str(x[0][0]), x[0][1])):
    iso = [x for x in v if x[1] <= maxd and x[2] == 0]
    if iso:
        print("%-12s %4d  %3d loads, %3d waited for on their own  %s" % (loc[0], loc[1], len(v), len(iso), v[:6]))
if slow:
    ops = ("v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32", "v_lshl_add_u64")
    c = collections.Counter((loc, t.split()[0]) for k, t, loc in ins if k == "I" and t.split()[0] in ops)
    print("slow instructions by source line:")
    for (loc, op), n in c.most_common(25):
        print("  %-12s %4d  %-16s %d" % (loc[0], loc[1], op, n))
