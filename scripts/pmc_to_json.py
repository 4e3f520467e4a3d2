#!/usr/bin/env python3
"""profiles/pmc_step_kernel.json from the output directory of scripts/pmc_passes.sh: HBM bytes per launch of the lean step
kernel with the guide's corrections (FETCH_SIZE / WRITE_SIZE are in KB; the read counter is calibrated against a kernel that
streams a known byte count, lmc_stream_probe), stamped with the fingerprint of the kernel sources it was measured on
(bench.py only reports `traffic` when the fingerprint matches the tree).
usage: python scripts/pmc_to_json.py <pmc outdir> <label of the profile files in profiles/>"""
import json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

out, label = sys.argv[1], sys.argv[2]
summ = json.load(open(os.path.join(out, "summary.json")))
cal = json.load(open(os.path.join(out, "calib.json")))
probe = next(v for k, v in cal.items() if "stream_probe" in k)
GiB_KB = (1 << 30) / 1024.0
read_corr = GiB_KB / probe["FETCH_SIZE"]   # the probe reads exactly 1 GiB per launch
write_corr = GiB_KB / probe["WRITE_SIZE"]  # ... and writes 1 GiB
# the dominant kernel of the run: the lean instantiation without light sub-paths (<true,false,false,true>) on environment-lit scenes
# such as the headline one, the general one (<true,false,false>) otherwise
names = [n for n in summ if n.replace(" ", "").startswith("voidk_step_small<true,false,false")]
k = summ[max(names, key=lambda n: summ[n].get("launches", 0))]
fetch = k["FETCH_SIZE"] * 1024.0 * read_corr
write = k["WRITE_SIZE"] * 1024.0 * write_corr
# chain-steps per launch of the measured run: the bench line each pass printed (pass1.log); the wasted-traffic ratio must pair the
# counter figure with THIS number, not with the steps of some other window
steps_per_launch = None
try:
    for ln in open(os.path.join(out, "pass1.log")):
        if ln.startswith("{") and '"roofline"' in ln:
            r = json.loads(ln)["roofline"]
            # under the counters the window's own figure can come out empty (the per-launch brackets are not collected); the serialised launches
            # behind the window run the same work lists
            # the lean kernel's own work list: the figure of the serialised launches behind the window (under the counters the launches are serialised
            # and the window's "dominant kernel" can come out as another launch, or empty)
            steps_per_launch = r.get("standalone", {}).get("chain_steps_per_launch") or r["chain_steps_per_launch"]
except Exception:
    pass
d = {
    "kernel": max(names, key=lambda n: summ[n].get("launches", 0)).split("(")[0].replace("void ", ""),
    "chain_steps_per_launch": steps_per_launch,
    "traffic_over_algorithmic": ((fetch + write) / (bench.ALGO_BYTES_PER_STEP * steps_per_launch)) if steps_per_launch else None,
    "source": "profiles/%s_pmc_summary.json (rocprofv3 --pmc passes of `bench.py --no-cpu-baseline --no-rmse --steps 32 --warmup 40`, last third of the launches)" % label,
    "fetch_bytes_per_launch": fetch,
    "write_bytes_per_launch": write,
    "hbm_bytes_per_launch": fetch + write,
    "corrections": "FETCH_SIZE KB x 1024 x %.3f, WRITE_SIZE KB x 1024 x %.3f (profiles/%s_pmc_calibration.json: lmc_stream_probe moves 1 GiB each way per launch)" % (read_corr, write_corr, label),
    "kernel_source_sha16": bench.kernel_source_sha(),
}
json.dump(d, open(os.path.join(ROOT, "profiles", "pmc_step_kernel.json"), "w"), indent=1)
json.dump(summ, open(os.path.join(ROOT, "profiles", "%s_pmc_summary.json" % label), "w"), indent=1)
json.dump(cal, open(os.path.join(ROOT, "profiles", "%s_pmc_calibration.json" % label), "w"), indent=1)
print(json.dumps(d, indent=1))
