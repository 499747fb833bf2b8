#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of profiles/collect.sh (gpurun_out/<tag>/) into the committed
summaries: profiles/<tag>_kernel_stats.csv (the --kernel-trace --stats table of the default
bench command), profiles/<tag>_bench.json (the bench line of the same box) and
profiles/<tag>_traffic.json (HBM traffic per launch per kernel from the FETCH_SIZE and
WRITE_SIZE PMC passes).

Both counters are in KiB. On gfx950 FETCH_SIZE tallies the 128-byte requests of wide (16 B /
lane) streaming reads at 64 bytes (MI355X_MICROARCH.md, section HBM), so for STREAMING kernels
it is doubled. For per-lane record gathers the calibration of this repository
(profiles/calib_gather.hip, profiles/r02_calib_gather.json: 60e6 random 96-byte records, counter
= 0.98-1.00 x the bytes requested) says the counter is exact, so for the GATHER kernels listed
below it is not doubled. Every entry carries both readings (traffic_bytes_fetch_x1 / _x2);
traffic_bytes is the calibrated one for the kernel's class.

usage: python profiles/summarise.py <tag>"""
import collections
import csv
import json
import os
import re
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(root, "..", "gpurun_out", tag)


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name


def pmc(path, counter):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                per[short(r["Kernel_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: (sum(v.values()) / len(v), len(v)) for k, v in per.items()}


shutil.copy(os.path.join(src, "stats", "s_kernel_stats.csv"), os.path.join(root, tag + "_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(root, tag + "_bench.json"))
fetch = pmc(os.path.join(src, "fetch", "f_counter_collection.csv"), "FETCH_SIZE")
write = pmc(os.path.join(src, "write", "w_counter_collection.csv"), "WRITE_SIZE")
bench = json.load(open(os.path.join(src, "bench.json")))
GATHER = ("k_bp_dual", "k_np_prefilter", "k_np_fex", "k_np_roots", "k_broad_phase", "k_bvh_refit", "k_bvh_depth", "k_bvh_pack", "k_gather_aabb", "k_gather_u32")
out = {
    "tag": tag,
    "workload": bench["config"]["workload"],
    "note": "per launch; traffic_bytes = (f * FETCH_SIZE + WRITE_SIZE) * 1024 with f = 1 for record-gather kernels (calibrated: profiles/r02_calib_gather.json) and f = 2 for streaming kernels (MI355X_MICROARCH.md)",
    "kernels": {},
}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("k_"):
        continue
    f, nf = fetch.get(k, (0.0, 0))
    w, nw = write.get(k, (0.0, 0))
    gather = any(k.startswith(g) for g in GATHER)
    out["kernels"][k] = {"fetch_size_kib": round(f, 1), "write_size_kib": round(w, 1), "fetch_factor": 1 if gather else 2, "traffic_bytes": int(((1 if gather else 2) * f + w) * 1024),
                         "traffic_bytes_fetch_x1": int((f + w) * 1024), "traffic_bytes_fetch_x2": int((2 * f + w) * 1024), "launches_sampled": nf, "note": "fetch x%d (%s)" % (1 if gather else 2, "record gathers" if gather else "streaming")}
# The dual traversal is one dispatch per level: bench.py times a whole traversal as one unit, so
# its traffic is the sum over the levels = mean per dispatch x dispatches per traversal (one
# k_bp_dual_init per traversal).
if "k_bp_dual" in out["kernels"] and "k_bp_dual_init" in out["kernels"]:
    per = out["kernels"]["k_bp_dual"]["launches_sampled"] / max(1, out["kernels"]["k_bp_dual_init"]["launches_sampled"])
    d = out["kernels"]["k_bp_dual"]
    d["dispatches_per_traversal"] = round(per, 2)
    d["traffic_bytes_per_dispatch"] = d["traffic_bytes"]
    for key in ("traffic_bytes", "traffic_bytes_fetch_x1", "traffic_bytes_fetch_x2"):
        d[key] = int(d[key] * per)
json.dump(out, open(os.path.join(root, tag + "_traffic.json"), "w"), indent=1)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["traffic_bytes"])[:12]:
    print("%-22s fetch %10.1f KiB write %10.1f KiB traffic %8.1f MB" % (k, v["fetch_size_kib"], v["write_size_kib"], v["traffic_bytes"] / 1e6))
