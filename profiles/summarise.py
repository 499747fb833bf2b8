#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of profiles/collect.sh (gpurun_out/<tag>/) into the committed
summaries: profiles/<tag>_kernel_stats.csv (the --kernel-trace --stats table of the default
bench command), profiles/<tag>_bench.json (the bench line of the same box) and
profiles/<tag>_traffic.json (HBM traffic per launch per kernel from the FETCH_SIZE and
WRITE_SIZE PMC passes).

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB, and on
gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md, section HBM), so
it is doubled. The correction is calibrated for wide (16 B / lane) streaming reads; for the
gather-heavy kernels here it is an upper bound on the read side.

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
out = {
    "tag": tag,
    "workload": bench["config"]["workload"],
    "note": "per launch; traffic_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 correction, MI355X_MICROARCH.md)",
    "kernels": {},
}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("k_"):
        continue
    f, nf = fetch.get(k, (0.0, 0))
    w, nw = write.get(k, (0.0, 0))
    out["kernels"][k] = {"fetch_size_kib": round(f, 1), "write_size_kib": round(w, 1), "traffic_bytes": int((2 * f + w) * 1024), "launches_sampled": nf}
# The dual traversal is one dispatch per level: bench.py times a whole traversal as one unit, so
# its traffic is the sum over the levels = mean per dispatch x dispatches per traversal (one
# k_bp_dual_init per traversal).
if "k_bp_dual" in out["kernels"] and "k_bp_dual_init" in out["kernels"]:
    per = out["kernels"]["k_bp_dual"]["launches_sampled"] / max(1, out["kernels"]["k_bp_dual_init"]["launches_sampled"])
    d = out["kernels"]["k_bp_dual"]
    d["dispatches_per_traversal"] = round(per, 2)
    d["traffic_bytes_per_dispatch"] = d["traffic_bytes"]
    d["traffic_bytes"] = int(d["traffic_bytes"] * per)
json.dump(out, open(os.path.join(root, tag + "_traffic.json"), "w"), indent=1)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["traffic_bytes"])[:12]:
    print("%-22s fetch %10.1f KiB write %10.1f KiB traffic %8.1f MB" % (k, v["fetch_size_kib"], v["write_size_kib"], v["traffic_bytes"] / 1e6))
