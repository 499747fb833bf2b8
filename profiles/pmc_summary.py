#!/usr/bin/env python3
"""Summarise a rocprofv3 results.db (rocpd sqlite): per kernel, mean duration and mean PMC values.
usage: pmc_summary.py results.db [kernel-substring]"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
rows = c.execute("select * from counters_collection").fetchall()
ix = {n: i for i, n in enumerate(cols)}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for r in rows:
    k = r[ix["kernel_name"]] if "kernel_name" in ix else r[ix["name"]]
    if flt not in k:
        continue
    agg[k][r[ix["counter_name"]]].append((r[ix["dispatch_id"]], r[ix["value"]]))
    if "start" in ix:
        dur[k][r[ix["dispatch_id"]]] = r[ix["end"]] - r[ix["start"]]
for k, cs in agg.items():
    d = dur.get(k, {})
    print(k[:90], "dispatches", len(d), "mean_us", round(sum(d.values()) / max(len(d), 1) / 1e3, 2))
    for cn, vals in sorted(cs.items()):
        per = collections.defaultdict(float)
        for did, v in vals:
            per[did] += v
        print("   %-28s mean/dispatch %.4g" % (cn, sum(per.values()) / len(per)))
