#!/usr/bin/env python3
"""Merge the per-call smooth_traffic.json files tools/pmc_smooth.sh leaves under gpurun_out/<tag>_* into
profiles/smooth_traffic.json (the table bench.py reads).  Calls that were not re-measured keep their entries.
usage: pmc_smooth_merge.py <tag>"""
import glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
dst = os.path.join(ROOT, "profiles", "smooth_traffic.json")
table = json.load(open(dst)) if os.path.exists(dst) else {"calls": {}}
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag + "_*", "smooth_traffic.json"))):
    d = json.load(open(f))
    key = "%s_%s_%s" % ("x".join(str(v) for v in d["shape"]), d["call"], d["direction"])
    table["calls"][key] = {k: d[k] for k in ("traffic_over_algorithmic", "traffic_bytes_per_call", "algorithmic_bytes_per_call")}
    table["calls"][key]["traffic_over_algorithmic"] = round(d["traffic_over_algorithmic"], 3)
    print(key, table["calls"][key]["traffic_over_algorithmic"])
json.dump(table, open(dst, "w"), indent=1, sort_keys=True)
