#!/usr/bin/env python3
"""Per-kernel and per-call HBM traffic of the smoothing kernels from the two counter passes of tools/pmc_smooth.sh.
usage: pmc_smooth_summary.py <dir> fwd|bwd temporal|conv3d B H T W C"""
import collections, csv, json, os, sys
out, direction, which = sys.argv[1], sys.argv[2], sys.argv[3]
B, H, T, W, C = (int(a) for a in sys.argv[4:9])
n = B * H * T * W * C
alg = (8 if direction == "fwd" else 12) * n
# the bench script runs 5 warm-up + reps calls of the selected call (fwd: 200, bwd: 100 after one forward)
reps = {"fwd": 205, "bwd": 105}[direction]
res = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for row in csv.DictReader(open(os.path.join(out, c + "_counters.csv"))):
        if row.get("Counter_Name") != c:
            continue
        name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("kccot::", "")
        if name.startswith(("smooth", "maxnorm", "reduce_blockmax", "conv_axis", "divide_by", "copy_with")):
            res[name][c].append(float(row["Counter_Value"]) * 1024)        # KiB -> bytes
print("== %s %s  [%d,%d,%d,%d,%d]  algorithmic %.1f MB per call" % (which, direction, B, H, T, W, C, alg / 1e6))
tot = 0.0
summary = {}
for name, d in sorted(res.items()):
    launches = len(d["FETCH_SIZE"])
    f = sum(d["FETCH_SIZE"]) / max(launches, 1)
    w = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1)
    per_call = launches / reps
    tot += (2 * f + w) * per_call
    summary[name] = {"launches_per_bench_call": per_call, "fetch_raw_bytes": f, "fetch_x2_bytes": 2 * f, "write_bytes": w}
    print("  %-70s %5.2f launches/call  fetch %8.2f MB (raw %8.2f)  write %8.2f MB" % (name[:70], per_call, 2 * f / 1e6, f / 1e6, w / 1e6))
print("  per call: %.1f MB of HBM-side traffic = %.2f x the algorithmic %.1f MB" % (tot / 1e6, tot / alg, alg / 1e6))
json.dump({"shape": [B, H, T, W, C], "direction": direction, "call": which, "algorithmic_bytes_per_call": alg, "traffic_bytes_per_call": tot,
           "traffic_over_algorithmic": tot / alg, "kernels": summary}, open(os.path.join(out, "smooth_traffic.json"), "w"), indent=1)
