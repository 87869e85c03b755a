#!/usr/bin/env python3
"""Summarise the FETCH_SIZE / WRITE_SIZE passes of tools/pmc.sh into per-kernel HBM bytes per launch.

gfx950 corrections (MI355X_MICROARCH.md, section HBM): the counters are in KiB; FETCH_SIZE reports
exactly half of the bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled for the
kernels whose reads are global_load_dwordx4 streams; WRITE_SIZE is exact for 16-B-per-lane stores."""
import csv, json, os, sys, collections
out = sys.argv[1]
WIDE_READERS = ("gram128_partial", "gram128_partial_x3", "gram128_partial_x3ws", "apply_coeffs_mfma", "apply_coeffs_x3", "cost_direct_partial", "gram_tile_x3", "gram_q256", "rows_gram", "apply_coeffs_x3_m256", "apply_coeffs_x3_m256n128")
res = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = os.path.join(out, c + "_counters.csv")
    if not os.path.exists(p):
        continue
    for row in csv.DictReader(open(p)):
        if row.get("Counter_Name") != c:
            continue
        name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("kccot::", "")
        res[name][c].append(float(row["Counter_Value"]))
summary = {}
for name, d in sorted(res.items()):
    if not name.startswith(("gram", "apply", "cost_", "sinkhorn", "causal", "build", "mixed", "conv_axis", "divide")):
        continue
    f = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1) * 1024
    w = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1) * 1024
    corr = 2.0 if name.split("<")[0] in WIDE_READERS else 1.0
    summary[name] = {"fetch_raw_bytes": f, "fetch_bytes": f * corr, "write_bytes": w, "hbm_bytes_per_launch": f * corr + w,
                     "launches": len(d["FETCH_SIZE"])}
    print("%-28s fetch %.2f MB (raw %.2f, x%.0f)  write %.2f MB  total %.2f MB  [%d launches]"
          % (name, f * corr / 1e6, f / 1e6, corr, w / 1e6, (f * corr + w) / 1e6, len(d["FETCH_SIZE"])))
json.dump(summary, open(os.path.join(out, "hbm_traffic_all.json"), "w"), indent=1)
g = [v for k, v in summary.items() if k.startswith("gram128_partial")]
if g:
    json.dump({"gram128_partial_bytes_per_launch": g[0]["hbm_bytes_per_launch"], "detail": g[0],
               "source": "rocprofv3 --pmc pass '%s' (tools/pmc.sh; copy of its summary under profiles/); read from this "
                         "file by bench.py, not re-measured in the bench run" % os.path.basename(os.path.normpath(out)),
               "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, KiB units, FETCH_SIZE x2 (gfx950 wide-read correction)"},
              open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
