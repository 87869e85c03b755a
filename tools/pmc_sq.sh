#!/bin/bash
# SQ / GRBM counters of the large-batch kernels (one pass, no trace domains besides the kernel trace):
# MFMA-busy, wave cycles, wait buckets, active cycles -> matrix-pipe utilisation and the effective clock.
# usage: tools/pmc_sq.sh <tag> <cfg...>      (cfg names of tools/bench_configs.py)
set -o pipefail
TAG=${1:-pmc_sq}; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -- python tools/bench_configs.py "$@" > "$OUT/sq.log" 2>&1 || { echo "pass failed"; tail -20 "$OUT/sq.log"; exit 1; }
find "$OUT/sq" -name "*counter_collection.csv" | head -1 | xargs -r -I{} cp {} "$OUT/sq_counters.csv"
find "$OUT/sq" -name "*kernel_trace.csv" | head -1 | xargs -r -I{} cp {} "$OUT/sq_kernel_trace.csv"
python - "$OUT" <<'PY'
import csv, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(out + "/sq_counters.csv")):
    name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("kccot::", "")
    if not name.startswith(("gram", "apply", "sinkhorn", "coeffs")):
        continue
    acc[name + " grid=" + row.get("Grid_Size", "?")][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/sq_summary.txt", "w") as f:
    for k, d in sorted(acc.items()):
        m = {c: sum(v) / len(v) for c, v in d.items()}
        line = "%-60s launches %d  " % (k, len(next(iter(d.values())))) + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(m.items()))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m and m["SQ_BUSY_CYCLES"]:
            line += "  | mfma_busy/sq_busy=%.3f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_BUSY_CYCLES"])
        print(line); f.write(line + "\n")
PY
