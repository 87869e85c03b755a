#!/bin/bash
# SQ counters of the configs[1] kernels (bench.py, eager launches): matrix-pipe busy / co-execution with the VALU,
# VALU and LDS active cycles, wait buckets.  One pass, kernel trace only.  usage: tools/pmc_sq_bench.sh <tag>
set -o pipefail
TAG=${1:-pmc_sqb}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export KCCOT_BENCH_EAGER=1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d "$OUT/sq" -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-train --no-configs --no-pmc > "$OUT/sq.log" 2>&1 || { echo "pass failed"; tail -20 "$OUT/sq.log"; exit 1; }
find "$OUT/sq" -name "*counter_collection.csv" | head -1 | xargs -r -I{} cp {} "$OUT/sq_counters.csv"
python - "$OUT" <<'PY'
import csv, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(out + "/sq_counters.csv")):
    name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("kccot::", "")
    if not name.startswith(("gram", "apply", "sinkhorn", "coeffs")):
        continue
    acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out + "/sq_summary.txt", "w") as f:
    for k, d in sorted(acc.items()):
        m = {c: sum(v) / len(v) for c, v in d.items()}
        line = "%-34s launches %4d  " % (k, len(next(iter(d.values())))) + "  ".join("%s=%.4g" % (c.replace("SQ_", ""), v) for c, v in sorted(m.items()))
        print(line); f.write(line + "\n")
PY
