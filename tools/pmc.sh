#!/bin/bash
# HBM traffic of the loss-path kernels from the L2 memory-side counters, one counter per pass
# (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -o pipefail
TAG=${1:-pmc}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export KCCOT_BENCH_EAGER=1   # plain launches (no graph replay) so that every dispatch is attributed to its kernel
for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$C" -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-train --no-configs --no-pmc > "$OUT/$C.log" 2>&1 || { echo "$C pass failed"; tail -20 "$OUT/$C.log"; exit 1; }
    find "$OUT/$C" -name "*counter_collection.csv" | head -1 | xargs -r -I{} cp {} "$OUT/${C}_counters.csv"
done
python tools/pmc_summary.py "$OUT" | tee "$OUT/summary.txt"
