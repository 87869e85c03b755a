#!/bin/bash
# L2 -> fabric read traffic (FETCH_SIZE) of the large-batch kernels: tools/pmc.sh for tools/bench_configs.py.
# usage: tools/pmc_configs.sh <tag> <cfg...>
set -o pipefail
TAG=${1:-pmc_cfg}; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$C" -- python tools/bench_configs.py "$@" > "$OUT/$C.log" 2>&1 || { echo "$C pass failed"; tail -20 "$OUT/$C.log"; exit 1; }
    find "$OUT/$C" -name "*counter_collection.csv" | head -1 | xargs -r -I{} cp {} "$OUT/${C}_counters.csv"
done
python tools/pmc_summary.py "$OUT" | tee "$OUT/summary.txt"
