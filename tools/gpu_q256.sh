#!/bin/bash
# GPU session for the 256-row pair-tile Gram kernel: parity tests that reach it, cost-stage timing, kernel trace.
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT" || exit 1
TAG=${1:-q256}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize_grads.py -q -x -p no:cacheprovider \
    -k "blocked_mfma or larger_batches or full_size or gram_sums or large_batch" > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -n 15 "$OUT/pytest.log"
for shape in "256 64 30 64 3" "512 128 48 128 3"; do
    timeout -k 10 300 python tools/bench_gram.py $shape 2>&1 | tee -a "$OUT/bench_gram.txt"
done
timeout -k 10 300 python tools/bench_configs.py cfg4 cfg5 2>&1 | tee "$OUT/bench_configs.jsonl"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python tools/bench_configs.py cfg4 cfg5 > "$OUT/prof.log" 2>&1
f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv" && head -n 14 "$OUT/kernel_stats.csv" | cut -c1-60,200- | awk -F'",' '{print substr($1,1,60) " | " $2 " " $3 " " $4}' 
exit 0
