#!/bin/bash
# A/B of the wave layout of the 256-row Gram tiles (option cost_tile256_w8): parity test, cost-stage time per setting,
# per-kernel time from a kernel trace.  usage: tools/gpu_q256_w8.sh [tag]
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT" || exit 1
TAG=${1:-q256w8}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "tile256" > "$OUT/pytest.log" 2>&1 || { tail -30 "$OUT/pytest.log"; exit 1; }
tail -n 3 "$OUT/pytest.log"
for shape in "256 64 30 64 3" "512 128 48 128 3"; do
  for w8 in 0 1 2 3; do
    KCCOT_OPTIONS="cost_tile256_w8=$w8" timeout -k 10 300 python tools/bench_gram.py $shape 2>&1 | grep "cost stage" | tee -a "$OUT/bench_gram.txt"
  done
done
export TMPDIR=/tmp
for w8 in 0 3; do
  export KCCOT_OPTIONS="cost_tile256_w8=$w8"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof$w8" -- python tools/bench_gram.py 512 128 48 128 3 > "$OUT/prof$w8.log" 2>&1 || exit 1
  f=$(find "$OUT/prof$w8" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_w8_$w8.csv" && rm -rf "$OUT/prof$w8"
  echo "== w8=$w8"; grep -i "gram_q256" "$OUT/kernel_stats_w8_$w8.csv" | awk -F',' '{print $1, $2, $3, $4}' | cut -c1-160
done
exit 0
