#!/bin/bash
# HBM traffic of the large-batch cost stage (gram_tile_x3 and friends): FETCH_SIZE / WRITE_SIZE in separate passes over
# tools/bench_gram.py at B = 256 (configs[3] shape) and B = 512 (configs[4] shape); summary by tools/pmc_summary.py.
set -o pipefail
export TMPDIR=/tmp
for shape in "256 64 30 64 3" "512 128 48 128 3"; do
  tag=$(echo $shape | cut -d' ' -f1)
  OUT=gpurun_out/${1:-pmc_tile}_B$tag
  mkdir -p "$OUT"
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$C" -- python3 tools/bench_gram.py $shape > "$OUT/$C.log" 2>&1 || { echo "$C pass failed"; tail -20 "$OUT/$C.log"; exit 1; }
    find "$OUT/$C" -name "*counter_collection.csv" | head -1 | xargs -r -I{} cp {} "$OUT/${C}_counters.csv"
    find "$OUT/$C" -name "*kernel_trace.csv" -delete
  done
  echo "== B=$tag  (algorithmic: 2 B K 4 bytes = $(python3 -c "B,H,T,W,C=map(int,'$shape'.split()); print('%.1f MB' % (2*B*H*T*W*C*4/1e6))"))"
  python3 tools/pmc_summary.py "$OUT"
done
