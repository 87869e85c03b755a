#!/bin/bash
# HBM-side traffic of every KernelSmoothing kernel: FETCH_SIZE and WRITE_SIZE in separate rocprofv3
# passes over tools/bench_smooth.py (forward) and tools/bench_smooth_bwd.py (backward) at the configs[1] and configs[3]
# shapes; per kernel: bytes per launch, launches per call; per call: total against the algorithmic bytes (forward: read +
# write the tensor = 8 n; backward: read gout and out, write din = 12 n).  FETCH_SIZE is doubled (gfx950 counts a wide
# coalesced read at half its bytes, MI355X_MICROARCH.md section HBM; the 8-byte pieces of the temporal walks are
# uncalibrated -- raw values are printed beside the corrected ones).  usage: tools/pmc_smooth.sh <tag> [fwd|bwd]
# then tools/pmc_smooth_merge.py <tag> (here, after the call) folds the results into profiles/smooth_traffic.json
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT" || exit 1
TAG=${1:-pmc_smooth}
DIRS=${2:-fwd bwd}
export TMPDIR=/tmp
for shape in "64 64 30 64 1" "256 64 30 64 3"; do
  s=$(echo $shape | tr ' ' 'x')
  for dir in $DIRS; do
   for which in temporal conv3d; do
    script=tools/bench_smooth.py; [ $dir = bwd ] && script=tools/bench_smooth_bwd.py
    OUT=gpurun_out/${TAG}_${s}_${which}_$dir
    mkdir -p "$OUT"
    for C in FETCH_SIZE WRITE_SIZE; do
      timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$C" -- python3 $script $shape $which > "$OUT/$C.log" 2>&1 || { echo "$C pass failed"; tail -20 "$OUT/$C.log"; exit 1; }
      find "$OUT/$C" -name "*counter_collection.csv" | head -1 | xargs -r -I{} cp {} "$OUT/${C}_counters.csv"
      rm -rf "$OUT/$C"
    done
    python3 tools/pmc_smooth_summary.py "$OUT" $dir $which $shape
   done
  done
done
