#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace --stats) of tools/ab_cost_stage.py under the option sets given as arguments
# usage: tools/prof_ab_fold.sh "name=value[,name=value]" ...
export TMPDIR=/tmp
mkdir -p gpurun_out/r4prof
for o in "$@"; do
  KCCOT_OPTIONS="$o" timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "gpurun_out/r4prof/$o" -- python tools/ab_cost_stage.py 100 > "gpurun_out/r4prof/$o.log" 2>&1
  f=$(find "gpurun_out/r4prof/$o" -name "*kernel_stats.csv" | head -1)
  echo "== $o"; cut -c1-150 "$f" | head -8
done
