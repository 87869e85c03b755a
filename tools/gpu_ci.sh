#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof summary.  Usage: tools/gpu_ci.sh [tag]
# Stops at the first step that is killed or times out (never starts another GPU step after that).
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
step() {  # name, timeout, command...
    local name=$1 to=$2; shift 2
    echo "=== $name" | tee -a "$OUT/steps.log"
    timeout -k 10 "$to" "$@" > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "rc=$rc" | tee -a "$OUT/steps.log"
    tail -n 25 "$OUT/$name.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name killed/timeout: stopping"; exit $rc; fi
    return 0
}
step pytest 900 python -m pytest tests -m gpu -q --tb=short --maxfail=40 -p no:cacheprovider
step smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
step bench 600 python bench.py --steps 50 --warmup 10
export TMPDIR=/tmp
step rocprof 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-configs --no-pmc
find "$OUT/prof" -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} "$OUT/kernel_stats.csv"
[ -f "$OUT/kernel_stats.csv" ] && head -n 25 "$OUT/kernel_stats.csv"
exit 0
