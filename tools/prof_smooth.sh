#!/bin/bash
# per-kernel durations of KernelSmoothing forward (rocprofv3 --kernel-trace --stats): tools/prof_smooth.sh tag [B H T W C]
export TMPDIR=/tmp
tag=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_smooth_$tag -- python3 tools/bench_smooth.py "$@" > gpurun_out/prof_smooth_$tag.log 2>&1 || { tail -5 gpurun_out/prof_smooth_$tag.log; exit 1; }
grep "^temporal\|^conv3d" gpurun_out/prof_smooth_$tag.log
f=$(find gpurun_out/prof_smooth_$tag -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    print("%-90s calls %5s avg %9.1f ns" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])))
PY
