#!/bin/bash
# per-kernel durations of the full training iteration (default find mode + shipped find-db; FAST_FIND=1: MIOPEN_FIND_MODE=2): rocprofv3 --kernel-trace --stats
export TMPDIR=/tmp; [ -n "$FAST_FIND" ] && export MIOPEN_FIND_MODE=2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -- python3 tools/bench_train.py --json --iters 2 --kernel 3d > gpurun_out/prof_train.log 2>&1 || { tail -5 gpurun_out/prof_train.log; exit 1; }
grep "^{" gpurun_out/prof_train.log
f=$(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1); find gpurun_out/prof_train -name "*kernel_trace.csv" -delete
python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms over %d kernel names, %d launches" % (tot/1e6, len(rows), sum(int(r["Calls"]) for r in rows)))
for r in rows[:25]:
    print("%-100s calls %6s total %8.1f ms  %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"])/1e6, 100*float(r["TotalDurationNs"])/tot))
PY
