#!/bin/bash
# needs kccotgan_amd/csrc/libkccot_old.so: tools/build_old_lib.sh <commit> (run here, the .so travels with the snapshot)
# per-kernel average durations of the configs[1] step for two library builds (rocprofv3 --kernel-trace --stats)
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lib in old new; do
  p=$PWD/kccotgan_amd/csrc/libkccot.so; [ $lib = old ] && p=$PWD/kccotgan_amd/csrc/libkccot_old.so
  export KCCOT_LIB_PATH=$p
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ab_$lib -- python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-train --no-configs > gpurun_out/prof_ab_$lib.log 2>&1 || { tail -5 gpurun_out/prof_ab_$lib.log; exit 1; }
  echo "== $lib"
  f=$(find gpurun_out/prof_ab_$lib -name "*kernel_stats.csv" | head -1)
  python3 - $f <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print("%-60s calls %5s avg %9.1f ns" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])))
PY
done
