#!/bin/bash
# per-kernel durations of the configs[4] loss step (B = 512) with the video gradient on 256 x 256 tiles (apply_q256 = 1) and without
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_cfg5
for o in 0 1; do
  KCCOT_OPTIONS="apply_q256=$o" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg5/o$o -- python3 tools/bench_configs.py cfg5 > gpurun_out/prof_cfg5/o$o.log 2>&1
  echo "== apply_q256=$o  $(grep -i 'ms' gpurun_out/prof_cfg5/o$o.log | tail -1 | cut -c1-200)"
  f=$(find gpurun_out/prof_cfg5/o$o -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.reader(open(sys.argv[1])))[1:9]:
    print("   %-52s calls %4s avg %9.3f ms" % (r[0].replace("void kccot::", "")[:52], r[1], float(r[3]) / 1e6))
PY
done
