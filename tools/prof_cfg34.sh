export TMPDIR=/tmp
mkdir -p gpurun_out/prof_cfg34
for c in cfg3 cfg4; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg34/$c -- python3 tools/bench_configs.py $c > gpurun_out/prof_cfg34/$c.log 2>&1
  f=$(find gpurun_out/prof_cfg34/$c -name "*kernel_stats.csv" | head -1)
  echo "== $c $(grep -o '"ms_fwd_bwd": [0-9.]*' gpurun_out/prof_cfg34/$c.log | tail -1)"
  python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.reader(open(sys.argv[1])))[1:12]:
    print("   %-60s calls %4s avg %9.1f us  %5s %%" % (r[0].replace("void kccot::", "")[:60], r[1], float(r[3]) / 1e3, r[4]))
PY
done
