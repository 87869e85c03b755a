#!/bin/bash
# GPU check of the folded smoothing backward: tests, then kernel-only timing with the option on / off.
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_smoothing_golden.py tests/test_gpu_parity.py -m gpu -q -k "smooth or rows_gram" > gpurun_out/fold_tests.log 2>&1 || { tail -40 gpurun_out/fold_tests.log; exit 1; }
tail -3 gpurun_out/fold_tests.log
for shape in "64 64 30 64 1" "256 64 30 64 3" "16 64 30 64 1"; do
  for fold in 1 2 0; do
    echo "== $shape fold=$fold"
    KCCOT_OPTIONS="smooth_bwd_fold=$fold" timeout -k 10 300 python tools/bench_smooth_bwd.py $shape
  done
done 2>&1 | tee gpurun_out/fold_bench.log
bash tools/pmc_smooth.sh r3fold bwd 2>&1 | tee gpurun_out/r3fold_pmc_smooth_bwd.txt
