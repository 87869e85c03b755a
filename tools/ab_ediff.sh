#!/bin/bash
# A/B of the tiled Gram with E = fake - real materialised once (KCCOT_GRAM_EDIFF_MINB=128) vs subtracted while staging (=0):
# bit-equality tests first, then the cost stage at B = 256 / 384 / 512 (tools/bench_gram.py), alternating, two rounds.
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tiled or ediff or materialised or full_size_configs or large_batch" > gpurun_out/r03ac_tests.log 2>&1; tail -3 gpurun_out/r03ac_tests.log
for rep in 1 2; do
for minb in 0 128; do
  echo "== KCCOT_GRAM_EDIFF_MINB=$minb"
  KCCOT_GRAM_EDIFF_MINB=$minb python tools/bench_gram.py 256 64 30 64 3 &&
  KCCOT_GRAM_EDIFF_MINB=$minb python tools/bench_gram.py 384 64 30 64 3 &&
  KCCOT_GRAM_EDIFF_MINB=$minb python tools/bench_gram.py 512 128 48 128 3 || exit 1
done
done
