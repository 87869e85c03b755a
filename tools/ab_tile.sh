#!/bin/bash
# A/B of the tiled Gram between libkccot_old.so (tools/build_old_lib.sh <commit>) and the working tree: parity tests, then the
# cost stage at B = 128 / 256 / 512, alternating, two rounds.
set -o pipefail
# needs kccotgan_amd/csrc/libkccot_old.so: tools/build_old_lib.sh <commit> (run here, the .so travels with the snapshot)
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tiled or gram or full_size or large_batch or cost" > gpurun_out/r02y_tests.log 2>&1; tail -3 gpurun_out/r02y_tests.log
for rep in 1 2; do
for lib in old new; do
  p=$PWD/kccotgan_amd/csrc/libkccot.so; [ $lib = old ] && p=$PWD/kccotgan_amd/csrc/libkccot_old.so
  echo "== $lib"
  KCCOT_LIB_PATH=$p python tools/bench_gram.py 128 64 30 64 1 &&
  KCCOT_LIB_PATH=$p python tools/bench_gram.py 256 64 30 64 3 &&
  KCCOT_LIB_PATH=$p python tools/bench_gram.py 512 128 48 128 3 || exit 1
done
done
