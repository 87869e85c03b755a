#!/bin/bash
# timing ablations of smooth_fused3 (diag twin; KCCOT_F3_ABLATE bits: 1 no loads in the walk, 2 no T stage, 4 no W-stage LDS reads,
# 8 no division, 16 no stores, 32 maxima pass only, 64 writing pass only, 128 no LDS writes of the plane, 256 no H stencil / emit,
# 512 no barriers)
export KCCOT_LIB_PATH=$PWD/kccotgan_amd/csrc/libkccot_diag.so
for shape in "${@:-512 128 48 128 3}"; do
for a in 0 32 33 34 36 39 167 295 423 551 935; do
  echo "shape $shape ablate $a: $(KCCOT_F3_ABLATE=$a timeout -k 10 120 python3 tools/bench_smooth.py $shape conv3d 2>&1 | grep -o 'T=[0-9]*: [0-9.]* us')"
done
done
