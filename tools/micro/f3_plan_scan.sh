#!/bin/bash
# tile scan of smooth_fused3 (diag twin: KCCOT_F3_PLAN = "wt,hseg")
export KCCOT_LIB_PATH=$PWD/kccotgan_amd/csrc/libkccot_diag.so
for p in 32,32 64,32 64,16 64,64 32,64 32,16; do
  echo "configs[2] plan $p: $(KCCOT_F3_PLAN=$p timeout -k 10 120 python3 tools/bench_smooth.py 128 64 30 64 3 conv3d 2>&1 | grep -o 'T=30: [0-9.]* us')"
done
for p in 64,32 64,64 64,16 32,64; do
  echo "configs[3] plan $p: $(KCCOT_F3_PLAN=$p timeout -k 10 120 python3 tools/bench_smooth.py 256 64 30 64 3 conv3d 2>&1 | grep -o 'T=30: [0-9.]* us')"
done
for p in 32,128 32,64 32,32; do
  echo "configs[4] plan $p: $(KCCOT_F3_PLAN=$p timeout -k 10 120 python3 tools/bench_smooth.py 512 128 48 128 3 conv3d 2>&1 | grep -o 'T=48: [0-9.]* us')"
done
for p in 32,16 64,8 64,16 32,8 64,32; do
  echo "configs[1] plan $p: $(KCCOT_F3_PLAN=$p timeout -k 10 120 python3 tools/bench_smooth.py 64 64 30 64 1 conv3d 2>&1 | grep -o 'T=30: [0-9.]* us')"
done
