#!/bin/bash
# threads per workgroup of the fused walks at a fixed tile (diag twin: KCCOT_F3_NT): more waves per SIMD against idle lanes
export KCCOT_LIB_PATH=$PWD/kccotgan_amd/csrc/libkccot_diag.so
for nt in 384 448 512; do
  echo "configs[4] fwd nt=$nt: $(KCCOT_F3_NT=$nt timeout -k 10 120 python3 tools/bench_smooth.py 512 128 48 128 3 conv3d 2>&1 | grep -o 'T=48: [0-9.]* us')"
  echo "configs[4] bwd nt=$nt: $(KCCOT_F3_NT=$nt timeout -k 10 120 python3 tools/bench_smooth_bwd.py 512 128 48 128 3 conv3d 2>&1 | grep -o 'T=48: [0-9.]* us')"
done
