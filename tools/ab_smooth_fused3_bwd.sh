#!/bin/bash
# same-box A/B of the 3-D KernelSmoothing backward: all three adjoint stages in one pass (smooth_fused3 = 2, smooth_fused3_adj)
# against the chain of per-axis adjoint stages (= 0), statistics folded in both (smooth_bwd_fold = 2)
SHAPES=("64 64 30 64 1" "64 64 30 64 3" "128 64 30 64 3" "256 64 30 64 3" "64 128 48 128 3" "512 128 48 128 3")
for shape in "${SHAPES[@]}"; do
  for o in 0 2 0 2; do
    KCCOT_OPTIONS="smooth_fused3=$o,smooth_bwd_fold=2" timeout -k 10 120 python3 tools/bench_smooth_bwd.py $shape conv3d 2>&1 | grep conv3d
  done
done
