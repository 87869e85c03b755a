#!/bin/bash
# same-box A/B of the 3-D KernelSmoothing forward: one fused pass per phase (smooth_fused3 = 1) against the per-axis chain (= 0);
# default: the BASELINE frame shapes; "scan": batch sizes in between (where does the fused pass start to win?)
mkdir -p gpurun_out/f3
if [ "$1" = "scan" ]; then
  SHAPES=("16 64 30 64 3" "32 64 30 64 3" "64 64 30 64 3" "256 64 30 64 1" "512 64 30 64 1" "32 128 48 128 3" "64 128 48 128 3")
else
  SHAPES=("64 64 30 64 1" "128 64 30 64 3" "256 64 30 64 3" "512 128 48 128 3")
fi
for shape in "${SHAPES[@]}"; do
  for o in 0 1 0 1; do
    KCCOT_OPTIONS="smooth_fused3=$o" timeout -k 10 120 python3 tools/bench_smooth.py $shape conv3d 2>&1 | grep conv3d
  done
done
