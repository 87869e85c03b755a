timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "fused_3d or random_shapes or handed_in" 2>&1 | tail -2
for shape in "512 128 48 128 3" "256 64 30 64 3" "64 64 30 64 1" "64 64 30 64 3"; do
  echo "fwd $(KCCOT_OPTIONS=smooth_fused3=2 timeout -k 10 120 python3 tools/bench_smooth.py $shape conv3d 2>&1 | grep conv3d | cut -c1-60)"
  echo "$(timeout -k 10 120 python3 tools/bench_smooth_bwd.py $shape conv3d 2>&1 | grep conv3d | cut -c1-60)"
done
