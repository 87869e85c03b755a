for shape in "4 64 30 64 1" "8 64 30 64 1" "16 64 30 64 1" "32 64 30 64 1" "64 64 30 64 1" "4 64 30 64 3" "8 64 30 64 3" "16 64 30 64 3" "32 64 30 64 3"; do
  for o in "smooth_fused3=0" "smooth_fused3=2,smooth_bwd_fold=2"; do
    echo "$o: $(KCCOT_OPTIONS=$o timeout -k 10 120 python3 tools/bench_smooth_bwd.py $shape conv3d 2>&1 | grep conv3d)"
  done
done
