#!/bin/bash
# needs kccotgan_amd/csrc/libkccot_old.so: tools/build_old_lib.sh <commit> (run here, the .so travels with the snapshot)
# bit-level and timing A/B of two library builds on the loss backward (video gradient kernels)
set -o pipefail
OLD=$PWD/kccotgan_amd/csrc/libkccot_old.so
for shape in "64 64 30 64 1" "48 32 12 32 3" "128 64 30 64 1"; do
  tag=$(echo $shape | tr ' ' '_')
  KCCOT_LIB_PATH=$OLD python tools/dump_loss_grads.py gpurun_out/ab_old_$tag.npz $shape &&
  python tools/dump_loss_grads.py gpurun_out/ab_new_$tag.npz $shape &&
  python tools/dump_loss_grads.py --compare gpurun_out/ab_old_$tag.npz gpurun_out/ab_new_$tag.npz || exit 1
done
for rep in 1 2; do
  for lib in old new; do
    p=$PWD/kccotgan_amd/csrc/libkccot.so; [ $lib = old ] && p=$OLD
    KCCOT_LIB_PATH=$p python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train --no-configs > gpurun_out/ab_apply_$lib.json 2>/dev/null || exit 1
    python - $lib gpurun_out/ab_apply_$lib.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "ms/step %.4f" % d["ms_per_step"], {k: v for k, v in d.get("kernels_us", {}).items()} if "kernels_us" in d else "")
PY
  done
done
