export TMPDIR=/tmp
mkdir -p gpurun_out/r4prof
for a in 0 1; do
  KCCOT_L3_ABLATE=$a KCCOT_LIB_PATH=$PWD/kccotgan_amd/csrc/libkccot_diag.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4prof/abl$a -- python tools/ab_cost_stage.py 100 > gpurun_out/r4prof/abl$a.log 2>&1
  f=$(find gpurun_out/r4prof/abl$a -name "*kernel_stats.csv" | head -1)
  echo "== ablate $a"; grep apply_coeffs "$f" | cut -c1-120
done
