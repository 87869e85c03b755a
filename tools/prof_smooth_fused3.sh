#!/bin/bash
# rocprofv3 --kernel-trace --stats of the 3-D KernelSmoothing calls (forward: tools/bench_smooth.py, backward: bench_smooth_bwd.py)
# at the configs[1], configs[3] and configs[4] shapes -> gpurun_out/prof_f3/<shape>_<dir>_kernel_stats.csv
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_f3
for shape in "64 64 30 64 1" "256 64 30 64 3" "512 128 48 128 3"; do
  s=$(echo $shape | tr ' ' 'x')
  for dir in fwd bwd; do
    script=tools/bench_smooth.py; [ $dir = bwd ] && script=tools/bench_smooth_bwd.py
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_f3/${s}_$dir -- python3 $script $shape conv3d > gpurun_out/prof_f3/${s}_$dir.log 2>&1 || exit 1
    f=$(find gpurun_out/prof_f3/${s}_$dir -name "*kernel_stats.csv" | head -1)
    cp "$f" gpurun_out/prof_f3/${s}_${dir}_kernel_stats.csv
    rm -rf gpurun_out/prof_f3/${s}_$dir
    echo "== $s $dir: $(grep conv3d gpurun_out/prof_f3/${s}_$dir.log | cut -c1-80)"
    head -6 gpurun_out/prof_f3/${s}_${dir}_kernel_stats.csv | cut -c1-150
  done
done
