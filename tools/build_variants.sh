#!/bin/bash
# Experiment builds of the library: tools/build_variants.sh name "-DFLAG=.. -DFLAG=.." [name flags ...]
# -> kccotgan_amd/csrc/libkccot_<name>.so (git-ignored; loaded with KCCOT_LIB_PATH by the tools and bench.py)
set -e
cd "$(dirname "$0")/../kccotgan_amd/csrc"
SRCS="api.hip cost.hip cost_mfma.hip cost_tiled.hip cost_bwd.hip sinkhorn.hip sinkhorn_gen.hip sinkhorn_coop.hip loss.hip martingale.hip smooth.hip convlstm.hip layernorm.hip"
while [ $# -ge 2 ]; do
    name=$1; flags=$2; shift 2
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $flags -shared -o libkccot_$name.so $SRCS &
done
wait
ls -la libkccot_*.so
