#!/bin/bash
# Timing-only ablation: the B = 512 Gram (gram_q256<*>) and video-gradient (apply_q256) kernels with every
# v_mfma_f32_32x32x16_bf16 replaced by two v_mfma_f32_16x16x32_bf16 on quarter tiles (gram_q.h, KCCOT_ABLATE_MFMA16; results are
# garbage, FLOPs / operand reads / registers equal).  Question: does the 1.12x of tools/micro/mfma_shape.hip survive in the
# kernels?   build: here;  run: GPU box
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd); cd "$ROOT"
if [ "$1" = "build" ]; then
    mkdir -p build/abl
    for f in cost_bwd_q256 cost_tile256; do
        /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DKCCOT_ABLATE_MFMA16 -c kccotgan_amd/csrc/$f.hip -o build/abl/m16_$f.o
    done
    OBJS=$(ls kccotgan_amd/csrc/obj/*.o | grep -v "diag_\|cost_tile256.o\|cost_bwd_q256.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build/libkccot_mfma16.so $OBJS build/abl/m16_cost_tile256.o build/abl/m16_cost_bwd_q256.o
else
    export TMPDIR=/tmp
    mkdir -p gpurun_out/mfma16
    for v in base mfma16 base mfma16; do
        lib=$ROOT/kccotgan_amd/csrc/libkccot.so; [ $v = mfma16 ] && lib=$ROOT/build/libkccot_mfma16.so
        KCCOT_LIB_PATH=$lib timeout -k 10 200 python3 tools/check_apply_q256.py 512 2359296 > gpurun_out/mfma16/apply_$v.log 2>&1
        KCCOT_LIB_PATH=$lib timeout -k 10 200 python3 tools/bench_gram.py 512 128 48 128 3 > gpurun_out/mfma16/gram_$v.log 2>&1
        echo "== $v: apply $(grep -o '"ms_q256": [0-9.]*' gpurun_out/mfma16/apply_$v.log)   gram: $(grep 'cost stage' gpurun_out/mfma16/gram_$v.log)"
    done
fi
