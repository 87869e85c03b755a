#!/bin/bash
# Timing-only ablations of apply_coeffs_x3_m256n128 (results WRONG): builds build/libkccot_apabl<N>.so from a patched COPY of
# kccotgan_amd/csrc/cost_bwd.hip.  bit 0: no W fragment loads (consumers)   bit 1: no global loads of the stack (producers)
# bit 2: no MFMAs.   usage: tools/micro/apply_ablate.sh build | run
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd "$ROOT"
VARIANTS="0 1 2 3 4"
if [ "$1" = "build" ]; then
    mkdir -p build/abl
    python3 - <<'PY'
s = open("kccotgan_amd/csrc/cost_bwd.hip").read()
a = s.index("void apply_coeffs_x3_m256n128(")
head, body = s[:a], s[a:]
def rep(x, y):
    global body
    assert x in body, x
    body = body.replace(x, y, 1)
rep("    auto ldA = [&](int g, int slot) {\n        const int64_t o = (int64_t)512 * g;\n",
    "    auto ldA = [&](int g, int slot) {\n        if (APPLY_ABL & 1) return;\n        const int64_t o = (int64_t)512 * g;\n")
rep("        auto load_stage = [&](int64_t tile, int c) {\n            const int g0 = c * AY_ROWS;\n",
    "        auto load_stage = [&](int64_t tile, int c) {\n            if (APPLY_ABL & 2) { tile = blockIdx.x & 7; c = 0; }   // always the same few KB: served by L2\n            const int g0 = c * AY_ROWS;\n")
body = body.replace("acc[0][2 * pair] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(", "if (!(APPLY_ABL & 4)) acc[0][2 * pair] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(", 1)
s = (head + body).replace('#include "common.h"', '#include "../../kccotgan_amd/csrc/common.h"').replace('#include "cost_internal.h"', '#include "../../kccotgan_amd/csrc/cost_internal.h"').replace('#include "options.h"', '#include "../../kccotgan_amd/csrc/options.h"')
open("build/abl/cost_bwd_abl.hip", "w").write(s)
PY
    OBJS=$(ls kccotgan_amd/csrc/obj/*.o | grep -v "diag_\|cost_bwd.o")
    for v in $VARIANTS; do
        ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I kccotgan_amd/csrc -DAPPLY_ABL=$v -c build/abl/cost_bwd_abl.hip -o build/abl/ap$v.o \
          && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build/libkccot_apabl$v.so $OBJS build/abl/ap$v.o && echo "built $v" ) &
    done
    wait
else
    for v in $VARIANTS; do
        echo -n "ABL=$v  "
        KCCOT_LIB_PATH=$ROOT/build/libkccot_apabl$v.so timeout -k 10 200 python3 tools/bench_apply_rows.py big 2>&1 | grep "^{" | cut -c1-200
    done
fi
