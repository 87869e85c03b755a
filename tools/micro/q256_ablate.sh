#!/bin/bash
# Timing-only ablations of gram_q256 (results are WRONG in every variant): builds build/libkccot_abl<N>.so from a patched
# COPY of kccotgan_amd/csrc/cost_tile256.hip (the product source carries no hooks) and times the cost stage with each.
#   bit 0: no global loads      bit 1: no split / LDS writes      bit 2: no workgroup barriers
#   bit 3: no MFMAs (fragments still read)      bit 4: no LDS fragment reads (MFMAs on stale registers)
#   bit 5: EPAIR does not store E (round 4: the upper bound of what NOT materialising E could save in the E-writing launch)
# usage: tools/micro/q256_ablate.sh build   (here, no GPU)   |   tools/micro/q256_ablate.sh run "<B H T W C>"   (GPU box)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd "$ROOT"
VARIANTS=${Q256_VARIANTS:-"0 1 2 4 16 19 32"}
if [ "$1" = "build" ]; then
    mkdir -p build/abl
    python3 - <<'PY'
import re
s = open("kccotgan_amd/csrc/cost_tile256.hip").read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b, 1)
rep('    auto issue = [&](float4 (&G)[NP], int g) {\n', '    auto issue = [&](float4 (&G)[NP], int g) {\n        if (Q256_ABL & 1) return;\n')
rep('    auto emit_even = [&](float4 (&G)[NP], int g, unsigned char* zs) {\n',
    '    auto emit_even = [&](float4 (&G)[NP], int g, unsigned char* zs) {\n        if (Q256_ABL & 2) { for (int p = 0; p < NP; ++p) carry[p] = make_float2(G[p].z, G[p].w); return; }\n')
rep('    auto emit_odd = [&](unsigned char* zs) {\n', '    auto emit_odd = [&](unsigned char* zs) {\n        if (Q256_ABL & 2) return;\n')
s = s.replace('        lds_barrier();\n', '        if (!(Q256_ABL & 4)) lds_barrier();\n')
rep('__device__ __forceinline__ void q256_mfma6(qf32x16& acc, const QFrag& a, const QFrag& b) {   // smallest terms first\n',
    '__device__ __forceinline__ void q256_mfma6(qf32x16& acc, const QFrag& a, const QFrag& b) {   // smallest terms first\n'
    '    if (Q256_ABL & 8) { asm volatile("" :: "v"(a.h), "v"(a.m), "v"(a.l), "v"(b.h), "v"(b.m), "v"(b.l)); return; }\n')
rep('__device__ __forceinline__ QFrag q256_frag(const unsigned char* zs, int off) {\n    QFrag f;\n',
    '__device__ __forceinline__ QFrag q256_frag(const unsigned char* zs, int off) {\n    QFrag f;\n'
    '    if (Q256_ABL & 16) { asm volatile("" : "=v"(f.h), "=v"(f.m), "=v"(f.l)); return f; }\n')
rep('                __builtin_amdgcn_raw_buffer_store_b128(', '                if (!(Q256_ABL & 32)) __builtin_amdgcn_raw_buffer_store_b128(')
s = s.replace('#include "common.h"', '#include "../../kccotgan_amd/csrc/common.h"').replace('#include "cost_internal.h"', '#include "../../kccotgan_amd/csrc/cost_internal.h"').replace('#include "options.h"', '#include "../../kccotgan_amd/csrc/options.h"')
open("build/abl/cost_tile256_abl.hip", "w").write(s)
PY
    sed -i 's#"../../include/kccot.h"#"../../include/kccot.h"#' build/abl/cost_tile256_abl.hip
    OBJS=$(ls kccotgan_amd/csrc/obj/*.o | grep -v "diag_\|cost_tile256.o")
    for v in $VARIANTS; do
        ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I kccotgan_amd/csrc -DQ256_ABL=$v -c build/abl/cost_tile256_abl.hip -o build/abl/q$v.o \
          && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build/libkccot_abl$v.so $OBJS build/abl/q$v.o && echo "built $v" ) &
    done
    wait
elif [ "$1" = "clock" ]; then
    # effective shader clock per variant and kernel: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration
    shift
    export TMPDIR=/tmp
    for v in ${2:-0 1 19}; do
        rm -rf gpurun_out/abl_clock$v
        KCCOT_LIB_PATH=$ROOT/build/libkccot_abl$v.so timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/abl_clock$v -- python3 tools/bench_gram.py $1 > gpurun_out/abl_clock$v.log 2>&1
        python3 - "$v" <<'PY'
import csv, glob, sys, collections
v = sys.argv[1]
d = "gpurun_out/abl_clock%s" % v
cc = glob.glob(d + "/*/*counter_collection.csv")[0]
kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(kt))}
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(cc)):
    name = r["Kernel_Name"].split("(")[0].replace("void kccot::", "")
    if "gram_q256<" not in name: continue
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        acc[name]["ns"] += dur[r["Dispatch_Id"]]; n[name] += 1
for k, m in sorted(acc.items()):
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    print("ABL=%s %-22s launches %d  %.2f ms  clock %.2f GHz  mfma_busy %.3f  wait_any/wave %.2f  wait_inst/wave %.2f  active/wave %.2f" % (
        v, k, n[k], m["ns"] / n[k] / 1e6, cyc / m["ns"], m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024) if cyc else 0,
        m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"]))
PY
    done
else
    shift
    for v in $VARIANTS; do
        echo -n "ABL=$v  "
        KCCOT_LIB_PATH=$ROOT/build/libkccot_abl$v.so timeout -k 10 120 python tools/bench_gram.py $1 2>&1 | grep "cost stage"
    done
fi
