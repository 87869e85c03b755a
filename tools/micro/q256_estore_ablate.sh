#!/bin/bash
# Round 4, VERDICT r3 item 7: what would NOT materialising E = fake - real buy at B = 512?  Timing-only ablation (results of the
# later launches are wrong): a patched COPY of cost_tile256.hip whose E-writing launch (gram_q256<EPAIR>) skips its E stores.
# The difference to the product library is the UPPER bound of the saving inside that launch; the other launches (DIAG / OFF
# pairs with an E panel) would each have to load F and X instead of E and subtract: 22 panel-chunk loads instead of 16.
#   tools/micro/q256_estore_ablate.sh build        (here, no GPU)
#   tools/micro/q256_estore_ablate.sh run "<B H T W C>"   (GPU box): per-kernel durations of both builds
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd "$ROOT"
if [ "$1" = "build" ]; then
    mkdir -p build/abl
    python3 - <<'PY'
s = open("kccotgan_amd/csrc/cost_tile256.hip").read()
a = '                __builtin_amdgcn_raw_buffer_store_b128('
assert s.count(a) == 1
s = s.replace(a, '                if (!Q256_NO_ESTORE) __builtin_amdgcn_raw_buffer_store_b128(')
for h in ("common.h", "cost_internal.h", "options.h", "gram_q.h"):
    s = s.replace('#include "%s"' % h, '#include "../../kccotgan_amd/csrc/%s"' % h)
open("build/abl/cost_tile256_estore.hip", "w").write(s)
PY
    OBJS=$(ls kccotgan_amd/csrc/obj/*.o | grep -v "diag_\|cost_tile256.o")
    for v in 0 1; do
        /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I kccotgan_amd/csrc -DQ256_NO_ESTORE=$v -c build/abl/cost_tile256_estore.hip -o build/abl/es$v.o
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o build/libkccot_estore$v.so $OBJS build/abl/es$v.o && echo "built $v"
    done
else
    shift
    export TMPDIR=/tmp
    mkdir -p gpurun_out/estore
    for v in 0 1; do
        KCCOT_PRIME_LIB=$ROOT/build/libkccot_estore0.so KCCOT_LIB_PATH=$ROOT/build/libkccot_estore$v.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/estore/v$v -- python3 tools/bench_gram.py $1 > gpurun_out/estore/v$v.log 2>&1
        echo "== E stores $([ $v = 1 ] && echo removed || echo kept): $(grep 'cost stage' gpurun_out/estore/v$v.log)"
        f=$(find gpurun_out/estore/v$v -name "*kernel_stats.csv" | head -1)
        grep "gram_q256" "$f" | python3 -c "
import sys, csv
for r in csv.reader(sys.stdin):
    print('   %-45s calls %s avg %.3f ms' % (r[0][:45], r[1], float(r[3]) / 1e6))"
    done
fi
