#!/bin/bash
# A/B of experiment builds (tools/build_variants.sh): kernel-only time of the configs[1] cost kernel from bench.py.
# usage: tools/ab_variants.sh outdir reps name[:ENV=V[,ENV=V]] ...
set -o pipefail
OUT=gpurun_out/$1; REPS=$2; shift 2
mkdir -p "$OUT"
for rep in $(seq 1 $REPS); do
for spec in "$@"; do
    name=${spec%%:*}; envs=""
    [ "$spec" != "$name" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
    lib=$PWD/kccotgan_amd/csrc/libkccot_$name.so
    [ "$name" = "product" ] && lib=$PWD/kccotgan_amd/csrc/libkccot.so
    tag=$(echo "$spec" | tr ':,=' '___')_$rep
    env KCCOT_LIB_PATH=$lib $envs timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train --no-configs > "$OUT/$tag.json" 2> "$OUT/$tag.err" || { echo "$tag failed"; tail -5 "$OUT/$tag.err"; exit 1; }
    python - "$tag" "$OUT/$tag.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r=d["roofline"]
print("%-40s ms/step %.4f  cost-kernel %.2f us  frac %.3f" % (sys.argv[1], d["ms_per_step"], r.get("kernel_us", 0.0), r["frac"]))
PY
done
done
