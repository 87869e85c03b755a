#!/bin/bash
# A/B of the tuning knobs on one box, one process per setting (same device): prints one line each.
set -o pipefail
OUT=gpurun_out/${1:-ab}
mkdir -p "$OUT"
run() {
    local tag=$1; shift
    env "$@" timeout -k 10 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > "$OUT/$tag.json" 2> "$OUT/$tag.err" || { echo "$tag failed"; tail -5 "$OUT/$tag.err"; return 1; }
    python - "$tag" "$OUT/$tag.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r=d["roofline"]
print("%-22s ms/step %.4f  cost-kernel %.1f us  cost-stage %.1f us  hbm-frac %.3f" % (sys.argv[1], d["ms_per_step"], r["kernel_us"], r["cost_stage_us"], r["hbm_frac"]))
PY
}
for rep in 1 2; do
run default_$rep A=1 &&
run ws0_$rep KCCOT_GRAM_WS=0 &&
run wgs256_$rep KCCOT_GRAM_WGS=300 &&
run wgs480_$rep KCCOT_GRAM_WGS=480 &&
run f32_$rep KCCOT_GRAM_F32=1 || exit 1
done
