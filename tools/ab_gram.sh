#!/bin/bash
# A/B of the Gram kernel knobs: kernel-only time from bench.py's roofline block.
set -o pipefail
OUT=gpurun_out/${1:-abgram}
mkdir -p "$OUT"
run() {
    local tag=$1; shift
    env "$@" timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > "$OUT/$tag.json" 2> "$OUT/$tag.err" || { echo "$tag failed"; tail -5 "$OUT/$tag.err"; return 1; }
    python - "$tag" "$OUT/$tag.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r=d["roofline"]
print("%-22s ms/step %.4f  cost-kernel %.2f us  cost-stage %.1f us  hbm-frac %.3f" % (sys.argv[1], d["ms_per_step"], r["kernel_us"], r["cost_stage_us"], r["hbm_frac"]))
PY
}
for rep in 1 2; do
run default_$rep A=1 &&
run deep_$rep KCCOT_GRAM_DEEP=1 &&
run wgs256_$rep KCCOT_GRAM_WGS=256 &&
run wgs480_$rep KCCOT_GRAM_WGS=480 || exit 1
done
