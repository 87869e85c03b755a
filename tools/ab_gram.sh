#!/bin/bash
# A/B of the Gram kernel knobs: kernel-only time from bench.py's roofline block.
set -o pipefail
OUT=gpurun_out/${1:-abgram}
mkdir -p "$OUT"
run() {
    local tag=$1; shift
    env "$@" timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-train --no-configs > "$OUT/$tag.json" 2> "$OUT/$tag.err" || { echo "$tag failed"; tail -5 "$OUT/$tag.err"; return 1; }
    python - "$tag" "$OUT/$tag.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r=d["roofline"]
print("%-22s ms/step %.4f  cost-kernel %.2f us  frac %.3f" % (sys.argv[1], d["ms_per_step"], r.get("kernel_us", 0.0), r["frac"]))
PY
}
for rep in 1 2; do
run shallow_$rep KCCOT_GRAM_DEEP=0 &&
run deep_$rep KCCOT_GRAM_DEEP=1 &&
run deep_wgs128_$rep KCCOT_GRAM_DEEP=1 KCCOT_GRAM_WGS=128 &&
run deep_wgs192_$rep KCCOT_GRAM_DEEP=1 KCCOT_GRAM_WGS=192 &&
run deep_wgs256_$rep KCCOT_GRAM_DEEP=1 KCCOT_GRAM_WGS=256 || exit 1
done
