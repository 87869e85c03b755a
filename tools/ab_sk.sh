#!/bin/bash
# A/B of the Sinkhorn lane geometry (lanes per line) at n = 64: graph-replayed step, shortcut off.
set -o pipefail
OUT=gpurun_out/${1:-absk}
mkdir -p "$OUT"
run() {
    local tag=$1; shift
    env "$@" timeout -k 10 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > "$OUT/$tag.json" 2> "$OUT/$tag.err" || { echo "$tag failed"; tail -5 "$OUT/$tag.err"; return 1; }
    python - "$tag" "$OUT/$tag.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
w=d["with_exact_shortcut"]
print("%-12s ms/step %.4f   near(shortcut) %.4f  far(shortcut) %.4f" % (sys.argv[1], d["ms_per_step"], w["near"]["ms_per_step"], w["far"]["ms_per_step"]))
PY
}
for rep in 1 2; do
run lpr16_$rep A=1 &&
run lpr8_$rep KCCOT_SK_LPR=8 &&
run lpr4_$rep KCCOT_SK_LPR=4 || exit 1
done
