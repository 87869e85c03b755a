#!/bin/bash
# Same-box A/B of the configs[1] cost stage: alternates the library builds given as arguments (paths relative to the repo
# root; default: csrc/libkccot_base.so = a build of an earlier commit, then the current csrc/libkccot.so), three rounds.
# usage: tools/ab_cost_stage.sh [tag] [lib ...]
set -o pipefail
TAG=${1:-ab}; shift
LIBS=("$@"); [ ${#LIBS[@]} -eq 0 ] && LIBS=(kccotgan_amd/csrc/libkccot_base.so kccotgan_amd/csrc/libkccot.so)
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
for round in 1 2 3; do
  for entry in "${LIBS[@]}"; do          # "path" or "path:name=value,name=value" (KCCOT_OPTIONS for that run)
    l=${entry%%:*}; o=""; [ "$l" != "$entry" ] && o=${entry#*:}
    KCCOT_OPTIONS="$o" KCCOT_LIB_PATH="$PWD/$l" timeout -k 10 200 python tools/ab_cost_stage.py 200 2>>"$OUT/err.log" | tee -a "$OUT/ab.jsonl" || exit $?
  done
done
