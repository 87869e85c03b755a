#!/bin/bash
# same-box A/B of the configs[3] / configs[4] loss step with the video gradient on 256 x 256 tiles (apply_q256 = 1) and without
mkdir -p gpurun_out/ab_cfg5
for rep in 1 2; do
  for o in 0 1; do
    KCCOT_OPTIONS="apply_q256=$o" timeout -k 10 300 python3 tools/bench_configs.py cfg4 cfg5 > gpurun_out/ab_cfg5/o${o}_$rep.log 2>&1
    python3 - gpurun_out/ab_cfg5/o${o}_$rep.log $o $rep <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print("apply_q256=%s rep %s  %s  B=%d  %s" % (sys.argv[2], sys.argv[3], d["config"], d["B"], {k: round(v, 3) for k, v in d.items() if k.startswith("ms")}))
PY
  done
done
