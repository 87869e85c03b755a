#!/bin/bash
# Rehearsal of `bench.py --gpus N` on ONE GPU: N ranks share cuda:0, collectives over gloo (host-staged: the timings mean
# nothing, the code path is the driver's).  N = 2 (headline sharded step; the `configs` block of every BASELINE config the ranks
# divide -- multi-GB shapes are skipped under gloo -- with all protocols and per-phase times; the weak-scaling loss line; the
# data-parallel train-steps/s children: 2 parents + 2 children on the card) and N = 4 (the same without the trainer children: a
# box admits six GPU processes).  usage: tools/gpu_bench_rehearsal.sh [tag]
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT" || exit 1
OUT=gpurun_out/${1:-rehearsal}
mkdir -p "$OUT"
export KCCOT_BENCH_BACKEND=gloo
for n in 2 4; do
  timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) \
      bench.py --gpus $n --steps 5 --warmup 2 $([ $n -gt 2 ] && echo --no-train) > "$OUT/bench_n$n.log" 2>&1 || { echo "N=$n failed"; tail -30 "$OUT/bench_n$n.log"; exit 1; }
  grep '^{' "$OUT/bench_n$n.log" | tail -1 > "$OUT/bench_n$n.json"
  python3 - "$OUT/bench_n$n.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("N=%d headline %.1f %s (%.3f ms)" % (d["n_gpus"], d["value"], d["unit"], d["ms_per_step"]))
for name, c in (d.get("configs") or {}).items():
    for proto, r in c["protocols"].items():
        print("  %-40s B=%-4d %-14s %s" % (name, c["B"], proto, ("%.2f ms  " % r["ms_fwd_bwd"] + " ".join("%s=%.2f" % kv for kv in r["phases_ms_max_over_ranks"].items())) if "ms_fwd_bwd" in r else r))
print("  dp train:", json.dumps(d.get("train_steps_per_sec_data_parallel"))[:700])
PY
done
