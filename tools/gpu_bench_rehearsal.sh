#!/bin/bash
# Rehearsal of `bench.py --gpus N` on ONE GPU: N ranks share cuda:0, collectives over gloo (host-staged: the timings mean
# nothing, the code path is the driver's).  N = 2 (configs[1] sharded step) and N = 4 (adds the configs[2] block with both
# protocols).  usage: tools/gpu_bench_rehearsal.sh [tag]
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT" || exit 1
OUT=gpurun_out/${1:-rehearsal}
mkdir -p "$OUT"
export KCCOT_BENCH_BACKEND=gloo
for n in 2 4; do
  timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) \
      bench.py --gpus $n --steps 5 --warmup 2 --no-train > "$OUT/bench_n$n.log" 2>&1 || { echo "N=$n failed"; tail -30 "$OUT/bench_n$n.log"; exit 1; }
  grep '^{' "$OUT/bench_n$n.log" | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=%d' % d['n_gpus'], d['value'], d['unit'], d['ms_per_step'], json.dumps(d.get('configs'))[:600])"
done
