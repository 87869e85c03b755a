for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-configs --no-pmc 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('K=20', d['ms_per_step'], d['settle_steps_before_warmup'], d['event_timing']['median_ms'])"; done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --no-train --no-configs 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('N=2', d['ms_per_step'], d['settle_steps_before_warmup'])"
