#!/bin/bash
# bench.py under torch.distributed with one rank (RCCL initialised, sums, reduces): one reduce of the whole image against the
# image finished in two row bands (only the frames of the last launch are rendered band by band).
# usage (GPU box, repo root): reduce_bands_ab.sh
for round in 1 2; do for b in 1 2; do for st in 20 36; do
  printf "steps %s reduce-bands %s: " $st $b
  MASTER_ADDR=127.0.0.1 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 2965$b \
    bench.py --gpus 1 --steps $st --warmup 5 --no-cpu-baseline --reduce-bands $b 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['timing']['region_ms'], d['reduce'])"
done; done; done
