#!/bin/bash
# round 5: bench.py with and without its process groups at world size 1 (the N > 1 code path on a one-GPU box), then the pieces one by one (scripts/r05_dist_probe.py)
cd "$(dirname "$0")/.."
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']/1e6,2), 'M', d['ms_per_step']*1e3, 'us/step  frac(events)', d['roofline']['frac'], d['config'].get('step_mode_at_end'))"; }
B="--no-also --no-cpu-baseline"
for round in 1 2; do
python bench.py --steps 20 --warmup 5 $B 2>/dev/null | pr "plain --steps 20          "
python bench.py --force-dist --steps 20 --warmup 5 $B 2>/dev/null | pr "--force-dist --steps 20   "
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2954$round bench.py --gpus 1 --force-dist --steps 20 --warmup 5 $B 2>/dev/null | pr "torchrun --force-dist 20  "
python bench.py --steps 2000 --warmup 200 $B 2>/dev/null | pr "plain --steps 2000        "
python bench.py --force-dist --steps 2000 --warmup 200 $B 2>/dev/null | pr "--force-dist --steps 2000 "
done
for m in none lazy+gather nccl all; do timeout -k 10 120 python scripts/r05_dist_probe.py $m 2>/dev/null | grep us/step; done
