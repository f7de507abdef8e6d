#!/bin/bash
# closed-loop pilot bench under alternative kernel choices (trs_pilot_tuning fields), alternating with the defaults, two rounds
# usage: scripts/tuning_ab.sh "frame5_f=1" "chain_nb=0" ...
cd "$(dirname "$0")/.."
pl() { python bench.py --no-cpu-baseline --pilot "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,3), 'M', d['roofline']['avg_step_us'], 'us', d['roofline']['frac'])"; }
for round in 1 2; do
for t in "" "$@"; do
  extra=(); [ -n "$t" ] && extra=(--pilot-tuning "$t")
  echo -n "[${t:-defaults}] 1024x120x160: "; pl --steps 150 --warmup 60 "${extra[@]}"
  echo -n "[${t:-defaults}] 512x240x320+d: "; pl --steps 60 --warmup 30 --envs-per-gpu 512 --img-h 240 --img-w 320 --depth "${extra[@]}"
done
done
