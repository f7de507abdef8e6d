#!/bin/bash
# the pilot loop with dense1 + tail in one launch: per-layer trace at both frame formats, A/B against the two older forms, untraced bench lines
cd "$(dirname "$0")/.."
out=gpurun_out/r02_pilot_dense.txt
: > $out
PL_TAG=dense2 bash scripts/pilot_layers.sh >> $out 2>&1
TRS_PILOT_DENSE=1 PL_TAG=dense1 bash scripts/pilot_layers.sh >> $out 2>&1
TRS_PILOT_DENSE=0 PL_TAG=dense0 bash scripts/pilot_layers.sh >> $out 2>&1
PL_TAG=dense2_c5 bash scripts/pilot_layers.sh --envs-per-gpu 512 --img-h 240 --img-w 320 --depth >> $out 2>&1
for i in 1 2; do
python3 bench.py --no-cpu-baseline --pilot --envs-per-gpu 1024 --steps 200 --warmup 20 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('untraced 1024x120x160:', d['value'], d['roofline']['achieved'], d['roofline']['frac'])" >> $out
done
python3 bench.py --no-cpu-baseline --pilot --envs-per-gpu 512 --img-h 240 --img-w 320 --depth --steps 100 --warmup 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('untraced 512x240x320+depth:', d['value'], d['roofline']['achieved'], d['roofline']['frac'])" >> $out
cat $out
