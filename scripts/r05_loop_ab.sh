#!/bin/bash
# closed pilot loop, whole step by HIP events, in-tree library against scripts/ab_bin/libtrsim_prev.so (scripts/build_prev.sh <rev>), alternating on one box
cd "$(dirname "$0")/.."
one() { python3 bench.py --pilot --no-cpu-baseline "$@" | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ', round(d['value']), 'env-steps/s', d['roofline']['avg_step_us'], 'us per step')"; }
for round in $(seq 1 ${ROUNDS:-3}); do for v in new prev; do
  lib=$PWD/triton-racer-sim_amd/csrc/libtrsim.so; [ $v = prev ] && lib=$PWD/scripts/ab_bin/libtrsim_prev.so
  echo "$v 1024 x 120x160"; TRS_HIP_LIB=$lib one --steps 200 --warmup 60
  echo "$v 512 x 240x320 + depth"; TRS_HIP_LIB=$lib one --steps 80 --warmup 30 --envs-per-gpu 512 --img-h 240 --img-w 320 --depth
done; done
