#!/bin/bash
# same-box A/B of the physics step: configs[1] (256 envs, no camera) posted tick by tick / per launch, and the closed loop's env step
cd "$(dirname "$0")/.."
B="python bench.py --no-cpu-baseline --no-also"
run() { $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', d['value'], 'env-steps/s', d['ms_per_step']*1e3, 'us per step')"; }
for r in 1 2; do for v in prev tree; do
  lib=$PWD/triton-racer-sim_amd/csrc/libtrsim.so; [ $v = prev ] && lib=$PWD/scripts/ab_bin/libtrsim_prev.so
  echo "== $v"
  TRS_HIP_LIB=$lib run --envs-per-gpu 256 --steps 4000 --no-render
  TRS_HIP_LIB=$lib run --envs-per-gpu 256 --steps 4000 --no-render --step-mode launch
  TRS_HIP_LIB=$lib run --step-mode launch
done; done
