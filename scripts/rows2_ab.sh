#!/bin/bash
# same-box A/B of the raster's ground rows (two rows per iteration against one): resident 1024 / 512 / 256 envs, launch mode, lock step, closed loop
cd "$(dirname "$0")/.."
B="python bench.py --no-cpu-baseline --no-also"
run() { $B "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ', ' '.join(sys.argv[1:]) or '(default)', d['value'], 'env-steps/s', round(d['ms_per_step']*1e3,3), 'us per step')" "$@"; }
for r in 1 2; do for v in prev tree; do
  lib=$PWD/triton-racer-sim_amd/csrc/libtrsim.so; [ $v = prev ] && lib=$PWD/scripts/ab_bin/libtrsim_prev.so
  echo "== $v"
  export TRS_HIP_LIB=$lib
  run
  run --envs-per-gpu 512
  run --envs-per-gpu 256
  run --envs-per-gpu 4096 --steps 500
  run --step-mode launch
  run --steps 600 --depth
  run --pilot --steps 200 --warmup 60
done; done
