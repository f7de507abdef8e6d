#!/bin/bash
# small shards (512 / 256 / 128 envs per GPU, resident worker, every step posted on its own): the in-tree build against scripts/ab_bin/libtrsim_prev.so, alternating
cd "$(dirname "$0")/.."
for round in 1 2 3; do for v in new prev; do
  lib=$PWD/triton-racer-sim_amd/csrc/libtrsim.so; [ $v = prev ] && lib=$PWD/scripts/ab_bin/libtrsim_prev.so
  echo -n "$v "
  for n in 1024 512 256 128; do
    TRS_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-also --envs-per-gpu $n --steps 3000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('| $n envs', round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,3), 'us', end=' ')"
  done; echo
done; done
