#!/bin/bash
# A/B of two prebuilt libraries: scripts/ab_bin/libtrsim_base.so (baseline) vs the in-tree build, alternating runs
cd "$(dirname "$0")/.."
run() { python bench.py --no-cpu-baseline --no-also "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,2), 'us', d['roofline']['frac'])"; }
for round in 1 2; do for v in base new; do
  lib=$PWD/triton-racer-sim_amd/csrc/libtrsim.so; [ $v = base ] && lib=$PWD/scripts/ab_bin/libtrsim_base.so
  for b in "--envs-per-gpu 1024 --steps 2000" "--envs-per-gpu 512 --steps 2000" "--envs-per-gpu 256 --steps 2000" "--envs-per-gpu 1024 --steps 2000 --step-mode launch" "--envs-per-gpu 1024 --steps 2000 --steps-per-launch 8 --step-mode launch" "--envs-per-gpu 512 --steps 300 --img-h 240 --img-w 320 --depth"; do
    echo -n "$v $b : "; TRS_HIP_LIB=$lib run $b
  done
done; done
