#!/bin/bash
# A/B of the step kernel: library built from an older git ref vs the working tree, alternating runs in one session.
#   here (has .git):    scripts/step_ab.sh prepare <ref>      -> copies that ref's csrc into scripts/ab_old/ (git-ignored)
#   on the GPU box:     gpurun -- 'bash scripts/step_ab.sh'
set -e
cd "$(dirname "$0")/.."
if [ "$1" = prepare ]; then
  mkdir -p scripts/ab_old
  for f in trsim_hip.hip trsim_pilot.hip trsim_tables.cpp trsim_tables.hpp trsim_internal.hpp; do git show "${2:-HEAD}:triton-racer-sim_amd/csrc/$f" > scripts/ab_old/$f; done
  exit 0
fi
C=triton-racer-sim_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -ffp-contract=off -fno-fast-math"
/opt/rocm/bin/hipcc $FLAGS -I. -o /tmp/libtrsim_new.so $C/trsim_hip.hip $C/trsim_resident.hip $C/trsim_comm.hip $C/trsim_pilot.hip $C/trsim_tables.cpp -ldl -Iinclude 2>/dev/null &
( cd scripts/ab_old && /opt/rocm/bin/hipcc $FLAGS -o /tmp/libtrsim_old.so trsim_hip.hip trsim_pilot.hip trsim_tables.cpp 2>/dev/null ) &
wait
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M', d['ms_per_step']*1e3, 'us', d['roofline']['frac'])"; }
for round in 1 2 3; do
 for cfg in "--envs-per-gpu 1024 --steps 2000" "--envs-per-gpu 1024 --steps 2000 --steps-per-launch 8" "--envs-per-gpu 4096 --steps 500" "--envs-per-gpu 16384 --steps 128" "--envs-per-gpu 16384 --steps 128 --steps-per-launch 8"; do
  for v in old new; do echo -n "$v $cfg : "; TRS_HIP_LIB=/tmp/libtrsim_$v.so run $cfg; done
 done
done
