#!/bin/bash
# A/B: raster / physics waves per workgroup of the step kernel (alternating runs in one session)
cd "$(dirname "$0")/.."
C=triton-racer-sim_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -ffp-contract=off -fno-fast-math"
CFGS=${WAVES_CFGS:-"10,5 8,4"}
for cfg in $CFGS; do r=${cfg%,*}; ph=${cfg#*,}
  /opt/rocm/bin/hipcc $FLAGS -DTRS_RASTER_WAVES=$r -DTRS_PHYS_WAVES=$ph -o /tmp/libtrsim_w${r}_$ph.so $C/trsim_hip.hip $C/trsim_resident.hip $C/trsim_comm.hip $C/trsim_pilot.hip $C/trsim_tables.cpp -ldl -Iinclude 2>/dev/null &
done; wait
run() { python bench.py --no-cpu-baseline --no-also "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,2), 'us', d['roofline']['frac'])"; }
for round in 1 2 3; do for cfg in $CFGS; do r=${cfg%,*}; ph=${cfg#*,}
  for b in "--envs-per-gpu 1024 --steps 2000" "--envs-per-gpu 1024 --steps 2000 --steps-per-launch 8" "--envs-per-gpu 512 --steps 2000" "--envs-per-gpu 256 --steps 2000" "--envs-per-gpu 4096 --steps 500"; do
    echo -n "raster=$r phys=$ph $b : "; TRS_HIP_LIB=/tmp/libtrsim_w${r}_$ph.so run $b
  done
done; done
