#!/bin/bash
# A/B of the cache-policy bits on the image stores (timing only; all variants produce identical images)
set -e
cd "$(dirname "$0")/.."
C=triton-racer-sim_amd/csrc
for a in 0 2 16 17 18; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -ffp-contract=off -fno-fast-math -DTRS_STORE_AUX=$a -o /tmp/libtrsim_aux$a.so $C/trsim_hip.hip $C/trsim_resident.hip $C/trsim_comm.hip $C/trsim_pilot.hip $C/trsim_tables.cpp -ldl -Iinclude 2>/dev/null &
done
wait
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"; }
for round in 1 2; do
for a in 0 2 16 17 18; do
  for cfg in "--envs-per-gpu 1024 --steps 2000" "--envs-per-gpu 512 --steps 2000" "--envs-per-gpu 16384 --steps 128"; do
    echo -n "aux=$a $cfg : "; TRS_HIP_LIB=/tmp/libtrsim_aux$a.so run $cfg
  done
done
done
