#!/bin/bash
# Timing-only ablation builds of the step kernel (outputs are wrong by construction; never used by tests).
set -e
cd "$(dirname "$0")/.."
C=triton-racer-sim_amd/csrc
for a in 1 2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -ffp-contract=off -fno-fast-math -DTRS_ABLATE=$a -o /tmp/libtrsim_ab$a.so $C/trsim_hip.hip $C/trsim_resident.hip $C/trsim_comm.hip $C/trsim_pilot.hip $C/trsim_tables.cpp -ldl -Iinclude 2>/dev/null
done
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"; }
for cfg in "--envs-per-gpu 16384 --steps 128" "--envs-per-gpu 1024 --steps 2048 --steps-per-launch 16" "--envs-per-gpu 1024 --steps 2000"; do
  echo "== $cfg"
  echo -n "full       "; run $cfg
  echo -n "no-store   "; TRS_HIP_LIB=/tmp/libtrsim_ab1.so run $cfg
  echo -n "store-only "; TRS_HIP_LIB=/tmp/libtrsim_ab2.so run $cfg
done
