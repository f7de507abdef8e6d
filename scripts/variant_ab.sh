#!/bin/bash
set -e
cd "$(dirname "$0")/.."
C=triton-racer-sim_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -ffp-contract=off -fno-fast-math"
/opt/rocm/bin/hipcc $F -DTRS_SINGLE_VARIANT -o /tmp/libtrsim_single.so $C/trsim_hip.hip $C/trsim_pilot.hip $C/trsim_tables.cpp 2>/dev/null &
/opt/rocm/bin/hipcc $F -o /tmp/libtrsim_both.so $C/trsim_hip.hip $C/trsim_pilot.hip $C/trsim_tables.cpp 2>/dev/null &
wait
run() { python bench.py --no-cpu-baseline --envs-per-gpu 1024 --steps 3000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"; }
for r in 1 2 3; do
  echo -n "single: "; TRS_HIP_LIB=/tmp/libtrsim_single.so run
  echo -n "both:   "; TRS_HIP_LIB=/tmp/libtrsim_both.so run
done
