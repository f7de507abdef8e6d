#!/bin/bash
# build the library of a commit (default HEAD) into scripts/ab_bin/libtrsim_prev.so for same-box A/B runs (scripts/pilot_lib_ab.sh, scripts/lib_ab.sh)
set -e
cd "$(dirname "$0")/.."
rev=${1:-HEAD}
tmp=$(mktemp -d)
git archive "$rev" triton-racer-sim_amd/csrc include | tar -x -C "$tmp"
mkdir -p scripts/ab_bin
src=$tmp/triton-racer-sim_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -ldl -fno-slp-vectorize \
  -I"$tmp/include" -o scripts/ab_bin/libtrsim_prev.so $src/trsim_hip.hip $src/trsim_resident.hip $src/trsim_comm.hip $src/trsim_pilot.hip $src/trsim_tables.cpp
rm -rf "$tmp"
echo "built scripts/ab_bin/libtrsim_prev.so from $rev"
