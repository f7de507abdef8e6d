#!/bin/bash
# A/B of two prebuilt libraries on the pilot loop's per-layer kernel times: scripts/ab_bin/libtrsim_prev.so against the in-tree build,
# alternating runs on the same box (box-to-box spread of this loop is +-6 %: only same-box pairs count).  ROUNDS=3 by default.
cd "$(dirname "$0")/.."
for round in $(seq 1 ${ROUNDS:-3}); do for v in new prev; do
  lib=$PWD/triton-racer-sim_amd/csrc/libtrsim.so; [ $v = prev ] && lib=$PWD/scripts/ab_bin/libtrsim_prev.so
  TRS_HIP_LIB=$lib PL_TAG=${v}_a bash scripts/pilot_layers.sh 2>&1 | grep -v amdgpu.ids
  TRS_HIP_LIB=$lib PL_TAG=${v}_b bash scripts/pilot_layers.sh --envs-per-gpu 512 --img-h 240 --img-w 320 --depth 2>&1 | grep -v amdgpu.ids
done; done
