#!/bin/bash
# A/B of the conv4-7 chain on the two MFMA shapes, same box, alternating (bench.py --pilot --pilot-tuning chain_mfma=16|32): per-layer kernel times under the tracer.
# 16w8 = the 16x16x32 chain built with 8 waves per workgroup (scripts/ab_bin/libtrsim_r05_c16w8.so, -DTRS_C16_WAVES=8 -DTRS_CHAIN16_R=4)
cd "$(dirname "$0")/.."
for round in $(seq 1 ${ROUNDS:-2}); do for m in 32 16 16w8; do
  lib=$PWD/triton-racer-sim_amd/csrc/libtrsim.so; t=$m
  [ $m = 16w8 ] && { lib=$PWD/scripts/ab_bin/libtrsim_r05_c16w8.so; t=16; }
  echo "#### chain_mfma=$m"
  TRS_HIP_LIB=$lib PL_TAG=r05c_$m bash scripts/pilot_layers.sh --pilot-tuning chain_mfma=$t 2>&1 | grep -v amdgpu.ids | grep "conv4\|all kernels\|bench"
done; done
