#!/bin/bash
# Where do the fused head's LDS bank-conflict cycles come from?  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of trs_conv12_band_kernel in ablation builds
# (results wrong on purpose, only the counters matter): fa1 = no conv1 phase, fa2 = no conv2 phase, c2a1 = conv2 without fragment refills (reads of the
# first kC2Depth k-steps only).  `build` on the CPU box, the rest on the GPU box.
cd "$(dirname "$0")/.."
SRC=triton-racer-sim_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function -ldl -Iinclude"
mkdir -p scripts/ab_bin
if [ "$1" = build ]; then
  /opt/rocm/bin/hipcc $FLAGS -DTRS_FUSE_ABLATE=1 -o scripts/ab_bin/libtrsim_r05_fa1.so $SRC/trsim_hip.hip $SRC/trsim_resident.hip $SRC/trsim_comm.hip $SRC/trsim_pilot.hip $SRC/trsim_tables.cpp &
  /opt/rocm/bin/hipcc $FLAGS -DTRS_FUSE_ABLATE=2 -o scripts/ab_bin/libtrsim_r05_fa2.so $SRC/trsim_hip.hip $SRC/trsim_resident.hip $SRC/trsim_comm.hip $SRC/trsim_pilot.hip $SRC/trsim_tables.cpp &
  /opt/rocm/bin/hipcc $FLAGS -DTRS_C2_ABLATE=1 -o scripts/ab_bin/libtrsim_r05_c2a1.so $SRC/trsim_hip.hip $SRC/trsim_resident.hip $SRC/trsim_comm.hip $SRC/trsim_pilot.hip $SRC/trsim_tables.cpp &
  wait
  exit 0
fi
for v in cur fa1 fa2 c2a1; do
  lib=$PWD/triton-racer-sim_amd/csrc/libtrsim.so; [ $v != cur ] && lib=$PWD/scripts/ab_bin/libtrsim_r05_$v.so
  echo "#### $v"
  TRS_HIP_LIB=$lib PL_TAG=r05h_$v PL_SQ_ONLY=1 bash scripts/pilot_pmc.sh "$@" 2>&1 | grep -v amdgpu.ids
done
