#!/bin/bash
# Phase times of the fused head (trs_conv12_band_kernel) by ablation builds made HERE (results are wrong on purpose, only the kernel's time
# matters): TRS_FUSE_ABLATE 1 = no conv1 phase, 2 = no conv2 phase.  Builds go to scripts/ab_bin (git-ignored) when `build` is given — run
# that on the CPU box, the timing part on the GPU box.  HEAD_SRC=dir takes another source tree (an older commit extracted by git archive).
cd "$(dirname "$0")/.."
SRC=${HEAD_SRC:-.}/triton-racer-sim_amd/csrc; INC=${HEAD_SRC:-.}/include; TAG=${HEAD_TAG:-cur}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function -ldl -I$INC"
mkdir -p scripts/ab_bin
if [ "$1" = build ]; then
  for a in 1 2; do
    /opt/rocm/bin/hipcc $FLAGS -DTRS_FUSE_ABLATE=$a -o scripts/ab_bin/libtrsim_${TAG}_fa$a.so $SRC/trsim_hip.hip $SRC/trsim_resident.hip $SRC/trsim_comm.hip $SRC/trsim_pilot.hip $SRC/trsim_tables.cpp || exit 1
  done
  exit 0
fi
for tag in "$@"; do for a in 1 2; do
  echo "#### $tag ablate=$a (1 no conv1 phase, 2 no conv2 phase)"
  TRS_HIP_LIB=$PWD/scripts/ab_bin/libtrsim_${tag}_fa$a.so PL_TAG=${tag}_fa$a bash scripts/pilot_layers.sh 2>&1 | grep "conv1+2"
done; done
