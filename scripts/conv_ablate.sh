#!/bin/bash
# ablation builds of trs_conv_lt_kernel (results are wrong on purpose; only the per-layer times matter)
cd "$(dirname "$0")/.."
SRC="triton-racer-sim_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -ffp-contract=off -fno-fast-math"
for a in 0 1 2 3 4; do
  /opt/rocm/bin/hipcc $FLAGS -DTRS_CONV_ABLATE=$a -o /tmp/libtrsim_abl$a.so $SRC/trsim_hip.hip $SRC/trsim_resident.hip $SRC/trsim_comm.hip $SRC/trsim_pilot.hip $SRC/trsim_tables.cpp -ldl -Iinclude || exit 1
  echo "#### ablate=$a (0 full, 1 no steady loads, 2 no LDS transpose, 3 no MFMA, 4 no stores)"
  TRS_HIP_LIB=/tmp/libtrsim_abl$a.so PL_TAG=abl$a scripts/pilot_layers.sh | grep -v "env step\|tail\|dense1"
done
