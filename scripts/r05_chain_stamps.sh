#!/bin/bash
# phase stamps of trs_conv_chain16_kernel: `build` here, run on the GPU box
cd "$(dirname "$0")/.."
SRC=triton-racer-sim_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function -ldl -Iinclude"
if [ "$1" = build ]; then
  for a in 0 1 2 3; do
    /opt/rocm/bin/hipcc $FLAGS -DTRS_C16_STAMPS=1 -DTRS_C16_ABLATE=$a ${C16_EXTRA} -o scripts/ab_bin/libtrsim_r05_c16stamps$a.so $SRC/trsim_hip.hip $SRC/trsim_resident.hip $SRC/trsim_comm.hip $SRC/trsim_pilot.hip $SRC/trsim_tables.cpp &
  done
  wait
  exit 0
fi
for a in 0 1 2 3; do
  echo "#### ablate $a (1 = no weight refills, 2 = pixel fragments of the first k-steps only, 3 = no MFMA)"
  TRS_HIP_LIB=$PWD/scripts/ab_bin/libtrsim_r05_c16stamps$a.so python3 bench.py --no-cpu-baseline --pilot --envs-per-gpu 1024 --steps 4 --warmup 2 --pilot-tuning chain_mfma=16 2>&1 | grep "chain16 workgroup" | tail -4
done
