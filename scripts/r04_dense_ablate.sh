#!/bin/bash
# round 4: what bounds dense1 at 512 x 240x320 (37 us for 72 MB of activations)?  Timing-only builds: 1 = no weight stream, 2 = activations from cache
cd "$(dirname "$0")/.."
for a in 1 2; do
  TRS_HIP_LIB=$PWD/scripts/ab_bin/libtrsim_dab$a.so PL_TAG=dab$a bash scripts/pilot_layers.sh --envs-per-gpu 512 --img-h 240 --img-w 320 --depth 2>&1 | grep -E "==|dense1|tail|all kernels"
done
