#!/bin/bash
# round 4: dense1 with 64 frames per workgroup (default where K is long) against 32 (trs_pilot_tuning.dense = 2), 512 x 240x320 + depth
cd "$(dirname "$0")/.."
for d in 1 2; do
  PL_TAG=nf$d bash scripts/pilot_layers.sh --envs-per-gpu 512 --img-h 240 --img-w 320 --depth --pilot-tuning dense=$d 2>&1 | grep -E "==|dense1|tail|all kernels|bench"
done
