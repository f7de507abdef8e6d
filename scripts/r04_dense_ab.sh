#!/bin/bash
# round 4: dense1 at 512 x 240x320 + depth under more K slices (more bytes in flight per CU); per-layer times by rocprofv3 kernel trace
cd "$(dirname "$0")/.."
for ks in 0 32 48 64; do
  PL_TAG=d$ks bash scripts/pilot_layers.sh --envs-per-gpu 512 --img-h 240 --img-w 320 --depth --pilot-tuning ksplit=$ks 2>&1 | grep -v amdgpu.ids
done
