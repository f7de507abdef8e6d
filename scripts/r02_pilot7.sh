#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r02_pilot7
mkdir -p $O
for a in 1 2; do
  echo "== fused head, TRS_FUSE_ABLATE=$a (1 = no conv1 phase, 2 = no conv2 phase)"
  TRS_HIP_LIB=$PWD/scripts/ab_bin/libtrsim_fuse$a.so PL_TAG=fa$a timeout -k 10 300 bash scripts/pilot_layers.sh 2>&1 | grep "conv1+2\|all kernels"
done | tee $O/fuse_ablate.txt
