#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r02_pilot3
mkdir -p $O
for a in 0 1 2 3 4; do
  echo "== ablate $a (0 = product; 1 no weight refills, 2 one LDS pixel read per item, 3 no stores, 4 no staging)"
  if [ $a = 0 ]; then unset TRS_HIP_LIB; else export TRS_HIP_LIB=$PWD/scripts/ab_bin/libtrsim_fa$a.so; fi
  TRS_PILOT_FRAME_F=4 PL_TAG=abl$a timeout -k 10 300 bash scripts/pilot_layers.sh 2>&1 | grep "conv4\|conv5\|conv6\|conv7\|all kernels"
done | tee $O/ablate.txt
