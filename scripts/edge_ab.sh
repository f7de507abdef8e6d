#!/bin/bash
# phase ablation of trs_preprocess_edge_kernel (timing-only builds scripts/ab_bin/libtrsim_edgeK.so, -DTRS_EDGE_ABLATE=K)
cd "$(dirname "$0")/.."
for k in 0 1 2 3 4 5; do
  lib=$PWD/scripts/ab_bin/libtrsim_edge$k.so; [ $k = 0 ] && lib=$PWD/triton-racer-sim_amd/csrc/libtrsim.so
  [ -f $lib ] || continue
  echo "== TRS_EDGE_ABLATE=$k (1 no Sobel, 2 no suppression, 3 no hysteresis, 4 no output phase, 5 no trim phase)"
  TRS_HIP_LIB=$lib python scripts/preprocess_bench.py 1024 120 160 2>/dev/null | grep Canny
  TRS_HIP_LIB=$lib python scripts/preprocess_bench.py 256 240 320 2>/dev/null | grep Canny
done
