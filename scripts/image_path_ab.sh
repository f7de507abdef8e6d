#!/bin/bash
# same-box A/B of the whole image path on stored frames (scripts/preprocess_bench.py, every row): the library of a previous commit against the tree's
cd "$(dirname "$0")/.."
for r in 1 2; do for v in prev tree; do
  lib=$PWD/triton-racer-sim_amd/csrc/libtrsim.so; [ $v = prev ] && lib=$PWD/scripts/ab_bin/libtrsim_prev.so
  echo "== $v"; TRS_HIP_LIB=$lib python scripts/preprocess_bench.py 1024 120 160 2>/dev/null | grep -v amdgpu
  TRS_HIP_LIB=$lib python scripts/preprocess_bench.py 256 240 320 2>/dev/null | grep -v amdgpu
done; done
