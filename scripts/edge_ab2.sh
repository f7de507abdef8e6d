#!/bin/bash
# Canny kernel: in-tree build against other builds of the same ABI (scripts/ab_bin/libtrsim_<tag>.so), alternating; + parity of the image path
cd "$(dirname "$0")/.."
for round in 1 2; do
for tag in tree "$@"; do
  lib=$PWD/scripts/ab_bin/libtrsim_$tag.so; [ $tag = tree ] && lib=$PWD/triton-racer-sim_amd/csrc/libtrsim.so
  [ -f $lib ] || continue
  echo "== $tag"
  TRS_HIP_LIB=$lib python scripts/preprocess_bench.py 1024 120 160 2>/dev/null | grep -E "Canny|HSV"
  TRS_HIP_LIB=$lib python scripts/preprocess_bench.py 256 240 320 2>/dev/null | grep -E "Canny|HSV"
done
done
