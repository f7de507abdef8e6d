#!/bin/bash
# phase stamps of the fused head from -DTRS_BAND_STAMPS=1 builds: scripts/ab_bin/libtrsim_st_<tag>.so for each tag given; also the kernel's traced time
cd "$(dirname "$0")/.."
for tag in "$@"; do
  lib=$PWD/scripts/ab_bin/libtrsim_st_$tag.so
  echo "#### $tag"
  TRS_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-also --pilot --envs-per-gpu 1024 --steps 2 --warmup 1 2>&1 | grep "band head\|conv1 tile loop" | tail -n 7 | sort
  TRS_HIP_LIB=$lib PL_TAG=st_$tag bash scripts/pilot_layers.sh 2>&1 | grep "conv1+2"
done
