#!/bin/bash
# kernel time of the fused head in the ablation builds of scripts/r05_head_lds.sh (same box, two rounds)
cd "$(dirname "$0")/.."
for round in 1 2; do for v in cur fa1 fa2 c2a1; do
  lib=$PWD/triton-racer-sim_amd/csrc/libtrsim.so; [ $v != cur ] && lib=$PWD/scripts/ab_bin/libtrsim_r05_$v.so
  echo "#### $v"
  TRS_HIP_LIB=$lib PL_TAG=r05t_$v bash scripts/pilot_layers.sh "$@" 2>&1 | grep "conv1+2\|all kernels"
done; done
