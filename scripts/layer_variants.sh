#!/bin/bash
# traced per-layer times of the pilot loop for prebuilt variants scripts/ab_bin/libtrsim_v_<tag>.so (and `prev`), alternating, ROUNDS times;
# LAYER = the pattern of the lines to show (default: conv3), extra arguments of bench.py in ARGS
cd "$(dirname "$0")/.."
for round in $(seq 1 ${ROUNDS:-2}); do for tag in "$@"; do
  lib=$PWD/scripts/ab_bin/libtrsim_v_$tag.so; [ $tag = prev ] && lib=$PWD/scripts/ab_bin/libtrsim_prev.so
  echo -n "$tag: "; TRS_HIP_LIB=$lib PL_TAG=v_$tag bash scripts/pilot_layers.sh $ARGS 2>&1 | grep "${LAYER:-conv3}\|all kernels" | tr '\n' ' '; echo
done; done
