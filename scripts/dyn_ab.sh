#!/bin/bash
# same-box A/B of the dynamic-brightness step: the library of a previous commit (scripts/build_prev.sh) against the tree's, alternating
cd "$(dirname "$0")/.."
for r in 1 2 3; do
  echo "== prev"; TRS_HIP_LIB=$PWD/scripts/ab_bin/libtrsim_prev.so python scripts/filter_bench.py ${DYN_MODE:-} 2>/dev/null | grep -E "dynamic"
  echo "== tree"; python scripts/filter_bench.py ${DYN_MODE:-} 2>/dev/null | grep -E "dynamic"
done
