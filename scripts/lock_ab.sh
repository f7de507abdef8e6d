#!/bin/bash
# same-box A/B of the lock-step tick: the library of a previous commit (scripts/build_prev.sh) against the tree's
cd "$(dirname "$0")/.."
for r in 1 2; do
  echo "== prev"; TRS_HIP_LIB=$PWD/scripts/ab_bin/libtrsim_prev.so python scripts/lock_bench.py 2>/dev/null | tail -2
  echo "== tree"; python scripts/lock_bench.py 2>/dev/null | tail -2
done
