#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r02_pilot2
mkdir -p $O
echo "== pilot tests" && timeout -k 10 500 python -m pytest tests/test_pilot.py -x -q > $O/tests.log 2>&1; rc=$?; tail -5 $O/tests.log; [ $rc -eq 0 ] || exit $rc
echo "== pilot tests (deep)" && TRS_PILOT_FRAME_DEEP=1 timeout -k 10 500 python -m pytest tests/test_pilot.py -x -q -k "forward or fused" > $O/tests_deep.log 2>&1; rc=$?; tail -3 $O/tests_deep.log; [ $rc -eq 0 ] || exit $rc
for deep in 0 1; do for f in 4; do
  echo "== F=$f deep=$deep"
  if [ $deep = 1 ]; then export TRS_PILOT_FRAME_DEEP=1; else unset TRS_PILOT_FRAME_DEEP; fi
  TRS_PILOT_FRAME_F=$f PL_TAG=f${f}d$deep timeout -k 10 300 bash scripts/pilot_layers.sh 2>&1 | grep "conv4\|conv5\|conv6\|conv7\|all kernels\|bench"
done; done | tee $O/sweep.txt
