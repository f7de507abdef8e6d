#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r02_pilot5
mkdir -p $O
echo "== pilot tests" && timeout -k 10 500 python -m pytest tests/test_pilot.py tests/test_pilot_types.py -x -q > $O/tests.log 2>&1; rc=$?; tail -5 $O/tests.log; [ $rc -eq 0 ] || exit $rc
for ks in 0 8 16 24 72; do
  echo "== TRS_PILOT_KSPLIT=$ks (0 = default)"
  if [ $ks = 0 ]; then unset TRS_PILOT_KSPLIT; else export TRS_PILOT_KSPLIT=$ks; fi
  PL_TAG=ks$ks timeout -k 10 300 bash scripts/pilot_layers.sh 2>&1 | grep "dense1\|tail\|all kernels\|bench"
done | tee $O/ksplit.txt
