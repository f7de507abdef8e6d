#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r02_pilot6
mkdir -p $O
echo "== pilot tests" && timeout -k 10 500 python -m pytest tests/test_pilot.py -x -q > $O/tests.log 2>&1; rc=$?; tail -5 $O/tests.log; [ $rc -eq 0 ] || exit $rc
PL_TAG=h1 timeout -k 10 300 bash scripts/pilot_layers.sh 2>&1 | tee $O/layers_120.txt
