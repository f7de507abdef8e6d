#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r02_pilot1
mkdir -p $O
echo "== pilot tests" && timeout -k 10 500 python -m pytest tests/test_pilot.py tests/test_pilot_types.py -x -q > $O/tests.log 2>&1; rc=$?; tail -12 $O/tests.log; [ $rc -eq 0 ] || exit $rc
PL_TAG=frame timeout -k 10 300 bash scripts/pilot_layers.sh 2>&1 | tee $O/layers_frame.txt
PL_TAG=lt TRS_PILOT_FRAME_LAYERS=0 timeout -k 10 300 bash scripts/pilot_layers.sh 2>&1 | tee $O/layers_lt.txt
PL_TAG=frame5 timeout -k 10 300 bash scripts/pilot_layers.sh --envs-per-gpu 512 --img-h 240 --img-w 320 --depth 2>&1 | tee $O/layers_frame5.txt
for f in 1 2 3 4; do echo "== TRS_PILOT_FRAME_F=$f"; TRS_PILOT_FRAME_F=$f PL_TAG=f$f timeout -k 10 300 bash scripts/pilot_layers.sh 2>&1 | grep "conv4\|conv5\|conv6\|conv7\|all kernels\|bench" ; done | tee $O/frame_f_sweep.txt
