#!/bin/bash
# round 2, second GPU call: resident mode with counted drains + one-read dispatcher; uniform rows first in the launch kernel
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r02_call2
mkdir -p $O
echo "== resident tests" && timeout -k 10 420 python -m pytest tests/test_resident.py -x -q > $O/resident_tests.log 2>&1; rc=$?; tail -15 $O/resident_tests.log; [ $rc -eq 0 ] || { echo "resident tests rc=$rc"; exit $rc; }
echo "== bench (launch mode + also legs)" && timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_new.json 2> $O/bench_new.err; rc=$?; cat $O/bench_new.json; [ $rc -eq 0 ] || { tail -5 $O/bench_new.err; exit $rc; }
for cfg in "--resident" "--envs-per-gpu 512" "--envs-per-gpu 512 --resident" "--envs-per-gpu 256 --resident" "--envs-per-gpu 4096 --steps 500 --resident" "--envs-per-gpu 4096 --steps 500" "--steps 20 --warmup 5" "--steps 20 --warmup 5 --resident" "--steps 600 --depth --resident" "--steps 600 --depth" "--steps 200 --img-h 240 --img-w 320 --depth --resident --envs-per-gpu 512" "--steps 200 --img-h 240 --img-w 320 --depth --envs-per-gpu 512"; do
  echo "== $cfg"; timeout -k 10 200 python bench.py --no-cpu-baseline --no-also $cfg 2>> $O/bench_new.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])" || exit 1
done | tee $O/sweep.txt
echo "== gpu parity + coresidency" && timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_coresidency.py tests/test_components.py -x -q > $O/gpu_tests.log 2>&1; rc=$?; tail -5 $O/gpu_tests.log; exit $rc
