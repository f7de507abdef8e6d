#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r02_call6
mkdir -p $O
echo "== new config/comm/stream tests + resident" && timeout -k 10 600 python -m pytest tests/test_configs_gpu.py tests/test_resident.py -x -q > $O/tests.log 2>&1; rc=$?; tail -15 $O/tests.log; [ $rc -eq 0 ] || { echo "rc=$rc"; exit $rc; }
for cfg in "--resident" "--envs-per-gpu 512 --resident" "--envs-per-gpu 256 --resident" "--steps 20 --warmup 5 --resident" "--steps 600 --depth --resident"; do
  echo "== $cfg"; timeout -k 10 200 python bench.py --no-cpu-baseline --no-also $cfg 2>> $O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])" || exit 1
done | tee $O/sweep.txt
echo "== also legs" && timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_also.json 2>> $O/bench.err && python -c "
import json; d=json.load(open('$O/bench_also.json')); print(d['value']); [print(k, v['env_steps_per_s'], v['frac_of_hbm_peak'], v.get('us_per_call'), v.get('lock_step_us_per_call')) for k,v in d['also'].items()]"
echo "== full gpu suite" && timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; tail -6 $O/gpu_tests.log; exit $rc
