#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r02_call7
mkdir -p $O
echo "== resident + parity tests" && timeout -k 10 600 python -m pytest tests/test_resident.py tests/test_gpu_parity.py tests/test_coresidency.py -x -q > $O/tests.log 2>&1; rc=$?; tail -5 $O/tests.log; [ $rc -eq 0 ] || exit $rc
for cfg in "" "--step-mode launch" "--envs-per-gpu 512" "--envs-per-gpu 512 --step-mode launch" "--envs-per-gpu 256" "--envs-per-gpu 2048 --steps 1000" "--envs-per-gpu 4096 --steps 500" "--steps 20 --warmup 5" "--steps 600 --depth" "" "--step-mode launch"; do
  echo "== $cfg"; timeout -k 10 200 python bench.py --no-cpu-baseline --no-also $cfg 2>> $O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])" || exit 1
done | tee $O/sweep.txt
echo "== also legs" && timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_also.json 2>> $O/bench.err && python -c "
import json; d=json.load(open('$O/bench_also.json')); print(d['value']); [print(k, v['env_steps_per_s'], v['frac_of_hbm_peak'], v.get('us_per_call', v.get('us_per_step')), v.get('lock_step_us_per_call')) for k,v in d['also'].items()]"
