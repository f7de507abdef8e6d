#!/bin/bash
# round-2 wrap-up on one MI355X: whole GPU suite, default bench (+ also legs + CPU baseline), sweep, pilot benches, profiles
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r02_final
mkdir -p $O
echo "== full gpu suite" && timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; tail -4 $O/gpu_tests.log; [ $rc -eq 0 ] || exit $rc
echo "== default bench" && timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; rc=$?; python -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic']); [print(' ', k, v.get('env_steps_per_s'), v.get('frac_of_hbm_peak', v.get('frac_of_mfma_peak')), v.get('us_per_call', v.get('us_per_step')), v.get('lock_step_us_per_call')) for k,v in d['also'].items()]; print(' cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"; [ $rc -eq 0 ] || exit $rc
for cfg in "--steps 20 --warmup 5" "--step-mode launch" "--step-mode launch --steps 20 --warmup 5" "--envs-per-gpu 512" "--envs-per-gpu 512 --step-mode launch" "--envs-per-gpu 256" "--envs-per-gpu 2048 --steps 1000" "--envs-per-gpu 4096 --steps 500" "--envs-per-gpu 4096 --steps 500 --step-mode launch" "--envs-per-gpu 16384 --steps 128" "--steps 600 --depth" "--steps 600 --depth --step-mode launch" "--envs-per-gpu 512 --steps 200 --img-h 240 --img-w 320 --depth" "--envs-per-gpu 1024 --steps 200 --img-h 240 --img-w 320 --depth" "--envs-per-gpu 256 --steps 4000 --no-render" "--envs-per-gpu 256 --steps 4000 --no-render --steps-per-launch 16" "--envs-per-gpu 65536 --steps 256 --no-render --steps-per-launch 16"; do
  echo "== $cfg"; timeout -k 10 200 python bench.py --no-cpu-baseline --no-also $cfg 2>> $O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['config']['step_mode'])" || exit 1
done | tee $O/sweep.txt
echo "== pilot" && timeout -k 10 300 python bench.py --no-cpu-baseline --pilot --steps 300 --warmup 30 > $O/bench_pilot.json 2>> $O/bench.err && python -c "
import json; d=json.load(open('$O/bench_pilot.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
timeout -k 10 300 python bench.py --no-cpu-baseline --pilot --steps 200 --warmup 20 --envs-per-gpu 512 --img-h 240 --img-w 320 --depth > $O/bench_pilot5.json 2>> $O/bench.err && python -c "
import json; d=json.load(open('$O/bench_pilot5.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
PL_TAG=final timeout -k 10 300 bash scripts/pilot_layers.sh 2>&1 | tee $O/pilot_layers_120.txt
PL_TAG=final5 timeout -k 10 300 bash scripts/pilot_layers.sh --envs-per-gpu 512 --img-h 240 --img-w 320 --depth 2>&1 | tee $O/pilot_layers_240.txt
echo "== image path" && (timeout -k 10 200 python scripts/preprocess_bench.py 2>&1 | grep -v amdgpu; timeout -k 10 200 python scripts/preprocess_bench.py 256 240 320 2>&1 | grep -v amdgpu) | tee $O/image_path.txt
echo "== config 1" && timeout -k 10 200 python scripts/config1.py 2>&1 | grep -v amdgpu.ids | tee $O/config1.txt
echo "== profile (resident)" && timeout -k 10 500 bash scripts/profile.sh r02_final > $O/profile.log 2>&1; tail -3 $O/profile.log
