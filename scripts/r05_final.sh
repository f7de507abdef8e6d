#!/bin/bash
# round 5: the measurements published under profiles/r05_* (one MI355X, one gpurun call; scripts/publish_r05.py copies the outputs from gpurun_out/r05f_*)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
echo "[bench]"; date
python bench.py > $O/r05f_bench_1gpu.json 2> $O/r05f_bench_1gpu.err
python bench.py --steps 20 --warmup 5 > $O/r05f_bench_steps20.json 2>/dev/null
python bench.py --gpus 1 --spawn --steps 20 --warmup 5 --no-also --no-cpu-baseline > $O/r05f_bench_spawn.json 2>/dev/null
python bench.py --gpus 2 --share-gpu --steps 20 --warmup 5 > $O/r05f_bench_2ranks_rehearsal.json 2>/dev/null
python bench.py --pilot --steps 200 --warmup 60 --no-cpu-baseline > $O/r05f_bench_pilot_1024x120x160.json 2>/dev/null
python bench.py --pilot --steps 80 --warmup 30 --no-cpu-baseline --envs-per-gpu 512 --img-h 240 --img-w 320 --depth > $O/r05f_bench_pilot_512x240x320_depth.json 2>/dev/null
echo "[profile of the bench command]"; date
bash scripts/profile.sh r05f_resident > $O/r05f_profile_resident.log 2>&1
echo "[sweep]"; date
{
B="python bench.py --no-cpu-baseline --no-also"
for cfg in "--steps 20 --warmup 5" "" "--step-mode launch" "--envs-per-gpu 512" "--envs-per-gpu 256" "--envs-per-gpu 2048 --steps 1000" "--envs-per-gpu 4096 --steps 500" "--steps 600 --depth" "--envs-per-gpu 512 --steps 200 --img-h 240 --img-w 320 --depth" "--steps-per-launch 8 --step-mode launch" "--envs-per-gpu 256 --steps 4000 --no-render" "--envs-per-gpu 256 --steps 4000 --no-render --step-mode launch" "--envs-per-gpu 256 --steps 4000 --no-render --step-mode launch --steps-per-launch 16" "--envs-per-gpu 65536 --steps 256 --no-render --step-mode launch --steps-per-launch 16"; do
  echo "== $cfg"; timeout -k 10 120 $B $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], 'frac(events)', r['frac'], 'frac(wall)', r['frac_by_wall_clock'], 'launch_us', r['avg_launch_us'], d['config']['step_mode'])"
done
} > $O/r05f_sweep.txt 2>&1
echo "[pilot]"; date
{
PL_TAG=r05fa bash scripts/pilot_layers.sh
PL_TAG=r05fb bash scripts/pilot_layers.sh --envs-per-gpu 512 --img-h 240 --img-w 320 --depth
} > $O/r05f_pilot_layers.txt 2>&1
{ PL_TAG=r05fa bash scripts/pilot_pmc.sh; PL_TAG=r05fb PL_ENVS=512 bash scripts/pilot_pmc.sh --img-h 240 --img-w 320 --depth; } > $O/r05f_pilot_pmc.txt 2>&1
python scripts/pilot_precision.py > $O/r05f_pilot_precision.txt 2>&1
echo "[arbitration]"; date
python -m pytest tests/test_resident_arbitration.py -q -m gpu -s 2>&1 | grep -v amdgpu.ids > $O/r05f_resident_arbitration.txt
date
