#!/bin/bash
# round 3: the measurements published under profiles/r03_* (one MI355X, one gpurun call; copy the outputs from gpurun_out/r03f_* afterwards)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
echo "[bench]"; date
python bench.py > $O/r03f_bench_1gpu.json 2> $O/r03f_bench_1gpu.err
python bench.py --steps 20 --warmup 5 > $O/r03f_bench_steps20.json 2>/dev/null
python bench.py --pilot --steps 200 --warmup 60 --no-cpu-baseline > $O/r03f_bench_pilot_1024x120x160.json 2>/dev/null
python bench.py --pilot --steps 80 --warmup 30 --no-cpu-baseline --envs-per-gpu 512 --img-h 240 --img-w 320 --depth > $O/r03f_bench_pilot_512x240x320_depth.json 2>/dev/null
echo "[profile of the bench command]"; date
bash scripts/profile.sh r03f_resident > $O/r03f_profile_resident.log 2>&1
bash scripts/profile.sh r03f_4096 --envs-per-gpu 4096 --steps 500 --warmup 500 > $O/r03f_profile_4096.log 2>&1
echo "[sweep]"; date
{
B="python bench.py --no-cpu-baseline --no-also"
for cfg in "--steps 20 --warmup 5" "" "--step-mode launch" "--envs-per-gpu 512" "--envs-per-gpu 512 --step-mode launch" "--envs-per-gpu 256" "--envs-per-gpu 2048 --steps 1000" "--envs-per-gpu 4096 --steps 500" "--envs-per-gpu 4096 --steps 500 --step-mode launch" "--envs-per-gpu 16384 --steps 128" "--steps 600 --depth" "--envs-per-gpu 512 --steps 200 --img-h 240 --img-w 320 --depth" "--envs-per-gpu 1024 --steps 200 --img-h 240 --img-w 320 --depth" "--steps-per-launch 8 --step-mode launch" "--envs-per-gpu 256 --steps 4000 --no-render" "--envs-per-gpu 256 --steps 4000 --no-render --steps-per-launch 16" "--envs-per-gpu 65536 --steps 256 --no-render --steps-per-launch 16"; do
  echo "== $cfg"; timeout -k 10 120 $B $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], 'frac(events)', r['frac'], 'frac(wall)', r['frac_by_wall_clock'], 'launch_us', r['avg_launch_us'], d['config']['step_mode'])"
done
} > $O/r03f_sweep.txt 2>&1
echo "[steady state]"; date
{ python scripts/resident_steady.py 1024 2000; python scripts/resident_steady.py 512 2000; python scripts/resident_steady.py 4096 500; } > $O/r03f_steady_state.txt 2>&1
echo "[pilot]"; date
{
PL_TAG=r03fa bash scripts/pilot_layers.sh
PL_TAG=r03fb bash scripts/pilot_layers.sh --envs-per-gpu 512 --img-h 240 --img-w 320 --depth
} > $O/r03f_pilot_layers.txt 2>&1
{ PL_TAG=r03fa bash scripts/pilot_pmc.sh; PL_TAG=r03fb PL_ENVS=512 bash scripts/pilot_pmc.sh --img-h 240 --img-w 320 --depth; } > $O/r03f_pilot_pmc.txt 2>&1
python scripts/pilot_precision.py > $O/r03f_pilot_precision.txt 2>&1
{ STAMP_STEPS=40 TRS_HIP_LIB=$PWD/scripts/ab_bin/libtrsim_bstamps.so python scripts/band_stamps.py 1024 120 160 | tail -17; STAMP_STEPS=40 TRS_HIP_LIB=$PWD/scripts/ab_bin/libtrsim_bstamps.so python scripts/band_stamps.py 512 240 320 | tail -17; STAMP_STEPS=40 TRS_HIP_LIB=$PWD/scripts/ab_bin/libtrsim_cstamps.so python scripts/band_stamps.py 1024 120 160 | tail -10; } > $O/r03f_band_stamps.txt 2>&1
echo "[image path]"; date
{
python scripts/preprocess_bench.py 1024 120 160
python scripts/preprocess_bench.py 256 240 320
TRS_HIP_LIB=$PWD/scripts/ab_bin/libtrsim_stamps.so python scripts/edge_stamps.py 1024 120 160 | tail -1
TRS_HIP_LIB=$PWD/scripts/ab_bin/libtrsim_stamps.so python scripts/edge_stamps.py 256 240 320 | tail -1
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  rm -rf $O/pmc_r03f_img_$name
  rocprofv3 --pmc $pass --output-format csv -d $O/pmc_r03f_img_$name -o p -- python3 scripts/preprocess_bench.py 1024 120 160 > /dev/null 2> $O/pmc_r03f_img_$name.err
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_r03f_img_*/")):
    fs = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not fs: print("no csv in", d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        k = "edge" if "preprocess_edge" in k else "preprocess" if "trs_preprocess_kernel" in k else "normalize" if "normalize" in k else None
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print("counters per dispatch (1024 frames of 120x160), kernel:", k, "(last dispatch of the script's heaviest configuration of that kernel)")
        for c, v in cs.items():
            print("   %-28s n=%4d  last %.4g" % (c, len(v), v[-1]))
PY
} > $O/r03f_image_path.txt 2>&1
echo "[frame filter]"; date
{ python scripts/filter_bench.py; python scripts/filter_bench.py resident; } > $O/r03f_fused_filter.txt 2>&1
echo "[config 1]"; date
python scripts/config1.py > $O/r03f_config1.txt 2>&1
date
