#!/bin/bash
# round 3, first GPU call: the numbers every later change is compared with, all on ONE box.
#   1. same-box store ceiling (bare kernels)                          -> gpurun_out/r03_store_ceiling.txt
#   2. env step at 1024 / 4096 / 16384 envs, resident, store policy A/B -> gpurun_out/r03_store_ab.txt
#   3. PMC passes of the 4096-env resident run                          -> gpurun_out/prof_r03_4096/
#   4. pilot loop per layer at both frame sizes                          -> gpurun_out/r03_pilot_layers.txt
#   5. image path timings + one PMC pass of the Canny layer              -> gpurun_out/r03_image_path.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "[1] store ceiling"; date
timeout -k 10 240 scripts/ab_bin/store_ceiling > gpurun_out/r03_store_ceiling.txt 2>&1 || echo "store_ceiling failed"
tail -5 gpurun_out/r03_store_ceiling.txt
echo "[2] store policy A/B, resident mode"; date
run() { timeout -k 10 120 python bench.py --no-cpu-baseline --no-also "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,2), 'us', d['roofline']['frac'], d['roofline']['avg_launch_us'])"; }
{
for round in 1 2; do
for a in 17 0 2 19; do
  lib=$PWD/scripts/ab_bin/libtrsim_aux$a.so; [ $a = 17 ] && lib=$PWD/triton-racer-sim_amd/csrc/libtrsim.so
  [ -f $lib ] || continue
  for cfg in "--envs-per-gpu 1024 --steps 2000" "--envs-per-gpu 4096 --steps 500" "--envs-per-gpu 16384 --steps 128" "--envs-per-gpu 4096 --steps 500 --step-mode launch"; do
    echo -n "aux=$a $cfg : "; TRS_HIP_LIB=$lib run $cfg
  done
done
done
} > gpurun_out/r03_store_ab.txt 2>&1
cat gpurun_out/r03_store_ab.txt
echo "[3] PMC 4096 envs"; date
bash scripts/profile.sh r03_4096 --envs-per-gpu 4096 --steps 500 --warmup 500 > gpurun_out/r03_prof4096.log 2>&1
tail -30 gpurun_out/r03_prof4096.log
echo "[4] pilot per layer"; date
{
PL_TAG=r03a bash scripts/pilot_layers.sh
PL_TAG=r03b bash scripts/pilot_layers.sh --envs-per-gpu 512 --img-h 240 --img-w 320 --depth
for i in 1 2; do
python bench.py --no-cpu-baseline --pilot --steps 100 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('untraced 1024x120x160:', d['value'], d['roofline']['frac'], d['roofline']['avg_step_us'])"
python bench.py --no-cpu-baseline --pilot --steps 60 --warmup 10 --envs-per-gpu 512 --img-h 240 --img-w 320 --depth 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('untraced 512x240x320+depth:', d['value'], d['roofline']['frac'], d['roofline']['avg_step_us'])"
done
} > gpurun_out/r03_pilot_layers.txt 2>&1
cat gpurun_out/r03_pilot_layers.txt
echo "[5] image path"; date
{
python scripts/preprocess_bench.py 1024 120 160
python scripts/preprocess_bench.py 256 240 320
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  rm -rf gpurun_out/pmc_r03_img_$name
  rocprofv3 --pmc $pass --output-format csv -d gpurun_out/pmc_r03_img_$name -o p -- python3 scripts/preprocess_bench.py 1024 120 160 > /dev/null 2> gpurun_out/pmc_r03_img_$name.err
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_r03_img_*/")):
    fs = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not fs: print("no csv in", d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if "preprocess" not in k and "normalize" not in k: continue
        print(k)
        for c, v in cs.items():
            # per-dispatch values of the LAST half of the dispatches (the timed loop of the heaviest configuration comes last per kernel)
            print("   %-28s n=%4d  mean %.4g  last %.4g" % (c, len(v), sum(v) / len(v), v[-1]))
PY
} > gpurun_out/r03_image_path.txt 2>&1
cat gpurun_out/r03_image_path.txt
date
