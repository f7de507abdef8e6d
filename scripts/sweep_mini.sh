set -e
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline"
for cfg in "--envs-per-gpu 1024 --steps 2000" "--envs-per-gpu 1024 --steps 2000 --steps-per-launch 8" "--envs-per-gpu 512 --steps 2000" "--envs-per-gpu 256 --steps 2000" "--envs-per-gpu 4096 --steps 500" "--envs-per-gpu 16384 --steps 128" "--envs-per-gpu 1024 --steps 600 --depth" "--envs-per-gpu 1024 --steps 200 --img-h 240 --img-w 320 --depth"; do
  echo "== $cfg"; timeout -k 10 120 $B $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
done
