set -e
B="python bench.py --no-cpu-baseline --no-render"
for cfg in "--envs-per-gpu 256 --steps 4000" "--envs-per-gpu 256 --steps 4000 --steps-per-launch 16" "--envs-per-gpu 256 --steps 10000 --steps-per-launch 100" "--envs-per-gpu 65536 --steps 256 --steps-per-launch 16"; do
  echo "== $cfg"; timeout -k 10 120 $B $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
