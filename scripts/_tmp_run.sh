cd /root/repo
timeout -k 10 500 python -m pytest tests/test_pilot.py -x -q -m gpu 2>&1 | tail -5 || exit 1
for rep in 1 2; do for m in 0 1; do
TRS_PILOT_FRAME5=$m python3 bench.py --no-cpu-baseline --pilot --envs-per-gpu 1024 --steps 300 --warmup 30 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('frame5 $m untraced 1024x120x160:', d['value'], d['ms_per_step'])"
done; done
PL_TAG=f5 bash scripts/pilot_layers.sh 2>&1
