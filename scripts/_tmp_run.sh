cd /root/repo
timeout -k 10 600 python -m pytest tests/test_pilot.py -x -q -m gpu 2>&1 | tail -3 || exit 1
for rep in 1 2 3; do for m in 2 3 auto; do
if [ $m = auto ]; then unset TRS_PILOT_CHAIN_NT; else export TRS_PILOT_CHAIN_NT=$m; fi
python3 bench.py --no-cpu-baseline --pilot --envs-per-gpu 1024 --steps 300 --warmup 30 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chain nt=$m untraced 1024x120x160:', d['value'], d['ms_per_step'])"
done; done
unset TRS_PILOT_CHAIN_NT
PL_TAG=nta bash scripts/pilot_layers.sh 2>&1 | grep "conv4-7\|all kernels"
