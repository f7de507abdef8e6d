#!/bin/bash
# one box of the pool: the closed pilot loop at both configurations + the main line (profiles/r05_pilot_boxes.txt collects several calls)
cd "$(dirname "$0")/.."
one() { python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step']*1e3,2), 'us')"; }
echo -n "main 1024 x 120x160 (resident): "; one --no-also
echo -n "pilot 1024 x 120x160: "; one --pilot --steps 200 --warmup 60
echo -n "pilot 512 x 240x320 + depth: "; one --pilot --steps 80 --warmup 30 --envs-per-gpu 512 --img-h 240 --img-w 320 --depth
