#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r02_call8
mkdir -p $O
for d in 0 1 2 3 0; do
  echo "== TRS_RESIDENT_DIAG=$d"; TRS_RESIDENT_DIAG=$d timeout -k 10 200 python bench.py --no-cpu-baseline --no-also 2>> $O/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done | tee $O/diag.txt
