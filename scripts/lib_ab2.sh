#!/bin/bash
# in-tree build against scripts/ab_bin/libtrsim_<tag>.so on the pilot loop and the main line, alternating
cd "$(dirname "$0")/.."
pl() { python bench.py --no-cpu-baseline --pilot "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,3), 'M', d['roofline']['avg_step_us'], 'us', d['roofline']['frac'])"; }
for round in 1 2; do
for tag in tree "$@"; do
  lib=$PWD/scripts/ab_bin/libtrsim_$tag.so; [ $tag = tree ] && lib=$PWD/triton-racer-sim_amd/csrc/libtrsim.so
  [ -f $lib ] || continue
  echo -n "$tag pilot 1024x120x160: "; TRS_HIP_LIB=$lib pl --steps 150 --warmup 60
  echo -n "$tag pilot 512x240x320+d: "; TRS_HIP_LIB=$lib pl --steps 60 --warmup 30 --envs-per-gpu 512 --img-h 240 --img-w 320 --depth
done
done
