#!/bin/bash
# per-layer conv kernel times: persistent LDS-band kernels on (default) vs off (TRS_PILOT_NO_PERSIST=1)
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
for mode in persist generic; do
  rm -rf gpurun_out/prof_pl_$mode
  if [ $mode = generic ]; then export TRS_PILOT_NO_PERSIST=1; else unset TRS_PILOT_NO_PERSIST; fi
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_pl_$mode -o p -- python3 bench.py --no-cpu-baseline --pilot --envs-per-gpu 1024 --steps 30 --warmup 5 "$@" > /dev/null 2>&1
  echo "== $mode"
  python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/prof_pl_$mode/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "conv_" in r["Kernel_Name"] or "tail" in r["Kernel_Name"] or "trs_step" in r["Kernel_Name"]]
by=collections.OrderedDict()
for r in rows:
    k=(("persist" if "persist" in r["Kernel_Name"] else "generic" if "conv_mfma" in r["Kernel_Name"] else "other"), r["Grid_Size_X"], r["Grid_Size_Y"])
    by.setdefault(k,[]).append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
steps=len([r for r in rows if "tail" in r["Kernel_Name"]])
tot=0
for k,v in by.items():
    tot+=sum(v)
    print(f"  {k[0]:8s} grid={k[1]:>8s}x{k[2]}  launches/step={len(v)/steps:4.1f}  mean {sum(v)/len(v)/1e3:8.1f} us  per step {sum(v)/steps/1e3:8.1f} us")
print(f"  all kernels per step: {tot/steps/1e3:.1f} us")
PY
done
