#!/bin/bash
# per-layer conv kernel times with the LDS input band on (mask 127) and off (mask 0)
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
for mask in 127 0; do
  rm -rf gpurun_out/prof_pl_$mask
  TRS_PILOT_LDSA_MASK=$mask rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_pl_$mask -o p -- python3 bench.py --no-cpu-baseline --pilot --envs-per-gpu 1024 --steps 30 --warmup 5 "$@" > /dev/null 2>&1
  echo "== LDSA mask $mask"
  python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/prof_pl_$mask/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "conv_mfma" in r["Kernel_Name"] or "tail" in r["Kernel_Name"] or "trs_step" in r["Kernel_Name"]]
by=collections.OrderedDict()
for r in rows:
    k=(r["Kernel_Name"].split("(")[0][-34:], r["Grid_Size_X"], r["Grid_Size_Y"], r["LDS_Block_Size"])
    by.setdefault(k,[]).append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
tot=0
for k,v in by.items():
    m=sum(v)/len(v)/1e3; tot+=m
    print(f"  {k[0]:36s} grid={k[1]:>8s}x{k[2]} lds={k[3]:>7s}  {m:8.1f} us")
print(f"  sum {tot:.1f} us")
PY
done
