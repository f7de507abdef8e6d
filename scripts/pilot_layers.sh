#!/bin/bash
# per-layer kernel times of the pilot loop (rocprofv3 kernel trace); extra args go to bench.py; TRS_HIP_LIB picks another build of the library, PL_TAG names the output
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
tag=${PL_TAG:-run}
rm -rf gpurun_out/prof_pl_$tag
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_pl_$tag -o p -- python3 bench.py --no-cpu-baseline --pilot --envs-per-gpu 1024 --steps 30 --warmup 5 "$@" > gpurun_out/prof_pl_$tag.json 2>/dev/null
echo "== $tag $*"
python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/prof_pl_$tag/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "trs_conv" in r["Kernel_Name"] or "tail" in r["Kernel_Name"] or "trs_step" in r["Kernel_Name"] or "trs_pilot_dense" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# a step = the pilot's launches between two env steps
seq=[]; cur=[]
for r in rows:
    if "trs_step" in r["Kernel_Name"] and cur:
        seq.append(cur); cur=[]
    cur.append(r)
seq=[s for s in seq if len(s)==len(seq[-1])]
layouts={10:["env step","conv1","conv2","conv3","conv4","conv5","conv6","conv7","dense1","tail"],
         9:["env step","conv1+2","conv3","conv4","conv5","conv6","conv7","dense1","tail"],
         8:["env step","conv1+2","conv3","conv4","conv5","conv6","conv7","dense+tail"],
         7:["env step","conv1+2","conv3","conv4","conv5-7","dense1","tail"],
         6:["env step","conv1+2","conv3","conv4-7","dense1","tail"]}
names=fused=layouts.get(len(seq[-1]),[str(i) for i in range(len(seq[-1]))])
tot=0
for j in range(len(seq[-1])):
    d=[int(s[j]["End_Timestamp"])-int(s[j]["Start_Timestamp"]) for s in seq]
    r=seq[-1][j]
    kn=r["Kernel_Name"]
    kind="fused" if "conv12" in kn else "chain" if "conv_chain" in kn else "frame5" if "frame5" in kn else "dense" if "pilot_dense" in kn else "frame" if "conv_frame" in kn else "span" if "span" in kn else "lt" if "conv_lt" in kn else "u8" if "conv_u8" in kn else "chunked" if "conv_mfma" in kn else "-"
    nm=names[j] if len(seq[-1])==len(names) else (fused[j] if len(seq[-1])==len(fused) else str(j))
    tot+=sum(d)/len(d)
    print(f"  {nm:9s} {kind:8s} grid={r['Grid_Size_X']:>8s}x{r['Grid_Size_Y']} wg={r['Workgroup_Size_X']:>4s} lds={r.get('LDS_Block_Size','?'):>7s}  mean {sum(d)/len(d)/1e3:8.1f} us")
print(f"  all kernels per step: {tot/1e3:.1f} us   (steps seen: {len(seq)})")
PY
cat gpurun_out/prof_pl_$tag.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  bench:', d['value'], d['unit'], d['roofline']['achieved'], d['roofline']['unit'])"
