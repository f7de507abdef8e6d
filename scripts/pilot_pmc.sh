#!/bin/bash
# hardware counters per pilot layer (separate --pmc passes, no tracing domains); PL_TAG names the output
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
tag=${PL_TAG:-pmc}
mkdir -p gpurun_out
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
pass() {  # name, counters...
  local name=$1; shift
  rm -rf gpurun_out/pmc_${tag}_$name
  rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}_$name -o p -- python3 bench.py --no-cpu-baseline --pilot --envs-per-gpu 1024 --steps 6 --warmup 2 > /dev/null 2>gpurun_out/pmc_${tag}_$name.err
}
pass sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU &&
pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE &&
python3 - <<PY
import csv,glob,collections
names=["env step","conv1","conv2","conv3","conv4","conv5","conv6","conv7","dense1","tail"]
for name in ("sq","tcp"):
    fs=glob.glob("gpurun_out/pmc_${tag}_%s/**/*counter_collection.csv"%name,recursive=True)
    if not fs: print("no csv for",name); continue
    rows=list(csv.DictReader(open(fs[0])))
    rows=[r for r in rows if "trs_conv" in r["Kernel_Name"] or "tail" in r["Kernel_Name"] or "trs_step" in r["Kernel_Name"]]
    # group by dispatch id -> counters
    disp=collections.OrderedDict()
    for r in rows:
        d=disp.setdefault(int(r["Dispatch_Id"]),{"k":r["Kernel_Name"]})
        d[r["Counter_Name"]]=d.get(r["Counter_Name"],0)+float(r["Counter_Value"])
    ds=[disp[k] for k in sorted(disp)]
    # split into steps at 'tail'
    steps=[];cur=[]
    for d in ds:
        cur.append(d)
        if "tail" in d["k"]: steps.append(cur);cur=[]
    fused=["env step","conv1+2","conv3","conv4","conv5","conv6","conv7","dense1","tail"]
    steps=[s for s in steps if len(s) in (len(names),len(fused))]
    if not steps: print("no complete step for",name); continue
    last=steps[-1]
    if len(last)==len(fused): names=fused
    ctrs=[c for c in last[1] if c!="k"]
    print("== %s (last step)"%name)
    print("  %-9s "%"layer"+" ".join("%22s"%c[-22:] for c in ctrs))
    for nm,d in zip(names,last):
        print("  %-9s "%nm+" ".join("%22.4g"%d.get(c,0) for c in ctrs))
PY
