#!/bin/bash
# hardware counters per pilot layer (separate --pmc passes, no tracing domains); PL_TAG names the output
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
tag=${PL_TAG:-pmc}
EXTRA=("$@")
mkdir -p gpurun_out
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
pass() {  # name, counters...
  local name=$1; shift
  rm -rf gpurun_out/pmc_${tag}_$name
  rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}_$name -o p -- python3 bench.py --no-cpu-baseline --pilot --envs-per-gpu ${PL_ENVS:-1024} --steps 6 --warmup 2 "${EXTRA[@]}" > /dev/null 2>gpurun_out/pmc_${tag}_$name.err
}
pass sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU &&
{ [ -n "$PL_SQ_ONLY" ] || pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE; } &&
python3 - <<PY
import csv,glob,collections
def label(k):
    for key,nm in (("trs_step","env step"),("conv12","conv1+2"),("frame5","conv3 (frame5)"),("span","conv3 (span)"),("conv_chain","conv4-7 chain"),
                   ("conv_frame","3x3 frame"),("pilot_dense","dense1"),("conv_mfma","dense1 (chunked)"),("tail","tail"),("conv_u8","conv1"),("conv_lt","conv (lt)")):
        if key in k: return nm
    return None
for name in ("sq","tcp"):
    fs=glob.glob("gpurun_out/pmc_${tag}_%s/**/*counter_collection.csv"%name,recursive=True)
    if not fs: print("no csv for",name); continue
    rows=[r for r in csv.DictReader(open(fs[0])) if label(r["Kernel_Name"])]
    disp=collections.OrderedDict()
    for r in rows:
        d=disp.setdefault(int(r["Dispatch_Id"]),{"k":r["Kernel_Name"]})
        d[r["Counter_Name"]]=d.get(r["Counter_Name"],0)+float(r["Counter_Value"])
    ds=[disp[k] for k in sorted(disp)]
    # the last complete step = the dispatches behind the last env step but one
    idx=[i for i,d in enumerate(ds) if "trs_step" in d["k"]]
    if len(idx)<2: print("no complete step for",name); continue
    last=ds[idx[-2]:idx[-1]]
    ctrs=[c for c in last[0] if c!="k"]
    print("== %s (one closed-loop step)"%name)
    print("  %-16s "%"kernel"+" ".join("%22s"%c[-22:] for c in ctrs))
    for d in last:
        print("  %-16s "%label(d["k"])+" ".join("%22.4g"%d.get(c,0) for c in ctrs))
PY
