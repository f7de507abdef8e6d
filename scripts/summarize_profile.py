#!/usr/bin/env python3
"""Summarise a scripts/profile.sh output directory: per-kernel stats + mean PMC values per dispatch."""
import csv, glob, os, sys, collections
out = sys.argv[1]
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", os.path.relpath(f, out))
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            if i < 8: print("  ", ",".join(row[:8]))
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    by = collections.defaultdict(list)
    for r in rows:
        by[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in by.items():
        v.sort()
        print(f"== trace {k[:60]}: n={len(v)} mean={sum(v)/len(v)/1e3:.3f}us median={v[len(v)//2]/1e3:.3f}us min={v[0]/1e3:.3f}us")
    step = [r for r in rows if "trs_step" in r["Kernel_Name"]]
    if step:
        r = step[-1]
        print("   regs:", {k: r[k] for k in r if k in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Workgroup_Size", "Grid_Size", "Accum_VGPR_Count")})
        ts = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
        gaps = [ts[i + 1][0] - ts[i][1] for i in range(len(ts) - 1)]
        gaps.sort()
        print(f"   inter-kernel gap: median={gaps[len(gaps)//2]/1e3:.3f}us mean={sum(gaps)/len(gaps)/1e3:.3f}us")
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if "trs_step" not in k: continue
        print("== pmc", os.path.basename(os.path.dirname(os.path.dirname(f))), k[:50])
        for c, v in d.items():
            print(f"   {c}: mean/dispatch={sum(v)/len(v):.1f} n={len(v)}")

# HBM traffic per launch for bench.py's roofline.traffic: WRITE_SIZE + 2 x FETCH_SIZE (KiB; gfx950 FETCH_SIZE reads 1/2
# of a wide coalesced stream, /opt/skills/guides/MI355X_MICROARCH.md section HBM), mean over the step-kernel dispatches
import json
vals = {}
for name in ("WRITE_SIZE", "FETCH_SIZE"):
    for f in glob.glob(os.path.join(out, f"pmc_{name}", "**", "*counter_collection.csv"), recursive=True):
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "trs_step" in r["Kernel_Name"] and r["Counter_Name"] == name]
        if v: vals[name] = sum(v) / len(v)
if len(vals) == 2:
    traffic = (vals["WRITE_SIZE"] + 2 * vals["FETCH_SIZE"]) * 1024
    print(f"== traffic per launch: {traffic:.0f} B (WRITE_SIZE {vals['WRITE_SIZE']:.1f} KiB, FETCH_SIZE {vals['FETCH_SIZE']:.1f} KiB x2)")
    key = None
    try:   # the bench line of the traced run names the workload the counters belong to
        with open(os.path.join(out, "bench_trace.json")) as fh:
            cfg = [json.loads(l) for l in fh if l.strip().startswith("{")][-1]["config"]
        key = f"{cfg['envs_per_gpu']}x{cfg['img_h']}x{cfg['img_w']}x{cfg['steps_per_launch']}" + ("+depth" if cfg.get("depth") else "")
    except Exception as exc:
        print("   (no bench line to key the traffic figure by:", exc, ")")
    json.dump({"traffic_bytes_per_launch": traffic, "key": key, **vals}, open(os.path.join(out, "traffic.json"), "w"))
    # profiles/pmc_traffic.json is what bench.py reads for roofline.traffic: {"per_launch": {key: bytes}}
    if key:
        repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        dst = os.path.join(repo, "gpurun_out", "pmc_traffic.json")
        cur = {"per_launch": {}}
        for cand in (dst, os.path.join(repo, "profiles", "pmc_traffic.json")):
            if os.path.exists(cand):
                cur = json.load(open(cand)); break
        cur.setdefault("per_launch", {})[key] = traffic
        json.dump(cur, open(dst, "w"), indent=1)
        print("== merged into gpurun_out/pmc_traffic.json under", key, "(copy to profiles/ to publish)")
