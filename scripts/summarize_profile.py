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
    step = [r for r in rows if "trs_step" in r["Kernel_Name"] or "trs_worker_kernel" in r["Kernel_Name"]]
    for r in rows:
        if "trs_worker_kernel" in r["Kernel_Name"]:
            print(f"   dispatch trs_worker_kernel: {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.1f} us")
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
        if "trs_step" not in k and "trs_worker_kernel" not in k: continue
        print("== pmc", os.path.basename(os.path.dirname(os.path.dirname(f))), k[:50])
        for c, v in d.items():
            print(f"   {c}: mean/dispatch={sum(v)/len(v):.1f} n={len(v)}")

# HBM traffic for bench.py's roofline.traffic: WRITE_SIZE + 2 x FETCH_SIZE (KiB; gfx950 FETCH_SIZE reads 1/2 of a wide
# coalesced stream, /opt/skills/guides/MI355X_MICROARCH.md section HBM), mean over the dispatches of the timed kernel, divided by
# the env-steps one launch completes (the bench line of the traced run says which kernel, how many, and names the workload)
import json
line = None
try:
    with open(os.path.join(out, "bench_trace.json")) as fh:
        line = [json.loads(l) for l in fh if l.strip().startswith("{")][-1]
except Exception as exc:
    print("   (no bench line in bench_trace.json:", exc, ")")
kern = line["roofline"]["kernel"].split()[0] if line else "trs_step"
vals = {}
for name in ("WRITE_SIZE", "FETCH_SIZE"):
    for f in glob.glob(os.path.join(out, f"pmc_{name}", "**", "*counter_collection.csv"), recursive=True):
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"] and r["Counter_Name"] == name]
        if line and line["roofline"]["launches"] == 1 and len(v) > 1:
            v = v[-1:]                                       # resident mode: the LAST worker dispatch is the timed region's
        if v: vals[name] = sum(v) / len(v)
if len(vals) == 2 and line:
    traffic = (vals["WRITE_SIZE"] + 2 * vals["FETCH_SIZE"]) * 1024
    per_step = traffic / line["roofline"]["env_steps_per_launch"]
    key = line["roofline"].get("traffic_key")
    print(f"== traffic per launch of {kern}: {traffic:.0f} B = {per_step:.1f} B per env-step (algorithmic {line['roofline']['bytes_per_env_step']}); WRITE_SIZE {vals['WRITE_SIZE']:.1f} KiB, FETCH_SIZE {vals['FETCH_SIZE']:.1f} KiB x2; key {key}")
    json.dump({"traffic_bytes_per_launch": traffic, "bytes_per_env_step": per_step, "key": key, **vals}, open(os.path.join(out, "traffic.json"), "w"))
    if key:   # profiles/pmc_traffic.json is what bench.py reads for roofline.traffic
        repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        dst = os.path.join(repo, "gpurun_out", "pmc_traffic.json")
        cur = {"per_env_step": {}, "source": {}}
        for cand in (dst, os.path.join(repo, "profiles", "pmc_traffic.json")):
            if os.path.exists(cand):
                cur = json.load(open(cand)); break
        cur.setdefault("per_env_step", {})[key] = per_step
        cur.setdefault("source", {})[key] = f"{os.path.basename(out)}: (WRITE_SIZE + 2 x FETCH_SIZE) KiB -> bytes per launch of {kern} / {line['roofline']['env_steps_per_launch']} env-steps per launch"
        json.dump(cur, open(dst, "w"), indent=1)
        print("== merged into gpurun_out/pmc_traffic.json under", key, "(copy to profiles/ to publish)")
