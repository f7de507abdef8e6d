#!/usr/bin/env python3
"""Print the GPU timeline of the last few physics / raster dispatches from a rocprofv3 kernel trace csv."""
import csv, sys, glob, os
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "trs_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-40:-8]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = "P" if "physics" in r["Kernel_Name"] else "R"
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{name} q={r.get('Queue_Id','?'):>3} start={s/1e3:9.2f}us end={e/1e3:9.2f}us dur={(e-s)/1e3:7.2f}us")
