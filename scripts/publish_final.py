#!/usr/bin/env python3
"""Copy what scripts/r02_final.sh measured (gpurun_out/r02_final, gpurun_out/prof_r02_final) into profiles/ under the published names."""
import json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(R, "gpurun_out", "r02_final"), os.path.join(R, "profiles")
cp = lambda a, b: shutil.copyfile(os.path.join(O, a), os.path.join(P, b))
cp("bench_default.json", "r02_bench_1gpu.json"); cp("bench_pilot.json", "r02_bench_pilot_1024x120x160.json")
cp("bench_pilot5.json", "r02_bench_pilot_512x240x320_depth.json"); cp("config1.txt", "r02_config1.txt")
hdr = open(os.path.join(P, "r02_sweep_final.txt")).readline()
open(os.path.join(P, "r02_sweep_final.txt"), "w").write(hdr + open(os.path.join(O, "sweep.txt")).read())
d, d5 = (json.load(open(os.path.join(P, f))) for f in ("r02_bench_pilot_1024x120x160.json", "r02_bench_pilot_512x240x320_depth.json"))
old = open(os.path.join(P, "r02_pilot_final.txt")).read()
keep = old[old.index("# state before this round's last changes"):] if "# state before this round's last changes" in old else ""
head = (f"# r02 final pilot loop (scripts/pilot_layers.sh: rocprofv3 kernel trace; the bench figure under the tracer is ~10 % below the untraced one: this box, untraced: "
        f"{d['value'] / 1e6:.2f} M env-steps/s = {d['roofline']['achieved']:.0f} TFLOP/s = {d['roofline']['frac']:.3f} of the bf16 peak and {d5['value'] / 1e6:.3f} M = "
        f"{d5['roofline']['achieved']:.0f} TFLOP/s = {d5['roofline']['frac']:.3f}, profiles/r02_bench_pilot_*.json; boxes of this pool differ by ~5 % in this loop: the same build "
        f"measured 4.95-5.28 M and 0.82-0.87 M)\n")
open(os.path.join(P, "r02_pilot_final.txt"), "w").write(head + open(os.path.join(O, "pilot_layers_120.txt")).read() + open(os.path.join(O, "pilot_layers_240.txt")).read() + keep)
ip = os.path.join(O, "image_path.txt")
if os.path.exists(ip):
    old_ip = open(os.path.join(P, "r02_image_path.txt")).read()
    hdr_ip = "".join(l for l in old_ip.splitlines(True) if l.startswith("#"))
    open(os.path.join(P, "r02_image_path.txt"), "w").write(hdr_ip + open(ip).read())
Q = os.path.join(R, "gpurun_out", "prof_r02_final")
shutil.copyfile(os.path.join(Q, "summary.txt"), os.path.join(P, "r02_worker_kernel_1024envs_rocprofv3_summary.txt"))
shutil.copyfile(os.path.join(Q, "trace", "trace_kernel_stats.csv"), os.path.join(P, "r02_worker_kernel_1024envs_kernel_stats.csv"))
shutil.copyfile(os.path.join(Q, "bench_trace.json"), os.path.join(P, "r02_worker_kernel_1024envs_bench_under_rocprofv3.json"))
a, b = json.load(open(os.path.join(P, "pmc_traffic.json"))), json.load(open(os.path.join(R, "gpurun_out", "pmc_traffic.json")))
a.setdefault("per_env_step", {}).update(b.get("per_env_step", {}))
json.dump(a, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
print("published:", d["value"], d5["value"], a["per_env_step"])
