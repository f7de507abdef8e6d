#!/usr/bin/env python3
"""Copy what scripts/r04_final.sh measured (gpurun_out/r04f_*, gpurun_out/prof_r04f_resident) into profiles/ under the published names."""
import json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")


def cp(a, b, header=None):
    src = os.path.join(O, a)
    if not os.path.exists(src):
        print("missing:", a)
        return
    text = "".join(l for l in open(src, errors="replace") if "amdgpu.ids" not in l)
    open(os.path.join(P, b), "w").write((header or "") + text)


cp("r04f_bench_1gpu.json", "r04_bench_1gpu.json")
cp("r04f_bench_steps20.json", "r04_bench_steps20_warmup5.json")
cp("r04f_bench_spawn.json", "r04_bench_self_launch_1gpu.json")
cp("r04f_bench_2ranks_rehearsal.json", "r04_bench_2ranks_rehearsal_one_gpu.json")
cp("r04f_bench_pilot_1024x120x160.json", "r04_bench_pilot_1024x120x160.json")
cp("r04f_bench_pilot_512x240x320_depth.json", "r04_bench_pilot_512x240x320_depth.json")
cp("r04f_sweep.txt", "r04_sweep.txt", "# round 4 sweep (scripts/r04_final.sh): env-steps/s, ms per step, frac of 8 TB/s by HIP events / by wall clock, average launch us, step mode\n")
cp("r04f_pilot_layers.txt", "r04_pilot_layers.txt", "# round 4: per-kernel times of one closed-loop step under the rocprofv3 kernel tracer (scripts/pilot_layers.sh; the untraced loop is ~5-8 % faster: profiles/r04_bench_pilot_*.json)\n")
cp("r04f_pilot_pmc.txt", "r04_pilot_pmc.txt", "# round 4: hardware counters per kernel of one closed-loop step (scripts/pilot_pmc.sh: separate --pmc passes, no tracing domains)\n")
cp("r04f_pilot_precision.txt", "r04_pilot_precision.txt")
cp("r04f_image_path.txt", "r04_image_path.txt", "# round 4: image path (scripts/preprocess_bench.py); the Canny layer was not changed this round\n")
cp("r04f_fused_filter.txt", "r04_fused_filter.txt", "# round 4: cam/processed_img per step (scripts/filter_bench.py): the dynamic-brightness filter with its tables in LDS, uniform rows filtered once (26.3-26.9 -> 23.1-23.6 us resident),\n# and phase A's channel sums by a class-count table + the raw palette by channel + one v_dot4_u32_u8 per channel (-> 21.3-21.4 us; launch mode 27.9 -> 25.9),\n# then the phases' LDS reads in flight together, class bits in a shift register, DPP sums (-> 15.5 us; launch mode 22.9; profiles/r04_dyn_stamps.txt)\n")
cp("r04f_config1.txt", "r04_config1.txt")
Q = os.path.join(O, "prof_r04f_resident")
if os.path.exists(os.path.join(Q, "summary.txt")):
    shutil.copyfile(os.path.join(Q, "summary.txt"), os.path.join(P, "r04_worker_kernel_1024envs_rocprofv3_summary.txt"))
    shutil.copyfile(os.path.join(Q, "trace", "trace_kernel_stats.csv"), os.path.join(P, "r04_worker_kernel_1024envs_kernel_stats.csv"))
    shutil.copyfile(os.path.join(Q, "bench_trace.json"), os.path.join(P, "r04_worker_kernel_1024envs_bench_under_rocprofv3.json"))
    a, b = json.load(open(os.path.join(P, "pmc_traffic.json"))), json.load(open(os.path.join(O, "pmc_traffic.json")))
    a.setdefault("per_env_step", {}).update(b.get("per_env_step", {}))
    json.dump(a, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
    print("traffic:", a["per_env_step"])
for extra, name, hdr in (("r04_dense_ab.txt", "r04_pilot_dense.txt", "# round 4, dense1 at 512 x 240x320 + depth (37 us at the start of the round).  (1) more K slices = more workgroups per CU: no gain\n"),):
    if os.path.exists(os.path.join(O, extra)):
        parts = [hdr + open(os.path.join(O, extra)).read()]
        for more, h2 in (("r04_dense_ablate.txt", "# (2) timing-only builds (scripts/r04_dense_ablate.sh): dab1 = no weight stream (the first chunk's weights serve every chunk), dab2 = activations re-read from cache\n"),
                         ("r04_dense_nf.txt", "# (3) 64 frames per workgroup (every weight fragment feeds two MFMAs; the default where K is long) against 32 (trs_pilot_tuning.dense = 2)\n")):
            if os.path.exists(os.path.join(O, more)):
                parts.append(h2 + open(os.path.join(O, more)).read())
        open(os.path.join(P, name), "w").write("".join(l for l in "".join(parts).splitlines(True) if "amdgpu.ids" not in l))
if os.path.exists(os.path.join(O, "r04_dyn_ablate.txt")):
    open(os.path.join(P, "r04_fused_filter_ablation.txt"), "w").write(
        "# round 4, BEFORE the change: where the dynamic-brightness step (26.9 us against 9.4 raw, resident worker, 1024 envs) spends its time; timing-only builds of\n"
        "# raster_dyn_batch (scripts/r04_dyn_ablate.sh): 1 = phase A without its classification, 2 = phase B without the filter arithmetic, 3 = phase C without its stores\n"
        + open(os.path.join(O, "r04_dyn_ablate.txt")).read())
print("published")
