#!/bin/bash
# round 4: where the dynamic-brightness step spends its time inside raster_dyn_batch - a diagnostic build (-DTRS_DYN_STAMPS, built into scripts/ab_bin/libtrsim_dynst.so
# by the caller) accumulates s_memtime ticks (10 ns) of workgroup 7's first raster thread per phase; resident worker, 1024 envs, every step posted on its own
cd "$(dirname "$0")/.."
TRS_HIP_LIB=$PWD/scripts/ab_bin/libtrsim_dynst.so python - <<'PY'
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from triton_racer_sim_amd.env import BatchedEnv
env = BatchedEnv(n_envs=1024, auto_reset=True)
env.set_step_mode(True)
env.step_synthetic(6000, 1)
env.set_frame_filter({"preprocessing_color_filter_enabled": True, "preprocessing_contrast_enhancement_ratio": 1.2, "preprocessing_dynamic_brightness_enabled": True})
steps_fn = lambda k: env.step_synthetic(k, 1)
steps_fn(200); env.sync()
s0 = env.fetch("stats").copy()
t0 = time.perf_counter(); steps_fn(2000); env.sync(); dt = time.perf_counter() - t0
s1 = env.fetch("stats")
d = (s1 - s0).astype(np.float64)
nb = d[52]
names = ["A (classify + sums)", "barrier 1", "B (palettes)", "barrier 2", "C (shade + store)"]
print(f"{dt / 2000 * 1e6:.2f} us per step by wall clock; batches of workgroup 7 seen: {int(nb)}")
tot = 0.0
for k, nm in enumerate(names):
    us = d[46 + k] / nb / 2100.0   # s_memtime counts shader clocks here (~2.1 GHz under this load), not 100 MHz
    tot += us
    print(f"  {nm:24s} {us:6.2f} us per batch")
print(f"  inside raster_dyn_batch  {tot:6.2f} us per batch (one batch of 4 envs per step and workgroup at 1024 envs)")
PY
