#!/bin/bash
# Diagnostic: s_memtime stamps of workgroup 7 (wave 0 = raster team, wave 10 = first physics wave) in pipelined launches.
set -e
cd "$(dirname "$0")/.."
C=triton-racer-sim_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -ffp-contract=off -fno-fast-math -DTRS_STAMPS=1 -o /tmp/libtrsim_stamps.so $C/trsim_hip.hip $C/trsim_pilot.hip $C/trsim_tables.cpp 2>/dev/null
TRS_HIP_LIB=/tmp/libtrsim_stamps.so python - "$@" <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from triton_racer_sim_amd.env import BatchedEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
env = BatchedEnv(n_envs=n, auto_reset=True)
import os
if os.environ.get("STAMPS_DYN"):
    env.set_frame_filter({"preprocessing_dynamic_brightness_enabled": True, "preprocessing_color_filter_enabled": True, "preprocessing_contrast_enhancement_ratio": 1.2})
env.step_synthetic(50, 1)
names = ["entry", "DMA + poses requested", "barrier passed (all staged)", "physics done (phys wave)", "(unused)", "raster done", "dyn: rows classified + sums (A)", "dyn: barrier A passed", "dyn: palettes built + barrier B"]
acc = []
for _ in range(10):
    env.step_synthetic(6, 1)          # pipelined: last full launch before the raster-only flush is what remains in the slots
    st = env.fetch("stats").astype(np.int64)
    acc.append(np.stack([st[8:17], st[32:41]]))
a = np.median(np.array(acc), axis=0)
base = a[0, 0]
print(f"n_envs={n} (ticks from wave 0 entry; NOTE the final launch of a call is raster-only, so physics slots are from the launch before)")
for i, nm in enumerate(names):
    print(f"  {nm:28s} raster wave0 {a[0,i]-base:9.0f}   physics wave10 {a[1,i]-base:9.0f}")
PY
