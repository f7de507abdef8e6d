#!/bin/bash
# Diagnostic: per-phase s_memtime stamps of workgroup 7 (wave 0 and wave 1) for single-step launches.
set -e
cd "$(dirname "$0")/.."
C=triton-racer-sim_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -ffp-contract=off -fno-fast-math -DTRS_STAMPS=1 -o /tmp/libtrsim_stamps.so $C/trsim_hip.hip $C/trsim_tables.cpp 2>/dev/null
TRS_HIP_LIB=/tmp/libtrsim_stamps.so python - "$@" <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from triton_racer_sim_amd.env import BatchedEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
render = (sys.argv[2] != '0') if len(sys.argv) > 2 else True
env = BatchedEnv(n_envs=n, auto_reset=True, render=render)
env.step_synthetic(50, 1)
names = ["entry", "prologue issued", "phase0 done", "barrier1 passed", "phaseA done", "barrier2 passed", "phaseA2(+map write) done", "barrier3 passed", "raster done", "barrier4 passed"]
acc = []
for _ in range(20):
    env.step_synthetic(1, 1)
    st = env.fetch("stats").astype(np.int64)
    acc.append(np.stack([st[8:18], st[24:34]]))
a = np.median(np.array(acc), axis=0)
base = a[0, 0]
print(f"n_envs={n} render={render}  (cycles from wave 0 entry; ~2.4 cycles/ns)")
for i, nm in enumerate(names):
    print(f"  {nm:28s} wave0 {a[0,i]-base:9.0f}   wave1 {a[1,i]-base:9.0f}")
PY
