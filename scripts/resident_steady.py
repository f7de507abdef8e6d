#!/usr/bin/env python3
"""Resident worker: steady-state wall clock (completion flags, worker stays) against one event-bracketed worker launch, per chunk."""
import sys, time
sys.path.insert(0, ".")
from triton_racer_sim_amd.env import BatchedEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
env = BatchedEnv(n_envs=n, auto_reset=True)
env.set_step_mode(True)
env.step_synthetic(200, 1); env.sync()
B = 57688
for rep in range(6):
    t0 = time.perf_counter(); env.step_synthetic(K, 1); t1 = time.perf_counter(); env.sync(); t2 = time.perf_counter()
    print(f"steady wall chunk {rep}: {1e6 * (t2 - t0) / K:7.3f} us/step  (posting returned after {1e6 * (t1 - t0) / K:7.3f})  frac {B * n * K / (t2 - t0) / 8e12:.4f}")
for rep in range(3):
    env.event_record(0); t0 = time.perf_counter(); env.step_synthetic(K, 1); env.event_record(1); ms = env.event_elapsed_ms(0, 1); t2 = time.perf_counter()
    print(f"one launch by events {rep}: {1e3 * ms / K:7.3f} us/step   wall incl. launch + exit {1e6 * (t2 - t0) / K:7.3f}  frac {B * n * K / (ms * 1e-3) / 8e12:.4f}")
for rep in range(3):
    t0 = time.perf_counter(); env.step_synthetic(K, 1); env.sync(); t2 = time.perf_counter()
    print(f"steady wall again {rep}: {1e6 * (t2 - t0) / K:7.3f} us/step  frac {B * n * K / (t2 - t0) / 8e12:.4f}")
