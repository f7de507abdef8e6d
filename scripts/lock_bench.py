#!/usr/bin/env python3
"""Lock step (post the controls, wait for the frame, post again: trs_step_wait) against queued posts, resident worker, host wall clock per tick."""
import sys, time
sys.path.insert(0, ".")
import torch
from triton_racer_sim_amd.env import BatchedEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
env = BatchedEnv(n_envs=n, auto_reset=True)
env.set_step_mode(True)
st_ = torch.zeros(n, device="cuda"); th_ = torch.full((n,), 0.5, device="cuda")
torch.cuda.synchronize()
ps, pt = st_.data_ptr(), th_.data_ptr()
env.step_synthetic(4000, 1); env.sync()
for rep in range(3):
    for _ in range(300): env.step_device_wait(ps, pt)
    t0 = time.perf_counter()
    for _ in range(3000): env.step_device_wait(ps, pt)
    lock = (time.perf_counter() - t0) / 3000 * 1e6
    t0 = time.perf_counter()
    env.step_synthetic(3000, 1); env.sync()
    q = (time.perf_counter() - t0) / 3000 * 1e6
    print(f"n_envs={n}: lock step {lock:6.2f} us per tick   queued (synthetic controls) {q:6.2f} us per step", flush=True)
