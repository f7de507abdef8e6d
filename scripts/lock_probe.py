#!/usr/bin/env python3
"""Diagnostic (a -DTRS_RESIDENT_DIAG=4 build, loaded through TRS_HIP_LIB): where one raster wave (workgroup 7, wave 0) of the resident worker spends its clocks
in LOCK STEP (post, wait for the frame, post) against queued posts.  Shader clocks (s_memtime), ~2.1 GHz."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from triton_racer_sim_amd.env import BatchedEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
env = BatchedEnv(n_envs=n, auto_reset=True)
env.set_step_mode(True)
st_ = torch.zeros(n, device="cuda"); th_ = torch.full((n,), 0.5, device="cuda")
torch.cuda.synchronize()


def run(label, fn, k):
    env.step_synthetic(300, 1); env.sync()
    env.set_step_mode(False); env.set_step_mode(True)      # the warm-up worker leaves: its probe values are in `base`
    base = env.fetch("stats").astype(np.int64)
    t0 = time.perf_counter(); fn(k); env.sync(); wall = time.perf_counter() - t0
    st = env.fetch("stats").astype(np.int64) - base
    steps, tot = st[44], st[43]
    print(f"{label}: {wall / k * 1e6:.2f} us per step (host wall); raster wave over {steps} steps: whole loop {tot / max(steps, 1):.0f} clk/step")
    for name, v in (("waiting for the post", st[40]), ("waiting for poses", st[41]), ("counted store wait", st[42])):
        print(f"    {name:22s} {v / max(steps, 1):9.0f} clk/step  {100.0 * v / max(tot, 1):5.1f} %")


def lock(k):
    for _ in range(k):
        env.step_device_wait(st_.data_ptr(), th_.data_ptr())


def queued(k):
    for _ in range(k):
        env.step_device(st_.data_ptr(), th_.data_ptr())


run("queued posts", queued, 2000)
run("lock step   ", lock, 2000)
