#!/usr/bin/env python3
"""Diagnostic (TRS_RESIDENT_DIAG=4): where one raster wave of the resident worker spends its clocks."""
import os, sys
os.environ["TRS_RESIDENT_DIAG"] = "4"
sys.path.insert(0, ".")
import numpy as np
from triton_racer_sim_amd.env import BatchedEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
env = BatchedEnv(n_envs=n, auto_reset=True)
env.set_step_mode(True)
env.step_synthetic(200, 1); env.sync()
env.set_step_mode(False); env.set_step_mode(True)      # the warm-up worker leaves (its probe values are discarded below)
base = env.fetch("stats").astype(np.int64)              # quiesces
import time
t0 = time.perf_counter()
env.step_synthetic(2000, 1); env.sync()
wall = time.perf_counter() - t0
st = env.fetch("stats").astype(np.int64) - base
steps = st[44]
print(f"n_envs={n}: {wall / 2000 * 1e6:.2f} us per step (host wall); raster wave (wg 7, wave 0) over {steps} steps:")
tot = st[43]
for name, v in (("waiting for the post", st[40]), ("waiting for poses", st[41]), ("counted store wait", st[42])):
    print(f"  {name:22s} {v / max(steps, 1):9.0f} clk/step  {100.0 * v / max(tot, 1):5.1f} %")
print(f"  {'whole loop':22s} {tot / max(steps, 1):9.0f} clk/step")
