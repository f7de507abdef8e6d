#!/usr/bin/env python3
"""Closed-loop rate of the three KerasPilot model types that have their own tail (cnn_2d_speed_control, cnn_2d_speed_as_feature,
cnn_2d_full_house), 1024 envs x 120x160, random-init weights: what the extra dense branches and the second head cost."""
import os, sys, time, json, importlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_pilot import make_weights
from test_pilot_types import make_named_weights
pkg = importlib.import_module("triton-racer-sim_amd")
n, steps = 1024, 300
out = {}
for kind in ("cnn_2d_speed_control", "cnn_2d_speed_as_feature", "cnn_2d_full_house"):
    env = pkg.BatchedEnv(n_envs=n, auto_reset=True)
    if kind == "cnn_2d_speed_control":
        env.pilot_load(make_weights(120, 160, seed=1))
    else:
        env.pilot_load(make_named_weights(120, 160, kind, seed=1)[0])
    cfg = {"model_type": kind}
    env.step_synthetic(2, 1)
    env.step_pilot(30, cfg); env.sync()
    t0 = time.perf_counter(); env.step_pilot(steps, cfg); env.sync(); dt = time.perf_counter() - t0
    out[kind] = {"env_steps_per_s": round(n * steps / dt, 1), "us_per_step": round(dt / steps * 1e6, 2)}
    env.close()
print(json.dumps(out))
