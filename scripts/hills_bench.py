#!/usr/bin/env python3
"""The env step on a track with elevation (mountain_track: per-env row tables from the slope ahead) against the same step on the flat generated track:
1024 envs x 120x160, resident worker (every step posted on its own) and one launch per step; host wall clock between completion flags / stream sync."""
import sys, time
sys.path.insert(0, ".")
from triton_racer_sim_amd.env import BatchedEnv

N, STEPS = 1024, 2000
for track in ("generated_track", "mountain_track"):
    for resident in (True, False):
        for depth in (False, True):
            env = BatchedEnv(n_envs=N, auto_reset=True, track=track, depth=depth)
            env.set_step_mode(resident, 100000)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.08:
                env.step_synthetic(400, 1); env.sync()
            t1 = time.perf_counter()
            env.step_synthetic(STEPS, 1); env.sync()
            us = (time.perf_counter() - t1) * 1e6 / STEPS
            print(f"{track:16s} {'resident' if resident else 'launch  '} {'rgb+depth' if depth else 'rgb      '} {us:7.2f} us per step  {N / us:7.2f} M env-steps/s  LDS {env.map_info.lds_bytes} B")
            env.close()
