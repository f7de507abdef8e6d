#!/usr/bin/env python3
"""Batched LocationTracker (trs_locate: host points in, host indices out) — queries per second incl. the PCIe copies."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from triton_racer_sim_amd.env import BatchedEnv

for track in ("generated_track.json", "mountain_track.json"):
    env = BatchedEnv(n_envs=1, track=track, render=False)
    pts = np.asarray(env.track_points, dtype=np.float64) if hasattr(env, "track_points") else None
    rng = np.random.default_rng(0)
    n = 1 << 20
    if pts is None:
        import json, os
        here = os.path.join("triton-racer-sim_amd", "track_data", track)
        pts = np.asarray(json.load(open(here)), dtype=np.float64)
    q = pts[rng.integers(0, len(pts), n)] + rng.normal(0, 0.5, (n, 3))
    env.locate(q[:1024])
    t0 = time.perf_counter(); idx = env.locate(q); dt = time.perf_counter() - t0
    print(f"{track:22s} {n} queries in {dt * 1e3:7.2f} ms = {n / dt / 1e6:6.1f} M queries/s (host -> device -> host); reference: 323 / 658 us per query on the CPU")
    env.close()
