#!/usr/bin/env python3
"""Soak of the per-GPU worker arbitration (round 5): three handles of ONE process on one GPU (two with a camera, one physics-only), a random interleaving of
posted steps (queued and lock step), fetches, mode changes, quiesces and resets for SECONDS seconds; every handle has a CPU-oracle twin that gets the same
calls, compared bit for bit at checkpoints.  Any TRS_ERR_DEVICE, stall or mismatch fails."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from triton_racer_sim_amd import _ffi
from triton_racer_sim_amd.env import BatchedEnv

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
oracle = _ffi.Api(ctypes.CDLL(os.path.join(ROOT, "oracle", "libtrsim_oracle.so")), "trso_")
specs = [dict(n_envs=96, auto_reset=True), dict(n_envs=64, auto_reset=True, img_h=60, img_w=80, env_id_base=500, track="mountain_track"), dict(n_envs=128, auto_reset=True, render=False, env_id_base=900)]
pairs = [(BatchedEnv(**kw), BatchedEnv(_api=oracle, **kw)) for kw in specs]
for g, _ in pairs:
    g.set_step_mode(True)
rng = np.random.default_rng(77)
t0 = time.time()
ops = steps = checks = 0
worst = 0.0
while time.time() - t0 < SECONDS:
    i = int(rng.integers(0, len(pairs)))
    g, o = pairs[i]
    op = rng.uniform()
    t1 = time.perf_counter()
    if op < 0.55:
        k = int(rng.integers(1, 13))
        g.step_synthetic(k, 1); o.step_synthetic(k, 1); steps += k
    elif op < 0.75:
        n = g.n
        st, th = rng.uniform(-1, 1, n).astype(np.float32), rng.uniform(0, 1, n).astype(np.float32)
        g.step(st, th, 0.0); g.sync(); o.step(st, th, 0.0); steps += 1
    elif op < 0.82:
        g.sync()
    elif op < 0.88:
        name = ["pos_x", "seg_idx", "speed"][int(rng.integers(0, 3))]
        a, b = g.fetch(name), o.fetch(name)
        assert np.array_equal(a, b) if a.dtype.kind in "iu" else np.max(np.abs(a - b)) <= 1e-5, (name, i)
        checks += 1
    elif op < 0.92:
        g.quiesce()
    elif op < 0.96:
        on = bool(rng.integers(0, 2))
        g.set_step_mode(on)
    else:
        mask = (rng.uniform(0, 1, g.n) < 0.2).astype(np.uint8)
        g.reset(mask); o.reset(mask)
    worst = max(worst, time.perf_counter() - t1)
    ops += 1
    if ops % 400 == 0:
        for j, (gg, oo) in enumerate(pairs):
            for name in ("seg_idx", "done", "ep_len"):
                assert np.array_equal(gg.fetch(name), oo.fetch(name)), (name, j, ops)
            if gg.cfg.render:
                assert np.array_equal(gg.fetch("img"), oo.fetch("img")), ("img", j, ops)
        checks += 1
for j, (gg, oo) in enumerate(pairs):
    for name in ("pos_x", "pos_z", "yaw", "speed", "cte", "ep_return"):
        assert np.max(np.abs(gg.fetch(name) - oo.fetch(name))) <= 1e-5, (name, j)
    if gg.cfg.render:
        assert np.array_equal(gg.fetch("img"), oo.fetch("img")), ("img", j)
print(f"arbitration soak OK: {ops} operations, {steps} env steps on 3 handles in {time.time() - t0:.1f} s, {checks} checkpoints equal to the oracle, "
      f"longest single call {worst * 1e3:.1f} ms (includes the oracle twin), modes at the end {[g.step_mode() for g, _ in pairs]}")
