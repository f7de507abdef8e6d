#!/usr/bin/env python3
"""Round 4 soak of the new resident paths against the oracle (test infrastructure: the checker, not the product): many worker generations
(short lifetimes), mixed queue depths (synthetic bursts, host controls, lock step), physics-only and dynamic-brightness workers."""
import ctypes, os, sys, time
sys.path.insert(0, ".")
import numpy as np
from triton_racer_sim_amd import _ffi
from triton_racer_sim_amd.env import BatchedEnv

oracle = _ffi.Api(ctypes.CDLL(os.path.join("oracle", "libtrsim_oracle.so")), "trso_")
rng = np.random.default_rng(404)


def compare(g, o, what, image):
    for name in ("seg_idx", "done", "ep_len"):
        assert np.array_equal(g.fetch(name), o.fetch(name)), (what, name)
    for name in ("pos_x", "pos_y", "pos_z", "speed", "cte", "yaw", "ep_return", "last_return", "steer_filt"):
        d = float(np.max(np.abs(g.fetch(name).astype(np.float64) - o.fetch(name))))
        assert d <= 1e-5, (what, name, d)
    if image:
        assert np.array_equal(g.fetch("img"), o.fetch("img")), (what, "img")


def soak(n, render, dyn, rounds, life_us):
    kw = dict(n_envs=n, render=render, auto_reset=True)
    g, o = BatchedEnv(**kw), BatchedEnv(_api=oracle, **kw)
    if dyn:
        f = {"preprocessing_dynamic_brightness_enabled": True, "preprocessing_color_filter_enabled": True, "preprocessing_contrast_enhancement_ratio": 1.2}
        g.set_frame_filter(f); o.set_frame_filter(f)
    g.set_step_mode(True, idle_us=int(rng.integers(200, 3000)))
    g.resident_lifetime(life_us)
    total = 0
    t0 = time.time()
    for r in range(rounds):
        kind = r % 4
        if os.environ.get("SOAK_TRACE"): print(f"    round {r} kind {kind}", flush=True)
        if kind == 0:
            k = int(rng.integers(1, 400))
            g.step_synthetic(k, 1); o.step_synthetic(k, 1)
        elif kind == 1:
            k = int(rng.integers(1, 12))
            for _ in range(k):
                st, th = rng.uniform(-1, 1, n).astype(np.float32), rng.uniform(-0.2, 1, n).astype(np.float32)
                rs = (rng.uniform(0, 1, n) < 0.01) if rng.uniform() < 0.2 else None
                g.step(st, th, 0.0, reset=rs); o.step(st, th, 0.0, reset=rs)
        elif kind == 2:
            k = int(rng.integers(1, 20))
            for _ in range(k):
                g.step_synthetic(1, 1); g.sync(); o.step_synthetic(1, 1)       # lock step
        else:
            k = 0
            time.sleep(float(rng.uniform(0, 0.004)))                         # idle exits
        total += k
        if r % 25 == 24:
            compare(g, o, f"n={n} render={render} dyn={dyn} round {r}", render and r % 100 == 99)
            print(f"  ... n={n} render={render} dyn={dyn}: round {r}, {total} steps, {time.time() - t0:.1f} s", flush=True)
    compare(g, o, f"n={n} render={render} dyn={dyn} end", render)
    assert int(g.fetch("stats")[2]) == 0
    print(f"soak n={n} render={render} dyn={dyn} lifetime {life_us} us: {rounds} rounds, {total} steps, {time.time() - t0:.1f} s: HIP == oracle", flush=True)
    g.close(); o.close()


soak(256, False, False, 600, 700)
soak(101, False, False, 400, 300)
soak(20, True, True, 300, 1500)        # (the oracle renders and filters every frame on the CPU: few envs)
soak(24, True, False, 300, 900)
