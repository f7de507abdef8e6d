"""Race screen (GPU): every product kernel beside a busy second stream gives what it gives alone.

Single-stream parity cannot see faults that need workgroups of OTHER kernels on the same CU (different LDS leftovers, issue
timing, register-file neighbours); one such fault sat in the rasteriser (profiles/r01_coresidency_screen.txt).  Each case runs
an operation alone, then again while another handle keeps the pilot loop (MFMA convolutions, the step kernel, small tail
kernels) running from a thread, and compares bit for bit."""
import contextlib
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def busy_neighbour(make_env):
    from test_pilot import make_weights
    other = make_env("hip", n_envs=96, auto_reset=True)
    other.pilot_load(make_weights(120, 160, seed=9))
    other.step_synthetic(3, 1)
    stop = threading.Event()

    def loop():
        while not stop.is_set():
            other.step_pilot(8)
            other.sync()

    t = threading.Thread(target=loop)
    t.start()
    try:
        yield
    finally:
        stop.set()
        t.join()


def run_env(make_env, rounds, per_call, per_launch, fields, setup=None, **kw):
    env = make_env("hip", auto_reset=True, **kw)
    if setup:
        setup(env)
    out = []
    for _ in range(rounds):
        env.step_synthetic(per_call, per_launch)
        out.append([env.fetch(f) for f in fields])
    return out


CASES = {
    "one step per launch": dict(rounds=150, per_call=1, per_launch=1, fields=("img", "pos_x", "speed", "cte", "seg_idx"), n_envs=96),
    "eight steps per launch": dict(rounds=40, per_call=8, per_launch=8, fields=("img", "pos_x", "yaw", "ep_return"), n_envs=200),
    "pipelined launches": dict(rounds=40, per_call=6, per_launch=2, fields=("img", "pos_z", "speed"), n_envs=96),
    "depth frames": dict(rounds=60, per_call=1, per_launch=1, fields=("img", "depth", "pos_x"), n_envs=64, depth=True),
    "240x320 + depth": dict(rounds=30, per_call=1, per_launch=1, fields=("img", "depth"), n_envs=48, img_h=240, img_w=320, depth=True),
    "dynamic brightness in the step kernel": dict(rounds=60, per_call=1, per_launch=1, fields=("img", "pos_x"), n_envs=101,
                                                  setup=lambda e: e.set_frame_filter({"preprocessing_dynamic_brightness_enabled": True, "preprocessing_color_filter_enabled": True})),
    "physics only": dict(rounds=60, per_call=16, per_launch=16, fields=("pos_x", "pos_z", "yaw", "speed", "cte", "seg_idx"), n_envs=256, render=False),
}


@pytest.mark.parametrize("case", list(CASES))
def test_env_steps_beside_a_busy_stream(make_env, case):
    kw = dict(CASES[case])
    alone = run_env(make_env, **kw)
    with busy_neighbour(make_env):
        beside = run_env(make_env, **kw)
    for k, (a, b) in enumerate(zip(alone, beside)):
        for f, x, y in zip(kw["fields"], a, b):
            assert np.array_equal(x, y), (case, f, "call", k)


def test_image_path_and_queries_beside_a_busy_stream(make_env):
    env = make_env("hip", n_envs=64, auto_reset=True)
    env.step_synthetic(20, 1)
    frames = env.fetch("img")
    rng = np.random.default_rng(4)
    noise = rng.integers(0, 256, frames.shape, dtype=np.uint8)
    cfgs = [
        {"preprocessing_dynamic_brightness_enabled": True, "preprocessing_color_filter_enabled": True},
        {"preprocessing_edge_detection_enabled": True, "preprocessing_contrast_enhancement_ratio": 1.2},
    ]
    pts = rng.uniform([40, 0, -10], [95, 1, 85], (20000, 3))
    spd = rng.uniform(0, 20, 64).astype(np.float32)

    def everything():
        out = []
        for _ in range(25):
            for cfg in cfgs:
                out += [env.preprocess_host(frames, cfg), env.preprocess_host(noise, cfg)]
            out += [env.normalize_host(frames), env.locate(pts)]
            out += list(env.driver_assist_host(np.linspace(-1, 1, 64), np.full(64, 0.7), np.zeros(64), spd, mode="steering"))
        return out

    alone = everything()
    with busy_neighbour(make_env):
        beside = everything()
    for k, (x, y) in enumerate(zip(alone, beside)):
        assert np.array_equal(np.asarray(x), np.asarray(y)), k
