"""Long form of tests/test_coresidency.py (GPU; by hand): every product operation, many repetitions, beside a handle that
keeps the pilot loop running on its own stream — results must equal the same operation's results alone.

    python scripts/coresidency_soak.py [repetitions]        ->  one line per case, mismatching repetitions (expect 0)"""
import os
import sys
import threading

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np

from test_pilot import make_weights
from triton_racer_sim_amd.env import BatchedEnv

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 1000


def env_case(rounds, per_call, per_launch, fields, setup=None, **kw):
    def run():
        env = BatchedEnv(auto_reset=True, **kw)
        if setup:
            setup(env)
        out = []
        for _ in range(rounds):
            env.step_synthetic(per_call, per_launch)
            out.append([env.fetch(f) for f in fields])
        env.close()
        return out
    return run


def image_case():
    env = BatchedEnv(n_envs=64, auto_reset=True)
    env.step_synthetic(20, 1)
    frames = env.fetch("img")
    rng = np.random.default_rng(4)
    noise = rng.integers(0, 256, frames.shape, dtype=np.uint8)
    cfgs = [{"preprocessing_dynamic_brightness_enabled": True, "preprocessing_color_filter_enabled": True},
            {"preprocessing_edge_detection_enabled": True, "preprocessing_contrast_enhancement_ratio": 1.2}]
    pts = rng.uniform([40, 0, -10], [95, 1, 85], (20000, 3))
    spd = rng.uniform(0, 20, 64).astype(np.float32)

    def run():
        out = []
        for _ in range(max(1, REPS // 20)):
            for cfg in cfgs:
                out.append([env.preprocess_host(frames, cfg), env.preprocess_host(noise, cfg)])
            out.append([env.normalize_host(frames), env.locate(pts)])
            out.append(list(env.driver_assist_host(np.linspace(-1, 1, 64), np.full(64, 0.7), np.zeros(64), spd, mode="steering")))
        return out
    return run


CASES = {
    "one step per launch": env_case(REPS, 1, 1, ("img", "pos_x", "speed", "cte", "seg_idx"), n_envs=96),
    "eight steps per launch": env_case(REPS // 4, 8, 8, ("img", "pos_x", "yaw", "ep_return"), n_envs=200),
    "pipelined launches": env_case(REPS // 4, 6, 2, ("img", "pos_z", "speed"), n_envs=96),
    "depth frames": env_case(REPS // 2, 1, 1, ("img", "depth", "pos_x"), n_envs=64, depth=True),
    "240x320 + depth": env_case(REPS // 4, 1, 1, ("img", "depth"), n_envs=48, img_h=240, img_w=320, depth=True),
    "dynamic brightness in the step kernel": env_case(REPS // 2, 1, 1, ("img", "pos_x"), n_envs=101,
                                                      setup=lambda e: e.set_frame_filter({"preprocessing_dynamic_brightness_enabled": True, "preprocessing_color_filter_enabled": True})),
    "physics only": env_case(REPS // 2, 16, 16, ("pos_x", "pos_z", "yaw", "speed", "cte", "seg_idx"), n_envs=256, render=False),
    "image path, queries, control glue": image_case(),
}


def main():
    bad_total = 0
    for name, run in CASES.items():
        alone = run()
        other = BatchedEnv(n_envs=96, auto_reset=True)
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
            beside = run()
        finally:
            stop.set()
            t.join()
            other.close()
        bad = sum(int(not all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(a, b))) for a, b in zip(alone, beside))
        bad_total += bad
        print(f"{name:42s} mismatching repetitions: {bad} of {len(alone)}", flush=True)
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
