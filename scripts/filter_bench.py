#!/usr/bin/env python3
"""cam/processed_img per step: fused behind the rasteriser (static palette filter, dynamic-brightness instantiation) vs the
separate trs_preprocess kernel; device time by HIP events on the handle's stream."""
import sys
sys.path.insert(0, ".")
from triton_racer_sim_amd.env import BatchedEnv

N, STEPS = 1024, 600
RESIDENT = len(sys.argv) > 1 and sys.argv[1] == "resident"      # steps posted to the resident worker instead of one launch per step
STATIC = {"preprocessing_color_filter_enabled": True, "preprocessing_contrast_enhancement_ratio": 1.2}
DYN = dict(STATIC, preprocessing_dynamic_brightness_enabled=True)


def timed(env, fn):
    fn(50)
    env.sync(); env.event_record(0); fn(STEPS); env.event_record(1); env.sync()
    return env.event_elapsed_ms(0, 1) * 1e3 / STEPS


env = BatchedEnv(n_envs=N, auto_reset=True)
if RESIDENT:
    env.set_step_mode(True)
env.step_synthetic(6000, 1)                                     # GPU clocks up
rows = []
rows.append(("raw frames (no filter)", timed(env, lambda k: env.step_synthetic(k, 1))))
env.set_frame_filter(STATIC)
rows.append(("trim + HSV masks fused (palette filtered on the host)", timed(env, lambda k: env.step_synthetic(k, 1))))
env.set_frame_filter(DYN)
rows.append(("+ dynamic brightness fused (per-env palette in the kernel)", timed(env, lambda k: env.step_synthetic(k, 1))))
env.set_frame_filter(enabled=False)
pc = env.pre_config(DYN)


def separate(k):
    for _ in range(k):
        env.step_synthetic(1, 1)
        env.preprocess_latest(pc)


rows.append(("raw frames + separate trs_preprocess kernel (same filter)", timed(env, separate)))
print("step mode:", "resident worker (every step posted on its own)" if RESIDENT else "one launch per step")
for name, us in rows:
    print(f"{name:62s} {us:8.2f} us / step   {N / us:7.2f} M env-steps/s")
