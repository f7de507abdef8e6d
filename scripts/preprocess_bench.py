#!/usr/bin/env python3
"""trs_preprocess / trs_normalize on the env's own frames (device resident): us per batch, GB/s of algorithmic traffic."""
import sys
sys.path.insert(0, ".")
import ctypes as C
from triton_racer_sim_amd.env import BatchedEnv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
H = int(sys.argv[2]) if len(sys.argv) > 2 else 120
W = int(sys.argv[3]) if len(sys.argv) > 3 else 160
env = BatchedEnv(n_envs=N, auto_reset=True, img_h=H, img_w=W)
env.step_synthetic(20, 1)
CFGS = {
    "identity trim": {},
    "trim (contrast 1.2)": {"preprocessing_contrast_enhancement_ratio": 1.2},
    "trim + dynamic brightness": {"preprocessing_contrast_enhancement_ratio": 1.2, "preprocessing_dynamic_brightness_enabled": True},
    "trim + dynamic + masks + Canny": {"preprocessing_contrast_enhancement_ratio": 1.2, "preprocessing_dynamic_brightness_enabled": True, "preprocessing_color_filter_enabled": True,
                                       "preprocessing_edge_detection_enabled": True},
    "trim + dynamic + HSV masks": {"preprocessing_contrast_enhancement_ratio": 1.2, "preprocessing_dynamic_brightness_enabled": True, "preprocessing_color_filter_enabled": True},
    "trim + HSV masks (no dynamic)": {"preprocessing_contrast_enhancement_ratio": 1.2, "preprocessing_color_filter_enabled": True},
}
frame = env.H * env.W * 3
for name, cfg in CFGS.items():
    pc = env.pre_config(cfg)
    for _ in range(5):
        env.preprocess_latest(pc)
    env.sync(); env.event_record(0)
    for _ in range(50):
        env.preprocess_latest(pc)
    env.event_record(1); env.sync()
    us = env.event_elapsed_ms(0, 1) * 1e3 / 50
    print(f"{name:32s} {us:8.2f} us per {N} frames   {2 * frame * N / us / 1e3:7.1f} GB/s (read + write once)")

# pilot-side normalisation float32(img) / 255 (keras_pilot.py:49-55): 1 byte in, 4 bytes out per channel value
hip = C.CDLL(None)                                      # hipMalloc of the runtime libtrsim.so already uses (global symbols; no torch import: slow on a fresh box)
dst = C.c_void_p()
assert hip.hipMalloc(C.byref(dst), C.c_size_t(4 * frame * N)) == 0
for _ in range(5):
    env.api.check(env.api.normalize(env._h, None, dst, N), "normalize")
env.sync(); env.event_record(0)
for _ in range(50):
    env.api.check(env.api.normalize(env._h, None, dst, N), "normalize")
env.event_record(1); env.sync()
us = env.event_elapsed_ms(0, 1) * 1e3 / 50
print(f"{'normalize (u8 -> f32 / 255)':32s} {us:8.2f} us per {N} frames   {5 * frame * N / us / 1e3:7.1f} GB/s (1 B read + 4 B written per value)")
hip.hipFree(dst)
