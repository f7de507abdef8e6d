#!/usr/bin/env python3
"""Phase clocks of trs_preprocess_edge_kernel (a -DTRS_EDGE_STAMPS build given in TRS_HIP_LIB prints them from the device)."""
import sys
sys.path.insert(0, ".")
from triton_racer_sim_amd.env import BatchedEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
H = int(sys.argv[2]) if len(sys.argv) > 2 else 120
W = int(sys.argv[3]) if len(sys.argv) > 3 else 160
env = BatchedEnv(n_envs=N, auto_reset=True, img_h=H, img_w=W)
env.step_synthetic(20, 1)
pc = env.pre_config({"preprocessing_contrast_enhancement_ratio": 1.2, "preprocessing_dynamic_brightness_enabled": True, "preprocessing_color_filter_enabled": True,
                     "preprocessing_edge_detection_enabled": True})
for _ in range(3):
    env.preprocess_latest(pc)
    env.sync()
