#!/usr/bin/env python3
"""Phase clocks of the fused head / the conv4-7 chain (a -DTRS_BAND_STAMPS / -DTRS_CHAIN_STAMPS build given in TRS_HIP_LIB prints
them from the device).  usage: band_stamps.py [envs] [H] [W] [tuning_field=value ...]"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_pilot import make_weights
from triton_racer_sim_amd.env import BatchedEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
H = int(sys.argv[2]) if len(sys.argv) > 2 else 120
W = int(sys.argv[3]) if len(sys.argv) > 3 else 160
env = BatchedEnv(n_envs=N, auto_reset=True, img_h=H, img_w=W)
tune = dict(kv.split("=") for kv in sys.argv[4:])                 # e.g. chain_nt=2 (fields of trs_pilot_tuning)
if tune:
    env.pilot_tuning(**{k: int(v) for k, v in tune.items()})
env.pilot_load(make_weights(H, W, seed=1))
env.step_synthetic(4, 1)
import os
for _ in range(int(os.environ.get("STAMP_STEPS", "3"))):   # STAMP_STEPS=400: the clocks of a loaded chip (read the last lines)
    env.step_pilot(1)
    env.sync()
