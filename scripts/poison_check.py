"""Staging screen (GPU; by hand; needs a -DTRS_DEBUG_PROBES build as TRS_HIP_LIB): the whole LDS of every CU is filled with a
pattern before every step, so a kernel that reads LDS bytes its own staging has not written yet shows wrong frames or states
(against the same steps without the poison)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

from triton_racer_sim_amd.env import BatchedEnv

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
CASES = [dict(n_envs=96), dict(n_envs=1024), dict(n_envs=77, depth=True), dict(n_envs=48, img_h=240, img_w=320, depth=True), dict(n_envs=256, render=False)]
for kw in CASES:
    fields = ("img", "pos_x", "speed", "seg_idx") if kw.get("render", True) else ("pos_x", "speed", "seg_idx")
    ref = BatchedEnv(auto_reset=True, **kw)
    want = []
    for i in range(steps):
        ref.step_synthetic(1, 1)
        want.append([ref.fetch(f) for f in fields])
    ref.close()
    for pattern in (0x00000000, 0xFFFFFFFF, 0x5A5A5A5A):
        env = BatchedEnv(auto_reset=True, **kw)
        poison = env.api.cdll.trs_debug_poison_lds
        poison.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_int]
        bad = 0
        for i in range(steps):
            assert poison(env._h, pattern, 160 * 1024) == 0
            env.step_synthetic(1, 1)
            bad += int(not all(np.array_equal(env.fetch(f), w) for f, w in zip(fields, want[i])))
        env.close()
        print(f"{kw}: LDS poisoned with {pattern:#010x} before every step: wrong steps {bad} of {steps}", flush=True)
