import sys, traceback
sys.path.insert(0, '.')
import numpy as np
from triton_racer_sim_amd.env import BatchedEnv
g = BatchedEnv(n_envs=48, auto_reset=True)      # env first, torch afterwards: the order that used to fail
import torch
g.step_synthetic(5, 1); g.sync()
for name in ("ep_return", "img"):
    h = g.device_array(name)
    print(name, h.__cuda_array_interface__)
    try:
        t = torch.as_tensor(h, device="cuda")
        print(" ok", t.shape, t.dtype, bool(np.array_equal(t.cpu().numpy(), g.fetch(name))))
    except Exception:
        traceback.print_exc()
try:
    h = g.preprocess_latest({})
    print(h.__cuda_array_interface__)
    t = torch.as_tensor(h, device="cuda"); print(" ok", t.shape)
except Exception:
    traceback.print_exc()
