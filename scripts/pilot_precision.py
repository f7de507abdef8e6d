#!/usr/bin/env python3
"""Pilot precision on record: max |HIP - fp32 PyTorch| per output over a batch of rendered + noise frames, at both
frame sizes, next to what a bf16 / an fp16 operand-and-activation pipeline would give in emulation (PyTorch fp32 arithmetic with the
weights and every stored activation rounded to that format: the product's stated arithmetic, and the fp16 alternative)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import torch
import torch.nn.functional as F
from test_pilot import SPEC, make_weights, torch_pure, torch_tail
from triton_racer_sim_amd.env import BatchedEnv


def emulated(frames, ws, fmt):
    rnd = (lambda t: t.bfloat16().float()) if fmt == "bf16" else (lambda t: t.half().float())
    x = torch.from_numpy(np.ascontiguousarray(frames, dtype=np.float32))
    for i, (k, s, cin, cout) in enumerate(SPEC):
        wk = torch.from_numpy(ws[2 * i])
        if i == 0:
            wk = wk / 255.0 if fmt == "bf16" else wk          # bf16: 1/255 folded into conv1's weights (as the kernel does); fp16: applied to the sum
        wk = rnd(wk).permute(3, 2, 0, 1).contiguous()
        y = F.conv2d(x.permute(0, 3, 1, 2), wk, None, stride=s)
        if i == 0 and fmt != "bf16":
            y = y / 255.0
        y = F.relu(y + torch.from_numpy(ws[2 * i + 1]).view(1, -1, 1, 1))
        x = rnd(y).permute(0, 2, 3, 1).contiguous()
    h1 = F.relu(x.reshape(x.shape[0], -1) @ rnd(torch.from_numpy(ws[14])) + torch.from_numpy(ws[15]))
    return torch_tail(h1.numpy(), ws)


for (h, w) in ((120, 160), (240, 320)):
    n = 48
    env = BatchedEnv(n_envs=n, img_h=h, img_w=w, auto_reset=True)
    worst = {"hip": np.zeros(2), "bf16": np.zeros(2), "fp16": np.zeros(2)}
    rms = {"hip": np.zeros(2), "bf16": np.zeros(2), "fp16": np.zeros(2)}
    cnt = 0
    for seed in (3, 4, 5):
        ws = make_weights(h, w, seed=seed)
        env.pilot_load(ws)
        env.step_synthetic(30, 1)
        frames = env.fetch("img")
        rng = np.random.default_rng(seed)
        frames = np.concatenate([frames[:32], rng.integers(0, 256, (16, h, w, 3), dtype=np.uint8)])
        pure = torch_pure(frames, ws)
        outs = {"hip": env.pilot_forward_host(frames), "bf16": emulated(frames, ws, "bf16"), "fp16": emulated(frames, ws, "fp16")}
        for k, o in outs.items():
            d = np.abs(o - pure)
            worst[k] = np.maximum(worst[k], d.max(0))
            rms[k] += (d ** 2).sum(0)
        cnt += len(frames)
        scale = np.abs(pure).max(0)
    print(f"== {h}x{w}: {cnt} frames (2/3 rendered, 1/3 noise), 3 weight sets; outputs (steering, speed/20); |output| up to {scale}")
    for k in ("hip", "bf16", "fp16"):
        label = {"hip": "HIP kernels (fp16 operands / activations)", "bf16": "emulated bf16 pipeline", "fp16": "emulated fp16 pipeline"}[k]
        print(f"  {label:44s} max |x - fp32| = {worst[k][0]:.3e}, {worst[k][1]:.3e}   rms = {np.sqrt(rms[k][0] / cnt):.3e}, {np.sqrt(rms[k][1] / cnt):.3e}")
    env.close()
