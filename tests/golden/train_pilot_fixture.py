#!/usr/bin/env python3
"""Generates tests/golden/pilot_trained_120x160.npz: weights of the reference's cnn_2d_speed_control architecture
(Keras_2D_CNN.get_model(input_shape=(120, 160, 3), num_outputs=2), components/keras_train.py:127-174) TRAINED the way the
reference trains it (components/keras_train.py:264-299: inputs = recorded frames / 255, targets = [steering, speed / 20],
mean squared error, Adam) — on records of a scripted driver in the build's own CPU oracle, because no Keras, no TensorFlow and
no recorded tub exist in this image.  PyTorch on the CPU stands in for Keras; the arrays are stored in Keras layouts
([KH][KW][CIN][COUT] kernels, [IN][OUT] dense matrices behind an NHWC flatten) as binary16, which is what the library rounds
them to anyway.

Why: every other pilot test uses Glorot-random weights, whose outputs barely depend on the image.  A
trained network has the weight and activation statistics of a real model, and it DRIVES: tests/test_pilot_trained.py runs it in
the closed loop on the GPU and checks that the cars stay on the road.

Run here (CPU only, ~10-15 minutes on 8 cores): python tests/golden/train_pilot_fixture.py
"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

import __graft_entry__ as g
from triton_racer_sim_amd import _ffi
from triton_racer_sim_amd.env import BatchedEnv

SPEC = [(5, 2, 3, 24), (5, 2, 24, 32), (5, 2, 32, 64), (3, 1, 64, 64), (3, 1, 64, 64), (3, 1, 64, 128), (3, 1, 128, 128)]
H, W = 120, 160


def wrap(a):
    return (a + np.pi) % (2 * np.pi) - np.pi


def expert(env, track_yaw, target_speed):
    """The scripted driver: steer against the cross-track error and the heading error six track points ahead, hold a speed."""
    seg, cte, yaw, spd = env.fetch("seg_idx"), env.fetch("cte"), env.fetch("yaw"), env.fetch("speed")
    herr = wrap(yaw - track_yaw[(seg + 6) % len(track_yaw)])
    steer = np.clip(-(1.0 * cte + 2.0 * herr), -1, 1).astype(np.float32)
    thr = np.where(spd < target_speed, 0.6, 0.1).astype(np.float32)
    return steer, thr


def collect(n_envs=256, steps=420, every=5, seed=0):
    oracle = _ffi.Api(ctypes.CDLL(g.build_oracle()), "trso_")
    env = BatchedEnv(n_envs=n_envs, auto_reset=True, _api=oracle)
    tan = env.fetch("tangent")
    track_yaw = np.arctan2(tan[:, 0], tan[:, 1])
    rng = np.random.default_rng(seed)
    target = rng.uniform(4.0, 8.0, n_envs).astype(np.float32)            # each car holds its own speed: the speed output has something to learn
    env.step(0.0, 0.0, 0.0)
    frames, ys = [], []
    wobble = np.zeros(n_envs, np.float32)
    for t in range(steps):
        steer, thr = expert(env, track_yaw, target)
        if t % every == 0 and t >= 20:
            frames.append(env.fetch("img").copy())
            ys.append(np.stack([steer, env.fetch("speed") / 20.0], 1).astype(np.float32))
        # the car does not follow the expert exactly: a slowly varying disturbance takes it off the centre line, the LABEL stays the expert's answer
        wobble = 0.9 * wobble + 0.1 * rng.normal(0, 0.8, n_envs).astype(np.float32)
        env.step(np.clip(steer + wobble, -1, 1), thr, 0.0)
    X = np.concatenate(frames)
    Y = np.concatenate(ys)
    print(f"collected {X.shape[0]} frames; steering std {Y[:, 0].std():.3f}, speed/20 mean {Y[:, 1].mean():.3f}; off-track resets seen: {int(env.fetch('ep_len').size)} envs", flush=True)
    return X, Y, env, track_yaw


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.convs = torch.nn.ModuleList([torch.nn.Conv2d(cin, cout, k, stride=s) for k, s, cin, cout in SPEC])
        ih, iw = H, W
        for k, s, _, _ in SPEC:
            ih, iw = (ih - k) // s + 1, (iw - k) // s + 1
        dims = [ih * iw * 128, 100, 50, 25, 2]
        self.dense = torch.nn.ModuleList([torch.nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])])
        for m in list(self.convs) + list(self.dense):                        # Keras defaults: Glorot-uniform kernels, zero biases
            torch.nn.init.xavier_uniform_(m.weight)
            torch.nn.init.zeros_(m.bias)

    def forward(self, x_u8_nhwc):
        x = x_u8_nhwc.float().div(255.0).permute(0, 3, 1, 2)                 # keras_pilot.py:49-50 / keras_train.py:41-42
        for c in self.convs:
            x = F.relu(c(x))
        x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)                    # Keras Flatten on NHWC
        for i, d in enumerate(self.dense):
            x = d(x)
            if i < 3:
                x = F.relu(x)
        return x

    def keras_arrays(self):
        ws = []
        for c in self.convs:
            ws += [c.weight.detach().permute(2, 3, 1, 0).contiguous().numpy(), c.bias.detach().numpy()]
        for d in self.dense:
            ws += [d.weight.detach().t().contiguous().numpy(), d.bias.detach().numpy()]
        return ws


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    X, Y, env, track_yaw = collect()
    net = Net()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    Xt, Yt = torch.from_numpy(X), torch.from_numpy(Y)
    n = Xt.shape[0]
    t0 = time.time()
    for epoch in range(4):
        perm = torch.randperm(n)
        tot = 0.0
        for i in range(0, n - 63, 64):
            idx = perm[i:i + 64]
            loss = F.mse_loss(net(Xt[idx]), Yt[idx])
            opt.zero_grad(); loss.backward(); opt.step()
            tot += float(loss)
        print(f"epoch {epoch}: mean loss {tot / (n // 64):.5f}  ({time.time() - t0:.0f} s)", flush=True)
        if epoch == 2:
            for gr in opt.param_groups:
                gr["lr"] = 3e-4
    # does it drive?  the network alone in the loop (oracle env, fp32 PyTorch), KerasPilot's post-processing restated inline
    import math
    n_eval = 32
    oracle = _ffi.Api(ctypes.CDLL(g.ORACLE_LIB), "trso_")
    ev = BatchedEnv(n_envs=n_eval, auto_reset=True, _api=oracle)
    ev.step(0.0, 0.0, 0.0)
    dones = 0
    net.eval()
    with torch.no_grad():
        for t in range(300):
            out = net(torch.from_numpy(ev.fetch("img"))).numpy()
            spd = ev.fetch("speed")
            steer = np.clip(out[:, 0], -1, 1)
            thr = np.array([math.atan((o * 20 * 1.1 - s) * 2) / (math.pi / 2) for o, s in zip(out[:, 1], spd)], np.float32)
            thr[(thr > -0.2) & (thr < 0.0)] = 0.0
            ev.step(steer.astype(np.float32), thr, 0.0)
            dones += int(ev.fetch("done").sum())
    print(f"closed loop on the oracle, {n_eval} cars x 300 ticks: off-track events {dones}, mean speed {ev.fetch('speed').mean():.2f}, mean |cte| {np.abs(ev.fetch('cte')).mean():.3f}", flush=True)
    ws = net.keras_arrays()
    out = os.path.join(ROOT, "tests", "golden", "pilot_trained_120x160.npz")
    np.savez_compressed(out, **{f"a{i:02d}": w.astype(np.float16) for i, w in enumerate(ws)})
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
