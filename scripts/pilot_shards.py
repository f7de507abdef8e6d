#!/usr/bin/env python3
"""Closed pilot loop with the envs of one GPU split over K handles (shards) on K streams: each shard's launch gaps and kernel tails are
filled by the other's kernels.  Shards are independent by construction (RNG and start poses keyed by global env id), so the frames and
states are those of one handle of the same envs (checked here against a single handle for the first shard's rows).
usage: pilot_shards.py [--envs 1024] [--shards 2] [--steps 300] [--img-h 120 --img-w 160] [--depth]"""
import argparse, importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("triton-racer-sim_amd")
BatchedEnv = pkg.BatchedEnv

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=1024); ap.add_argument("--shards", type=int, default=2)
ap.add_argument("--steps", type=int, default=300); ap.add_argument("--warmup", type=int, default=30)
ap.add_argument("--img-h", type=int, default=120); ap.add_argument("--img-w", type=int, default=160)
ap.add_argument("--depth", action="store_true"); ap.add_argument("--check", action="store_true")
a = ap.parse_args()
rng = np.random.default_rng(0)
spec = [(5, 2, 3, 24), (5, 2, 24, 32), (5, 2, 32, 64), (3, 1, 64, 64), (3, 1, 64, 64), (3, 1, 64, 128), (3, 1, 128, 128)]
ws, ih, iw, macs = [], a.img_h, a.img_w, 0
for k, s_, cin, cout in spec:
    ih, iw = (ih - k) // s_ + 1, (iw - k) // s_ + 1
    lim = (6.0 / (k * k * (cin + cout))) ** 0.5
    ws += [rng.uniform(-lim, lim, (k, k, cin, cout)).astype("float32"), np.zeros(cout, "float32")]
    macs += ih * iw * cout * k * k * cin
dims = [ih * iw * 128, 100, 50, 25, 2]
for a_, b_ in zip(dims[:-1], dims[1:]):
    lim = (6.0 / (a_ + b_)) ** 0.5
    ws += [rng.uniform(-lim, lim, (a_, b_)).astype("float32"), np.zeros(b_, "float32")]
    macs += a_ * b_

def make(k):
    n = a.envs // k
    envs = [BatchedEnv(n_envs=n, env_id_base=j * n, img_h=a.img_h, img_w=a.img_w, depth=a.depth, auto_reset=True) for j in range(k)]
    for e in envs:
        e.pilot_load(ws)
    return envs

def run(envs, steps, chunk=10):
    # interleave the shards' launches in chunks so that every stream has work queued all the time
    done = 0
    while done < steps:
        c = min(chunk, steps - done)
        for e in envs:
            e.step_pilot(c)
        done += c

res = {}
for k in sorted({1, a.shards}):
    envs = make(k)
    run(envs, a.warmup)
    for e in envs: e.sync()
    t0 = time.perf_counter()
    run(envs, a.steps)
    for e in envs: e.sync()
    wall = time.perf_counter() - t0
    rate = a.envs * a.steps / wall
    res[k] = dict(env_steps_per_s=round(rate, 1), us_per_step=round(wall / a.steps * 1e6, 2), tflops=round(2 * macs * rate / 1e12, 1), frac_of_mfma_peak=round(2 * macs * rate / 2.5e15, 4))
    if a.check:
        res[k]["x"] = np.concatenate([e.fetch("pos_x") for e in envs]); res[k]["img"] = np.concatenate([e.fetch("img") for e in envs])
    for e in envs: e.close()
if a.check and len(res) == 2:
    ks = sorted(res)
    same = bool(np.array_equal(res[ks[0]]["x"], res[ks[1]]["x"]) and np.array_equal(res[ks[0]]["img"], res[ks[1]]["img"]))
    for k in ks: res[k].pop("x"); res[k].pop("img")
    res["identical_to_one_handle"] = same
print(json.dumps({"envs": a.envs, "img": [a.img_h, a.img_w], "depth": a.depth, "steps": a.steps, "shards": res}))
