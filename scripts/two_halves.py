#!/usr/bin/env python3
"""Measurement only: the closed loop (trs_step_pilot) of ONE handle of N envs against P handles of N/P envs each stepped from P host threads (every
handle launches on its own stream, so the P loops may interleave their kernels on the GPU).  Prints aggregate env-steps/s by wall clock.
usage: two_halves.py [H W depth N]"""
import sys, time, threading
sys.path.insert(0, ".")
import torch
from bench import pilot_weights
from triton_racer_sim_amd.env import BatchedEnv

H, W, depth, N = (int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3])), int(sys.argv[4])) if len(sys.argv) > 4 else (120, 160, False, 1024)
ws, _ = pilot_weights(H, W)
K = 300 if H == 120 else 80

def make(n):
    e = BatchedEnv(n_envs=n, img_h=H, img_w=W, depth=depth, auto_reset=True, device=0)
    e.pilot_load(ws)
    e.step_synthetic(2, 1)
    e.step_pilot(K // 2)
    e.sync()
    return e

def run(parts):
    envs = [make(N // parts) for _ in range(parts)]
    best = 0.0
    for rep in range(3):
        bar = threading.Barrier(parts + 1)
        def work(e):
            bar.wait()
            e.step_pilot(K)
            e.sync()
        th = [threading.Thread(target=work, args=(e,)) for e in envs]
        for t in th: t.start()
        bar.wait()
        t0 = time.perf_counter()
        for t in th: t.join()
        dt = time.perf_counter() - t0
        best = max(best, (N // parts) * parts * K / dt)
    for e in envs: e.close()
    return best

for rnd in range(2):
    for parts in (1, 2, 4, 1, 2):
        r = run(parts)
        print(f"{H}x{W} depth={int(depth)} N={N} handles={parts}: {r/1e6:.3f} M env-steps/s ({N*1e6/r/1:.1f} us per step of all {N})", flush=True)
