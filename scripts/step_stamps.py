#!/usr/bin/env python3
"""Phase stamps of trs_step_kernel (a -DTRS_STAMPS=1 build of the library, TRS_HIP_LIB): workgroup 7's raster wave 0 and its first physics wave write s_memtime
(shader clocks on gfx950) at entry (0), staging issued (1), barrier passed (2), physics done (3, physics wave) and the end (5) into stats[8..].  Modes: the closed
loop with the pilot (the env step waits for this step's controls: physics -> raster inside one launch), one launch per step open loop, and a single consumer-paced call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from triton_racer_sim_amd.env import BatchedEnv
import bench

def show(tag, env):
    st = env.fetch("stats").astype(np.int64)
    r, p = st[8:8 + 8], st[32:32 + 8]
    t0 = min(x for x in (r[0], p[0]) if x > 0)
    f = lambda a, i: f"{(a[i] - t0):6d}" if a[i] > 0 else "     -"
    print(f"{tag:34s} raster wave 0: entry {f(r,0)} staged {f(r,1)} barrier {f(r,2)} end {f(r,5)} | physics wave: entry {f(p,0)} staged {f(p,1)} barrier {f(p,2)} physics done {f(p,3)} end {f(p,5)}   [clocks]")

for (n, h, w, depth) in ((1024, 120, 160, False), (512, 240, 320, True)):
    env = BatchedEnv(n_envs=n, img_h=h, img_w=w, depth=depth, auto_reset=True)
    ws, _ = bench.pilot_weights(h, w)
    env.pilot_load(ws)
    env.step_pilot(60); env.sync()
    env.step_pilot(1); env.sync()
    show(f"{n} x {h}x{w} closed loop", env)
    env.step_synthetic(200, 1); env.sync()
    env.step_synthetic(1, 1); env.sync()
    show(f"{n} x {h}x{w} open loop, 1 per launch", env)
    del env
