"""Race screen for the step kernel beside other kernels (GPU; run by hand, not a test — the fault it hunts is statistical).

Handle A runs the pilot loop (`trs_step_pilot`), handle B has no pilot kernels in its stream and is stepped with the
controls A's pilot produced, a third handle keeps a pilot loop running on its own stream from a thread.  Same floats in,
so A's and B's state and frames must be bit-identical at every step.

History: with the ray step of a row computed by an in-place `v_pk_mul_f32 ... op_sel:[0,1]` this screen found 1-3 % of the
steps with a few wrong pixels (lanes 48..63 of one wave, one row pass: dx = +-0), only while workgroups of other kernels
shared the CU; never alone, never with LDS poisoned (`trs_debug_poison_lds`), never with the scalar form that ships
(`ray_step` in csrc/trsim_hip.hip).  TRS_HIP_LIB=<another build> screens a variant, e.g. one built with -DTRS_PACKED_RAY_STEP.

    python scripts/coresidency_probe.py [steps]      ->  prints the number of differing frames (expect 0)
"""
import os
import sys
import threading

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np

from test_pilot import make_weights
from triton_racer_sim_amd.env import BatchedEnv

N = int(os.environ.get("PROBE_N", "96"))
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
WEIGHTS = make_weights(120, 160, seed=9)


def fresh(pilot=False):
    e = BatchedEnv(n_envs=N, auto_reset=True)
    if pilot:
        e.pilot_load(WEIGHTS)
    e.step_synthetic(5, 1)
    e.sync()
    return e


def main():
    a, b, g = fresh(pilot=True), fresh(), fresh(pilot=True)
    stop = threading.Event()

    def aggressor():
        while not stop.is_set():
            g.step_pilot(8)
            g.sync()

    t = threading.Thread(target=aggressor)
    t.start()
    bad = 0
    try:
        for i in range(STEPS):
            a.step_pilot(1)
            ia = a.fetch("img")
            b.step(a.fetch("ctl_steer"), a.fetch("ctl_thr"), a.fetch("ctl_brk"))
            ib = b.fetch("img")
            if not all(np.array_equal(a.fetch(f), b.fetch(f)) for f in ("pos_x", "pos_z", "yaw", "speed")):
                print(f"step {i}: states differ; stopping")
                bad += 1
                break
            if not np.array_equal(ia, ib):
                bad += 1
                d = np.argwhere((ia != ib).any(axis=3))
                for ee, rr in sorted(set(map(tuple, d[:, :2].tolist())))[:4]:
                    cc = d[(d[:, 0] == ee) & (d[:, 1] == rr)][:, 2]
                    tids = sorted(set(((rr % 12) * 40 + cc // 4).tolist()))            # 120x160: 12 rows x 40 column groups per pass
                    print(f"step {i}: env {ee} row {rr} cols {cc.min()}..{cc.max()} differ (raster threads {tids[0]}..{tids[-1]}, lanes {tids[0] % 64}..{tids[-1] % 64})")
    finally:
        stop.set()
        t.join()
        for e in (a, b, g):
            e.close()
    print(f"frames that differ: {bad} of {STEPS}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
