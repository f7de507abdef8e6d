"""ControlMultiplexer for N cars (reference components/controlmultiplexer.py:24-70).
The reference file cannot be imported here (pygame), and the reference holds no test for it, so parity is UNPINNED
by reference outputs: the C oracle is checked against an independent event-queue model (oracle/pyref.py), hand-derived
cases from the source, and the HIP kernel against the oracle, bit for bit."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyref  # noqa: E402

from triton_racer_sim_amd.components import BatchedControlMultiplexer, MUX_INPUTS, MUX_OUTPUTS  # noqa: E402

LOCKS = {"ai_launch_boost_throttle_enabled": True, "ai_launch_boost_throttle_value": 0.9, "ai_launch_boost_throttle_duration": 0.5,
         "ai_launch_lock_steering_enabled": True, "ai_launch_lock_steering_value": -0.25, "ai_launch_lock_steering_duration": 0.3}


def mode_script(n, ticks, seed, hold=(1, 9)):
    """Per car a piecewise-constant random mode sequence (with a few out-of-range codes)."""
    rng = np.random.default_rng(seed)
    out = np.zeros((ticks, n), np.uint8)
    for i in range(n):
        t = 0
        while t < ticks:
            m = rng.choice([0, 1, 2, 2, 2, 7], p=[0.25, 0.2, 0.2, 0.15, 0.15, 0.05])
            d = int(rng.integers(hold[0], hold[1]))
            out[t:t + d, i] = m
            t += d
    return out


def run(env, cfg, script, seed):
    rng = np.random.default_rng(seed)
    ticks, n = script.shape
    keep = (np.full(n, 0.5, np.float32), np.full(n, -0.5, np.float32), np.full(n, 0.125, np.float32))
    rows, ins = [], []
    for t in range(ticks):
        vals = rng.uniform(-1, 1, (6, n)).astype(np.float32)
        keep = env.control_mux_host(script[t], vals[:3], vals[3:], keep=keep, cfg=cfg)
        rows.append(np.stack(keep))
        ins.append(vals)
    return np.stack(rows), np.stack(ins)


def model_run(cfg, script, ins, hz=20):
    import math
    ticks, n = script.shape
    thr = (cfg.get("ai_launch_boost_throttle_enabled", False), np.float32(cfg.get("ai_launch_boost_throttle_value", 1.0)),
           max(1, math.ceil(cfg.get("ai_launch_boost_throttle_duration", 5) * hz)))
    st = (cfg.get("ai_launch_lock_steering_enabled", False), np.float32(cfg.get("ai_launch_lock_steering_value", 0.0)),
          max(1, math.ceil(cfg.get("ai_launch_lock_steering_duration", 3) * hz)))
    cars = [pyref.MuxModel(thr, st) for _ in range(n)]
    keep = [(np.float32(0.5), np.float32(-0.5), np.float32(0.125))] * n
    out = np.zeros((ticks, 3, n), np.float32)
    for t in range(ticks):
        for i, car in enumerate(cars):
            keep[i] = car.step(int(script[t, i]), tuple(ins[t, :3, i]), tuple(ins[t, 3:, i]), keep[i])
            out[t, :, i] = keep[i]
    return out


@pytest.mark.parametrize("cfg", [{}, LOCKS, dict(LOCKS, ai_launch_lock_steering_enabled=False)])
def test_oracle_equals_event_model(make_env, cfg):
    """hold >= 3 ticks keeps at most 4 lock ends pending per car (10-tick lock), inside the 8 the C state tracks."""
    env = make_env("oracle", n_envs=96, track=None, render=False)
    script = mode_script(96, 300, seed=4, hold=(3, 12))
    got, ins = run(env, cfg, script, seed=5)
    assert np.array_equal(got, model_run(cfg, script, ins))


def test_source_derived_cases(make_env):
    env = make_env("oracle", n_envs=1, track=None, render=False)
    usr, ai = (0.1, 0.2, 0.3), (-0.4, -0.5, -0.6)
    f = lambda m, cfg=None, keep=None: tuple(float(a[0]) for a in env.control_mux_host([m], usr, ai, keep=keep, cfg=cfg))
    near = lambda a, b: np.allclose(a, b, atol=1e-7)
    assert near(f("human"), usr)                                            # controlmultiplexer.py:26-27
    assert near(f("ai_steering"), (ai[0], usr[1], usr[2]))                  # :28-29
    assert near(f("ai"), ai)                                                # :30-31 (locks disabled by default, config.py:57,61)
    assert near(f("something else", keep=(9.0, 8.0, 7.0)), (9.0, 8.0, 7.0))  # empty tuple -> the pool keeps its values
    # launch locks: 0.5 s / 0.3 s at 20 Hz = 10 / 6 ticks, counted from the tick of the transition
    env.control_mux_reset()
    assert near(f("ai_steering", LOCKS), (ai[0], usr[1], usr[2]))
    seen = [f("ai", LOCKS) for _ in range(12)]
    for k in range(12):
        want = (-0.25 if k < 6 else ai[0], 0.9 if k < 10 else ai[1], ai[2])
        assert near(seen[k], want), k
    # already in AI: no new launch (:33 needs last_mode != AI); human -> ai triggers again and overrides a HUMAN tick too
    assert near(f("ai", LOCKS), ai)
    assert near(f("human", LOCKS), usr)
    assert near(f("ai", LOCKS), (-0.25, 0.9, ai[2]))
    assert near(f("human", LOCKS), (-0.25, 0.9, usr[2]))                    # :37-40 apply in every mode while a lock is active
    # a re-trigger while the first end is pending: the OLDER sleep ends the lock (each trigger has its own thread)
    env.control_mux_reset()
    only_thr = dict(LOCKS, ai_launch_lock_steering_enabled=False)
    seq = ["ai"] * 4 + ["human"] + ["ai"] * 10          # triggers at ticks 0 and 5; ends at 10 and 15
    thr = [f(m, only_thr)[1] for m in seq]
    assert near(thr[:4], [0.9] * 4) and near(thr[4], 0.9) and near(thr[5:10], [0.9] * 5)
    assert near(thr[10:], [ai[1]] * 5)                  # tick 10: the first thread clears the flag although the 2nd lock began at 5


def test_errors(make_env):
    env = make_env("oracle", n_envs=4, track=None, render=False)
    bad = env.mux_config(dict(LOCKS))
    bad.throttle_lock_ticks = 0
    with pytest.raises(RuntimeError, match="lock ticks"):
        env.control_mux_host([0] * 4, (0, 0, 0), (0, 0, 0), cfg=bad)
    with pytest.raises(RuntimeError, match="n_envs"):
        env.control_mux_host([0] * 5, (0, 0, 0), (0, 0, 0))


def test_component_ports_and_tick(oracle_api):
    part = BatchedControlMultiplexer(LOCKS, n_cars=3, _api=oracle_api)
    assert part.step_inputs == MUX_INPUTS and part.step_outputs == MUX_OUTPUTS and part.getName() == "Control Multiplexer"
    out = part.step(["human", "ai_steering", "ai"], [0.1, 0.2, 0.3], 0.5, None, [-0.1, -0.2, -0.3], -0.5, 0.25)
    assert np.allclose(out[0], [0.1, -0.2, -0.25]) and np.allclose(out[1], [0.5, 0.5, 0.9]) and np.allclose(out[2], [0.0, 0.0, 0.25])
    out = part.step(None, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)                       # unknown mode: values stay
    assert np.allclose(out[0], [0.1, -0.2, -0.25])
    part.onShutdown()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [{}, LOCKS])
def test_gpu_equals_oracle(make_env, cfg):
    """Includes rapid toggling (hold 1..2 ticks) so that more than 8 lock ends are pending: the bounded state must
    behave identically on both sides."""
    n = 3000
    g = make_env("hip", n_envs=n, track=None, render=False)
    o = make_env("oracle", n_envs=n, track=None, render=False)
    for hold in ((1, 3), (3, 12)):
        for env in (g, o):
            env.control_mux_reset()
        script = mode_script(n, 120, seed=11, hold=hold)
        got, _ = run(g, cfg, script, seed=12)
        want, _ = run(o, cfg, script, seed=12)
        assert np.array_equal(got, want), hold


@pytest.mark.gpu
def test_gpu_mux_feeds_step_on_device(make_env):
    """Closed loop on the device: mux outputs (device arrays) are what trs_step consumes."""
    import ctypes as C
    torch = pytest.importorskip("torch")
    n = 256
    g = make_env("hip", n_envs=n, auto_reset=True)
    o = make_env("oracle", n_envs=n, auto_reset=True)
    rng = np.random.default_rng(3)
    script = mode_script(n, 12, seed=2, hold=(2, 5))
    script[script > 2] = 0
    mc = g.mux_config(LOCKS)
    for t in range(12):
        vals = rng.uniform(-1, 1, (6, n)).astype(np.float32)
        vals[[2, 5]] = np.abs(vals[[2, 5]]) * 0.2
        d = [torch.as_tensor(v, device="cuda") for v in vals]
        dm = torch.as_tensor(script[t], device="cuda")
        outs = [torch.zeros(n, device="cuda") for _ in range(3)]
        torch.cuda.synchronize()                                              # torch's stream is not the handle's stream
        g.api.check(g.api.control_mux(g._h, C.byref(mc), dm.data_ptr(), *[x.data_ptr() for x in d], *[x.data_ptr() for x in outs], n), "control_mux")
        g.step_device(outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr())
        g.sync()                                                              # before torch may recycle these tensors
        mux = o.control_mux_host(script[t], vals[:3], vals[3:], cfg=LOCKS)
        o.step(*mux)
    for name in ("seg_idx", "done"):
        assert np.array_equal(g.fetch(name), o.fetch(name))
    for name in ("pos_x", "pos_z", "speed", "cte"):
        assert np.max(np.abs(g.fetch(name) - o.fetch(name))) <= 1e-5
    assert np.array_equal(g.fetch("img"), o.fetch("img"))
