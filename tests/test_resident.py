"""Resident step mode (``trs_set_step_mode(TRS_STEP_RESIDENT)``): a worker kernel stays on the GPU and every ``trs_step``
only posts its controls — the consumer-paced call of the reference's drive loop (``core/car.py:45-53``: one
``component.step`` per tick with that tick's values; ``components/gyminterface.py:66-76``) without a launch per step.

The results must not depend on HOW the steps reach the GPU: every test compares the resident path with the CPU oracle
(images, indices and flags bit-exact; pose / speed within 1e-5), through the C ABI.
"""
import time

import numpy as np
import pytest

from test_gpu_parity import assert_state_equal

pytestmark = pytest.mark.gpu


def assert_frames_equal(g, o, where=""):
    a, b = g.fetch("img"), o.fetch("img")
    assert np.array_equal(a, b), f"image differs {where}: {np.flatnonzero((a != b).reshape(a.shape[0], -1).any(axis=1))[:8]}"


def controls(rng, n):
    return (rng.uniform(-1, 1, n).astype(np.float32), rng.uniform(-0.2, 1, n).astype(np.float32),
            (rng.uniform(0, 1, n) * (rng.uniform(0, 1, n) < 0.1)).astype(np.float32))


def test_resident_synthetic_and_host_controls_equal_the_oracle(make_env):
    n = 64
    g, o = make_env("hip", n_envs=n, auto_reset=True), make_env("oracle", n_envs=n, auto_reset=True)
    g.set_step_mode(True)
    rng = np.random.default_rng(1)
    for env in (g, o):
        env.step_synthetic(11, 1)
    assert_state_equal(g, o, "after 11 synthetic steps")
    assert_frames_equal(g, o, "after 11 synthetic steps")
    for k in range(9):                                           # more posts than ring slots
        st, th, br = controls(rng, n)
        rs = (rng.uniform(0, 1, n) < 0.05) if k == 4 else None
        for env in (g, o):
            env.step(st, th, br, reset=rs)
    assert_state_equal(g, o, "after host-controlled steps")
    assert_frames_equal(g, o, "after host-controlled steps")
    assert g.fetch("stats")[2] == 0


def test_resident_1024_envs_many_steps_in_flight(make_env):
    """BASELINE configs[2] shape: 1024 envs (4 per workgroup, one workgroup per CU), posts queued 8 deep."""
    n = 1024
    g, o = make_env("hip", n_envs=n, auto_reset=True), make_env("oracle", n_envs=n, auto_reset=True)
    g.set_step_mode(True)
    for env in (g, o):
        env.step_synthetic(48, 1)
    assert_state_equal(g, o, "1024 envs x 48")
    assert_frames_equal(g, o, "1024 envs x 48")
    for env in (g, o):
        env.step_synthetic(1, 1)                                 # odd step count: the other frame buffer
    assert_frames_equal(g, o, "1024 envs x 49")


def test_resident_device_controls_rewritten_in_place(make_env):
    """Controls produced on the device by another stream into the SAME arrays every step (what a policy does): no step may
    see the values an earlier step read from those addresses."""
    torch = pytest.importorskip("torch")
    n = 512
    g, o = make_env("hip", n_envs=n, auto_reset=True), make_env("oracle", n_envs=n, auto_reset=True)
    g.set_step_mode(True)
    rng = np.random.default_rng(2)
    d_st, d_th, d_br = (torch.zeros(n, device="cuda") for _ in range(3))
    for k in range(40):
        st, th, br = controls(rng, n)
        d_st.copy_(torch.from_numpy(st)); d_th.copy_(torch.from_numpy(th)); d_br.copy_(torch.from_numpy(br))
        d_st.mul_(1.0)                                           # a kernel of torch's stream is the last writer
        torch.cuda.synchronize()
        g.step_device(d_st.data_ptr(), d_th.data_ptr(), d_br.data_ptr())
        g.sync()                                                 # the step has consumed them before they are rewritten
        o.step(st, th, br)
        if k % 13 == 12:
            assert_state_equal(g, o, f"device controls, step {k}")
    assert_state_equal(g, o, "device controls")
    assert_frames_equal(g, o, "device controls")
    frames = torch.as_tensor(g.device_array("img"), device="cuda")         # a consumer kernel launched after trs_sync sees the frame
    assert np.array_equal(frames.cpu().numpy(), o.fetch("img"))


def test_resident_tick_loop_with_fetch_every_step(make_env):
    """The N = 1 Car-loop pattern: step, take the whole output tuple, step ... with the worker resident throughout."""
    g, o = make_env("hip", n_envs=1), make_env("oracle", n_envs=1)
    g.set_step_mode(True)
    rng = np.random.default_rng(3)
    for k in range(60):
        st, th = float(rng.uniform(-1, 1)), float(rng.uniform(0, 1))
        for env in (g, o):
            env.step(st, th, 0.0, reset=(k == 30))
        a, b = g.fetch_outputs(), o.fetch_outputs()
        assert np.array_equal(a[0], b[0]), f"frame, tick {k}"
        for x, y in zip(a[1:6], b[1:6]):
            assert np.max(np.abs(x - y)) <= 1e-5
        assert np.array_equal(a[6], b[6]) and np.array_equal(a[7], b[7])


def test_resident_mixed_with_launches_resets_and_idle_exit(make_env):
    n = 96
    g, o = make_env("hip", n_envs=n, auto_reset=True), make_env("oracle", n_envs=n, auto_reset=True)
    g.set_step_mode(True, idle_us=300)
    rng = np.random.default_rng(4)
    for env in (g, o):
        env.step_synthetic(5, 1)
    time.sleep(0.05)                                             # the worker leaves by itself (idle) ...
    for env in (g, o):
        env.step_synthetic(3, 1)                                 # ... and the next post starts it again
    assert_state_equal(g, o, "after the idle exit")
    mask = rng.uniform(0, 1, n) < 0.3
    for env in (g, o):
        env.reset(mask)                                          # needs the stream: the worker is asked to leave first
        env.step_synthetic(4, 1)
    assert_state_equal(g, o, "after a reset between resident steps")
    g.set_step_mode(False)
    for env in (g, o):
        env.step_synthetic(6, 2)                                 # launches (pipelined) on the state the worker left
    g.set_step_mode(True)
    st, th, br = controls(rng, n)
    for env in (g, o):
        env.step(st, th, br, n_steps=3)                          # held controls: three posts
        env.step_synthetic(2, 1)
    assert_state_equal(g, o, "resident -> launches -> resident")
    assert_frames_equal(g, o, "resident -> launches -> resident")


@pytest.mark.parametrize("n,h,w,depth", [(2048, 120, 160, False), (24, 240, 320, True), (300, 60, 80, True)])
def test_resident_shapes(make_env, n, h, w, depth):
    """Several envs per workgroup (state in LDS, step-major physics), the depth frame, ragged last workgroup."""
    kw = dict(n_envs=n, img_h=h, img_w=w, depth=depth, auto_reset=True)
    g, o = make_env("hip", **kw), make_env("oracle", **kw)
    g.set_step_mode(True)
    for env in (g, o):
        env.step_synthetic(13, 1)
    assert_state_equal(g, o, f"{n} envs {h}x{w}")
    assert_frames_equal(g, o, f"{n} envs {h}x{w}")
    if depth:
        assert np.array_equal(g.fetch("depth").view(np.uint32), o.fetch("depth").view(np.uint32))


def test_resident_sequence_and_mode_errors(make_env, hip_api):
    n = 32
    g, o = make_env("hip", n_envs=n, auto_reset=True), make_env("oracle", n_envs=n, auto_reset=True)
    g.set_step_mode(True)
    rng = np.random.default_rng(5)
    k = 11
    st = rng.uniform(-1, 1, (k, n)).astype(np.float32)
    th = rng.uniform(0, 1, (k, n)).astype(np.float32)
    for env in (g, o):
        env.step_sequence(st, th, steps_per_launch=4)
    assert_state_equal(g, o, "sequence through the worker")
    assert_frames_equal(g, o, "sequence through the worker")
    assert hip_api.set_step_mode(g._h, 7, 0) != 0                # an unknown mode is refused (physics-only handles have their own worker since round 4)


def test_resident_worker_generations_under_load(make_env):
    """The worker leaves on its own when its lifetime is spent, even while posts keep coming; the host restarts it from the
    first step it did not take.  With a 1.5 ms lifetime a 4,000-step run crosses that seam dozens of times."""
    n = 128
    g, o = make_env("hip_hooks", n_envs=n, auto_reset=True), make_env("oracle", n_envs=n, auto_reset=True)   # (the test build of the library: csrc/libtrsim_testhooks.so)
    g.set_step_mode(True)
    g.resident_lifetime(1500)                                    # trs_resident_debug_lifetime
    for chunk in (1500, 1, 2499):
        for env in (g, o):
            env.step_synthetic(chunk, 1)
        assert_state_equal(g, o, f"after a chunk of {chunk}")
        assert_frames_equal(g, o, f"after a chunk of {chunk}")
    assert int(g.fetch("stats")[2]) == 0


def test_resident_after_reloading_the_track(make_env):
    """trs_load_track restarts the step counter: tags and completion flags of the earlier steps must not be taken for new ones."""
    from conftest import track_points
    n = 40
    g, o = make_env("hip", n_envs=n, auto_reset=True), make_env("oracle", n_envs=n, auto_reset=True)
    g.set_step_mode(True)
    for env in (g, o):
        env.step_synthetic(21, 1)
        env.load_track(track_points("mountain"))
    for env in (g, o):
        env.step_synthetic(10, 1)
    assert_state_equal(g, o, "after a track reload")
    assert_frames_equal(g, o, "after a track reload")


def test_pilot_loop_with_resident_mode_selected(make_env):
    """trs_step_pilot steps by launch whatever the mode (its convolutions need the CUs' LDS a worker would hold): the same loop,
    with resident steps before and after it."""
    from test_pilot import make_weights
    n = 24
    ws = make_weights(120, 160, seed=31)
    a, b = make_env("hip", n_envs=n), make_env("hip", n_envs=n)
    a.pilot_load(ws); b.pilot_load(ws)
    a.set_step_mode(True)
    for env in (a, b):
        env.step_synthetic(3, 1)                                     # a: posted to the worker; b: launches
        env.step_pilot(4)
        env.step_synthetic(2, 1)
    for name in ("pos_x", "pos_z", "yaw", "speed", "seg_idx", "img", "ep_return"):
        assert np.array_equal(a.fetch(name), b.fetch(name)), name


def test_pilot_steps_then_fetch_with_resident_mode_selected(make_env):
    """in resident mode trs_step_pilot steps by LAUNCH, so the newest steps have no completion flag — trs_sync,
    trs_copy_to_host and trs_fetch_outputs called right behind them must wait for the stream (they used to spin on a flag nobody
    writes until the 10 s timeout).  Compared with a launch-mode env."""
    from test_pilot import make_weights
    n = 24
    ws = make_weights(120, 160, seed=33)
    a, b = make_env("hip", n_envs=n), make_env("hip", n_envs=n)
    a.pilot_load(ws); b.pilot_load(ws)
    a.set_step_mode(True)
    for env in (a, b):
        env.step_synthetic(3, 1)                                     # a: posted to the worker
        env.step_pilot(5)                                            # a: launched although resident mode is selected
    t0 = time.perf_counter()
    a.sync()                                                         # straight behind the launched steps
    got = {name: a.fetch(name) for name in ("pos_x", "pos_z", "yaw", "speed", "seg_idx", "img", "ep_return")}
    outs = a.fetch_outputs() if hasattr(a, "fetch_outputs") else None
    assert time.perf_counter() - t0 < 5.0                            # no completion-flag timeout
    for name, v in got.items():
        assert np.array_equal(v, b.fetch(name)), name
    if outs is not None:
        assert np.array_equal(outs[0], b.fetch("img"))
    for env in (a, b):                                               # and the worker starts again behind the launched steps
        env.step_synthetic(2, 1)
        env.step_pilot(1)
        env.step_synthetic(1, 1)
    for name in ("pos_x", "speed", "seg_idx", "img"):
        assert np.array_equal(a.fetch(name), b.fetch(name)), name


def test_quiesce_keeps_the_mode_and_the_results(make_env):
    """trs_quiesce: the worker leaves (posted steps complete first), the mode stays resident, the next step starts a new worker."""
    n = 32
    g, o = make_env("hip", n_envs=n, auto_reset=True), make_env("oracle", n_envs=n, auto_reset=True)
    g.quiesce()                                                      # no-op before resident mode was ever selected
    g.set_step_mode(True)
    for k in (5, 1, 7):
        for env in (g, o):
            env.step_synthetic(k, 1)
        g.quiesce()
        g.quiesce()
    assert_state_equal(g, o, "after quiesce calls")
    assert_frames_equal(g, o, "after quiesce calls")


@pytest.mark.parametrize("resident", [True, False])
def test_step_wait_is_step_then_sync(make_env, resident):
    """``trs_step_wait`` (the lock-step tick in one FFI crossing): when it returns, the frame and the telemetry of that tick are
    complete in memory for a consumer on ANOTHER stream — in both step modes, same values as the oracle."""
    torch = pytest.importorskip("torch")
    n = 256
    g, o = make_env("hip", n_envs=n, auto_reset=True), make_env("oracle", n_envs=n, auto_reset=True)
    g.set_step_mode(resident)
    rng = np.random.default_rng(5)
    d_st, d_th, d_br = (torch.zeros(n, device="cuda") for _ in range(3))
    for k in range(12):
        st, th, br = controls(rng, n)
        d_st.copy_(torch.from_numpy(st)); d_th.copy_(torch.from_numpy(th)); d_br.copy_(torch.from_numpy(br))
        torch.cuda.synchronize()
        g.step_device_wait(d_st.data_ptr(), d_th.data_ptr(), d_br.data_ptr())
        o.step(st, th, br)
        frames = torch.as_tensor(g.device_array("img", sync=False), device="cuda")   # read by torch's stream right away: no trs_sync in between
        assert np.array_equal(frames.cpu().numpy(), o.fetch("img")), f"tick {k}"
    assert_state_equal(g, o, "step_wait ticks")
    with pytest.raises(RuntimeError):
        g.step_device_wait(0, 0)                                 # trs_step's errors come through


# ---- physics-only envs (cfg.render == 0, BASELINE configs[1]): trs_physics_worker_kernel, round 4 ----

def test_resident_physics_only_equals_the_oracle(make_env):
    """The test at the top of this file with ``render=False``: synthetic controls, then host controls with more posts than ring slots,
    a user reset in the middle; every state field against the oracle (indices and flags exact, pose within 1e-5)."""
    n = 64
    g, o = make_env("hip", n_envs=n, render=False, auto_reset=True), make_env("oracle", n_envs=n, render=False, auto_reset=True)
    g.set_step_mode(True)
    rng = np.random.default_rng(11)
    for env in (g, o):
        env.step_synthetic(11, 1)
    assert_state_equal(g, o, "physics only, after 11 synthetic steps")
    for k in range(19):
        st, th, br = controls(rng, n)
        rs = (rng.uniform(0, 1, n) < 0.05) if k == 4 else None
        for env in (g, o):
            env.step(st, th, br, reset=rs)
    assert_state_equal(g, o, "physics only, after host-controlled steps")
    assert np.array_equal(g.fetch("last_return"), o.fetch("last_return"))
    assert g.fetch("stats")[2] == 0


def test_resident_physics_256_envs_configs1_many_steps(make_env):
    """BASELINE configs[1]: 256 envs, physics only; 1000 posted steps (queued 8 deep), a ragged last workgroup (n % 4 != 0) on the way,
    then lock step with device controls rewritten in place."""
    torch = pytest.importorskip("torch")
    for n, steps in ((256, 1000), (101, 60)):
        g, o = make_env("hip", n_envs=n, render=False, auto_reset=True), make_env("oracle", n_envs=n, render=False, auto_reset=True)
        g.set_step_mode(True)
        for env in (g, o):
            env.step_synthetic(steps, 1)
        assert_state_equal(g, o, f"physics only, {n} envs x {steps}")
        rng = np.random.default_rng(12)
        d_st, d_th = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        for k in range(30):
            st, th, _ = controls(rng, n)
            d_st.copy_(torch.from_numpy(st)); d_th.copy_(torch.from_numpy(th))
            torch.cuda.synchronize()
            g.step_device_wait(d_st.data_ptr(), d_th.data_ptr())      # post, wait for the telemetry: the arrays may be rewritten afterwards
            o.step(st, th)
        assert_state_equal(g, o, f"physics only, {n} envs, lock step")


def test_resident_physics_worker_generations_idle_exit_and_launch_mix(make_env):
    """Workers that leave (idle, a short lifetime under load) and are restarted by the next post; launches and resets in between."""
    n = 96
    g, o = make_env("hip_hooks", n_envs=n, render=False, auto_reset=True), make_env("oracle", n_envs=n, render=False, auto_reset=True)
    g.set_step_mode(True, idle_us=300)
    for env in (g, o):
        env.step_synthetic(5, 1)
    time.sleep(0.05)                                             # the worker leaves by itself (idle) ...
    for env in (g, o):
        env.step_synthetic(7, 1)                                 # ... and the next post starts a new one
    assert_state_equal(g, o, "physics only, after an idle exit")
    mask = np.zeros(n, np.uint8); mask[::7] = 1
    for env in (g, o):
        env.reset(mask)                                          # asks the worker to leave (the reset needs the stream)
        env.step_synthetic(3, 1)
    g.set_step_mode(False)
    for env in (g, o):
        env.step_synthetic(16, 8)                                # launches
    g.set_step_mode(True)
    g.resident_lifetime(400)                                     # many worker generations under load
    for env in (g, o):
        env.step_synthetic(600, 1)
    assert_state_equal(g, o, "physics only, generations under load")
    assert g.fetch("stats")[2] == 0


def test_resident_abort_makes_every_wave_leave_and_the_host_gets_an_error(make_env):
    """every spin of the worker is bounded and the abort bit makes every wave leave — also the raster waves of the
    dynamic-brightness instantiation inside their team barriers (raster_dyn_batch).  The abort is injected from outside while steps
    keep coming (trs_resident_debug_abort); the host must get TRS_ERR_DEVICE within seconds instead of waiting for ever, the handle says
    so on every later step, and loading the track again makes it usable."""
    for kw in (dict(filter=True), dict(filter=False), dict(filter=False, render=False)):
        n = 1024 if kw.get("render", True) else 256
        g = make_env("hip_hooks", n_envs=n, auto_reset=True, render=kw.get("render", True))
        if kw["filter"]:
            g.set_frame_filter({"preprocessing_dynamic_brightness_enabled": True, "preprocessing_contrast_enhancement_ratio": 1.2})
        g.set_step_mode(True, idle_us=100000)
        g.step_synthetic(40, 1)
        g.sync()
        g.step_synthetic(6, 1)                                    # steps in flight when the abort lands
        g.resident_abort()
        t0 = time.perf_counter()
        with pytest.raises(RuntimeError, match="resident worker gave up|injected"):
            for _ in range(50):
                g.step_synthetic(8, 1)
                g.sync()
        assert time.perf_counter() - t0 < 8.0, "the worker did not leave promptly after the abort"
        with pytest.raises(RuntimeError, match="gave up earlier"):
            g.step_synthetic(1, 1)
        g.load_track("generated_track")                           # every env on a defined state again
        o = make_env("oracle", n_envs=n, auto_reset=True, render=kw.get("render", True))
        if kw["filter"]:
            o.set_frame_filter({"preprocessing_dynamic_brightness_enabled": True, "preprocessing_contrast_enhancement_ratio": 1.2})
        for env in (g, o):
            env.step_synthetic(9, 1)
        assert_state_equal(g, o, f"after an injected abort and a track reload {kw}")
        if kw.get("render", True):
            assert_frames_equal(g, o, f"after an injected abort and a track reload {kw}")
