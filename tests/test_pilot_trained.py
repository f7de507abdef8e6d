"""The pilot with TRAINED weights (tests/golden/pilot_trained_120x160.npz, made by tests/golden/train_pilot_fixture.py).

Every other pilot test draws Glorot-random weights: their outputs barely depend on the image, and no model a user would load looks
like them (the reference ships no model file, components/keras_pilot.py:26 loads whatever `manage.py train`
produced).  The fixture is the reference's cnn_2d_speed_control architecture (components/keras_train.py:127-174) trained with the
reference's recipe (frames / 255 -> [steering, speed / 20], mean squared error, Adam: keras_train.py:264-299) on records of a scripted
driver in the CPU oracle — PyTorch on the CPU standing in for Keras, which this image does not have.  What the tests pin:
the fixture itself (CPU), the fp16 forward pass against fp32 PyTorch on rendered frames, the fp16 range on a real network, the closed
loop against a loop that shares no code with the product (oracle env + PyTorch pilot + the reference's scalar post-processing), and
that the network DRIVES on the GPU: hundreds of cars for hundreds of ticks, on the road.
"""
import os

import numpy as np
import pytest

from test_pilot import SPEC, pilot_postprocess, torch_layer, torch_pure, torch_tail

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pilot_trained_120x160.npz")
H, W = 120, 160


def trained_weights():
    z = np.load(FIXTURE)
    return [z[f"a{i:02d}"].astype(np.float32) for i in range(22)]


def test_the_fixture_is_the_reference_architecture_and_not_a_random_draw():
    ws = trained_weights()
    ih, iw = H, W
    for i, (k, s, cin, cout) in enumerate(SPEC):
        assert ws[2 * i].shape == (k, k, cin, cout) and ws[2 * i + 1].shape == (cout,)
        ih, iw = (ih - k) // s + 1, (iw - k) // s + 1
    dims = [ih * iw * 128, 100, 50, 25, 2]
    for j, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
        assert ws[14 + 2 * j].shape == (a, b) and ws[15 + 2 * j].shape == (b,)
    assert all(np.isfinite(w).all() for w in ws)
    assert sum(w.size for w in ws) == 834_893                          # the parameter count of the 120x160 model
    # trained: the biases moved away from Keras's zeros, and the output layer's steering column is far from a Glorot draw's scale
    assert max(float(np.abs(ws[2 * i + 1]).max()) for i in range(11)) > 0.01
    assert float(np.abs(ws[21]).max()) > 0.01


def oracle_frames(make_env, n, ticks=40):
    """Frames of cars that have driven a little (scattered over the track, off the centre line)."""
    o = make_env("oracle", n_envs=n, img_h=H, img_w=W, auto_reset=True)
    rng = np.random.default_rng(4)
    o.step(0.0, 0.0, 0.0)
    for _ in range(ticks):
        o.step(rng.uniform(-0.5, 0.5, n).astype(np.float32), 0.5, 0.0)
    return o.fetch("img").copy(), o


@pytest.mark.gpu
def test_trained_forward_matches_fp32_and_stays_inside_fp16(make_env):
    """The product's fp16 pass against fp32 PyTorch (the reference's arithmetic, keras_pilot.py:49-55) on frames the rasteriser made, with a network whose
    outputs depend on the image: |difference| <= 2e-3 on outputs that spread over tenths; no activation saturates (trs_pilot_range_check)."""
    ws = trained_weights()
    n = 96
    frames, _ = oracle_frames(make_env, n)
    env = make_env("hip", n_envs=n, img_h=H, img_w=W)
    env.pilot_load(ws)
    out = env.pilot_forward_host(frames)
    pure = torch_pure(frames, ws)
    err = float(np.max(np.abs(out - pure)))
    print(f"trained pilot: max |HIP fp16 - fp32| = {err:.2e}; steering outputs std {pure[:, 0].std():.3f} range [{pure[:, 0].min():.2f}, {pure[:, 0].max():.2f}], "
          f"speed / 20 outputs mean {pure[:, 1].mean():.3f}")
    assert err <= 2e-3, err
    assert pure[:, 0].std() > 0.05                                      # the steering depends on what the camera sees
    assert 0.1 < pure[:, 1].mean() < 0.6
    assert env.pilot_range_check().sum() == 0
    peak = max(float(env.pilot_layer(i, (n,) + tuple(torch_layer_shape(i))).max()) for i in range(7))
    print(f"largest activation of the seven convolutions: {peak:.1f} (binary16 saturates at 65504)")
    assert 0.5 < peak < 6.0e4


def torch_layer_shape(i):
    ih, iw = H, W
    for k, s, _, cout in SPEC[:i + 1]:
        ih, iw = (ih - k) // s + 1, (iw - k) // s + 1
    return ih, iw, SPEC[i][3]


@pytest.mark.gpu
def test_trained_pilot_drives_in_the_closed_loop(make_env):
    """trs_step_pilot with the trained network: 256 cars x 600 ticks entirely on the device.  The cars reach the speed the network asks for, stay near the
    centre line and leave the road hardly ever (the scripted driver the network imitates loses a car only where the recorded centre line ends:
    its 3.2-unit closure gap) — with Glorot weights the same loop loses every car within tens of ticks."""
    ws = trained_weights()
    n, ticks = 256, 600
    cfg = {"spd_ctl_threshold": 1.1, "spd_ctl_reverse_multiplier": 1.0}
    env = make_env("hip", n_envs=n, img_h=H, img_w=W, auto_reset=True)
    env.pilot_load(ws)
    env.step_pilot(ticks, cfg)
    st = env.fetch("stats")
    speed, cte, ep_len = env.fetch("speed"), env.fetch("cte"), env.fetch("ep_len")
    print(f"trained pilot, {n} cars x {ticks} ticks: off-track events {int(st[0])}, resets {int(st[1])}, speed mean {speed.mean():.2f} (min {speed.min():.2f}), "
          f"mean |cte| {np.abs(cte).mean():.3f}, mean episode length {ep_len.mean():.0f}")
    assert int(st[0]) <= n                                              # less than one event per car in 600 ticks (a random network: thousands)
    assert 3.0 < float(speed.mean()) < 10.0
    assert float(np.abs(cte).mean()) < 0.6
    assert float(ep_len.mean()) > 300


@pytest.mark.gpu
def test_trained_closed_loop_against_oracle_env_plus_torch_pilot(make_env):
    """The closed loop against a loop that shares no code with the product: CPU oracle env + the fp32 PyTorch mirror (fp16-rounded weights and activations,
    any summation order) + KerasPilot's post-processing in scalar Python (keras_pilot.py:78-95) — no output-layer shaping needed here: a trained network
    steers by itself.  40 ticks, 16 cars.  Tolerances from the measured control error (printed), as in tests/test_configs_gpu.py: the two pilots differ by
    ~7e-4 per raw output (fp32 summation order in front of fp16 roundings); the steering is that output, the throttle is atan(2 (22 out - speed)) / (pi / 2) —
    a gain of 28 around zero — so 7.5e-3 on the steering and 2.0e-2 on the throttle were measured over the 40 ticks; asserted: about twice that."""
    ws = trained_weights()
    n, ticks = 16, 40
    cfg = {"spd_ctl_threshold": 1.1, "spd_ctl_reverse_multiplier": 1.0}

    def mirror_pilot(frames):
        x = frames
        for layer in range(8):
            x = torch_layer(layer, x, ws, mirror=True)
        return torch_tail(x, ws)

    o = make_env("oracle", n_envs=n, img_h=H, img_w=W)
    g = make_env("hip", n_envs=n, img_h=H, img_w=W)
    g.pilot_load(ws)
    o.step(0.0, 0.0, 0.0)
    g.step_pilot(1, cfg)                                               # tick 1: no frame yet -> (0, 0, 0)
    assert np.array_equal(g.fetch("img"), o.fetch("img"))
    worst = np.zeros(3)
    steer_seen = []
    for _ in range(ticks - 1):
        out = mirror_pilot(o.fetch("img"))
        spd = o.fetch("speed")
        ctl = np.array([pilot_postprocess(out[i], float(spd[i]), cfg) for i in range(n)], dtype=np.float32)
        o.step(ctl[:, 0], ctl[:, 1], ctl[:, 2])
        g.step_pilot(1, cfg)
        got = np.stack([g.fetch("ctl_steer"), g.fetch("ctl_thr"), g.fetch("ctl_brk")], 1)
        worst = np.maximum(worst, np.abs(got - ctl).max(0))
        steer_seen.append(ctl[:, 0].copy())
    steer_seen = np.array(steer_seen)
    errs = {name: float(np.max(np.abs(g.fetch(name) - o.fetch(name)))) for name in ("pos_x", "pos_z", "yaw", "speed", "cte")}
    print(f"trained closed loop, {ticks} ticks: worst control error (steer, thr, brk) {worst}, state errors {errs}, speed {o.fetch('speed').mean():.2f}, "
          f"|steer| max {np.abs(steer_seen).max():.3f} std {steer_seen.std():.4f}")
    assert steer_seen.std() > 0.02 and o.fetch("speed").min() > 1.0     # the controls matter
    assert worst[0] <= 1.5e-2 and worst[1] <= 4e-2 and worst[2] == 0.0
    assert errs["pos_x"] <= 2e-2 and errs["pos_z"] <= 2e-2 and errs["yaw"] <= 2e-2 and errs["speed"] <= 2e-2 and errs["cte"] <= 2e-2


@pytest.mark.gpu
def test_manage_py_drive_with_the_trained_model(tmp_path):
    """`python manage.py drive --model m.h5` (car_templates/manage.py:46-108) with this package's parts: KerasPilot -> joystick (everybody in AI mode) ->
    ControlMultiplexer -> GymInterface -> LocationTracker -> DataStorage in the reference's Car loop, ONE car, the model loaded from a file as the reference
    does (keras_pilot.py:26; here the .npz form of model.get_weights()).  400 ticks: the car drives a third of the lap on the road, and the tub it records is
    what `manage.py train` reads (img_k.jpg + record_k.json with the reference's keys)."""
    import json

    from conftest import load_golden
    from triton_racer_sim_amd.components import BatchedControlMultiplexer, HipGymInterface, HipKerasPilot, LocationTracker
    from triton_racer_sim_amd.core import Car, Component
    from triton_racer_sim_amd.recorder import DataStorage

    class Joystick(Component):
        def __init__(self, n_ticks):
            Component.__init__(self, inputs=[], outputs=["usr/mode", "usr/steering", "usr/throttle", "usr/breaking", "usr/reset", "usr/del_record", "usr/toggle_record"])
            self.k, self.n = 0, n_ticks

        def step(self, *args):
            self.k += 1
            if self.k > self.n:
                raise KeyboardInterrupt
            return "ai", 0.0, 0.0, 0.0, None, False, True

    class Scalars(Component):                                         # the batched mux answers with arrays of one car; GymInterface takes scalars
        def __init__(self):
            Component.__init__(self, inputs=["mux/steering", "mux/throttle", "mux/breaking"], outputs=["mux/steering", "mux/throttle", "mux/breaking"])

        def step(self, *args):
            return tuple(float(np.asarray(a).reshape(-1)[0]) for a in args)

    class Probe(Component):
        def __init__(self):
            Component.__init__(self, inputs=["gym/cte", "gym/speed", "loc/segment", "ai/steering"])
            self.rows = []

        def step(self, *args):
            self.rows.append(tuple(float(np.asarray(a).reshape(-1)[0]) if a is not None else 0.0 for a in args))

    ticks = 400
    cfg = dict(load_golden("config_keys.json")["values"])             # the reference's default config dict (G6)
    cfg.update(scene_name="generated_track", use_location_tracker=True, spd_ctl_threshold=1.1)
    pilot = HipKerasPilot(cfg, model_path=FIXTURE, model_type="cnn_2d_speed_control")
    gym = HipGymInterface(poll_socket_sleep_time=0.01, gym_config=cfg)
    mux = BatchedControlMultiplexer(cfg, n_cars=1)
    tracker = LocationTracker(track_data_path=cfg["track_data_file"])
    store, probe = DataStorage(storage_path=str(tmp_path / "records_1")), Probe()
    car = Car(loop_hz=1e9, verbose=False)
    for part in (pilot, Joystick(ticks), mux, Scalars(), gym, tracker, probe, store):
        car.addComponent(part)
    car.start()
    rows = np.array(probe.rows)
    seg0, seg1 = 1185, int(round(rows[-1, 2] / 10.0 * 1185))          # loc/segment = index / points x 10 (track_data_process.py:103-107); the parts are shut down by now
    print(f"manage.py drive, trained model, {ticks} ticks: final track point {seg1} of {seg0}, speed {rows[-1, 1]:.2f}, max |cte| {np.abs(rows[:, 0]).max():.3f}, "
          f"|ai/steering| max {np.abs(rows[:, 3]).max():.3f}, loc/segment {rows[-1, 2]:.2f}")
    assert len(rows) == ticks
    assert np.abs(rows[20:, 0]).max() < 1.0                            # on the road all the way (off-track threshold: include/trsim_spec.h)
    assert rows[-1, 1] > 3.0                                           # at the speed the network asks for
    assert seg1 > 300                                                  # a third of the 1185 track points behind it
    assert np.abs(rows[:, 3]).max() > 0.1                              # it steered
    rec = json.load(open(tmp_path / "records_1" / "record_300.json"))
    assert list(rec) == load_golden("datastorage_record.json")["records"]["record_0.json"]["keys"]
    assert os.path.exists(tmp_path / "records_1" / "img_300.jpg") and rec["gym/speed"] > 3.0
