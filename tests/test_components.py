"""Drop-in parts in the Car loop.  The CPU variants inject the oracle backend to exercise the HOST logic (port
names, Python types, None handling, record writing); the gpu variant runs the same loop on the HIP path and
compares the two frame by frame."""
import json
import os

import numpy as np
import pytest

from conftest import load_golden
from triton_racer_sim_amd.components import GYM_INPUTS, GYM_OUTPUTS, BatchedGymInterface, HipGymInterface, LocationTracker
from triton_racer_sim_amd.core import Car, Component
from triton_racer_sim_amd.recorder import DataStorage


class Driver(Component):
    """Stands in for joystick + multiplexer: emits mux/* and the recorder's usr/* flags."""

    def __init__(self, n_ticks):
        super().__init__(outputs=["mux/steering", "mux/throttle", "mux/breaking", "usr/reset", "usr/del_record", "usr/toggle_record"])
        self.k, self.n = 0, n_ticks

    def step(self, *args):
        self.k += 1
        if self.k > self.n:
            raise KeyboardInterrupt
        return 0.3 * np.sin(self.k / 5.0), 0.6, None if self.k % 2 else 0.0, self.k == 10, False, True


class Probe(Component):
    def __init__(self):
        super().__init__(inputs=GYM_OUTPUTS + ["loc/segment"])
        self.rows = []

    def step(self, *args):
        self.rows.append(args)


def drive(api, tmp_path, n_ticks=25):
    os.makedirs(tmp_path, exist_ok=True)
    cfg = dict(load_golden("config_keys.json")["values"])             # the reference's full default config dict (G6)
    cfg.update(scene_name="generated_track", use_location_tracker=True)
    gym = HipGymInterface(poll_socket_sleep_time=0.01, gym_config=cfg, _api=api)
    tracker = LocationTracker(track_data_path=cfg["track_data_file"], _api=api)
    store, probe = DataStorage(storage_path=str(tmp_path / "records_1")), Probe()
    car = Car(loop_hz=1e9, verbose=False)
    for part in (Driver(n_ticks), gym, tracker, probe, store):         # manage.py:54-108 order: controls, sim, tracker, storage
        car.addComponent(part)
    car.start()
    return gym, probe, tmp_path / "records_1"


def check_run(gym, probe, rec_dir, n_ticks=25):
    assert gym.step_inputs == GYM_INPUTS and gym.step_outputs == GYM_OUTPUTS and gym.threaded is False
    assert gym.getName() == "Gym Interface"
    assert len(probe.rows) == n_ticks
    img, x, y, z, speed, cte, seg = probe.rows[-1]
    assert isinstance(img, np.ndarray) and img.dtype == np.uint8 and img.shape == (120, 160, 3) and img.flags["C_CONTIGUOUS"]
    assert all(type(v) is float for v in (x, y, z, speed, cte, seg))   # Python floats: json.dump needs them
    assert 0.0 <= seg < 10.0 and 0.5 < y < 0.6 and speed > 1.0
    assert probe.rows[3][0] is not probe.rows[4][0]                    # a fresh array per frame, never overwritten
    assert not np.array_equal(probe.rows[3][0], probe.rows[-1][0])
    assert probe.rows[9][4] == 0.0 and probe.rows[8][4] > 0.0          # the 10th tick carried usr/reset: back at the start, v = 0
    rec = json.load(open(rec_dir / "record_5.json"))
    g3 = load_golden("datastorage_record.json")
    assert list(rec) == g3["records"]["record_0.json"]["keys"]
    assert rec["cam/img"] == "img_5.jpg" and rec["mux/break"] is None and os.path.exists(rec_dir / "img_5.jpg")
    assert rec["gym/x"] == probe.rows[5][1] and rec["loc/segment"] == probe.rows[5][6]


def test_car_loop_with_oracle_backend(oracle_api, tmp_path):
    gym, probe, rec_dir = drive(oracle_api, tmp_path)
    check_run(gym, probe, rec_dir)


def test_location_tracker_contract(oracle_api):
    lt = LocationTracker("track_data/generated_track.json", _api=oracle_api)
    g1 = load_golden("locate_generated.json")
    for k in range(0, 400, 13):
        assert lt.step(*g1["queries"][k]) == (g1["segment"][k],)
    with pytest.raises(TypeError):
        lt.step(None, 0.0, 0.0)                                        # the reference raises TypeError on None too
    assert lt.step_inputs == ["gym/x", "gym/y", "gym/z"] and lt.step_outputs == ["loc/segment"]


def test_batched_interface_ports(oracle_api):
    part = BatchedGymInterface(6, to_host=True, _api=oracle_api)
    out = part.step(None, None, None, None)                            # first tick: nothing on the bus yet
    assert len(out) == len(part.step_outputs) == 8 and out[0].shape == (6, 120, 160, 3) and out[6].dtype == np.int32
    out = part.step(np.linspace(-1, 1, 6), 0.5, None, None)
    assert out[4].shape == (6,) and (out[4] > 0).all()


@pytest.mark.gpu
def test_car_loop_on_gpu_equals_oracle(oracle_api, tmp_path):
    gym, probe, rec_dir = drive(None, tmp_path / "gpu")
    check_run(gym, probe, rec_dir)
    _, ref, _ = drive(oracle_api, tmp_path / "cpu")
    for a, b in zip(probe.rows, ref.rows):
        assert np.array_equal(a[0], b[0])                              # frame, byte for byte
        assert a[6] == b[6] and max(abs(p - q) for p, q in zip(a[1:6], b[1:6])) <= 1e-5


def test_sim_latency_is_a_delay_line(oracle_api):
    """sim_latency ms (gyminterface.py:96) -> ceil(ms * loop_hz / 1000) ticks of delay on the returned tuple."""
    from triton_racer_sim_amd.components import HipGymInterface
    now = HipGymInterface(gym_config={"sim_latency": 0}, _api=oracle_api)
    late = HipGymInterface(gym_config={"sim_latency": 120, "loop_hz": 20}, _api=oracle_api)      # 2.4 -> 3 ticks
    assert late.latency_ticks == 3
    seen_now, seen_late = [], []
    for k in range(8):
        args = (0.1 * ((k % 3) - 1), 0.6, None, False)
        seen_now.append(now.step(*args))
        seen_late.append(late.step(*args))
    for k in range(3):
        assert seen_late[k][0] is None and seen_late[k][1:] == (0.0, 0.0, 0.0, 0.0, 0.0)
    for k in range(3, 8):
        assert np.array_equal(seen_late[k][0], seen_now[k - 3][0]) and seen_late[k][1:] == seen_now[k - 3][1:]
    now.onShutdown(); late.onShutdown()


@pytest.mark.gpu
def test_device_resident_part_graph_equals_step_pilot_and_copies_no_frames():
    """pilot -> mux -> sim as SEPARATE parts of the reference's Car loop (car_templates/manage.py:46-75: KerasPilot, joystick,
    ControlMultiplexer, GymInterface, in that order), 256 cars, every inter-part value a device handle: equals the monolithic
    trs_step_pilot tick for tick, and after the first tick (no frame yet: the pilot answers (0, 0, 0) on the host, like
    keras_pilot.py:46-47) the library copies NO byte device -> host."""
    from test_pilot import make_weights
    from triton_racer_sim_amd.components import BatchedControlMultiplexer, BatchedGymInterface, HipKerasPilot
    from triton_racer_sim_amd.core import Car, Component
    from triton_racer_sim_amd.env import BatchedEnv

    class Joystick(Component):                                       # stands in for the joystick part: everybody in AI mode
        def __init__(self):
            Component.__init__(self, inputs=[], outputs=["usr/mode", "usr/steering", "usr/throttle", "usr/breaking", "usr/reset"])

        def step(self, *args):
            return "ai", 0.0, 0.0, 0.0, None

        def getName(self):
            return "Joystick"

    n, ticks = 256, 7
    ws = make_weights(120, 160, seed=21)
    cfg = {"spd_ctl_threshold": 1.1, "spd_ctl_reverse_multiplier": 1.0}
    gym = BatchedGymInterface(n, auto_reset=False, sync=False)
    pilot = HipKerasPilot(cfg, weights=ws, env=gym.env)
    mux = BatchedControlMultiplexer({}, n_cars=n, env=gym.env)
    car = Car(loop_hz=1e9, verbose=False)
    for part in (pilot, Joystick(), mux, gym):
        car.addComponent(part)
    car.tick()                                                        # tick 1: no frame yet
    before = gym.env.counters()
    for _ in range(ticks - 1):
        car.tick()
    after = gym.env.counters()
    assert after[0] == before[0], f"{after[0] - before[0]} bytes were copied device -> host inside the loop"
    assert after[1] - before[1] <= (ticks - 1) * n * 16, "more than modes + joystick values went host -> device"
    assert hasattr(car.pool.get_value("cam/img"), "__cuda_array_interface__") and hasattr(car.pool.get_value("ai/steering"), "__cuda_array_interface__")
    ref = BatchedEnv(n_envs=n, auto_reset=False)
    ref.pilot_load(ws)
    ref.step_pilot(ticks, cfg)
    for name in ("pos_x", "pos_z", "yaw", "speed", "seg_idx", "img"):
        assert np.array_equal(gym.env.fetch(name), ref.fetch(name)), name
    assert gym.env.fetch("speed").max() > 0.0
    ref.close()
    car.stop()


@pytest.mark.gpu
def test_pilot_part_uses_a_host_speed_beside_a_device_frame():
    """KerasPilot.step always uses ITS 'gym/speed' input (keras_pilot.py:78-90).  A pilot part that owns its env (no
    sim behind it) and gets a device 'cam/img' with a HOST speed used to drop the speed silently and read its own (all-zero)
    array; the value is now uploaded, and a missing one is an error."""
    from test_pilot import make_weights
    from triton_racer_sim_amd.components import HipKerasPilot
    from triton_racer_sim_amd.env import BatchedEnv
    n = 8
    ws = make_weights(120, 160, seed=12)
    cfg = {"spd_ctl_threshold": 1.1, "spd_ctl_reverse_multiplier": 1.0}
    sim = BatchedEnv(n_envs=n, auto_reset=True)
    sim.step_synthetic(9, 1)
    frames_dev = sim.device_array("img")                              # (waits for the sim's stream)
    frames = sim.fetch("img")
    speeds = np.array([0.0, 1.5, 3.0, 6.0, 9.0, 12.0, 18.0, 25.0], dtype=np.float32)
    part = HipKerasPilot(cfg, weights=ws, n_cars=n)                   # its own env: no track, no sim
    want = part.step(frames, speeds, None, None, "ai")               # host path: the reference's arithmetic on the host
    import torch
    handles = part.step(frames_dev, speeds, None, None, "ai")        # device frame, host speeds -> three device handles
    part.env.sync()
    got = [torch.as_tensor(g, device="cuda").cpu().numpy() for g in handles]
    assert np.max(np.abs(got[0] - want[0])) <= 1e-6
    assert np.max(np.abs(got[1] - want[1])) <= 1e-4, (got[1], want[1])   # float atan on the device against math.atan
    assert np.ptp(want[1]) > 0.1                                     # the throttles do depend on the speeds given
    with pytest.raises(ValueError, match="gym/speed"):
        part.step(frames_dev, None, None, None, "ai")
    part.onShutdown()
    sim.close()
