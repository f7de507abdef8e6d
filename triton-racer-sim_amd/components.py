"""Drop-in parts for the reference's ``Car`` loop, backed by the HIP env.

``HipGymInterface`` replaces ``GymInterface`` (reference ``components/gyminterface.py:47-104``): same
constructor keywords, same port names, same return order and Python types — but the external Unity
simulator, the TCP/JSON/JPEG round trip and the two 1 s start-up sleeps are gone: physics and camera run
in-process on the GPU.  ``LocationTracker`` replaces ``components/track_data_process.py:68-107`` with the
exact same integer index, computed by the HIP nearest-point kernel.

Swap in ``car_templates/manage.py:72-75``::

    from triton_racer_sim_amd.components import HipGymInterface as GymInterface
"""
import numpy as np

from .core import Component
from .env import SLOT_MODE, SLOT_PILOT_IN, SLOT_USR, BatchedEnv, device_ptr, is_device_array

GYM_INPUTS = ["mux/steering", "mux/throttle", "mux/breaking", "usr/reset"]       # gyminterface.py:52
GYM_OUTPUTS = ["cam/img", "gym/x", "gym/y", "gym/z", "gym/speed", "gym/cte"]     # gyminterface.py:52

# keys of the reference's gym_config the env understands (gyminterface.py:16-45, core/config.py:8-9,94-101)
DEFAULT_GYM_CONFIG = {
    "img_w": 160, "img_h": 120, "scene_name": "generated_track", "sim_latency": 0,
    "track_data_file": "track_data/generated_track.json", "hip_device": 0,
    "hip_resident": False,      # True: trs_set_step_mode(TRS_STEP_RESIDENT) - the per-tick step is POSTED to a worker kernel that stays on the GPU
}

_SCENE_TRACKS = {"generated_track": "generated_track.json", "mountain_track": "mountain_track.json"}


def _track_for(cfg):
    scene = cfg.get("scene_name", "generated_track")
    if scene in _SCENE_TRACKS:
        return _SCENE_TRACKS[scene]
    return cfg.get("track_data_file", "track_data/generated_track.json")


class HipGymInterface(Component):
    """One car (N = 1).  ``step(steering, throttle, breaking, reset) -> (img, x, y, z, speed, cte)``."""

    def __init__(self, poll_socket_sleep_time=0.01, gym_config=None, _api=None):
        self.gym_config = dict(DEFAULT_GYM_CONFIG)          # a private copy: the reference mutates its module-level
        self.gym_config.update(gym_config or {})            # default in place (gyminterface.py:50-51) — not kept
        Component.__init__(self, inputs=list(GYM_INPUTS), outputs=list(GYM_OUTPUTS), threaded=False)
        # sim_latency (gyminterface.py:96 sleeps that many ms before a telemetry frame is accepted): with the env's fixed tick
        # this is a delay line of ceil(latency_ms * loop_hz / 1000) ticks on the returned tuple (deterministic)
        self.latency = self.gym_config["sim_latency"]
        import collections, math
        self.latency_ticks = int(math.ceil(float(self.latency) * float(self.gym_config.get("loop_hz", 20)) / 1000.0)) if self.latency else 0
        self._delay = collections.deque()
        self.env = BatchedEnv(n_envs=1, track=_track_for(self.gym_config), device=self.gym_config.get("hip_device", 0),
                              img_h=int(self.gym_config["img_h"]), img_w=int(self.gym_config["img_w"]), render=True, _api=_api)
        if self.gym_config.get("hip_resident"):
            self.env.set_step_mode(True, idle_us=int(self.gym_config.get("hip_resident_idle_us", 0)))
        self.last_image = None
        self.pos_x = self.pos_y = self.pos_z = self.speed = self.cte = 0.0
        self.seg_idx = 0

    def step(self, *args):
        steering, throttle, breaking, reset = args[0], args[1], args[2], args[3]
        if breaking is None:                                  # gyminterface.py:70
            breaking = 0.0
        if steering is None or throttle is None:              # first tick: the mux has not produced anything yet
            steering, throttle = 0.0, 0.0
        self.env.step(float(steering), float(throttle), float(breaking), reset=bool(reset))
        # a fresh ndarray per frame, never overwritten by later steps (ownership rule of gyminterface.py:99); Python floats
        # because json.dump needs them (gyminterface.py:100-104); one synchronisation for the whole tuple
        img, x, y, z, speed, cte, seg, _ = self.env.fetch_outputs()
        self.last_image = img[0]
        self.pos_x, self.pos_y, self.pos_z, self.speed, self.cte = float(x[0]), float(y[0]), float(z[0]), float(speed[0]), float(cte[0])
        self.seg_idx = int(seg[0])
        out = (self.last_image, self.pos_x, self.pos_y, self.pos_z, self.speed, self.cte)
        if self.latency_ticks:
            self._delay.append(out)
            if len(self._delay) <= self.latency_ticks:                # nothing has "arrived" yet: the constructor's state
                return None, 0.0, 0.0, 0.0, 0.0, 0.0                  # (gyminterface.py:56,60-64)
            out = self._delay.popleft()
        return out

    def onStart(self):
        print(f"HipGymInterface: in-process env on GPU {self.env.device}; artificial latency {self.latency} ms = {self.latency_ticks} tick(s).")

    def onShutdown(self):
        self.env.close()

    def getName(self):
        return "Gym Interface"


class BatchedGymInterface(Component):
    """N cars per tick: the same ports carry arrays (controls: float32[N] or scalars; outputs stay on the
    device — ``cam/img`` etc. are ``__cuda_array_interface__`` handles unless ``to_host=True``)."""

    def __init__(self, n_envs, gym_config=None, to_host=False, auto_reset=True, env_id_base=0, sync=True, _api=None):
        self.gym_config = dict(DEFAULT_GYM_CONFIG)
        self.gym_config.update(gym_config or {})
        Component.__init__(self, inputs=list(GYM_INPUTS), outputs=list(GYM_OUTPUTS) + ["loc/index", "gym/done"], threaded=False)
        self.to_host = to_host
        # sync=False: hand the device handles out without waiting — for a part graph whose other parts work on THIS env's
        # stream (HipKerasPilot(env=...), BatchedControlMultiplexer(env=...)): the stream orders them, the host never waits
        self.sync = sync
        self.env = BatchedEnv(n_envs=n_envs, track=_track_for(self.gym_config), device=self.gym_config.get("hip_device", 0),
                              img_h=int(self.gym_config["img_h"]), img_w=int(self.gym_config["img_w"]), render=True,
                              auto_reset=auto_reset, env_id_base=env_id_base, _api=_api)

    def step(self, *args):
        steering, throttle, breaking, reset = args
        if steering is None or throttle is None:
            steering, throttle = 0.0, 0.0
        if is_device_array(steering):                          # 'mux/*' as device handles: no host copy of the controls either
            rs = None
            if reset is not None and is_device_array(reset):
                rs = device_ptr(reset)
            elif reset is not None and np.any(reset):
                self.env.reset(np.broadcast_to(np.asarray(reset).astype(bool), (self.env.n,)))
            self.env.step_device(device_ptr(steering), device_ptr(throttle), device_ptr(breaking) or 0, rs or 0)
        else:
            self.env.step(steering, throttle, breaking, reset=None if reset is None else reset)
        names = ["img", "pos_x", "pos_y", "pos_z", "speed", "cte", "seg_idx", "done"]
        if self.to_host:
            return self.env.fetch_outputs()
        # device handles: the step ran on the env's own stream, so order it before anyone looks (one wait for the tuple).
        # 'cam/img' alternates between two buffers: a handle stays valid while the NEXT step renders, not beyond.
        if self.sync:
            self.env.sync()
        return tuple(self.env.device_array(n, sync=False) for n in names)

    def onShutdown(self):
        self.env.close()

    def getName(self):
        return "Batched Gym Interface"


class LocationTracker(Component):
    """``gym/x, gym/y, gym/z -> loc/segment`` with the reference's exact index semantics
    (``components/track_data_process.py:89-107``: L1 over all three coordinates in binary64, best = 100,
    strict '<', first minimum wins), evaluated by the HIP nearest-point kernel."""

    def __init__(self, track_data_path, min_map=0, max_map=10, device=0, _api=None):
        Component.__init__(self, inputs=["gym/x", "gym/y", "gym/z"], outputs=["loc/segment"])
        self.env = BatchedEnv(n_envs=1, track=track_data_path, device=device, render=False, _api=_api)
        self.data = self.env.track
        self.max = max_map
        self.min = min_map

    def localize_index(self, points):
        return self.env.locate(points)

    def localize(self, point):
        if any(c is None for c in point):                 # the reference fails in abs(None - float) (track_data_process.py:104)
            raise TypeError("unsupported operand type(s) for -: 'NoneType' and 'float'")
        idx = int(self.env.locate(np.asarray(point, dtype=np.float64).reshape(1, 3))[0])
        return idx / float(len(self.data)) * (self.max - self.min) + self.min, 0.0

    def step(self, *args):
        segment, _ = self.localize((args[0], args[1], args[2]))   # raises TypeError on None like the reference
        return segment,

    def onShutdown(self):
        self.env.close()

    def getName(self):
        return "Location Tracker"


class HipImgPreprocessing(Component):
    """``cam/img -> cam/processed_img`` with the reference's hand-off semantics
    (``components/img_preprocessing.py:18-35``): ``step`` deposits the new frame and returns the frame processed
    from the PREVIOUS deposit (the reference's filter thread is one tick behind the loop; ``None`` until the
    first result exists).  The filter itself (trim, HSV masks, Canny, merge; ``:37-102``) runs on the GPU."""

    def __init__(self, cfg=None, device=0, _api=None):
        Component.__init__(self, inputs=["cam/img"], outputs=["cam/processed_img"], threaded=False)
        self.cfg = dict(cfg or {})
        self.env = BatchedEnv(n_envs=1, track=None, device=device, render=False, img_h=int(self.cfg.get("img_h", 120)),
                              img_w=int(self.cfg.get("img_w", 160)), _api=_api)
        self.pre = self.env.pre_config(self.cfg)
        self.processed_img = None
        self.running = True

    def step(self, *args):
        img_arr = args[0]
        previous = self.processed_img
        if img_arr is not None:
            self.processed_img = self.env.preprocess_host(np.asarray(img_arr)[None], self.pre)[0]
        return previous,

    def onShutdown(self):
        self.running = False
        self.env.close()

    def getName(self):
        return "Image Preprocessing"


MUX_INPUTS = ["usr/mode", "usr/steering", "usr/throttle", "usr/breaking", "ai/steering", "ai/throttle", "ai/breaking"]   # controlmultiplexer.py:9
MUX_OUTPUTS = ["mux/steering", "mux/throttle", "mux/breaking"]


class BatchedControlMultiplexer(Component):
    """``ControlMultiplexer`` (reference ``components/controlmultiplexer.py:6-70``) for N cars per tick: same ports,
    same config keys (``ai_launch_*``, ``core/config.py:57-63``); every port carries an array (or a scalar that is
    broadcast), ``usr/mode`` a sequence of ``DriveMode`` members / their strings / codes 0..2.  The selection and the
    AI-launch locks run on the GPU (``trs_control_mux``); a lock's duration in seconds becomes
    ``ceil(duration * loop_hz)`` ticks of the loop.  ``None`` controls are read as 0 (the batched ports are numeric)."""

    def __init__(self, cfg=None, n_cars=1, loop_hz=20, env=None, device=0, _api=None):
        Component.__init__(self, inputs=list(MUX_INPUTS), outputs=list(MUX_OUTPUTS), threaded=False)
        self.cfg = dict(cfg or {})
        self.n = int(n_cars)
        self._own_env = env is None
        self.env = env if env is not None else BatchedEnv(n_envs=self.n, track=None, device=device, render=False, _api=_api)
        self.mux = self.env.mux_config(self.cfg, loop_hz)
        self.last = (np.zeros(self.n, np.float32),) * 3

    def step(self, *args):
        mode = args[0]
        if mode is None or np.isscalar(mode) or hasattr(mode, "value"):
            mode = [mode] * self.n
        clean = lambda a: 0.0 if a is None else a
        usr, ai = tuple(clean(a) for a in args[1:4]), tuple(clean(a) for a in args[4:7])
        if any(is_device_array(a) for a in ai):
            # device-resident: 'ai/*' arrive as device handles (HipKerasPilot on the same env) and 'mux/*' leave as device handles;
            # only the per-car modes and the joystick values (a few bytes per car) are uploaded
            d_mode = self.env.scratch(SLOT_MODE, (self.n,), np.uint8)
            self.env.upload(d_mode, self.env.encode_modes(mode))
            d_usr = []
            for slot, a in zip(SLOT_USR, usr):
                if is_device_array(a):
                    d_usr.append(a)
                else:
                    buf = self.env.scratch(slot, (self.n,))
                    self.env.upload(buf, a)
                    d_usr.append(buf)
            d_ai = []
            for k, a in enumerate(ai):
                if is_device_array(a):
                    d_ai.append(a)
                else:                                          # e.g. a pilot that answered (0.0, 0.0, 0.0) on the host
                    buf = self.env.scratch(10 + k, (self.n,))
                    self.env.upload(buf, a)
                    d_ai.append(buf)
            return self.env.control_mux_device(d_mode, d_usr, d_ai, cfg=self.mux, n=self.n)
        self.last = self.env.control_mux_host(mode, usr, ai, keep=self.last, cfg=self.mux)
        return self.last

    def onShutdown(self):
        if self._own_env:
            self.env.close()

    def getName(self):
        return "Control Multiplexer"


PILOT_INPUTS = ["cam/img", "gym/speed", "loc/segment", "gym/cte", "usr/mode"]          # keras_pilot.py:18
PILOT_OUTPUTS = ["ai/steering", "ai/throttle", "ai/breaking"]


def load_keras_weights(path, by_name=False):
    """The arrays of ``model.get_weights()`` from an ``.npz`` (``np.savez(path, *model.get_weights())``) or, where h5py is
    installed (it is wherever the reference's TensorFlow is; not in this image), from the Keras HDF5 file the reference
    trains and loads (``keras_train.py:407``, ``keras_pilot.py:26``): layers in ``layer_names`` order, each layer's
    arrays in ``weight_names`` order — the order ``get_weights()`` uses."""
    if str(path).endswith(".npz"):
        with np.load(path) as z:
            if by_name:    # np.savez(path, **{f"{layer.name}/{i}": w for layer in model.layers for i, w in enumerate(layer.get_weights())})
                layers = {}
                for k in z.files:
                    name, _, idx = k.rpartition("/")
                    layers.setdefault(name, {})[int(idx)] = z[k]
                return {name: (d[0], d[1]) for name, d in layers.items() if len(d) == 2}
            return [z[k] for k in z.files]
    try:
        import h5py
    except ImportError as exc:
        raise RuntimeError(f"{path}: reading a Keras HDF5 model needs h5py; convert once with "
                           "np.savez('model.npz', *model.get_weights())") from exc
    out, named = [], {}
    with h5py.File(path, "r") as f:
        g = f["model_weights"] if "model_weights" in f else f
        for layer in g.attrs["layer_names"]:
            layer = layer.decode() if isinstance(layer, bytes) else layer
            arrs = []
            for name in g[layer].attrs["weight_names"]:
                name = name.decode() if isinstance(name, bytes) else name
                arrs.append(np.asarray(g[layer][name], dtype=np.float32))
            out += arrs
            if len(arrs) == 2:
                named[layer] = (arrs[0], arrs[1])
    return named if by_name else out


class HipKerasPilot(Component):
    """``KerasPilot`` for ``ModelType.CNN_2D_SPD_CTL`` and ``ModelType.CNN_2D`` (reference ``components/keras_pilot.py:17-153``;
    both run the same network, ``keras_train.py:386-395``): same ports, same
    ``spd_ctl_*`` / ``smooth_steering_*`` config keys, same rule "``(0.0, 0.0, 0.0)`` without a frame or outside the two AI
    modes".  The network (``Keras_2D_CNN.get_model``, ``keras_train.py:127-174``) runs on the GPU in fp16 MFMA
    convolutions (``trs_pilot_forward_host``); the post-processing (cap, x20, ``calcThrottle`` / ``calcBreak``, smooth
    steering; ``:78-95``) is the reference's scalar arithmetic on the host.

    ``weights``: the 22 arrays of ``model.get_weights()`` (kernel, bias of conv1..conv7, dense1..dense3, output_layer), or
    ``model_path``: an ``.npz`` with those arrays in that order (``np.savez(path, *model.get_weights())``) or the Keras
    ``.h5`` itself where h5py is installed (``load_keras_weights``).  Ports may carry one frame (N = 1, the
    reference's use) or a batch ``uint8[N,H,W,3]`` with per-car speeds; the outputs are then arrays."""

    def __init__(self, cfg=None, model_path=None, model_type="cnn_2d_speed_control", weights=None, n_cars=1, device=0, env=None):
        mt = getattr(model_type, "value", model_type)
        from ._ffi import PILOT_MODEL_TYPES
        if mt not in PILOT_MODEL_TYPES:
            raise ValueError(f"HipKerasPilot implements the model types {sorted(PILOT_MODEL_TYPES)} (utils/types.py ModelType)")
        self.model_type = mt
        Component.__init__(self, inputs=list(PILOT_INPUTS), outputs=list(PILOT_OUTPUTS), threaded=False)
        self.cfg = dict(cfg or {})
        if weights is None:
            if model_path is None:
                raise ValueError("weights or model_path (.npz of model.get_weights(), or the Keras .h5) is required")
            weights = load_keras_weights(model_path, by_name=mt in ("cnn_2d_speed_as_feature", "cnn_2d_full_house"))
        from ._ffi import PILOT_ARRAYS_OF_TYPE
        n_arrays = 2 * len(weights) if isinstance(weights, dict) else len(weights)
        if n_arrays != PILOT_ARRAYS_OF_TYPE[mt]:
            raise ValueError(f"model_type {mt!r} needs the {PILOT_ARRAYS_OF_TYPE[mt]} arrays of its architecture (kernel, bias per layer; "
                             f"see trs_pilot_load in include/trsim.h), got {n_arrays}")
        # env=: run on an existing env's handle (and stream) — the device-resident graph pilot -> mux -> sim of N cars, where
        # 'cam/img' arrives as a device handle and 'ai/*' leave as device handles (frames never visit the host)
        self._own_env = env is None
        self.env = env if env is not None else BatchedEnv(n_envs=int(n_cars), track=None, device=device, render=False,
                                                          img_h=int(self.cfg.get("img_h", 120)), img_w=int(self.cfg.get("img_w", 160)))
        self.env.pilot_load(weights)
        self.speed_control_threshold = float(self.cfg.get("spd_ctl_threshold", 1.1))
        self.speed_control_break = bool(self.cfg.get("spd_ctl_break", False))
        self.speed_control_reverse_multiplier = float(self.cfg.get("spd_ctl_reverse_multiplier", 1.0))
        self.speed_control_break_multiplier = float(self.cfg.get("spd_ctl_break_multiplier", 1.0))
        self.smooth_steering = bool(self.cfg.get("smooth_steering_enabled", False))
        self.smooth_steering_threshold = float(self.cfg.get("smooth_steering_threshold", 0.9))
        self.on = True

    def step(self, *args):
        from . import control
        img, mode = args[0], getattr(args[-1], "value", args[-1])
        if is_device_array(img):
            # device-resident batch: model + post-processing on the device frames (trs_pilot_act); per-car modes are uploaded as codes
            d_mode = None
            if not (np.isscalar(mode) or isinstance(mode, str)) or mode not in ("ai", "ai_steering"):
                modes = [mode] * self.env.n if (np.isscalar(mode) or isinstance(mode, str) or mode is None) else mode
                d_mode = self.env.scratch(SLOT_MODE, (self.env.n,), np.uint8)
                self.env.upload(d_mode, self.env.encode_modes(modes))
            pc = dict(self.cfg); pc["model_type"] = self.model_type
            # keras_pilot.py:78-90 always uses ITS inputs: a speed / segment that arrives as a host value beside a device frame is
            # uploaded (as modes and usr/* are); only a missing value (None) falls back to the env's own arrays — and a pilot
            # that owns its env has no sim behind those arrays, so it must be given the speed
            def as_device(value, slot, what):
                if is_device_array(value):
                    return value
                if value is None:
                    if self._own_env:
                        raise ValueError(f"HipKerasPilot: {what!r} is None beside a device 'cam/img' and this pilot does not share a sim's env "
                                         "(env=...): there is no speed / track index to read")
                    return None
                buf = self.env.scratch(slot, (self.env.n,), np.float32)
                self.env.upload(buf, np.asarray(value, dtype=np.float32))
                return buf
            speed = as_device(args[1], SLOT_PILOT_IN[0], "gym/speed")
            segment = None
            if self.model_type == "cnn_2d_full_house":                   # 'loc/segment'; None = from the shared env's own tracker index
                segment = as_device(args[2], SLOT_PILOT_IN[1], "loc/segment")
            return self.env.pilot_act_device(frames=img, speed=speed, mode=d_mode, cfg=pc, segment=segment)
        if img is None or mode not in ("ai", "ai_steering"):               # keras_pilot.py:46-48,139
            return 0.0, 0.0, 0.0
        img = np.asarray(img, dtype=np.uint8)
        single = img.ndim == 3
        nimg = 1 if single else img.shape[0]
        if self.model_type == "cnn_2d_speed_as_feature":                   # :67-71: the model also reads speed / 20
            raw = self.env.pilot_forward_host(img[None] if single else img, speed=np.broadcast_to(np.asarray(args[1], np.float32), (nimg,)))
        elif self.model_type == "cnn_2d_full_house":                       # :97-104: ... and 'loc/segment'
            raw = self.env.pilot_forward_host(img[None] if single else img, speed=np.broadcast_to(np.asarray(args[1], np.float32), (nimg,)),
                                              segment=np.broadcast_to(np.asarray(args[2], np.float32), (nimg,)))
        else:
            raw = self.env.pilot_forward_host(img[None] if single else img)    # [n, 2]: steering, speed / 20
        steering = np.clip(raw[:, 0].astype(np.float64), -1.0, 1.0)        # __cap (:142-145)
        if self.model_type in ("cnn_2d", "cnn_2d_speed_as_feature"):       # :56-76: (steering, throttle) capped, no brake
            throttle = np.clip(raw[:, 1].astype(np.float64), -1.0, 1.0)
            if self.smooth_steering:
                steering = np.where(steering > self.smooth_steering_threshold, 1.0,
                                    np.where(steering < -self.smooth_steering_threshold, -1.0, steering))
            if single:
                return float(steering[0]), float(throttle[0]), 0.0
            return steering, throttle, np.zeros_like(throttle)
        predicted = raw[:, 1].astype(np.float64) * 20                      # :83
        real = np.broadcast_to(np.asarray(args[1], dtype=np.float64), predicted.shape)
        throttle = control.calc_throttle(real, predicted * self.speed_control_threshold, self.speed_control_reverse_multiplier)
        breaking = np.zeros_like(throttle)
        if self.speed_control_break:                                       # :88-90
            throttle = np.where(predicted - real > 0.0, 1.0, 0.0)
            breaking = control.calc_break(real, predicted * self.speed_control_threshold, self.speed_control_break_multiplier)
        if self.smooth_steering:                                           # :147-153
            steering = np.where(steering > self.smooth_steering_threshold, 1.0,
                                np.where(steering < -self.smooth_steering_threshold, -1.0, steering))
        if single:
            return float(steering[0]), float(throttle[0]), float(breaking[0])
        return steering, throttle, breaking

    def onStart(self):
        if self.cfg.get("preprocessing_enabled"):
            print("[WARNING] Image preprocessing is enabled. Autopilot is fed with FILTERED image.")

    def onShutdown(self):
        self.on = False
        if self._own_env:
            self.env.close()

    def getName(self):
        return "Keras Pilot"
