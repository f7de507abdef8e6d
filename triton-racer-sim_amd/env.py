"""``BatchedEnv`` — one shard of N in-process envs on one MI355X, over the C ABI of ``include/trsim.h``.

It is the batched form of what the reference's per-car ``GymInterface`` does for one car
(``TritonRacerSim/components/gyminterface.py:49-104``): controls in, ``(image, x, y, z, speed, cte)``
out, plus ``LocationTracker``'s index (``components/track_data_process.py:89-107``).  All outputs stay
device-resident; host copies happen only on request (``fetch``).
"""
import ctypes as C
import json
import os

import numpy as np

from . import _ffi

_HERE = os.path.dirname(os.path.abspath(__file__))
TRACK_DIR = os.path.join(_HERE, "track_data")

_FIELD_DTYPES = {
    "img": np.uint8, "pos_x": np.float32, "pos_y": np.float32, "pos_z": np.float32, "speed": np.float32,
    "cte": np.float32, "yaw": np.float32, "vel": np.float32, "seg_idx": np.int32, "ep_return": np.float32,
    "last_return": np.float32, "ep_len": np.int32, "done": np.uint8, "map": np.uint32, "rowtab": np.float32,
    "palette": np.uint32, "tangent": np.float32, "steer_filt": np.float32, "stats": np.uint64, "depth": np.float32, "rowdepth": np.float32,
    "ctl_steer": np.float32, "ctl_thr": np.float32, "ctl_brk": np.float32, "dpitch": np.float32,
}


def resolve_track(track):
    """Accepts an ``(n,3)`` array, a JSON path, or a scene / file name as the reference configs use them
    (``scene_name`` ``core/config.py:94``, ``track_data_file`` ``core/config.py:101``)."""
    if isinstance(track, (list, tuple, np.ndarray)):
        pts = np.asarray(track, dtype=np.float64)
    else:
        name = str(track)
        candidates = [name, os.path.join(TRACK_DIR, name), os.path.join(TRACK_DIR, os.path.basename(name)),
                      os.path.join(TRACK_DIR, os.path.splitext(os.path.basename(name))[0] + ".json")]
        for c in candidates:
            if os.path.isfile(c):
                with open(c) as f:
                    pts = np.asarray(json.load(f), dtype=np.float64)
                break
        else:
            raise FileNotFoundError(f"no track data for {track!r} (looked in {TRACK_DIR})")
    if pts.ndim != 2 or pts.shape[1] != 3 or pts.shape[0] < 2:
        raise ValueError("track must be an (n>=2, 3) array of [x, y, z]")
    return np.ascontiguousarray(pts)


class _DevicePtr:
    """Zero-copy view of a device array for torch / cupy (``__cuda_array_interface__``).  The producing work has been
    ordered before the view is handed out (``BatchedEnv.device_array``: a host synchronisation, or a wait queued on the
    consumer's stream), so the interface carries no stream of its own."""

    def __init__(self, ptr, shape, dtype, owner):
        self._owner = owner
        self.__cuda_array_interface__ = {
            "shape": tuple(shape), "typestr": np.dtype(dtype).str, "data": (int(ptr), False), "version": 2, "strides": None,
        }


def device_ptr(x):
    """Device address of a zero-copy handle: anything with ``__cuda_array_interface__`` (this module's handles, torch / cupy
    arrays), or an integer; ``None`` / ``0`` stay ``None``."""
    if x is None:
        return None
    if hasattr(x, "__cuda_array_interface__"):
        return int(x.__cuda_array_interface__["data"][0]) or None
    return int(x) or None


def is_device_array(x):
    return hasattr(x, "__cuda_array_interface__")


# handle-owned scratch slots (trs_scratch) the device-resident parts use for the values that travel between them
SLOT_AI = (0, 1, 2)            # 'ai/steering', 'ai/throttle', 'ai/breaking'    (HipKerasPilot)
SLOT_MODE = 3                  # 'usr/mode' as uint8 codes                         (BatchedControlMultiplexer, HipKerasPilot)
SLOT_USR = (4, 5, 6)           # 'usr/steering', 'usr/throttle', 'usr/breaking'   (uploaded joystick values)
SLOT_MUX = (7, 8, 9)           # 'mux/steering', 'mux/throttle', 'mux/breaking'   (BatchedControlMultiplexer)
SLOT_PILOT_IN = (10, 11)       # 'gym/speed', 'loc/segment' that reached HipKerasPilot as HOST values beside a device frame


def _stream_ptr(stream):
    """A HIP stream as an integer: accepts ``torch.cuda.Stream`` (``.cuda_stream``), cupy streams (``.ptr``) or the raw
    pointer value; ``0`` is the legacy default stream."""
    for attr in ("cuda_stream", "ptr"):
        if hasattr(stream, attr):
            return int(getattr(stream, attr))
    return int(stream)


class BatchedEnv:
    def __init__(self, n_envs=1, track="generated_track", device=0, img_h=120, img_w=160, render=True,
                 auto_reset=False, env_id_base=0, seed=None, depth=False, _api=None, **overrides):
        # `_api` exists for tests only (they pass the oracle's function table to diff both through one
        # wrapper); the product path always binds the HIP library and raises if it is not built.
        self.api = _api if _api is not None else _ffi.load_hip_library()
        cfg = _ffi.TrsConfig()
        self.api.default_config(C.byref(cfg))
        cfg.n_envs, cfg.env_id_base, cfg.img_h, cfg.img_w = int(n_envs), int(env_id_base), int(img_h), int(img_w)
        cfg.render, cfg.auto_reset, cfg.depth = int(bool(render)), int(bool(auto_reset)), int(bool(depth))
        if seed is not None:
            cfg.seed = int(seed)
        for k, v in overrides.items():
            if not hasattr(cfg, k):
                raise TypeError(f"unknown env parameter {k!r}")
            setattr(cfg, k, v)
        self.cfg = cfg
        self.n, self.H, self.W = int(n_envs), int(img_h), int(img_w)
        self.device = int(device)
        self._h = C.c_void_p()
        self.api.check(self.api.create(C.byref(cfg), self.device, C.byref(self._h)), "create")
        self.track = None
        self.map_info = None
        if track is not None:
            self.load_track(track)

    # -- lifecycle ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.api.destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_track(self, track):
        pts = resolve_track(track)
        self.api.check(self.api.load_track(self._h, pts.ctypes.data, int(pts.shape[0])), "load_track")
        self.track = pts
        mi = _ffi.TrsMapInfo()
        self.api.check(self.api.map_info_get(self._h, C.byref(mi)), "map_info_get")
        self.map_info = mi

    @property
    def n_points(self):
        return 0 if self.track is None else int(self.track.shape[0])

    # -- stepping ----------------------------------------------------------------------------
    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        if m is not None and m.shape != (self.n,):
            raise ValueError("mask must have shape (n_envs,)")
        self.api.check(self.api.reset(self._h, None if m is None else m.ctypes.data), "reset")

    def _ctl(self, a, name, optional=False):
        if a is None:
            if optional:
                return None
            raise ValueError(f"{name} is required")
        arr = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float32), (self.n,)))
        return arr

    def step(self, steering, throttle, brake=None, reset=None, n_steps=1):
        """Host-array controls (numpy / scalars).  Asynchronous; ``fetch`` synchronises."""
        st, th, br = self._ctl(steering, "steering"), self._ctl(throttle, "throttle"), self._ctl(brake, "brake", True)
        rs = None if reset is None else np.ascontiguousarray(np.broadcast_to(np.asarray(reset).astype(bool), (self.n,)), dtype=np.uint8)
        self.api.check(self.api.step_host(self._h, st.ctypes.data, th.ctypes.data, None if br is None else br.ctypes.data,
                                          None if rs is None else rs.ctypes.data, int(n_steps)), "step_host")

    def step_device(self, d_steering, d_throttle, d_brake=0, d_reset=0, n_steps=1, stream=None):
        """Raw device pointers (ints), e.g. ``tensor.data_ptr()`` of float32 / uint8 CUDA tensors.  The env works on its own
        stream: pass ``stream=`` (the stream that PRODUCES the control tensors, e.g. ``torch.cuda.current_stream()``) and the
        step is ordered behind that stream's work (``trs_stream_wait_external``: an event wait, no host synchronisation in
        launch mode); without it the caller must have synchronised the producer."""
        if stream is not None:
            self.api.check(self.api.stream_wait_external(self._h, _stream_ptr(stream) or None), "stream_wait_external")
        self.api.check(self.api.step(self._h, int(d_steering), int(d_throttle), int(d_brake) or None, int(d_reset) or None,
                                     int(n_steps)), "step")

    def step_device_wait(self, d_steering, d_throttle, d_brake=0, d_reset=0, n_steps=1):
        """``step_device`` and ``sync`` in one call (``trs_step_wait``): the lock-step tick — post the controls, return when the
        frame and the telemetry are complete — with one FFI crossing."""
        self.api.check(self.api.step_wait(self._h, int(d_steering), int(d_throttle), int(d_brake) or None, int(d_reset) or None,
                                          int(n_steps)), "step_wait")

    def step_sequence(self, steering, throttle, brake=None, reset=None, steps_per_launch=8):
        """Open-loop action sequences: ``steering`` / ``throttle`` / ``brake`` are ``[n_steps, n_envs]`` host arrays, one
        control set per step (``trs_step_sequence_host``); the call runs ``steps_per_launch`` steps per kernel launch."""
        st = np.ascontiguousarray(steering, dtype=np.float32)
        k = st.shape[0]
        f = lambda a: np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float32), (k, self.n)))
        st, th = f(st), f(throttle)
        br = None if brake is None else f(brake)
        rs = None if reset is None else np.ascontiguousarray(np.broadcast_to(np.asarray(reset, dtype=np.uint8), (self.n,)))
        self.api.check(self.api.step_sequence_host(self._h, st.ctypes.data, th.ctypes.data, None if br is None else br.ctypes.data,
                                                   None if rs is None else rs.ctypes.data, int(k), int(steps_per_launch)), "step_sequence_host")

    def step_sequence_device(self, d_steering, d_throttle, d_brake=0, d_reset=0, n_steps=1, steps_per_launch=8):
        """Same with raw device pointers to ``float32[n_steps, n_envs]`` arrays."""
        self.api.check(self.api.step_sequence(self._h, int(d_steering), int(d_throttle), int(d_brake) or None, int(d_reset) or None,
                                              int(n_steps), int(steps_per_launch)), "step_sequence")

    def step_synthetic(self, n_steps=1, steps_per_launch=1):
        self.api.check(self.api.step_synthetic(self._h, int(n_steps), int(steps_per_launch)), "step_synthetic")

    def set_step_mode(self, resident=True, idle_us=0):
        """``resident=True``: a worker kernel stays on the GPU and every ``step*`` call only posts its controls
        (``trs_set_step_mode``: no launch, no kernel boundary, no table re-staging per step); ``False``: one launch per call."""
        self.api.check(self.api.set_step_mode(self._h, 1 if resident else 0, int(idle_us)), "set_step_mode")

    def sync(self):
        self.api.check(self.api.sync(self._h), "sync")

    def step_mode(self):
        """``("resident" | "launch", fell_back)`` (``trs_get_step_mode``).  ``fell_back``: resident mode was selected and the library went back to
        launches by itself because the GPU is shared with another process's resident worker (``include/trsim.h``)."""
        mode, fb = C.c_int(0), C.c_int(0)
        self.api.check(self.api.get_step_mode(self._h, C.byref(mode), C.byref(fb)), "get_step_mode")
        return ("resident" if mode.value == 1 else "launch"), bool(fb.value)

    def quiesce(self):
        """Resident mode: the worker kernel leaves the GPU (posted steps complete first); the next step starts a new one
        (``trs_quiesce``).  Call before work of another stream that needs the CUs the worker occupies (collectives)."""
        self.api.check(self.api.quiesce(self._h), "quiesce")

    # -- outputs -----------------------------------------------------------------------------
    def _shape(self, name):
        mi = self.map_info
        return {
            "img": (self.n, self.H, self.W, 3), "map": (mi.map_h, mi.map_words) if mi else None,
            "rowtab": (self.H, 2), "palette": (self.H, 4), "tangent": (self.n_points, 2), "stats": (64,), "depth": (self.n, self.H, self.W), "rowdepth": (self.H,),
            "dpitch": (self.n_points,),
        }.get(name, (self.n,))

    def fetch(self, name):
        """Synchronising device→host copy of one output field as a fresh numpy array."""
        shape = self._shape(name)
        out = np.empty(shape, dtype=_FIELD_DTYPES[name])
        self.api.check(self.api.copy_to_host(self._h, _ffi.FIELDS[name], out.ctypes.data, out.nbytes), f"copy_to_host({name})")
        return out

    def fetch_outputs(self, image=True):
        """``(img, x, y, z, speed, cte, seg_idx, done)`` of all envs as fresh numpy arrays in one synchronisation
        (``trs_fetch_outputs``): what ``GymInterface.step`` returns plus the tracker index and the done flag."""
        n = self.n
        img = np.empty((n, self.H, self.W, 3), np.uint8) if image else None
        f = [np.empty(n, np.float32) for _ in range(5)]
        seg, done = np.empty(n, np.int32), np.empty(n, np.uint8)
        self.api.check(self.api.fetch_outputs(self._h, img.ctypes.data if image else None, *[a.ctypes.data for a in f],
                                              seg.ctypes.data, done.ctypes.data), "fetch_outputs")
        return (img, *f, seg, done)

    def state_view(self):
        sv = _ffi.TrsStateView()
        self.api.check(self.api.get_state(self._h, C.byref(sv)), "get_state")
        return sv

    def device_array(self, name, stream=None, sync=True):
        """Zero-copy handle (``__cuda_array_interface__``) on a device-resident output; use
        ``torch.as_tensor(env.device_array('ep_return'), device='cuda')``.  The env's steps run on the env's own stream, so
        the view is ordered first: with ``stream=`` (the CONSUMER's stream) that stream is made to wait for the env's work
        queued so far (``trs_stream_signal_external``), otherwise the host waits (``sync=False`` skips even that: the caller
        orders).  ``'img'`` / ``'depth'`` alternate between two buffers: the frame of step s is overwritten by step s + 2."""
        if stream is not None:
            self.api.check(self.api.stream_signal_external(self._h, _stream_ptr(stream) or None), "stream_signal_external")
        elif sync:
            self.sync()
        sv = self.state_view()
        ptr = getattr(sv, name)
        if not ptr:
            raise RuntimeError(f"{name} is not available")
        return _DevicePtr(ptr, self._shape(name), _FIELD_DTYPES[name], self)

    def set_pose(self, x=None, y=None, z=None, yaw=None, v=None):
        arrs = [None if a is None else np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float32), (self.n,))) for a in (x, y, z, yaw, v)]
        self.api.check(self.api.set_pose(self._h, *[None if a is None else a.ctypes.data for a in arrs]), "set_pose")

    def locate(self, points):
        """Batched ``LocationTracker.__find_closest`` (``track_data_process.py:89-101``) → int32 indices."""
        q = np.ascontiguousarray(np.asarray(points, dtype=np.float64).reshape(-1, 3))
        out = np.empty(q.shape[0], dtype=np.int32)
        self.api.check(self.api.locate(self._h, q.ctypes.data, int(q.shape[0]), out.ctypes.data), "locate")
        return out

    def segment(self, idx, min_map=0.0, max_map=10.0):
        """``LocationTracker.__map`` (``track_data_process.py:106-107``): index → 'loc/segment' float."""
        return np.asarray(idx, dtype=np.float64) / float(self.n_points) * (max_map - min_map) + min_map

    # -- image path (ImgPreprocessing + pilot normalisation) ------------------------------------
    def pre_config(self, cfg=None):
        """``trs_pre_config`` from the reference's ``preprocessing_*`` keys (``core/config.py:15-28``)."""
        pc = _ffi.TrsPreConfig()
        self.api.default_pre_config(C.byref(pc))
        cfg = cfg or {}
        pc.dynamic_brightness = int(bool(cfg.get("preprocessing_dynamic_brightness_enabled", False)))
        pc.brightness_baseline = float(cfg.get("preprocessing_brightness_baseline", 550))
        pc.contrast_ratio = float(cfg.get("preprocessing_contrast_enhancement_ratio", 1.0))
        pc.contrast_offset = float(cfg.get("preprocessing_contrast_enhancement_offset", 125))
        pc.color_filter_enabled = int(bool(cfg.get("preprocessing_color_filter_enabled", False)))
        pc.edge_detection_enabled = int(bool(cfg.get("preprocessing_edge_detection_enabled", False)))
        pc.edge_threshold_a = int(cfg.get("preprocessing_edge_detection_threshold_a", 60))
        pc.edge_threshold_b = int(cfg.get("preprocessing_edge_detection_threshold_b", 100))
        pc.edge_dst_channel = int(cfg.get("preprocessing_edge_detection_destination_channel", 2))
        if "preprocessing_color_filter_hsvs" in cfg:
            bounds = cfg["preprocessing_color_filter_hsvs"]
            chans = cfg.get("preprocessing_color_filter_destination_channels", list(range(len(bounds))))
            if len(bounds) > 4 or len(chans) != len(bounds):
                raise ValueError("at most 4 colour filters, one destination channel each")
            pc.n_filters = len(bounds)
            for f, (lo, hi) in enumerate(bounds):
                for k in range(3):
                    pc.hsv_lo[f][k], pc.hsv_hi[f][k] = int(min(lo[k], 255)), int(min(hi[k], 255))
                pc.dst_channel[f] = int(chans[f])
        return pc

    def preprocess_host(self, frames, cfg=None):
        """``ImgPreprocessing.__process`` (no Canny) for host frames ``uint8[n,H,W,3]`` -> fresh ndarray."""
        src = np.ascontiguousarray(frames, dtype=np.uint8).reshape(-1, self.H, self.W, 3)
        dst = np.empty_like(src)
        pc = cfg if isinstance(cfg, _ffi.TrsPreConfig) else self.pre_config(cfg)
        self.api.check(self.api.preprocess_host(self._h, C.byref(pc), src.ctypes.data, dst.ctypes.data, int(src.shape[0])), "preprocess_host")
        return dst

    def preprocess_latest(self, cfg=None):
        """Process the env's latest frames on the device; returns a zero-copy handle on ``uint8[N,H,W,3]``."""
        pc = cfg if isinstance(cfg, _ffi.TrsPreConfig) else self.pre_config(cfg)
        out = C.c_void_p()
        self.api.check(self.api.preprocess(self._h, C.byref(pc), None, None, self.n, C.byref(out)), "preprocess")
        return _DevicePtr(out.value, (self.n, self.H, self.W, 3), np.uint8, self)

    def set_frame_filter(self, cfg=None, enabled=True):
        """``ImgPreprocessing`` fused behind the rasteriser (``trs_set_frame_filter``): from the next frame on, the
        env's ``img`` IS ``cam/processed_img`` at no extra cost (the palette is filtered, not the pixels).  Trim and
        HSV masks only; dynamic brightness and Canny are refused.  ``enabled=False`` returns to raw frames."""
        if not enabled:
            self.api.check(self.api.set_frame_filter(self._h, None), "set_frame_filter")
            return
        pc = cfg if isinstance(cfg, _ffi.TrsPreConfig) else self.pre_config(cfg)
        self.api.check(self.api.set_frame_filter(self._h, C.byref(pc)), "set_frame_filter")

    def normalize_host(self, frames):
        """Pilot-side ``float32(img) / 255`` (``keras_pilot.py:49-55``) -> ``float32[n,H,W,3]``."""
        src = np.ascontiguousarray(frames, dtype=np.uint8).reshape(-1, self.H, self.W, 3)
        dst = np.empty(src.shape, dtype=np.float32)
        self.api.check(self.api.normalize_host(self._h, src.ctypes.data, dst.ctypes.data, int(src.shape[0])), "normalize_host")
        return dst

    def driver_assist_host(self, steering, throttle, brake, speed, mode="steering", k=5):
        """``DriverAssistance.step`` (``components/driver_assistance.py:13-31``) for N cars on the device; returns new float32 arrays."""
        arrs = [np.array(a, dtype=np.float32, copy=True).reshape(-1) for a in (steering, throttle, brake, speed)]
        n = arrs[0].size
        self.api.check(self.api.driver_assist_host(self._h, {"steering": 0, "speed": 1}[mode], float(k), arrs[0].ctypes.data, arrs[1].ctypes.data,
                                                    arrs[2].ctypes.data, arrs[3].ctypes.data, n), "driver_assist_host")
        return arrs[0], arrs[1], arrs[2]

    # -- ControlMultiplexer for N cars (components/controlmultiplexer.py:24-43) --------------------
    MODES = {"human": 0, "ai_steering": 1, "ai": 2}

    def mux_config(self, cfg=None, loop_hz=20):
        """``trs_mux_config`` from the reference's ``ai_launch_*`` keys (``core/config.py:57-63``); a duration in
        seconds becomes ``ceil(duration * loop_hz)`` ticks of the fixed-step env (at least 1)."""
        import math
        mc = _ffi.TrsMuxConfig()
        self.api.default_mux_config(C.byref(mc))
        cfg = cfg or {}
        mc.throttle_lock_enabled = int(bool(cfg.get("ai_launch_boost_throttle_enabled", False)))
        mc.throttle_lock_value = float(cfg.get("ai_launch_boost_throttle_value", 1.0))
        mc.throttle_lock_ticks = max(1, math.ceil(float(cfg.get("ai_launch_boost_throttle_duration", 5)) * loop_hz))
        mc.steering_lock_enabled = int(bool(cfg.get("ai_launch_lock_steering_enabled", False)))
        mc.steering_lock_value = float(cfg.get("ai_launch_lock_steering_value", 0.0))
        mc.steering_lock_ticks = max(1, math.ceil(float(cfg.get("ai_launch_lock_steering_duration", 3)) * loop_hz))
        return mc

    @classmethod
    def encode_modes(cls, modes):
        """``usr/mode`` values (``DriveMode`` members, their ``.value`` strings or the integers 0..2) -> uint8 codes;
        anything else becomes 255 = 'leave this car's outputs alone'."""
        if isinstance(modes, np.ndarray) and modes.dtype == np.uint8:
            return np.ascontiguousarray(modes).reshape(-1)
        out = np.empty(len(modes), np.uint8)
        for i, m in enumerate(modes):
            m = getattr(m, "value", m)
            if isinstance(m, str):
                out[i] = cls.MODES.get(m, 255)
            elif isinstance(m, (int, np.integer)) and 0 <= int(m) <= 2:
                out[i] = int(m)
            else:
                out[i] = 255
        return out

    def control_mux_host(self, modes, usr, ai, keep=None, cfg=None, loop_hz=20):
        """One tick of ``ControlMultiplexer.step`` for n cars on the device.  ``usr`` / ``ai``: (steering, throttle,
        breaking) arrays; ``keep``: the values a car with an unknown mode retains (default zeros).  Returns the three
        ``mux/*`` arrays (float32)."""
        mode = self.encode_modes(modes)
        n = mode.size
        f = lambda a: np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float32), (n,)))
        ins = [f(a) for a in (*usr, *ai)]
        outs = [np.array(f(a), copy=True) for a in (keep if keep is not None else (0.0, 0.0, 0.0))]
        mc = cfg if isinstance(cfg, _ffi.TrsMuxConfig) else self.mux_config(cfg, loop_hz)
        self.api.check(self.api.control_mux_host(self._h, C.byref(mc), mode.ctypes.data, *[a.ctypes.data for a in ins],
                                                 *[a.ctypes.data for a in outs], n), "control_mux_host")
        return tuple(outs)

    def control_mux_reset(self):
        self.api.check(self.api.control_mux_reset(self._h), "control_mux_reset")

    # -- pilot in the loop (cnn_2d_speed_control) -------------------------------------------------
    def pilot_load(self, weights):
        """``weights``: 22 float32 arrays — kernel, bias of conv1..conv7, dense1..dense3, output_layer in Keras
        layouts (``[KH][KW][CIN][COUT]`` / ``[IN][OUT]``), e.g. ``model.get_weights()`` of the reference's
        ``Keras_2D_CNN.get_model(input_shape, 2)`` (``components/keras_train.py:127-174``)."""
        if isinstance(weights, dict):
            # by layer name: {"conv1": (kernel, bias), ...}.  For the model types with several inputs this is the safe form: the
            # order of model.get_weights() follows Keras's layer sorting, the library's order is fixed by name (include/trsim.h)
            names = _ffi.PILOT_LAYERS[2 * len(weights)]
            missing = [nm for nm in names if nm not in weights]
            if missing:
                raise ValueError(f"weights lack the layers {missing}")
            weights = [a for nm in names for a in weights[nm]]
        arrs = [np.ascontiguousarray(w, dtype=np.float32) for w in weights]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        self.api.check(self.api.pilot_load(self._h, ptrs, len(arrs)), "pilot_load")

    def pilot_tuning(self, **choices):
        """Override kernel choices of the NEXT ``pilot_load`` (``trs_pilot_tuning``, include/trsim.h) — tests that compare a kernel
        with the one it replaced, and measurements.  No arguments: back to the defaults."""
        if not choices:
            self.api.check(self.api.pilot_set_tuning(self._h, None), "pilot_set_tuning")
            return
        t = _ffi.TrsPilotTuning()
        self.api.default_pilot_tuning(C.byref(t))
        names = {f[0] for f in _ffi.TrsPilotTuning._fields_} - {"struct_size"}
        for k, v in choices.items():
            if k not in names:
                raise ValueError(f"trs_pilot_tuning has no field {k!r}")
            setattr(t, k, int(v))
        self.api.check(self.api.pilot_set_tuning(self._h, C.byref(t)), "pilot_set_tuning")

    def resident_lifetime(self, life_us):
        """Test hook (``trs_resident_debug_lifetime``): resident workers leave by themselves after ``life_us``.  Only in the test build of the library
        (``csrc/libtrsim_testhooks.so``, ``-DTRS_TEST_HOOKS``): bind it with ``BatchedEnv(_api=...)`` as ``tests/conftest.py`` does."""
        if not getattr(self.api, "has_test_hooks", False):
            raise RuntimeError("trs_resident_debug_lifetime is not part of libtrsim.so: bind csrc/libtrsim_testhooks.so (tests/conftest.py: make_env('hip_hooks', ...))")
        self.api.check(self.api.resident_debug_lifetime(self._h, int(life_us)), "resident_debug_lifetime")

    def resident_abort(self):
        """Test hook (``trs_resident_debug_abort``): the running worker's abort bit is set from outside (test build of the library only)."""
        if not getattr(self.api, "has_test_hooks", False):
            raise RuntimeError("trs_resident_debug_abort is not part of libtrsim.so: bind csrc/libtrsim_testhooks.so (tests/conftest.py: make_env('hip_hooks', ...))")
        self.api.check(self.api.resident_debug_abort(self._h), "resident_debug_abort")

    def pilot_config(self, cfg=None):
        pc = _ffi.TrsPilotConfig()
        self.api.default_pilot_config(C.byref(pc))
        cfg = cfg or {}
        pc.spd_ctl_threshold = float(cfg.get("spd_ctl_threshold", 1.1))
        pc.spd_ctl_break = int(bool(cfg.get("spd_ctl_break", False)))
        pc.spd_ctl_reverse_multiplier = float(cfg.get("spd_ctl_reverse_multiplier", 1.0))
        pc.spd_ctl_break_multiplier = float(cfg.get("spd_ctl_break_multiplier", 1.0))
        pc.smooth_steering_enabled = int(bool(cfg.get("smooth_steering_enabled", False)))
        pc.smooth_steering_threshold = float(cfg.get("smooth_steering_threshold", 0.9))
        mt = getattr(cfg.get("model_type", "cnn_2d_speed_control"), "value", cfg.get("model_type", "cnn_2d_speed_control"))
        if mt not in _ffi.PILOT_MODEL_TYPES:
            raise ValueError(f"model_type {mt!r}: the GPU pilot runs {sorted(_ffi.PILOT_MODEL_TYPES)}")
        pc.model_type = _ffi.PILOT_MODEL_TYPES[mt]
        return pc

    def pilot_forward_host(self, frames, speed=None, segment=None):
        """Raw model outputs ``float32[n, 2]`` for host frames ``uint8[n,H,W,3]``; ``speed`` ('gym/speed', divided by 20 inside)
        for ``cnn_2d_speed_as_feature``, ``speed`` and ``segment`` ('loc/segment') for ``cnn_2d_full_house``."""
        src = np.ascontiguousarray(frames, dtype=np.uint8).reshape(-1, self.H, self.W, 3)
        n = int(src.shape[0])
        out = np.empty((n, 2), dtype=np.float32)
        if speed is None and segment is None:
            self.api.check(self.api.pilot_forward_host(self._h, src.ctypes.data, n, out.ctypes.data), "pilot_forward_host")
            return out
        f = lambda a: None if a is None else np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float32), (n,)))
        sp, sg = f(speed), f(segment)
        self.api.check(self.api.pilot_forward_host_ex(self._h, src.ctypes.data, None if sp is None else sp.ctypes.data,
                                                      None if sg is None else sg.ctypes.data, n, out.ctypes.data), "pilot_forward_host_ex")
        return out

    def pilot_layer(self, layer, shape):
        out = np.empty(shape, dtype=np.float32)
        self.api.check(self.api.pilot_debug_layer(self._h, int(layer), out.ctypes.data, out.size), "pilot_debug_layer")
        return out

    def pilot_range_check(self):
        """fp16 saturations of the last forward pass per convolution (``trs_pilot_range_check``): ``uint64[8]``, [0..6] = conv1..conv7, [7] = sum."""
        out = np.zeros(8, np.uint64)
        self.api.check(self.api.pilot_range_check(self._h, out.ctypes.data), "pilot_range_check")
        return out

    def step_pilot(self, n_steps=1, cfg=None):
        """Closed loop: controls = KerasPilot.step(previous frame, speed), then one env step; all on the device."""
        pc = cfg if isinstance(cfg, _ffi.TrsPilotConfig) else self.pilot_config(cfg)
        self.api.check(self.api.step_pilot(self._h, C.byref(pc), int(n_steps)), "step_pilot")

    # -- device-resident part graphs (pilot -> mux -> sim without host copies of frames; car_templates/manage.py:46-75) ------
    def scratch(self, slot, shape, dtype=np.float32):
        """A handle-owned device buffer (``trs_scratch``) as a zero-copy handle; the same slot returns the same memory."""
        shape = tuple(np.atleast_1d(shape).tolist()) if not isinstance(shape, tuple) else shape
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ptr = C.c_void_p()
        self.api.check(self.api.scratch(self._h, int(slot), nbytes, C.byref(ptr)), "scratch")
        return _DevicePtr(ptr.value, shape, dtype, self)

    def upload(self, dst, values):
        """Host values into a device buffer on the env's stream (``trs_upload``); ``values`` is broadcast to the buffer's shape."""
        cai = dst.__cuda_array_interface__
        arr = np.ascontiguousarray(np.broadcast_to(np.asarray(values, dtype=np.dtype(cai["typestr"])), cai["shape"]))
        self.api.check(self.api.upload(self._h, device_ptr(dst), arr.ctypes.data, arr.nbytes), "upload")

    def counters(self):
        """``(device->host bytes, host->device bytes, steps)`` the library itself has copied / taken since creation."""
        out = (C.c_uint64 * 4)()
        self.api.check(self.api.counters(self._h, C.byref(out)), "counters")
        return int(out[0]), int(out[1]), int(out[2])

    def pilot_act_device(self, frames=None, speed=None, mode=None, cfg=None, n=None, segment=None):
        """``KerasPilot.step`` for n cars on the device (``trs_pilot_act``): frames / speed / mode are device handles (``None`` =
        the env's latest frames / own speed / all cars in an AI mode); returns the three ``ai/*`` handles (scratch slots)."""
        n = self.n if n is None else int(n)
        pc = cfg if isinstance(cfg, _ffi.TrsPilotConfig) else self.pilot_config(cfg)
        outs = [self.scratch(sl, (n,)) for sl in SLOT_AI]
        self.api.check(self.api.pilot_act(self._h, C.byref(pc), device_ptr(frames), device_ptr(speed), device_ptr(segment), device_ptr(mode),
                                          *[device_ptr(o) for o in outs], n), "pilot_act")
        return tuple(outs)

    def control_mux_device(self, mode, usr, ai, cfg=None, loop_hz=20, n=None):
        """One tick of ``ControlMultiplexer.step`` on device arrays (``trs_control_mux``).  ``mode``: uint8 device handle;
        ``usr`` / ``ai``: three device handles each; returns the three ``mux/*`` handles (scratch slots, values of cars with an
        unknown mode are kept from the previous tick)."""
        n = self.n if n is None else int(n)
        mc = cfg if isinstance(cfg, _ffi.TrsMuxConfig) else self.mux_config(cfg, loop_hz)
        outs = [self.scratch(sl, (n,)) for sl in SLOT_MUX]
        self.api.check(self.api.control_mux(self._h, C.byref(mc), device_ptr(mode), *[device_ptr(a) for a in usr], *[device_ptr(a) for a in ai],
                                            *[device_ptr(o) for o in outs], n), "control_mux")
        return tuple(outs)

    # -- the one exchange of the multi-GPU path (trs_comm_* / trs_allgather_returns: RCCL behind the C ABI) ----------------
    def comm_unique_id(self):
        """Rank 0: the 128-byte id every rank passes to ``comm_init`` (hand it over by any host channel)."""
        buf = C.create_string_buffer(_ffi.COMM_ID_BYTES)
        self.api.check(self.api.comm_get_unique_id(buf), "comm_get_unique_id")
        return buf.raw

    def comm_init(self, rank, world, unique_id=None):
        uid = None if unique_id is None else C.create_string_buffer(bytes(unique_id), _ffi.COMM_ID_BYTES)
        self.api.check(self.api.comm_init(self._h, int(rank), int(world), uid), "comm_init")
        self._comm_world = int(world)

    def comm_destroy(self):
        self.api.check(self.api.comm_destroy(self._h), "comm_destroy")
        self._comm_world = 0

    def allgather_returns(self):
        """``ep_return`` of every env of every shard, ordered by rank, as a fresh ``float32[world * n_envs]`` host array."""
        out = np.empty(self._comm_world * self.n, np.float32)
        self.api.check(self.api.allgather_returns(self._h, None, out.ctypes.data), "allgather_returns")
        return out

    def allgather_returns_device(self):
        """Same, device resident: a zero-copy handle on the handle-owned buffer (the gather is queued on the env's stream;
        the handle is handed out after a host synchronisation)."""
        ptr = C.c_void_p()
        self.api.check(self.api.allgather_returns(self._h, C.byref(ptr), None), "allgather_returns")
        self.sync()
        return _DevicePtr(ptr.value, (self._comm_world * self.n,), np.float32, self)

    # -- timing (HIP events on the handle's stream) -------------------------------------------
    def event_record(self, slot):
        self.api.check(self.api.event_record(self._h, int(slot)), "event_record")

    def event_elapsed_ms(self, a, b):
        ms = C.c_float()
        self.api.check(self.api.event_elapsed_ms(self._h, int(a), int(b), C.byref(ms)), "event_elapsed_ms")
        return float(ms.value)
