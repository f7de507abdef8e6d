"""ctypes binding of the C ABI declared in ``include/trsim.h``.

This is the reference-side stub a maintainer of Triton-Racer-Sim would add (the reference is
Python, so its FFI is ctypes).  ``bind(cdll, prefix)`` types every entry
point; ``load_hip_library()`` opens the in-tree HIP build and FAILS LOUDLY when it is missing —
there is no CPU fallback in the product.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(_HERE, "csrc", "libtrsim.so")


class TrsConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("n_envs", C.c_int32), ("env_id_base", C.c_int32),
        ("img_h", C.c_int32), ("img_w", C.c_int32),
        ("render", C.c_int32), ("auto_reset", C.c_int32), ("depth", C.c_int32),
        ("seed", C.c_uint64),
        ("dt", C.c_float), ("max_steer", C.c_float), ("inv_wheelbase", C.c_float), ("accel_max", C.c_float),
        ("drag_lin", C.c_float), ("roll_res", C.c_float), ("brake_max", C.c_float),
        ("v_max", C.c_float), ("v_rev_max", C.c_float), ("offtrack_cte", C.c_float),
        ("offtrack_penalty", C.c_float), ("cam_fwd", C.c_float),
        ("road_half", C.c_double), ("edge_half", C.c_double), ("centre_half", C.c_double),
        ("dash_period", C.c_double), ("dash_on", C.c_double), ("map_margin", C.c_double),
        ("fov_v_deg", C.c_double), ("cam_h", C.c_double), ("cam_pitch_deg", C.c_double), ("z_far", C.c_double),
    ]


class TrsStateView(C.Structure):
    _fields_ = [
        ("n_envs", C.c_int32), ("img_h", C.c_int32), ("img_w", C.c_int32), ("n_points", C.c_int32),
        ("img", C.c_void_p),
        ("pos_x", C.c_void_p), ("pos_y", C.c_void_p), ("pos_z", C.c_void_p), ("speed", C.c_void_p), ("cte", C.c_void_p),
        ("yaw", C.c_void_p), ("vel", C.c_void_p), ("seg_idx", C.c_void_p),
        ("ep_return", C.c_void_p), ("last_return", C.c_void_p), ("ep_len", C.c_void_p), ("done", C.c_void_p),
        ("step_count", C.c_uint64),
        ("depth", C.c_void_p),
    ]


class TrsMapInfo(C.Structure):
    _fields_ = [
        ("map_w", C.c_int32), ("map_h", C.c_int32), ("map_words", C.c_int32),
        ("cell", C.c_double), ("x0", C.c_double), ("z0", C.c_double),
        ("n_points", C.c_int32), ("lds_bytes", C.c_int32),
    ]


class TrsPreConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("dynamic_brightness", C.c_int32), ("brightness_baseline", C.c_double),
        ("contrast_ratio", C.c_float), ("contrast_offset", C.c_float),
        ("color_filter_enabled", C.c_int32), ("n_filters", C.c_int32),
        ("hsv_lo", (C.c_uint8 * 3) * 4), ("hsv_hi", (C.c_uint8 * 3) * 4), ("dst_channel", C.c_int32 * 4),
        ("edge_detection_enabled", C.c_int32), ("edge_threshold_a", C.c_int32), ("edge_threshold_b", C.c_int32),
        ("edge_dst_channel", C.c_int32),
    ]


# selectors of trs_copy_to_host: name -> (enum value, numpy dtype string, per-env? shape tag)
FIELDS = {
    "img": 0, "pos_x": 1, "pos_y": 2, "pos_z": 3, "speed": 4, "cte": 5, "yaw": 6, "vel": 7,
    "seg_idx": 8, "ep_return": 9, "last_return": 10, "ep_len": 11, "done": 12,
    "map": 13, "rowtab": 14, "palette": 15, "tangent": 16, "steer_filt": 17, "stats": 18, "depth": 19, "rowdepth": 20,
    "ctl_steer": 21, "ctl_thr": 22, "ctl_brk": 23, "dpitch": 24,
}

# every symbol include/trsim.h declares (suffix after the prefix)
SYMBOLS = [
    "default_config", "create", "destroy", "load_track", "reset", "step", "step_host", "step_synthetic", "step_sequence", "step_sequence_host",
    "set_step_mode", "get_step_mode", "quiesce", "step_wait", "get_state", "copy_to_host", "fetch_outputs", "set_pose", "locate", "map_info_get", "sync", "event_record",
    "event_elapsed_ms", "device_count", "last_error",
    "default_pre_config", "preprocess", "preprocess_host", "set_frame_filter", "normalize", "normalize_host",
    "driver_assist", "driver_assist_host",
    "default_mux_config", "control_mux", "control_mux_host", "control_mux_reset",
    "comm_get_unique_id", "comm_init", "comm_destroy", "allgather_returns", "stream_wait_external", "stream_signal_external",
    "scratch", "upload", "counters",
]


class TrsMuxConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("throttle_lock_enabled", C.c_int32), ("throttle_lock_value", C.c_float), ("throttle_lock_ticks", C.c_int32),
        ("steering_lock_enabled", C.c_int32), ("steering_lock_value", C.c_float), ("steering_lock_ticks", C.c_int32),
    ]


class TrsPilotConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("spd_ctl_threshold", C.c_float), ("spd_ctl_break", C.c_int32),
        ("spd_ctl_reverse_multiplier", C.c_float), ("spd_ctl_break_multiplier", C.c_float),
        ("smooth_steering_enabled", C.c_int32), ("smooth_steering_threshold", C.c_float),
        ("model_type", C.c_int32),
    ]


class TrsPilotTuning(C.Structure):
    """``trs_pilot_tuning`` (include/trsim.h): kernel choices of ``trs_pilot_load``; tests and measurements only."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("no_fuse", C.c_int32), ("fuse_band_r2", C.c_int32), ("fuse_wsplit_max", C.c_int32), ("fuse_roll", C.c_int32),
        ("span_layers_mask", C.c_int32), ("frame5", C.c_int32), ("frame_layers_mask", C.c_int32), ("chain_layers", C.c_int32), ("dense", C.c_int32), ("ksplit", C.c_int32),
    ]


COMM_ID_BYTES = 128                                                    # TRS_COMM_ID_BYTES

PILOT_MODEL_TYPES = {"cnn_2d_speed_control": 0, "cnn_2d": 1, "cnn_2d_speed_as_feature": 2, "cnn_2d_full_house": 3}   # TRS_PILOT_*; ModelType values (utils/types.py)

# layer names in the order trs_pilot_load takes their (kernel, bias) pairs (include/trsim.h), per architecture
PILOT_LAYERS = {
    22: ["conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7", "dense1", "dense2", "dense3", "output_layer"],
    28: ["conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7", "dense1", "dense2", "dense3", "output_layer", "feature1", "feature2", "feature3"],
    42: ["conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7", "dense1", "dense2", "dense3", "output_speed", "feature1", "feature2", "feature3",
         "current_spd_1", "current_spd_2", "current_spd_3", "dense4", "dense5", "dense6", "out_steering"],
}
PILOT_ARRAYS_OF_TYPE = {"cnn_2d_speed_control": 22, "cnn_2d": 22, "cnn_2d_speed_as_feature": 28, "cnn_2d_full_house": 42}


# HIP library only: the CNN pilot is a floating-point kernel whose checker is a PyTorch fp32 reference, not the C oracle
PILOT_SYMBOLS = ["default_pilot_config", "pilot_load", "pilot_forward", "pilot_forward_host", "pilot_forward_ex", "pilot_forward_host_ex",
                 "pilot_debug_layer", "pilot_range_check", "pilot_act", "step_pilot", "default_pilot_tuning", "pilot_set_tuning"]
# test hooks of the resident worker: only in csrc/libtrsim_testhooks.so (-DTRS_TEST_HOOKS), never in the product library
HOOK_SYMBOLS = ["resident_debug_lifetime", "resident_debug_abort"]
HIP_TESTHOOKS_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libtrsim_testhooks.so")


class Api:
    """Typed function table of one loaded library (``trs_*`` for HIP, ``trso_*`` for the test oracle)."""

    def __init__(self, cdll, prefix):
        self.cdll, self.prefix = cdll, prefix
        vp, i32, fp, u8p, dp = C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p
        sigs = {
            "default_config": (None, [C.POINTER(TrsConfig)]),
            "create": (i32, [C.POINTER(TrsConfig), i32, C.POINTER(vp)]),
            "destroy": (i32, [vp]),
            "load_track": (i32, [vp, dp, i32]),
            "reset": (i32, [vp, u8p]),
            "step": (i32, [vp, fp, fp, fp, u8p, i32]),
            "step_host": (i32, [vp, fp, fp, fp, u8p, i32]),
            "step_synthetic": (i32, [vp, i32, i32]),
            "step_sequence": (i32, [vp, fp, fp, fp, u8p, i32, i32]),
            "step_sequence_host": (i32, [vp, fp, fp, fp, u8p, i32, i32]),
            "set_step_mode": (i32, [vp, i32, i32]),
            "get_step_mode": (i32, [vp, C.POINTER(i32), C.POINTER(i32)]),
            "quiesce": (i32, [vp]),
            "step_wait": (i32, [vp, fp, fp, fp, u8p, i32]),
            "get_state": (i32, [vp, C.POINTER(TrsStateView)]),
            "copy_to_host": (i32, [vp, i32, vp, C.c_size_t]),
            "fetch_outputs": (i32, [vp] + [vp] * 8),
            "set_pose": (i32, [vp, fp, fp, fp, fp, fp]),
            "locate": (i32, [vp, dp, i32, vp]),
            "map_info_get": (i32, [vp, C.POINTER(TrsMapInfo)]),
            "sync": (i32, [vp]),
            "event_record": (i32, [vp, i32]),
            "event_elapsed_ms": (i32, [vp, i32, i32, C.POINTER(C.c_float)]),
            "device_count": (i32, [C.POINTER(i32)]),
            "last_error": (C.c_char_p, []),
            "default_pre_config": (None, [C.POINTER(TrsPreConfig)]),
            "preprocess": (i32, [vp, C.POINTER(TrsPreConfig), vp, vp, i32, C.POINTER(vp)]),
            "preprocess_host": (i32, [vp, C.POINTER(TrsPreConfig), vp, vp, i32]),
            "set_frame_filter": (i32, [vp, C.POINTER(TrsPreConfig)]),
            "normalize": (i32, [vp, vp, vp, i32]),
            "normalize_host": (i32, [vp, vp, vp, i32]),
            "driver_assist": (i32, [vp, i32, C.c_double, vp, vp, vp, vp, i32]),
            "driver_assist_host": (i32, [vp, i32, C.c_double, vp, vp, vp, vp, i32]),
            "default_mux_config": (None, [C.POINTER(TrsMuxConfig)]),
            "control_mux": (i32, [vp, C.POINTER(TrsMuxConfig), vp] + [vp] * 9 + [i32]),
            "control_mux_host": (i32, [vp, C.POINTER(TrsMuxConfig), vp] + [vp] * 9 + [i32]),
            "control_mux_reset": (i32, [vp]),
            "comm_get_unique_id": (i32, [vp]),
            "comm_init": (i32, [vp, i32, i32, vp]),
            "comm_destroy": (i32, [vp]),
            "allgather_returns": (i32, [vp, C.POINTER(vp), vp]),
            "stream_wait_external": (i32, [vp, vp]),
            "stream_signal_external": (i32, [vp, vp]),
            "scratch": (i32, [vp, i32, C.c_size_t, C.POINTER(vp)]),
            "upload": (i32, [vp, vp, vp, C.c_size_t]),
            "counters": (i32, [vp, C.POINTER(C.c_uint64 * 4)]),
        }
        pilot = {
            "default_pilot_config": (None, [C.POINTER(TrsPilotConfig)]),
            "pilot_load": (i32, [vp, C.POINTER(C.c_void_p), i32]),
            "pilot_forward": (i32, [vp, vp, i32, vp]),
            "pilot_forward_host": (i32, [vp, vp, i32, vp]),
            "pilot_debug_layer": (i32, [vp, i32, vp, C.c_size_t]),
            "pilot_range_check": (i32, [vp, vp]),
            "pilot_forward_ex": (i32, [vp, vp, vp, vp, i32, vp]),
            "pilot_forward_host_ex": (i32, [vp, vp, vp, vp, i32, vp]),
            "pilot_act": (i32, [vp, C.POINTER(TrsPilotConfig), vp, vp, vp, vp, vp, vp, vp, i32]),
            "step_pilot": (i32, [vp, C.POINTER(TrsPilotConfig), i32]),
            "default_pilot_tuning": (None, [C.POINTER(TrsPilotTuning)]),
            "pilot_set_tuning": (i32, [vp, C.POINTER(TrsPilotTuning)]),
        }
        hooks = {"resident_debug_lifetime": (i32, [vp, i32]), "resident_debug_abort": (i32, [vp])}
        for name, (res, args) in sigs.items():
            fn = getattr(cdll, prefix + name)
            fn.restype, fn.argtypes = res, args
            setattr(self, name, fn)
        self.has_pilot = hasattr(cdll, prefix + "pilot_load")
        if self.has_pilot:
            for name, (res, args) in pilot.items():
                fn = getattr(cdll, prefix + name)
                fn.restype, fn.argtypes = res, args
                setattr(self, name, fn)
        self.has_test_hooks = hasattr(cdll, prefix + "resident_debug_lifetime")
        if self.has_test_hooks:
            for name, (res, args) in hooks.items():
                fn = getattr(cdll, prefix + name)
                fn.restype, fn.argtypes = res, args
                setattr(self, name, fn)

    def check(self, rc, what):
        if rc != 0:
            msg = self.last_error()
            raise RuntimeError(f"{self.prefix}{what} failed ({rc}): {msg.decode() if msg else ''}")


def _share_torch_hip_runtime():
    """PyTorch-ROCm ships its own ``libamdhip64``; two HIP runtimes in one process do not both see the GPU
    (``RuntimeError: No HIP GPUs are available`` in whichever loads second).  When torch is installed but not
    imported yet, bind ITS runtime first so that ``libtrsim.so`` and a later ``import torch`` share it; with torch
    already imported (or absent) the loader does the right thing by itself."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if not spec or not spec.submodule_search_locations:
        return
    lib = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(lib):
        try:
            C.CDLL(lib, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_hip_library(path=None):
    """Open ``csrc/libtrsim.so`` (built by ``__graft_entry__.build()``); no fallback of any kind."""
    _share_torch_hip_runtime()
    path = path or os.environ.get("TRS_HIP_LIB") or HIP_LIB_PATH      # TRS_HIP_LIB: A/B another HIP build of the same ABI
    if not os.path.exists(path):
        raise RuntimeError(
            f"HIP extension not built: {path} is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). The product has no CPU fallback.")
    return Api(C.CDLL(path, mode=C.RTLD_LOCAL), "trs_")
