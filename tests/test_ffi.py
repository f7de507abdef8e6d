"""The C-ABI shared library: loads without a GPU, exports every symbol include/trsim.h declares, agrees with the
ctypes struct layouts, and fails LOUDLY when there is no device (no CPU fallback in the product)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from triton_racer_sim_amd import _ffi


def header_functions(test_hooks=False):
    """Functions include/trsim.h declares; the block inside #ifdef TRS_TEST_HOOKS only when asked for."""
    with open(os.path.join(ROOT, "include", "trsim.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    hooks = "".join(re.findall(r"#ifdef TRS_TEST_HOOKS(.*?)#endif", text, flags=re.S))
    if not test_hooks:
        text = re.sub(r"#ifdef TRS_TEST_HOOKS.*?#endif", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(trs_[a-z_0-9]+)\s*\(", text)))
    return names if not test_hooks else sorted(set(re.findall(r"\b(trs_[a-z_0-9]+)\s*\(", hooks)))


def test_header_declares_what_the_binding_binds():
    assert header_functions() == sorted("trs_" + s for s in _ffi.SYMBOLS + _ffi.PILOT_SYMBOLS)
    assert header_functions(test_hooks=True) == sorted("trs_" + s for s in _ffi.HOOK_SYMBOLS)


def exported(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted(l.split()[-1] for l in out.splitlines() if l.split()[-1].startswith("trs_"))


def test_hip_library_exports_exactly_the_header():
    """libtrsim.so = the drop-in surface, nothing else: the two trs_resident_debug_* test hooks live in csrc/libtrsim_testhooks.so (-DTRS_TEST_HOOKS) only."""
    from conftest import ensure_hip_library
    import __graft_entry__
    assert exported(ensure_hip_library()) == header_functions()
    hooks_lib = __graft_entry__.build_hip_testhooks()
    assert exported(hooks_lib) == sorted(header_functions() + header_functions(test_hooks=True))


def test_oracle_exports_the_same_abi(oracle_api):
    for s in _ffi.SYMBOLS:
        assert hasattr(oracle_api.cdll, "trso_" + s), s
    assert not oracle_api.has_pilot        # the CNN's checker is a PyTorch fp32 reference (tests/test_pilot.py), not the C oracle


def test_struct_layouts_match_c(oracle_api, hip_api):
    for api in (oracle_api, hip_api):
        cfg = _ffi.TrsConfig()
        api.default_config(ctypes.byref(cfg))
        assert cfg.struct_size == ctypes.sizeof(_ffi.TrsConfig)      # C sizeof(trs_config) == ctypes layout
        assert (cfg.img_h, cfg.img_w, cfg.render, cfg.seed) == (120, 160, 1, 0x5EED)
        assert abs(cfg.dt - 0.05) < 1e-9 and cfg.z_far == 40.0      # last field reads back -> no padding drift


def test_bad_arguments_return_errors_not_crashes(oracle_api):
    h = ctypes.c_void_p()
    cfg = _ffi.TrsConfig()
    oracle_api.default_config(ctypes.byref(cfg))
    cfg.struct_size = 8
    assert oracle_api.create(ctypes.byref(cfg), 0, ctypes.byref(h)) == -1
    assert b"struct_size" in oracle_api.last_error()
    oracle_api.default_config(ctypes.byref(cfg))
    cfg.img_w = 158
    assert oracle_api.create(ctypes.byref(cfg), 0, ctypes.byref(h)) == -1
    oracle_api.default_config(ctypes.byref(cfg))
    assert oracle_api.create(ctypes.byref(cfg), 0, ctypes.byref(h)) == 0
    assert oracle_api.step_synthetic(h, 1, 1) == -2                  # no track loaded
    assert oracle_api.destroy(h) == 0


def test_product_fails_loudly_without_gpu(hip_api):
    n = ctypes.c_int(-1)
    assert hip_api.device_count(ctypes.byref(n)) == 0
    if n.value > 0:
        pytest.skip("a GPU is present")
    from triton_racer_sim_amd.env import BatchedEnv
    with pytest.raises(RuntimeError, match="no HIP device"):
        BatchedEnv(n_envs=2)
    with pytest.raises(RuntimeError, match="HIP extension not built"):
        _ffi.load_hip_library("/nonexistent/libtrsim.so")
