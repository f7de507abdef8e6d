import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def oracle_api():
    """Function table of the CPU oracle (test infrastructure).  Built on demand with gcc."""
    from triton_racer_sim_amd import _ffi
    lib = os.path.join(ROOT, "oracle", "libtrsim_oracle.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return _ffi.Api(ctypes.CDLL(lib), "trso_")


def ensure_hip_library():
    """The in-tree HIP build is git-ignored; compile it on demand (hipcc cross-compiles gfx950 without a GPU)."""
    from triton_racer_sim_amd import _ffi
    if not os.path.exists(_ffi.HIP_LIB_PATH):
        import __graft_entry__
        __graft_entry__.build_hip()
    return _ffi.HIP_LIB_PATH


@pytest.fixture(scope="session")
def hip_api():
    """Function table of the HIP library; the GPU tests call the product through the C ABI."""
    from triton_racer_sim_amd import _ffi
    ensure_hip_library()
    return _ffi.load_hip_library()


@pytest.fixture(scope="session")
def hip_hooks_api():
    """Function table of the TEST build of the HIP library (csrc/libtrsim_testhooks.so: the product's sources + the two trs_resident_debug_* entry points)."""
    import ctypes
    from triton_racer_sim_amd import _ffi
    import __graft_entry__
    path = __graft_entry__.build_hip_testhooks()
    _ffi._share_torch_hip_runtime()
    return _ffi.Api(ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL), "trs_")


@pytest.fixture()
def make_env(oracle_api, request):
    """Factory for env pairs: make_env('hip', ...) / make_env('oracle', ...); make_env('hip_hooks', ...) binds the test build of the HIP library."""
    from triton_racer_sim_amd.env import BatchedEnv
    made = []

    def _make(kind, **kw):
        if kind != "oracle":
            ensure_hip_library()
        if kind == "hip_hooks":
            env = BatchedEnv(_api=request.getfixturevalue("hip_hooks_api"), **kw)
        else:
            env = BatchedEnv(_api=oracle_api, **kw) if kind == "oracle" else BatchedEnv(**kw)
        made.append(env)
        return env

    yield _make
    for env in made:
        env.close()


def track_points(name="generated"):
    return np.asarray(load_golden(f"track_{name}.json"), dtype=np.float64)
