"""Import name of the package whose sources live in ``triton-racer-sim_amd/``.

The directory name required by the build layout contains a hyphen, which Python cannot
import; this shim points the package search path at it so that
``import triton_racer_sim_amd`` (and ``triton_racer_sim_amd.env`` etc.) resolve there.
"""
import os as _os

_SRC = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "triton-racer-sim_amd")
__path__.insert(0, _SRC)

with open(_os.path.join(_SRC, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_SRC, "__init__.py"), "exec"))
del _f
