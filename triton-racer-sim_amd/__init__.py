"""trsim-mi355x — batched vehicle-dynamics + camera env for MI355X behind Triton-Racer-Sim's Component API.

Import name: ``triton_racer_sim_amd`` (sources live in ``triton-racer-sim_amd/``).
Nothing here imports the CPU oracle; without the built HIP extension the env constructors raise.
"""
__version__ = "0.1.0"

__all__ = ["BatchedEnv", "HipGymInterface", "BatchedGymInterface", "LocationTracker", "Component", "DataPool", "Car", "Profiler"]


def __getattr__(name):  # lazy: importing the package must not need numpy/ctypes side effects
    if name == "BatchedEnv":
        from .env import BatchedEnv
        return BatchedEnv
    if name in ("HipGymInterface", "BatchedGymInterface", "LocationTracker"):
        from . import components
        return getattr(components, name)
    if name in ("Component", "DataPool", "Car", "Profiler"):
        from . import core
        return getattr(core, name)
    raise AttributeError(name)
