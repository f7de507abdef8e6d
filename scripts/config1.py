#!/usr/bin/env python3
"""BASELINE config 1: ONE env through the Component / DataPool / Car loop (the reference's own shape), frame copied
to the host every tick.  Reports ticks/s with the sleep disabled and the achieved rate at the reference's 20 Hz."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from triton_racer_sim_amd.components import HipGymInterface, LocationTracker
from triton_racer_sim_amd.core import Car, Component


class Const(Component):
    def __init__(self):
        super().__init__(outputs=["mux/steering", "mux/throttle", "mux/breaking", "usr/reset"])
    def step(self, *a):
        return 0.05, 0.5, None, False


def run(loop_hz, ticks, resident=False, tracker=True):
    car = Car(loop_hz=loop_hz, verbose=False)
    gym = HipGymInterface(gym_config={"scene_name": "generated_track", "hip_resident": resident, "hip_resident_idle_us": 100000})
    for part in (Const(), gym) + ((LocationTracker("track_data/generated_track.json"),) if tracker else ()):
        car.addComponent(part)
    car.tick()
    t0 = time.perf_counter()
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        car.start(max_ticks=ticks + 1)
    dt = time.perf_counter() - t0
    return ticks / dt, car.pool.get_value("cam/img").shape, type(car.pool.get_value("gym/x")).__name__


r, shape, ty = run(1e9, 5000)
print(f"config 1, sleep disabled: {r:.0f} ticks/s (frame {shape} uint8 to host + 6 Python {ty}s + loc/segment per tick)")
print(f"  PCIe-inclusive image rate: {r * 57600 / 1e6:.1f} MB/s (latency-bound: ctypes + hipMemcpy round trips, not the 63 GB/s link)")
for res in (False, True):
    r2, _, _ = run(1e9, 5000, resident=res, tracker=False)
    print(f"config 1 without the LocationTracker part (its index already comes with the step), {'resident worker' if res else 'launch per tick'}: {r2:.0f} ticks/s")
r3, _, _ = run(1e9, 5000, resident=True)
print(f"config 1 as above with all three parts, resident worker: {r3:.0f} ticks/s")
r20, _, _ = run(20, 60)
print(f"config 1, 20 Hz pacing (car_templates/manage.py:38): {r20:.2f} ticks/s")
