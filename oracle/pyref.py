"""pyref.py — pure-Python restatement of what the REFERENCE executes per tick.  TEST INFRASTRUCTURE ONLY
(imported by tests/ and by bench.py's cpu_baseline leg; never by the product).

Restates, for one car:
  * the nearest-point search of LocationTracker
    (/root/reference/TritonRacerSim/components/track_data_process.py:89-104, map :106-107) — pinned by
    tests/golden/locate_*.json (outputs of the reference itself);
  * the sequential tick of Car.start over a dict blackboard
    (core/car.py:45-53, core/datapool.py:14-28) — pinned by tests/golden/car_trace.json.
"""
import json
import os
import time

_HERE = os.path.dirname(os.path.abspath(__file__))
_TRACK = os.path.join(_HERE, "..", "tests", "golden", "track_generated.json")


def find_closest(data, point):
    """Index of the L1-nearest sample; best starts at 100, strict '<' (first minimum wins)."""
    selected, best = 0, 100
    for i, c in enumerate(data):
        d = abs(point[0] - c[0]) + abs(point[1] - c[1]) + abs(point[2] - c[2])
        if d < best:
            selected, best = i, d
    return selected


def segment_of(idx, n, lo=0, hi=10):
    return idx / float(n) * (hi - lo) + lo


class _Part:
    def __init__(self, inputs, outputs, fn):
        self.step_inputs, self.step_outputs, self.fn = list(inputs), list(outputs), fn


def tick(pool, parts):
    """One pass of the reference's loop body: fetch inputs in declared order, step, store positionally;
    a None return stores nothing."""
    for part in parts:
        args = tuple(pool[name] for name in part.step_inputs)
        out = part.fn(*args)
        if out is not None:
            for k, name in enumerate(part.step_outputs):
                pool[name] = out[k]


def time_reference_loop(budget_s=2.0):
    """Ticks/s of [trivial sim bridge -> LocationTracker] for one car with the 20 Hz sleep removed:
    the in-repo per-tick Python cost of the reference (~323 us per tracker call)."""
    with open(_TRACK) as f:
        data = json.load(f)
    n = len(data)
    state = {"k": 0}

    def bridge(steering, throttle, breaking, reset):
        state["k"] = (state["k"] + 7) % n
        p = data[state["k"]]
        return None, p[0] + 0.01, p[1], p[2] - 0.01, 1.0, 0.0

    def tracker(x, y, z):
        return (segment_of(find_closest(data, (x, y, z)), n),)

    parts = [
        _Part(["mux/steering", "mux/throttle", "mux/breaking", "usr/reset"],
              ["cam/img", "gym/x", "gym/y", "gym/z", "gym/speed", "gym/cte"], bridge),
        _Part(["gym/x", "gym/y", "gym/z"], ["loc/segment"], tracker),
    ]
    pool = {}
    for part in parts:
        for name in part.step_inputs + part.step_outputs:
            pool[name] = None
    pool["gym/x"], pool["gym/y"], pool["gym/z"] = 0.0, 0.0, 0.0
    ticks, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        tick(pool, parts)
        ticks += 1
    return ticks / (time.perf_counter() - t0)


class MuxModel:
    """Event-queue model of ONE ControlMultiplexer
    (/root/reference/TritonRacerSim/components/controlmultiplexer.py:24-70) under ideal timing: tick k happens at
    time k / hz, a lock-end thread started at tick k clears its flag at tick k + ticks, before that tick's step.
    Independent of the C oracle's bookkeeping (a heap of end events instead of a trigger ring).  Parity unpinned
    (the reference file needs pygame to import)."""
    HUMAN, AI_STEERING, AI = 0, 1, 2

    def __init__(self, throttle=(False, 1.0, 100), steering=(False, 0.0, 60)):
        self.thr_en, self.thr_val, self.thr_ticks = throttle
        self.st_en, self.st_val, self.st_ticks = steering
        self.last_mode = self.HUMAN
        self.throttle_lock_active = False
        self.steering_lock_active = False
        self.events = []            # (tick, which)
        self.now = 0

    def step(self, mode, usr, ai, keep):
        """usr / ai: (steering, throttle, breaking); keep: what the pool holds (returned for an unknown mode)."""
        import heapq
        while self.events and self.events[0][0] <= self.now:
            _, which = heapq.heappop(self.events)
            if which == "t":
                self.throttle_lock_active = False
            else:
                self.steering_lock_active = False
        out = None
        if mode == self.HUMAN:
            out = (usr[0], usr[1], usr[2])
        elif mode == self.AI_STEERING:
            out = (ai[0], usr[1], usr[2])
        elif mode == self.AI:
            out = (ai[0], ai[1], ai[2])
        if self.last_mode != self.AI and mode == self.AI:
            if self.thr_en:
                self.throttle_lock_active = True
                heapq.heappush(self.events, (self.now + self.thr_ticks, "t"))
            if self.st_en:
                self.steering_lock_active = True
                heapq.heappush(self.events, (self.now + self.st_ticks, "s"))
        if out is not None:
            if self.steering_lock_active:
                out = (self.st_val, out[1], out[2])
            if self.throttle_lock_active:
                out = (out[0], self.thr_val, out[2])
        self.last_mode = mode
        self.now += 1
        return keep if out is None else out
