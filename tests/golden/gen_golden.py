#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE itself.

Runs only in the build container (needs /root/reference); the fixtures it writes are
committed, the reference never travels.  Nothing in tests/ or bench.py imports this file.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

Fixtures:
  G1 locate_<track>.json   LocationTracker (components/track_data_process.py:77-107) on
                           seeded query points -> integer index + 'loc/segment' float
  G2 car_trace.json        Car + DataPool ordering trace with probe components
                           (core/car.py:43-65, core/datapool.py:14-28)
  G3 datastorage_record.json  DataStorage record key order / file naming
                           (components/datastorage.py:13-33,67-79)
  G4 mapping.json          utils/mapping.py calcThrottle / calcBreak / three_segment_map
  G5 driver_assistance.json   components/driver_assistance.py:13-31
  G6 config_keys.json      core/config.py default dict (keys + JSON-able values)
  track_<name>.json        the recorded centre lines (car_templates/track_data/*.json), data only
"""
import io
import contextlib
import json
import os
import random
import sys
import tempfile
import time

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f)
    print("wrote", name)


def load_track(name):
    with open(os.path.join(REF, "TritonRacerSim/car_templates/track_data", name)) as f:
        return json.load(f)


def gen_locate(track_file, tag, n_random, seed):
    from TritonRacerSim.components.track_data_process import LocationTracker
    path = os.path.join(REF, "TritonRacerSim/car_templates/track_data", track_file)
    lt = LocationTracker(track_data_path=path)
    pts = lt.data
    n = len(pts)
    rng = random.Random(seed)
    queries = []
    # (a) on-track jitter
    for _ in range(n_random):
        p = pts[rng.randrange(n)]
        queries.append([p[0] + rng.uniform(-3, 3), p[1] + rng.uniform(-0.2, 0.2), p[2] + rng.uniform(-3, 3)])
    # (b) exact track points (every 7th) and all duplicate pairs
    for i in range(0, n, 7):
        queries.append(list(pts[i]))
    dups = [i for i in range(1, n) if pts[i] == pts[i - 1]]
    for i in dups[:200]:
        queries.append(list(pts[i]))
    # (c) far points: L1 >= 100 from everything -> index 0
    for _ in range(32):
        queries.append([rng.uniform(-3000, -2000), rng.uniform(-5, 5), rng.uniform(2000, 3000)])
    # (d) near-tie midpoints of consecutive distinct points, and of far-apart pairs
    for _ in range(256):
        i = rng.randrange(n - 1)
        a, b = pts[i], pts[i + 1]
        queries.append([(a[0] + b[0]) / 2, (a[1] + b[1]) / 2, (a[2] + b[2]) / 2])
    for _ in range(128):
        a, b = pts[rng.randrange(n)], pts[rng.randrange(n)]
        queries.append([(a[0] + b[0]) / 2, (a[1] + b[1]) / 2, (a[2] + b[2]) / 2])
    # (e) values that are exactly representable in binary32 (what the env feeds the tracker)
    import struct
    for _ in range(n_random // 2):
        p = pts[rng.randrange(n)]
        q = [p[0] + rng.uniform(-2, 2), p[1], p[2] + rng.uniform(-2, 2)]
        queries.append([struct.unpack("f", struct.pack("f", c))[0] for c in q])
    # (f) points hovering around L1 distance 100 from the nearest point
    for _ in range(64):
        p = pts[rng.randrange(n)]
        queries.append([p[0], p[1] + rng.uniform(99.0, 101.0) + 60.0, p[2]])
    idx, seg = [], []
    for q in queries:
        s = lt.step(*q)[0]
        seg.append(s)
        idx.append(int(round(s / 10.0 * n)))
    # recover the integer exactly through the private search (name-mangled)
    exact = [lt._LocationTracker__find_closest(q)[0] for q in queries]
    assert exact == idx, "segment->index round trip disagrees"
    dump(f"locate_{tag}.json", {"track": f"track_{tag}.json", "n_points": n, "queries": queries, "idx": exact, "segment": seg})


def gen_car_trace():
    from TritonRacerSim.components.component import Component
    from TritonRacerSim.core.car import Car

    trace = []

    class Stop(Exception):
        pass

    class A(Component):
        def __init__(self):
            super().__init__(inputs=["b/out"], outputs=["a/out"])
            self.k = 0

        def step(self, *args):
            trace.append(["A", args[0]])
            self.k += 1
            if self.k > 4:
                raise KeyboardInterrupt
            return (self.k * 10,)

        def getName(self):
            return "A"

    class B(Component):
        def __init__(self):
            super().__init__(inputs=["a/out"], outputs=["b/out"])
            self.k = 0

        def step(self, *args):
            trace.append(["B", args[0]])
            self.k += 1
            if self.k == 2:
                return None          # datapool.py:22 -> no write
            return (self.k,)

        def onShutdown(self):
            trace.append(["B.onShutdown", None])

        def getName(self):
            return "B"

    with contextlib.redirect_stdout(io.StringIO()):
        car = Car(loop_hz=1e9)
        car.addComponent(A())
        car.addComponent(B())
        car.start()
    dump("car_trace.json", {"trace": trace, "pool": car.pool.pool})


def gen_datastorage():
    import numpy as np
    from PIL import Image
    from TritonRacerSim.components.datastorage import DataStorage
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "records_1/")
    with contextlib.redirect_stdout(io.StringIO()):
        ds = DataStorage(storage_path=path)
        img = (np.arange(120 * 160 * 3, dtype=np.uint32) % 251).astype(np.uint8).reshape(120, 160, 3)
        # (cam/img, mux/throttle, mux/steering, mux/break, gym/speed, loc/segment, gym/x, gym/y, gym/z, gym/cte, del, toggle)
        ds.step(img, 0.5, -0.25, None, 3.5, 1.25, 47.5, 0.56, 46.7, 0.125, False, True)
        ds.step(img, 0.6, 0.25, None, 3.6, 1.5, 47.6, 0.56, 46.8, -0.125, False, True)
        for _ in range(400):
            if len(os.listdir(path)) >= 4:
                break
            time.sleep(0.01)
        time.sleep(0.05)
        ds.onShutdown()
    files = sorted(os.listdir(path))
    recs = {}
    for f in files:
        if f.endswith(".json"):
            with open(os.path.join(path, f)) as fh:
                raw = fh.read()
            recs[f] = {"raw": raw, "keys": list(json.loads(raw).keys())}
    im = Image.open(os.path.join(path, "img_0.jpg"))
    dump("datastorage_record.json", {"step_inputs": ds.step_inputs, "files": files, "records": recs,
                                     "jpeg_size": list(im.size), "jpeg_mode": im.mode})


def gen_mapping():
    from TritonRacerSim.utils import mapping
    rng = random.Random(11)
    thr, brk, tsm = [], [], []
    for _ in range(200):
        c, p, m = rng.uniform(0, 25), rng.uniform(0, 25), rng.choice([0.5, 1.0, 1.5])
        thr.append([c, p, m, mapping.calcThrottle(c, p, m)])
        brk.append([c, p, m, mapping.calcBreak(c, p, m)])
    for c, p, m in [(5, 6, 1), (5, 3, 1), (5, 4.95, 1), (5, 5, 1), (0, 0, 1)]:
        thr.append([c, p, m, mapping.calcThrottle(c, p, m)])
        brk.append([c, p, m, mapping.calcBreak(c, p, m)])
    for _ in range(100):
        v = rng.uniform(-1.5, 1.5)
        tsm.append([v, 330, 370, 400, mapping.three_segment_map(v, 330, 370, 400)])
    tsm.append([0, 330, 370, 400, mapping.three_segment_map(0, 330, 370, 400)])
    dump("mapping.json", {"calcThrottle": thr, "calcBreak": brk, "three_segment_map": tsm})


def gen_driver_assistance():
    from TritonRacerSim.components.driver_assistance import DriverAssistance
    rng = random.Random(5)
    out = {}
    for mode in ("steering", "speed"):
        da = DriverAssistance({"drive_assist_limit_mode": mode, "drive_assist_limit_k": 5})
        rows = []
        cases = [(rng.uniform(-1, 1), rng.uniform(-1, 1), rng.choice([0.0, 0.3]), rng.uniform(0, 25)) for _ in range(200)]
        cases += [(0.5, 0.5, None, 10.0), (0.5, 0.5, 0.0, 0), (0.0, 0.5, 0.0, 12.0), (None, 0.1, 0.0, 3.0), (0.9, 1.0, 0.0, 20.0), (-0.9, 1.0, 0.0, 20.0)]
        for c in cases:
            rows.append([list(c), list(da.step(*c))])
        out[mode] = rows
    dump("driver_assistance.json", out)


def gen_config():
    from TritonRacerSim.core.config import config
    dump("config_keys.json", {"keys": list(config.keys()), "values": json.loads(json.dumps(config))})


def main():
    for fname, tag in (("generated_track.json", "generated"), ("mountain_track.json", "mountain")):
        dump(f"track_{tag}.json", load_track(fname))
    gen_locate("generated_track.json", "generated", 4096, 1234)
    gen_locate("mountain_track.json", "mountain", 1024, 4321)
    gen_car_trace()
    gen_datastorage()
    gen_mapping()
    gen_driver_assistance()
    gen_config()


if __name__ == "__main__":
    main()
