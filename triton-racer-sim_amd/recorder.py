"""Record writer with the reference's tub layout (``components/datastorage.py:13-33,67-79``).

One record per recorded tick: ``record_{k}.json`` + ``img_{k}.jpg`` with ``k`` starting at 0
(``datastorage.py:100,111-112``; the reference's loaders start reading at 1 — ``keras_train.py:36`` — a
quirk of the reference that is preserved, not fixed).  The JSON holds the stored port values in declared
order followed by ``usr/del_record`` and ``usr/toggle_record``; the image value is replaced by its file
name (``:76-79``).  Key quirk kept: the default list stores ``mux/break`` although the multiplexer writes
``mux/breaking`` (``datastorage.py:13`` vs ``controlmultiplexer.py:9``), so it records ``null`` unless a
DriverAssistance part runs.

Differences that do not change the files: values are passed through ``float()`` / ``int()`` when they are
numpy scalars (``json.dump`` of ``np.float32`` raises in the reference, ``gyminterface.py:100-104`` avoids
it the same way), and ``onShutdown`` drains the queue instead of abandoning it.
"""
import json
import os
import queue
import threading

import numpy as np

from .core import Component

DEFAULT_TO_STORE = ["cam/img", "mux/throttle", "mux/steering", "mux/break", "gym/speed", "loc/segment",
                    "gym/x", "gym/y", "gym/z", "gym/cte"]


def _plain(v):
    if isinstance(v, np.generic):
        return v.item()
    return v


class DataStorage(Component):
    def __init__(self, to_store=None, storage_path=None):
        Component.__init__(self, inputs=list(DEFAULT_TO_STORE if to_store is None else to_store), threaded=False)
        self.step_inputs += ["usr/del_record", "usr/toggle_record"]
        if storage_path is None:
            raise ValueError("storage_path is required (the reference derives it from sys.path[0]/data/records_N)")
        self.storage_path = storage_path
        os.mkdir(self.storage_path)
        self.count = 0
        self._written = 0
        self._q = queue.Queue()
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._io_loop, daemon=True)
        self._thread.start()

    def step(self, *args):
        if args[-2]:                                   # usr/del_record: forget the last 100 (datastorage.py:27-28,82-87)
            self.count = max(self.count - 100, 0)
            self._q.put(("rewind", self.count))
        elif args[-1]:                                 # usr/toggle_record
            record = {name: args[i] for i, name in enumerate(self.step_inputs)}
            self._q.put(("store", record))
            self.count += 1

    def _store(self, k, record):
        key = "cam/img" if "cam/img" in record else "cam/processed_img"
        img = record.get(key)
        if img is not None:
            from PIL import Image
            Image.fromarray(np.asarray(img)).save(os.path.join(self.storage_path, f"img_{k}.jpg"))
            record[key] = f"img_{k}.jpg"
        with open(os.path.join(self.storage_path, f"record_{k}.json"), "w") as f:
            json.dump({name: _plain(v) for name, v in record.items()}, f)

    def _io_loop(self):
        while True:
            try:
                kind, payload = self._q.get(timeout=0.05)
            except queue.Empty:
                if self._stop.is_set():
                    return
                continue
            if kind == "rewind":
                self._written = payload
            else:
                self._store(self._written, payload)
                self._written += 1

    def flush(self):
        while not self._q.empty():
            threading.Event().wait(0.005)

    def onShutdown(self):
        self.flush()
        self._stop.set()
        self._thread.join(timeout=5)
        if not os.listdir(self.storage_path):          # datastorage.py:45-47
            os.rmdir(self.storage_path)

    def getName(self):
        return "Data Storage"


class BatchedDataStorage(Component):
    """N cars per tick -> N tubs ``<root>/records_{i}/`` (``i`` from 1 like the reference's folder numbering,
    ``datastorage.py:60-65``), each with the single-car layout above, so every tub loads with the reference's
    ``DataLoader`` classes.  Ports carry arrays of length N (``cam/img``: ``uint8[N,H,W,3]``; ``None`` allowed);
    ``usr/del_record`` / ``usr/toggle_record`` are scalars or per-car arrays."""

    def __init__(self, n_cars, to_store=None, storage_root=None):
        Component.__init__(self, inputs=list(DEFAULT_TO_STORE if to_store is None else to_store), threaded=False)
        self.step_inputs += ["usr/del_record", "usr/toggle_record"]
        if storage_root is None:
            raise ValueError("storage_root is required")
        os.makedirs(storage_root, exist_ok=True)
        self.n = int(n_cars)
        self.tubs = [DataStorage(to_store=self.step_inputs[:-2], storage_path=os.path.join(storage_root, f"records_{i + 1}")) for i in range(self.n)]

    def step(self, *args):
        def car(v, i):
            if v is None or np.isscalar(v) or isinstance(v, (bool, str)):
                return v
            return v[i]
        for i, tub in enumerate(self.tubs):
            tub.step(*[car(a, i) for a in args])

    def onShutdown(self):
        for tub in self.tubs:
            tub.onShutdown()

    def getName(self):
        return "Batched Data Storage"


# ---- readers: the file walk and label / feature choices of the reference's loaders ------------------------------
# (components/keras_train.py:33-57 DataLoader.load; :107-119 default names / labels / features; :264-299 variants)
LOADERS = {
    # name: (labels(record), features(record) or None)
    "default": (lambda r: (r["mux/steering"], r["mux/throttle"]), None),                                     # :113-119
    "speed_feature": (lambda r: (r["mux/steering"], r["mux/throttle"]), lambda r: (r["gym/speed"] / 20,)),   # :263-268
    "speed_ctl": (lambda r: (r["mux/steering"], r["gym/speed"] / 20), None),                                 # :270-275
    "full_house": (lambda r: (r["mux/steering"], r["gym/speed"] / 20),
                   lambda r: (r["gym/speed"] / 20, r["loc/segment"])),                                       # :288-297
}


def load_records(paths, loader="speed_ctl"):
    """What ``DataLoader.load`` collects before its train/validation split (``keras_train.py:33-57``): for every tub,
    records ``i = 1, 2, ...`` until ``img_{i}.jpg`` or ``record_{i}.json`` is missing — the walk starts at 1, so the
    writer's record 0 is never read (reference quirk, kept).  Returns ``(images float32[n,H,W,3] in [0,1],
    features float32[n,F] (F = 0 without features), labels float32[n,L])``."""
    from PIL import Image
    labels_of, features_of = LOADERS[loader]
    paths = [paths] if isinstance(paths, (str, os.PathLike)) else list(paths)
    for p in paths:
        if not os.path.exists(p):
            raise FileNotFoundError(f"Folder does not exists: {p}")                    # keras_train.py:26-28
    imgs, feats, labels = [], [], []
    for p in paths:
        i = 1
        while True:
            try:
                img = np.asarray(Image.open(os.path.join(p, f"img_{i}.jpg")), dtype=np.float32)
                img /= 255
                with open(os.path.join(p, f"record_{i}.json")) as f:
                    record = json.load(f)
            except FileNotFoundError:
                break
            imgs.append(img)
            labels.append(np.asarray(labels_of(record), dtype=np.float32))
            feats.append(np.asarray(features_of(record) if features_of else (), dtype=np.float32))
            i += 1
    if not imgs:
        return np.zeros((0, 0, 0, 3), np.float32), np.zeros((0, 0), np.float32), np.zeros((0, 2), np.float32)
    return np.stack(imgs), np.stack(feats), np.stack(labels)


class TrackDataProcessor:
    """Tub -> centre-line file (reference ``components/track_data_process.py:9-39``): walks ``record_1.json, record_2.json, ...``
    (from 1 — ``record_0.json`` is skipped exactly as the reference skips it) until the first missing file, collects
    ``[gym/x, gym/y, gym/z]`` per record and dumps the list as JSON: the format ``LocationTracker`` / ``trs_load_track`` load
    (``car_templates/track_data/*.json``).  Closes the loop for this build: drive (``HipGymInterface``), record
    (``DataStorage``), process, then load the result as a track."""

    def __init__(self, tub_path, output_path):
        self.tub_path = tub_path
        self.output_path = output_path
        if not os.path.exists(tub_path):
            raise FileNotFoundError("Cannot find tub {}".format(tub_path))
        self.line = []

    def process(self, verbose=True):
        i = 1
        while True:
            try:
                with open(os.path.join(self.tub_path, "record_{}.json".format(i))) as f:
                    data = json.load(f)
            except FileNotFoundError:
                break
            self.line.append([data["gym/x"], data["gym/y"], data["gym/z"]])
            i += 1
        if verbose:
            print(i, "points loaded, Saving to ", self.output_path)       # (the reference prints the index of the first missing record)
        with open(self.output_path, "w") as output_file:
            json.dump(self.line, output_file)
        return self.line
