"""The CPU oracle (and the pure-Python restatement) against vectors captured from the REFERENCE itself
(tests/golden/gen_golden.py).  This is what pins the oracle's nearest-point search."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden, track_points

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyref  # noqa: E402


@pytest.mark.parametrize("track", ["generated", "mountain"])
def test_oracle_locate_matches_reference(make_env, track):
    g1 = load_golden(f"locate_{track}.json")
    env = make_env("oracle", n_envs=1, track=track_points(track), render=False)
    idx = env.locate(g1["queries"])
    assert np.array_equal(idx, np.asarray(g1["idx"], dtype=np.int32))
    assert np.array_equal(env.segment(idx), np.asarray(g1["segment"]))        # idx / n * 10, bit for bit
    assert g1["n_points"] == len(track_points(track))


def test_golden_covers_the_edge_cases():
    g1 = load_golden("locate_generated.json")
    pts = track_points("generated")
    idx, q = np.asarray(g1["idx"]), np.asarray(g1["queries"])
    d = np.abs(q - pts[idx]).sum(-1)
    assert ((idx == 0) & (d >= 100)).sum() >= 32                      # "lost" queries fall back to index 0
    dup = [i for i in range(1, len(pts)) if np.array_equal(pts[i], pts[i - 1])]
    assert len(dup) == 268                                            # 268 zero-length steps
    asked = {tuple(p) for p in q.tolist()}
    dup_asked = [i for i in dup if tuple(pts[i]) in asked]
    assert len(dup_asked) >= 100                                      # duplicate pairs are queried exactly ...
    answers = {tuple(qq): int(i) for qq, i in zip(q.tolist(), idx.tolist())}
    assert all(answers[tuple(pts[i])] < i for i in dup_asked)         # ... and resolve to the EARLIER sample (strict '<')


def test_pyref_restatement_matches_reference_sample():
    g1 = load_golden("locate_generated.json")
    pts = load_golden("track_generated.json")
    for k in range(0, len(g1["queries"]), 97):
        assert pyref.find_closest(pts, g1["queries"][k]) == g1["idx"][k]
        assert pyref.segment_of(g1["idx"][k], len(pts)) == g1["segment"][k]


def test_pyref_tick_matches_car_trace():
    """oracle/pyref.tick (the CPU-baseline loop) reproduces the reference Car/DataPool trace G2."""
    g2 = load_golden("car_trace.json")
    trace = []
    state = {"a": 0, "b": 0}

    def fa(x):
        trace.append(["A", x]); state["a"] += 1
        return (state["a"] * 10,)

    def fb(x):
        trace.append(["B", x]); state["b"] += 1
        return None if state["b"] == 2 else (state["b"],)

    parts = [pyref._Part(["b/out"], ["a/out"], fa), pyref._Part(["a/out"], ["b/out"], fb)]
    pool = {"a/out": None, "b/out": None}
    for _ in range(4):
        pyref.tick(pool, parts)
    ref = [t for t in g2["trace"] if t[0] in ("A", "B")][:8]
    assert trace == ref
