"""Vectorised control-side glue for N cars (numpy, elementwise) with the reference's scalar semantics.

* ``calc_throttle`` / ``calc_break`` — the speed controller behind ``cnn_2d_speed_control``
  (``utils/mapping.py:23-35``, used at ``components/keras_pilot.py:86-90``);
* ``three_segment_map`` — ``utils/mapping.py:9-16``;
* ``driver_assistance`` — ``components/driver_assistance.py:13-31`` (both limit modes, ``None`` pass-through).

Pinned by fixtures G4 / G5 (tests/golden/mapping.json, driver_assistance.json), captured from the reference.
"""
import math

import numpy as np


def calc_throttle(current_spd, predicted_spd, multiplier):
    delta = np.asarray(predicted_spd, dtype=np.float64) - np.asarray(current_spd, dtype=np.float64)
    thr = np.asarray(multiplier, dtype=np.float64) * np.arctan(delta * 2) / (math.pi / 2)
    return np.where((thr > -0.2) & (thr < 0.0), 0.0, thr)


def calc_break(current_spd, predicted_spd, multiplier):
    delta = np.asarray(predicted_spd, dtype=np.float64) - np.asarray(current_spd, dtype=np.float64)
    brk = -1.0 * np.asarray(multiplier, dtype=np.float64) * np.arctan(delta * 1.0) / (math.pi / 2)
    return np.where(brk < 0.4, 0.0, brk)


def three_segment_map(val, min_map, mid_map, max_map):
    v = np.clip(np.asarray(val, dtype=np.float64), -1, 1)
    return np.where(v == 0, mid_map, np.where(v < 0, mid_map + (mid_map - min_map) * v, mid_map + (max_map - mid_map) * v))


def driver_assistance(steering, throttle, breaking, speed, mode="steering", k=5):
    """Returns ``(steering, throttle, breaking)`` arrays.  ``None`` in any argument passes everything through
    unchanged, like the reference's ``if None not in args`` guard."""
    if steering is None or throttle is None or breaking is None or speed is None:
        return steering, throttle, breaking
    st = np.array(steering, dtype=np.float64, copy=True, ndmin=1)
    th = np.array(throttle, dtype=np.float64, copy=True, ndmin=1)
    br = np.array(breaking, dtype=np.float64, copy=True, ndmin=1)
    sp = np.asarray(speed, dtype=np.float64).reshape(-1)
    st, th, br, sp = np.broadcast_arrays(st, th, br, sp)
    st, th, br = st.copy(), th.copy(), br.copy()
    if mode == "steering":
        ok = sp != 0
        with np.errstate(divide="ignore"):
            lim = np.where(ok, k / np.where(ok, sp, 1.0), np.inf)
        hi = ok & (st > lim)
        lo = ok & ~hi & (st < -lim)
        st = np.where(hi, lim, np.where(lo, -lim, st))
        th = np.where(hi | lo, -0.1, th)
    elif mode == "speed":
        ok = st != 0
        lim = np.where(ok, k / np.where(ok, st, 1.0), np.inf)
        over = ok & (sp > lim)
        th = np.where(over, 0.0, th)
        br = np.where(over, 0.0, br)
    return st, th, br
