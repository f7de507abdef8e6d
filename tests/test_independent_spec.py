"""Independent restatements of the two build-defined parts the oracle and the product used to share line for line
(VERDICT r02, weak 1-2): the class map and the physics step.  Both are restated here in vectorised numpy FROM THE TEXT of
include/trsim_spec.h — a different formulation on purpose (brute force over all segments, no reach pruning, no running
minimum; physics vectorised over envs with every branch as a mask) — and compared with the oracle's results and with the
product's host tables (csrc/trsim_tables.cpp through tests/host_tables_driver.cpp).  CPU only."""
import ctypes as C
import math
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT, track_points
from test_oracle_spec import np_sincos

F = np.float32


# ---------------------------------------------------------------------------------------------------------------- class map

def spec_grid(pts, margin=2.5, budget=96 * 1024, cell_min=0.125):
    """include/trsim_spec.h "class map", grid: cells of 0.125 * 2^k, the smallest k whose packed 2-bit map fits the budget."""
    q = dedup_polyline(pts)
    xmin, zmin = q.min(0)
    xmax, zmax = q.max(0)
    cell = cell_min
    while True:
        x0 = math.floor((xmin - margin) / cell) * cell
        z0 = math.floor((zmin - margin) / cell) * cell
        gw = math.ceil((xmax + margin - x0) / cell)
        gh = math.ceil((zmax + margin - z0) / cell)
        if ((gw + 15) // 16) * 4 * gh <= budget:
            return cell, x0, z0, gw, gh
        cell *= 2.0


def dedup_polyline(pts):
    """Raw points -> closed polyline in (x, z): consecutive equal points dropped, a closing duplicate of the first dropped."""
    xz = np.asarray(pts, dtype=np.float64)[:, [0, 2]]
    keep = np.ones(len(xz), bool)
    keep[1:] = (xz[1:] != xz[:-1]).any(1)
    q = xz[keep]
    if len(q) > 1 and (q[-1] == q[0]).all():
        q = q[:-1]
    return q


def brute_force_classes(pts, cells_ix, cells_iz, cell, x0, z0, road_half=2.0, edge_half=0.10, centre_half=0.075, dash_period=3.0, dash_on=1.5):
    """Class of each sampled cell by distance and arc length over ALL segments of the de-duplicated closed polyline.  Returns
    (class, margin): margin = distance of d (or of the dash phase) to the nearest class threshold, so that cells within rounding
    noise of a boundary can be told apart from real disagreements."""
    q = dedup_polyline(pts)
    a, b = q, np.roll(q, -1, axis=0)
    ab = b - a
    seg_len = np.hypot(ab[:, 0], ab[:, 1])
    s0 = np.concatenate([[0.0], np.cumsum(seg_len)[:-1]])
    c = np.stack([x0 + (cells_ix + 0.5) * cell, z0 + (cells_iz + 0.5) * cell], 1)          # [n, 2] cell centres
    cls = np.zeros(len(c), np.int64)
    margin = np.full(len(c), np.inf)
    for lo in range(0, len(c), 512):                                                       # [512, segments] blocks
        p = c[lo:lo + 512, None, :]
        t = np.clip(((p - a[None]) * ab[None]).sum(-1) / (seg_len ** 2)[None], 0.0, 1.0)
        foot = a[None] + t[..., None] * ab[None]
        d = np.hypot(*(np.moveaxis(p - foot, -1, 0)))
        k = d.argmin(1)                                                                    # first minimum = lowest segment index
        rows = np.arange(len(k))
        dmin = d[rows, k]
        arc = s0[k] + t[rows, k] * seg_len[k]
        phase = np.fmod(arc, dash_period)
        centre = (dmin <= centre_half) & (phase < dash_on)
        edge = ~centre & (np.abs(dmin - road_half) <= edge_half)
        road = ~centre & ~edge & (dmin < road_half)
        cls[lo:lo + 512] = np.where(centre, 3, np.where(edge, 2, np.where(road, 1, 0)))
        m = np.minimum.reduce([np.abs(dmin - centre_half), np.abs(np.abs(dmin - road_half) - edge_half), np.abs(dmin - road_half)])
        on_line = dmin <= centre_half + 1e-9
        m = np.where(on_line, np.minimum.reduce([m, np.abs(phase - dash_on), phase, dash_period - phase]), m)
        # two segments at (nearly) the same distance may carry different arc lengths: the choice is then a rounding matter
        d2 = d.copy(); d2[rows, k] = np.inf
        tie = (d2.min(1) - dmin < 1e-9) & on_line
        margin[lo:lo + 512] = np.where(tie, 0.0, m)
    return cls, margin


def unpack(map_words, ix, iz):
    return (map_words[iz, ix >> 4] >> ((ix & 15) * 2).astype(np.uint32)) & 3


def sample_cells(pts, cell, x0, z0, gw, gh, n_random, n_near, seed):
    """Cells all over the map plus cells scattered around the track (where every class occurs)."""
    rng = np.random.default_rng(seed)
    ix = rng.integers(0, gw, n_random)
    iz = rng.integers(0, gh, n_random)
    xz = np.asarray(pts)[:, [0, 2]]
    base = xz[rng.integers(0, len(xz), n_near)] + rng.uniform(-2.6, 2.6, (n_near, 2))
    ix = np.concatenate([ix, np.clip(np.floor((base[:, 0] - x0) / cell).astype(np.int64), 0, gw - 1)])
    iz = np.concatenate([iz, np.clip(np.floor((base[:, 1] - z0) / cell).astype(np.int64), 0, gh - 1)])
    return ix, iz


@pytest.fixture(scope="module")
def tables_driver(tmp_path_factory):
    if not shutil.which("g++"):
        pytest.skip("g++ not available")
    exe = tmp_path_factory.mktemp("host_tables_plain") / "driver"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-I", os.path.join(ROOT, "include"), "-o", str(exe),
                           os.path.join(ROOT, "tests", "host_tables_driver.cpp"), os.path.join(ROOT, "triton-racer-sim_amd", "csrc", "trsim_tables.cpp")])
    return str(exe)


@pytest.mark.parametrize("track", ["generated", "mountain"])
def test_class_map_against_a_brute_force_restatement(make_env, oracle_api, tables_driver, tmp_path, track):
    from triton_racer_sim_amd import _ffi
    pts = track_points(track)
    env = make_env("oracle", n_envs=1, track=pts, render=True)
    mi = env.map_info
    cell, x0, z0, gw, gh = spec_grid(pts)
    assert (mi.cell, mi.x0, mi.z0, mi.map_w, mi.map_h, mi.map_words) == (cell, x0, z0, gw, gh, (gw + 15) // 16)   # the grid itself
    oracle_map = env.fetch("map").reshape(gh, -1)
    # the product's host tables (no GPU needed): csrc/trsim_tables.cpp through the test driver
    cfg = _ffi.TrsConfig()
    oracle_api.default_config(C.byref(cfg))
    cfg.n_envs = 1
    (tmp_path / "cfg.bin").write_bytes(bytes(cfg))
    (tmp_path / "pts.bin").write_bytes(np.ascontiguousarray(pts, dtype=np.float64).tobytes())
    out = subprocess.run([tables_driver, str(tmp_path / "cfg.bin"), str(tmp_path / "pts.bin"), str(tmp_path / "t")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    product_map = np.fromfile(tmp_path / "t.map", dtype=np.uint32).reshape(gh, -1)

    ix, iz = sample_cells(pts, cell, x0, z0, gw, gh, n_random=3000, n_near=5000, seed=17 if track == "generated" else 18)
    want, margin = brute_force_classes(pts, ix, iz, cell, x0, z0)
    border = (ix == 0) | (ix == gw - 1) | (iz == 0) | (iz == gh - 1)
    want = np.where(border, 0, want)                                    # the map border is class 0 (lookups outside clamp to it)
    sure = border | (margin > 1e-7)                                     # cells farther than rounding noise from every class threshold
    assert sure.mean() > 0.995
    for name, m in (("oracle", oracle_map), ("product host tables", product_map)):
        got = unpack(m, ix, iz)
        wrong = np.nonzero(sure & (got != want))[0]
        assert wrong.size == 0, f"{name}: {wrong.size} of {sure.sum()} cells differ, e.g. cell ({ix[wrong[0]]}, {iz[wrong[0]]}): got {got[wrong[0]]}, want {want[wrong[0]]}"
    counts = np.bincount(want, minlength=4)
    assert (counts > 50).all(), counts                                  # every class was really sampled (grass, road, edge line, centre dashes)


# ---------------------------------------------------------------------------------------------------------------- physics

def np_sincos_vec(a):
    out = [np_sincos(v) for v in np.asarray(a, dtype=F)]
    return np.array([o[0] for o in out], dtype=F), np.array([o[1] for o in out], dtype=F)


def spec_tangents(pts):
    """Unit tangent at each raw point: next distinct minus previous distinct point in (x, z), closed loop; binary64, stored binary32."""
    x, z = pts[:, 0], pts[:, 2]
    n = len(pts)
    tang = np.zeros((n, 2), F)
    for i in range(n):
        j = i
        while True:
            j = (j + 1) % n
            if x[j] != x[i] or z[j] != z[i]:
                break
        b = i
        while True:
            b = (b - 1) % n
            if x[b] != x[i] or z[b] != z[i]:
                break
        tx, tz = x[j] - x[b], z[j] - z[b]
        ln = math.sqrt(tx * tx + tz * tz)
        if ln == 0.0:
            tx, tz = x[j] - x[i], z[j] - z[i]
            ln = math.sqrt(tx * tx + tz * tz)
        tang[i] = (tx / ln, tz / ln)
    return tang


def numpy_step(cfg, pts, tang, st, steer, thr, brk, reset_in, env_id_base=0):
    """One env step of include/trsim_spec.h for all envs at once, every branch as a mask; float32 arithmetic operation by
    operation (numpy float32 ops round like C's), the nearest point in binary64."""
    n = len(st["x"])
    dt, pi, two_pi = F(cfg.dt), F(3.14159274101257324), F(6.28318548202514648)
    do_reset = reset_in | (bool(cfg.auto_reset) & (st["done"] != 0))
    steer = np.clip(steer.astype(F), F(-1), F(1)); thr = np.clip(thr.astype(F), F(-1), F(1)); brk = np.clip(brk.astype(F), F(0), F(1))
    sd, cd = np_sincos_vec(steer * F(cfg.max_steer))
    tan_d = (sd / cd).astype(F)
    v = st["v"]
    a = (thr * F(cfg.accel_max) - F(cfg.drag_lin) * v).astype(F)
    v1 = (v + a * dt).astype(F)
    dv = ((F(cfg.roll_res) + brk * F(cfg.brake_max)) * dt).astype(F)
    v2 = np.where(v1 > 0, np.maximum((v1 - dv).astype(F), F(0)), np.where(v1 < 0, np.minimum((v1 + dv).astype(F), F(0)), F(0))).astype(F)
    hit = {"reverse": (v1 < 0) & ~do_reset, "stopped_by_drag": (v1 > 0) & (v1 - dv < 0) & ~do_reset, "brake": (brk > 0) & ~do_reset,
           "clamp_vmax": (v2 > F(cfg.v_max)) & ~do_reset, "clamp_vrev": (v2 < -F(cfg.v_rev_max)) & ~do_reset}
    v2 = np.clip(v2, -F(cfg.v_rev_max), F(cfg.v_max)).astype(F)
    yaw1 = (st["yaw"] + ((v2 * tan_d).astype(F) * F(cfg.inv_wheelbase)).astype(F) * dt).astype(F)
    hit["yaw_wrap_down"] = (yaw1 > pi) & ~do_reset
    yaw1 = np.where(yaw1 > pi, (yaw1 - two_pi).astype(F), yaw1).astype(F)
    hit["yaw_wrap_up"] = (yaw1 < -pi) & ~do_reset
    yaw1 = np.where(yaw1 < -pi, (yaw1 + two_pi).astype(F), yaw1).astype(F)
    s, c = np_sincos_vec(yaw1)
    x1 = (st["x"] + (v2 * s).astype(F) * dt).astype(F)
    z1 = (st["z"] + (v2 * c).astype(F) * dt).astype(F)
    y_in = st["y"].copy()
    # reset: the env's start pose, no integration
    gid = env_id_base + np.arange(n)
    si = (37 * gid) % len(pts)
    t64 = spec_start_tangent(pts, si)
    start_yaw = np.arctan2(t64[:, 0], t64[:, 1]).astype(F)             # heading = atan2(tx, tz): forward = (sin yaw, cos yaw) in (x, z)
    x1 = np.where(do_reset, pts[si, 0].astype(F), x1); z1 = np.where(do_reset, pts[si, 2].astype(F), z1)
    y_in = np.where(do_reset, pts[si, 1].astype(F), y_in)
    yaw1 = np.where(do_reset, start_yaw, yaw1).astype(F); v2 = np.where(do_reset, F(0), v2).astype(F)
    # nearest raw point: binary64 L1, best initialised to 100, strict '<', lowest index wins
    q = np.stack([x1.astype(np.float64), y_in.astype(np.float64), z1.astype(np.float64)], 1)
    d = (np.abs(q[:, None, 0] - pts[None, :, 0]) + np.abs(q[:, None, 1] - pts[None, :, 1])) + np.abs(q[:, None, 2] - pts[None, :, 2])
    idx = d.argmin(1)
    best = d[np.arange(n), idx]
    lost = ~(best < 100.0)
    idx = np.where(lost, 0, idx)
    hit["lost"] = lost
    y1 = pts[idx, 1].astype(F)
    cte = ((x1 - pts[idx, 0].astype(F)).astype(F) * tang[idx, 1] - (z1 - pts[idx, 2].astype(F)).astype(F) * tang[idx, 0]).astype(F)
    done = (np.abs(cte) > F(cfg.offtrack_cte)) | lost
    npnt = len(pts)
    dd = idx - st["seg"]
    half = npnt // 2
    hit["reward_wrap_fwd"] = (dd < -half) & ~do_reset
    hit["reward_wrap_back"] = (dd >= npnt - half) & ~do_reset
    dd = np.where(dd >= npnt - half, dd - npnt, dd)
    dd = np.where(dd < -half, dd + npnt, dd)
    reward = (dd.astype(F) - np.where(done, F(cfg.offtrack_penalty), F(0))).astype(F)
    new = dict(st)
    new["last_return"] = np.where(do_reset, st["ep_return"], st["last_return"]).astype(F)
    new["ep_return"] = np.where(do_reset, F(0), (st["ep_return"] + reward).astype(F)).astype(F)
    new["ep_len"] = np.where(do_reset, 0, st["ep_len"] + 1)
    new.update(x=x1, y=y1, z=z1, yaw=yaw1, v=v2, speed=np.abs(v2), cte=cte, seg=idx, done=done.astype(np.uint8))
    hit["reset"] = do_reset
    hit["done"] = done & ~do_reset
    return new, hit


def spec_start_tangent(pts, si):
    """(tx, tz) in binary64 at the start points (the start yaw is atan2(tx, tz) of the binary64 tangent, then binary32)."""
    x, z = pts[:, 0], pts[:, 2]
    n = len(pts)
    out = np.zeros((len(si), 2))
    for k, i in enumerate(si):
        j = i
        while True:
            j = (j + 1) % n
            if x[j] != x[i] or z[j] != z[i]:
                break
        b = i
        while True:
            b = (b - 1) % n
            if x[b] != x[i] or z[b] != z[i]:
                break
        tx, tz = x[j] - x[b], z[j] - z[b]
        ln = math.sqrt(tx * tx + tz * tz)
        out[k] = (tx / ln, tz / ln)
    return out


def fetch_state(env):
    return {"x": env.fetch("pos_x"), "y": env.fetch("pos_y"), "z": env.fetch("pos_z"), "yaw": env.fetch("yaw"), "v": env.fetch("vel"),
            "speed": env.fetch("speed"), "cte": env.fetch("cte"), "seg": env.fetch("seg_idx").astype(np.int64), "done": env.fetch("done"),
            "ep_return": env.fetch("ep_return"), "last_return": env.fetch("last_return"), "ep_len": env.fetch("ep_len").astype(np.int64)}


@pytest.mark.parametrize("auto_reset", [False, True])
def test_physics_steps_against_a_numpy_restatement(make_env, auto_reset):
    """60 steps of 64 envs from random states, every step checked against the vectorised restatement fed the oracle's own
    previous state (so rounding differences of the emulated fma cannot accumulate): reverse driving, braking to a stop, both
    speed clamps, yaw wrap in both directions, the reward wrap across the start line in both directions, off-track, lost
    (> 100 L1 from every point), user reset and auto reset are each hit."""
    n, steps = 64, 60
    pts = track_points("generated")
    tang = spec_tangents(pts)
    env = make_env("oracle", n_envs=n, render=False, auto_reset=auto_reset)
    assert np.array_equal(env.fetch("tangent").reshape(-1, 2).view(np.uint32), tang.view(np.uint32))   # the tangent table itself
    env.step(0.0, 0.0)                                                  # consumes the pending reset: start pose
    rng = np.random.default_rng(7 + int(auto_reset))
    # random states: on and beside the track near random points (some near the start line, from both sides), speeds from beyond
    # the reverse clamp to beyond v_max, headings near +-pi so that the yaw wraps
    k = rng.integers(0, len(pts), n)
    k[:8] = [0, 1, 2, len(pts) - 1, len(pts) - 2, len(pts) - 3, 3, len(pts) - 4]
    x = (pts[k, 0] + rng.uniform(-1.0, 1.0, n)).astype(F); z = (pts[k, 2] + rng.uniform(-1.0, 1.0, n)).astype(F)
    tz_yaw = np.arctan2(tang[k, 0], tang[k, 1])
    yaw = (tz_yaw + rng.choice([0.0, math.pi], n) + rng.uniform(-0.3, 0.3, n)).astype(np.float64)
    yaw = ((yaw + math.pi) % (2 * math.pi) - math.pi).astype(F)
    yaw[8:16] = F(3.14) * rng.choice([-1, 1], 8).astype(F)
    v = rng.uniform(-6.0, 27.0, n).astype(F)
    x[60:] += F(500.0)                                                  # four envs far away: "lost"
    env.set_pose(x=x, z=z, yaw=yaw, v=v)
    cfg = env.cfg
    seen = {}
    mism_idx = 0
    for t in range(steps):
        st = fetch_state(env)
        steer = rng.uniform(-1.3, 1.3, n).astype(F)
        thr = rng.uniform(-1.2, 1.2, n).astype(F)
        brk = np.where(rng.random(n) < 0.3, rng.uniform(0.0, 1.2, n), 0.0).astype(F)
        if t % 10 < 3:
            thr[:16] = F(1.0); brk[:16] = F(0.0)                        # some envs keep accelerating (v_max) ...
            thr[16:24] = F(-1.0); brk[16:24] = F(0.0)                   # ... or reversing (v_rev_max)
        reset = (rng.random(n) < 0.04)
        env.step(steer, thr, brk, reset=reset.astype(np.uint8))
        want, hit = numpy_step(cfg, pts, tang, st, steer, thr, brk, reset)
        got = fetch_state(env)
        for key, mask in hit.items():
            seen[key] = seen.get(key, 0) + int(np.count_nonzero(mask))
        same_idx = got["seg"] == want["seg"]
        mism_idx += int((~same_idx).sum())
        for key in ("x", "z", "yaw", "v", "speed"):
            assert np.max(np.abs(got[key] - want[key])) <= 1e-5, (t, key)
        assert np.max(np.abs(got["y"][same_idx] - want["y"][same_idx])) == 0
        assert np.max(np.abs(got["cte"][same_idx] - want["cte"][same_idx])) <= 1e-5, t
        ok = same_idx & (np.abs(np.abs(want["cte"]) - cfg.offtrack_cte) > 1e-4)
        assert np.array_equal(got["done"][ok], want["done"][ok]), t
        assert np.array_equal(got["ep_len"], want["ep_len"]), t
        assert np.max(np.abs(got["ep_return"][ok] - want["ep_return"][ok])) <= 1e-4, t
        assert np.max(np.abs(got["last_return"] - want["last_return"])) <= 1e-4, t
    assert mism_idx <= 2, mism_idx                                      # an exact L1 tie decided by the last bit, at most
    need = ["reverse", "stopped_by_drag", "brake", "clamp_vmax", "clamp_vrev", "yaw_wrap_down", "yaw_wrap_up", "reward_wrap_fwd",
            "reward_wrap_back", "lost", "done", "reset"]
    missing = [k_ for k_ in need if seen.get(k_, 0) == 0]
    assert not missing, (missing, seen)
