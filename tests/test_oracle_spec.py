"""Properties of the oracle's own (reference-unpinned) parts: the spec's trig, one-step arithmetic restated in
numpy float32, determinism, sharding invariance, reset semantics and image sanity.  These run on CPU."""
import math

import numpy as np

from conftest import track_points

F = np.float32


def np_sincos(a):
    """include/trsim_spec.h trs_sincos restated with numpy float32 scalars (fma emulated in float64)."""
    def fma(x, y, z):
        return F(np.float64(x) * np.float64(y) + np.float64(z))
    q = F(np.rint(F(a) * F(0.636619746685028076)))
    r = fma(q, F(-1.5707963705062866211), F(a))
    r = fma(q, F(4.3711388286737928865e-08), r)
    z = F(r * r)
    ps = fma(fma(F(-1.9515295891e-4), z, F(8.3321608736e-3)), z, F(-1.6666654611e-1))
    s = fma(F(r * z), ps, r)
    pc = fma(fma(F(2.443315711809948e-5), z, F(-1.388731625493765e-3)), z, F(4.166664568298827e-2))
    c = fma(F(z * z), pc, fma(z, F(-0.5), F(1.0)))
    return [(s, c), (c, -s), (-s, -c), (-c, s)][int(q) & 3]


def test_spec_sincos_is_accurate():
    worst = 0.0
    for a in np.linspace(-math.pi, math.pi, 4001):
        s, c = np_sincos(a)
        worst = max(worst, abs(float(s) - math.sin(float(F(a)))), abs(float(c) - math.cos(float(F(a)))))
    assert worst < 2.5e-7


def test_one_step_matches_numpy_restatement(make_env):
    """First integration step from the start pose, host controls, against float32 numpy arithmetic."""
    n = 32
    env = make_env("oracle", n_envs=n, render=False)
    env.step(0.0, 0.0)                                     # consumes the pending reset: start pose, v = 0
    x0, z0, yaw0 = env.fetch("pos_x"), env.fetch("pos_z"), env.fetch("yaw")
    steer = np.linspace(-1.2, 1.2, n).astype(F)
    thr = np.linspace(0.1, 1.0, n).astype(F)
    env.step(steer, thr, 0.0)
    cfg = env.cfg
    for i in range(n):
        st = F(min(max(steer[i], -1), 1)); th = F(min(max(thr[i], -1), 1))
        sd, cd = np_sincos(F(st * F(cfg.max_steer)))
        tan_d = F(sd / cd)
        a = F(F(th * F(cfg.accel_max)) - F(F(cfg.drag_lin) * F(0)))
        v1 = F(F(0) + F(a * F(cfg.dt)))
        dv = F(F(F(cfg.roll_res) + F(F(0) * F(cfg.brake_max))) * F(cfg.dt))
        v2 = max(F(v1 - dv), F(0)) if v1 > 0 else F(0)
        yaw1 = F(yaw0[i] + F(F(F(v2 * tan_d) * F(cfg.inv_wheelbase)) * F(cfg.dt)))
        if yaw1 > F(math.pi): yaw1 = F(yaw1 - F(2 * math.pi))
        if yaw1 < -F(math.pi): yaw1 = F(yaw1 + F(2 * math.pi))
        s, c = np_sincos(yaw1)
        x1 = F(x0[i] + F(F(v2 * s) * F(cfg.dt))); z1 = F(z0[i] + F(F(v2 * c) * F(cfg.dt)))
        assert abs(float(env.fetch("vel")[i]) - float(v2)) <= 1e-6
        assert abs(float(env.fetch("yaw")[i]) - float(yaw1)) <= 1e-6
        assert abs(float(env.fetch("pos_x")[i]) - float(x1)) <= 1e-5 and abs(float(env.fetch("pos_z")[i]) - float(z1)) <= 1e-5


def test_determinism_and_sharding_invariance(make_env):
    full = make_env("oracle", n_envs=96, auto_reset=True)
    full.step_synthetic(120, 1)
    again = make_env("oracle", n_envs=96, auto_reset=True)
    again.step_synthetic(120, 7)
    assert np.array_equal(full.fetch("img"), again.fetch("img"))
    for s in range(3):                                     # 3 shards of 32: RNG / start pose keyed by GLOBAL env id
        shard = make_env("oracle", n_envs=32, env_id_base=32 * s, auto_reset=True)
        shard.step_synthetic(120, 1)
        for name in ("pos_x", "pos_z", "yaw", "ep_return", "seg_idx", "done", "ep_len"):
            assert np.array_equal(shard.fetch(name), full.fetch(name)[32 * s:32 * (s + 1)]), name
        assert np.array_equal(shard.fetch("img"), full.fetch("img")[32 * s:32 * (s + 1)])
    st = full.fetch("stats")
    assert st[1] >= 96 and st[0] > 0                       # every env reset at least once; some left the track


def test_reset_and_done_semantics(make_env):
    env = make_env("oracle", n_envs=4, render=False, auto_reset=False)
    env.step(0.0, 1.0)                                     # step 1 = pending reset -> start pose, no reward
    assert np.array_equal(env.fetch("ep_len"), np.zeros(4, np.int32)) and not env.fetch("ep_return").any()
    pts = track_points("generated")
    assert np.array_equal(env.fetch("pos_x"), pts[[0, 37, 74, 111], 0].astype(F))
    for _ in range(300):                                   # full lock + throttle: the car leaves the road
        env.step(1.0, 1.0)
    assert env.fetch("done").all() and (np.abs(env.fetch("cte")) > 3.0).all()
    ret = env.fetch("ep_return").copy()
    env.step(0.0, 0.0, reset=[True, False, False, False])  # usr/reset truthy for env 0 only
    assert env.fetch("ep_len")[0] == 0 and env.fetch("last_return")[0] == ret[0] and env.fetch("done")[0] == 0
    assert env.fetch("ep_len")[1] == 301


def test_image_sanity(make_env):
    env = make_env("oracle", n_envs=3)
    env.step(0.0, 0.0)
    img = env.fetch("img")
    assert img.shape == (3, 120, 160, 3) and img.dtype == np.uint8
    sky, ground = img[:, :20], img[:, 100:]
    assert (sky[..., 2] > sky[..., 0]).all()               # sky is blue-ish
    road = np.array([92, 92, 98])
    near = ground.reshape(-1, 3).astype(int)
    assert (np.abs(near - road).sum(1) < 40).mean() > 0.5  # the car starts on the road: asphalt fills the near field
    pal = env.fetch("palette")
    assert (pal[:40] == pal[:40, :1]).all()                # sky rows: one colour for every class


def test_render_matches_numpy_restatement(make_env):
    """The oracle's camera loop restated independently with vectorised numpy from the oracle's own tables (class map, row
    table, palette) and the env's pose: include/trsim_spec.h 'one frame'.  fma is emulated through binary64 (a double
    rounding can differ from a true fma in the last bit), so the check allows a handful of boundary pixels."""
    n, H, W = 6, 120, 160
    env = make_env("oracle", n_envs=n, auto_reset=True)
    env.step_synthetic(40, 1)
    mi = env.map_info
    gmap = env.fetch("map").reshape(mi.map_h, mi.map_words)
    rowtab = env.fetch("rowtab").reshape(H, 2)
    pal = env.fetch("palette").reshape(H, 4)
    img = env.fetch("img")
    x, z, yaw = env.fetch("pos_x"), env.fetch("pos_z"), env.fetch("yaw")
    x0f, z0f, inv = F(mi.x0), F(mi.z0), F(1.0 / mi.cell)
    fwd = F(env.cfg.cam_fwd)

    def fma(a, b, c):
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)

    uf = (np.arange(W, dtype=F) + F(0.5) - F(W // 2))[None, :]
    bad = 0
    for i in range(n):
        s, c = np_sincos(yaw[i])
        camx = F(F(F(x[i] + F(fwd * s)) - x0f) * inv)
        camz = F(F(F(z[i] + F(fwd * c)) - z0f) * inv)
        lz, kk = rowtab[:, :1], rowtab[:, 1:]
        ax, az = fma(lz, np.full_like(lz, s), np.full_like(lz, camx)), fma(lz, np.full_like(lz, c), np.full_like(lz, camz))
        dx, dz = (kk * F(c)).astype(F), (-(kk * F(s))).astype(F)
        gx = fma(np.broadcast_to(uf, (H, W)), np.broadcast_to(dx, (H, W)), np.broadcast_to(ax, (H, W)))
        gz = fma(np.broadcast_to(uf, (H, W)), np.broadcast_to(dz, (H, W)), np.broadcast_to(az, (H, W)))
        ix = np.clip(np.floor(gx).astype(np.int64), 0, mi.map_w - 1)
        iz = np.clip(np.floor(gz).astype(np.int64), 0, mi.map_h - 1)
        cls = (gmap[iz, ix >> 4] >> ((ix & 15) * 2).astype(np.uint32)) & 3
        rgb = pal[np.arange(H)[:, None], cls]
        want = np.stack([rgb & 255, (rgb >> 8) & 255, (rgb >> 16) & 255], -1).astype(np.uint8)
        bad += int((want != img[i]).any(-1).sum())
    assert bad <= 3 * n, bad                                                # double-rounded fma at a cell boundary, at most
