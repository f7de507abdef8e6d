"""Tracks with elevation (include/trsim_spec.h, "tracks with elevation"; round 5) — the CPU half: an INDEPENDENT numpy restatement of
the paragraph against the C oracle.  Reference data: car_templates/track_data/mountain_track.json (y in 3.19 .. 7.35; the reference itself only passes y
through its telemetry, components/gyminterface.py:100-104 — the closed simulator behind it rendered the hills).

* the per-point view-pitch offsets dpitch[] (binary64 on the host: grade over +-L samples, the slope A samples ahead against the slope here);
* the per-env, per-frame row tables in binary32 with the spec's operation order, the fogged palette, the frame and the depth frame they give;
* a flat track (generated_track: 1 cm of height) is NOT hilly: dpitch all zeros, frames from the host's binary64 tables as before."""
import math

import numpy as np
import pytest

from conftest import track_points

L, A, MIN_RANGE, MAX_DP, FOG_MAX = 8, 24, 0.25, 0.2, 0.65
BASE = np.array([[58, 132, 62], [92, 92, 98], [236, 236, 236], [232, 200, 40]], np.float32)
FOG = np.array([176, 196, 208], np.float32)
SKY_TOP, SKY_HOR = np.array([104, 156, 228], np.float64), np.array([192, 216, 240], np.float64)
f32 = np.float32


def spec_dpitch(pts):
    n = len(pts)
    if pts[:, 1].max() - pts[:, 1].min() <= MIN_RANGE:
        return np.zeros(n, np.float32)
    nxt = np.roll(pts, -1, axis=0)
    h = np.sqrt((nxt[:, 0] - pts[:, 0]) ** 2 + (nxt[:, 2] - pts[:, 2]) ** 2)
    theta = np.zeros(n)
    for i in range(n):
        d = 0.0
        for q in range(-L, L):
            d += h[(i + q) % n]
        g = (pts[(i + L) % n, 1] - pts[(i - L) % n, 1]) / d if d > 1e-9 else 0.0
        theta[i] = math.atan(g)
    dp = np.clip(np.roll(theta, -A) - theta, -MAX_DP, MAX_DP)
    return dp.astype(np.float32)


def np_sincos(a):
    """trs_sincos of the spec on a binary32 scalar."""
    a = f32(a)
    q = np.rint(f32(a * f32(0.636619746685028076)))
    fma = lambda x, y, z: f32(np.float64(x) * np.float64(y) + np.float64(z))     # exact product + one rounding = binary32 fma for these magnitudes
    r = fma(q, f32(-1.5707963705062866211), a)
    r = fma(q, f32(4.3711388286737928865e-08), r)
    z = f32(r * r)
    ps = fma(fma(f32(-1.9515295891e-4), z, f32(8.3321608736e-3)), z, f32(-1.6666654611e-1))
    s = fma(f32(r * z), ps, r)
    pc = fma(fma(f32(2.443315711809948e-5), z, f32(-1.388731625493765e-3)), z, f32(4.166664568298827e-2))
    c = fma(f32(z * z), pc, fma(z, f32(-0.5), f32(1.0)))
    n = int(q) & 3
    return [(s, c), (c, -s), (-s, -c), (-c, s)][n]


def env_row_tables(P, H, inv_f, cam_h, z_far, inv_cell, sky, far):
    """Row tables of ONE env and frame (binary32, the spec's order): (row_lz, row_k, depth, palette[H][4])."""
    sp, cp = np_sincos(P)
    hh = f32(H / 2)
    lz, kk, dep = np.zeros(H, np.float32), np.zeros(H, np.float32), np.full(H, f32(z_far), np.float32)
    pal = np.zeros((H, 4), np.uint32)
    inv_zfar = f32(1.0 / z_far)
    for v in range(H):
        yn = f32(f32(hh - f32(f32(v) + f32(0.5))) * inv_f)
        dy = f32(f32(yn * cp) - sp)
        dz = f32(f32(yn * sp) + cp)
        if dy >= f32(-1e-6):
            pal[v] = sky[v]
            continue
        t = f32(f32(cam_h) / f32(-dy))
        zd = f32(t * dz)
        if zd > f32(z_far):
            pal[v] = far
            continue
        lz[v] = f32(zd * inv_cell); kk[v] = f32(f32(t * inv_f) * inv_cell); dep[v] = zd
        fw = f32(f32(FOG_MAX) * f32(zd * inv_zfar))
        om = f32(f32(1.0) - fw)
        for c in range(4):
            rgb = 0
            for ch in range(3):
                val = int(f32(f32(f32(BASE[c, ch] * om) + f32(FOG[ch] * fw)) + f32(0.5)))
                rgb |= val << (8 * ch)
            pal[v, c] = rgb
    return lz, kk, dep, pal


def render(lz, kk, pal, cam, class_map, W):
    camx, camz, s, c = cam
    H = len(lz)
    gh, mw = class_map.shape
    img = np.zeros((H, W, 3), np.uint8)
    fma = lambda x, y, z: f32(np.float64(x) * np.float64(y) + np.float64(z))
    for v in range(H):
        ax, az = fma(lz[v], s, camx), fma(lz[v], c, camz)
        dx, dz = f32(kk[v] * c), f32(-f32(kk[v] * s))
        for u in range(W):
            uf = f32(f32(u) + f32(0.5) - f32(W // 2))
            gx, gz = fma(uf, dx, ax), fma(uf, dz, az)
            ix, iz = int(math.floor(gx)), int(math.floor(gz))
            ix = min(max(ix, 0), GW[0] - 1); iz = min(max(iz, 0), gh - 1)
            cls = (int(class_map[iz, ix >> 4]) >> ((ix & 15) * 2)) & 3
            rgb = int(pal[v, cls])
            img[v, u] = (rgb & 255, (rgb >> 8) & 255, (rgb >> 16) & 255)
    return img


GW = [0]


@pytest.mark.parametrize("track", ["generated", "mountain"])
def test_dpitch_table_against_the_numpy_restatement(make_env, track):
    pts = track_points(track)
    env = make_env("oracle", n_envs=1, track=pts)
    want = spec_dpitch(pts)
    got = env.fetch("dpitch")
    assert np.array_equal(got, want)
    if track == "generated":
        assert not got.any()                                   # 1 cm of height: a flat track
    else:
        assert np.abs(got).max() > 0.02 and np.abs(got).max() <= np.float32(MAX_DP)    # the mountain track's crests and dips tilt the view by degrees


def test_hilly_frames_against_the_numpy_restatement(make_env):
    """Four envs on the mountain track, small frames (the numpy renderer is a per-pixel Python loop): frames and depth frames of the oracle against the
    restated per-env row tables; the tables must differ between envs (different slopes ahead) and between an env's frames."""
    pts = track_points("mountain")
    H, W, n = 24, 32, 4
    env = make_env("oracle", n_envs=n, track=pts, img_h=H, img_w=W, depth=True, env_id_base=11)
    dp = spec_dpitch(pts)
    cfg = env.cfg
    f = (H / 2.0) / math.tan(cfg.fov_v_deg * math.pi / 180.0 / 2.0)
    inv_f, pitch_f = f32(1.0 / f), f32(cfg.cam_pitch_deg * math.pi / 180.0)
    mi = env.map_info
    inv_cell = f32(1.0 / mi.cell)
    class_map = env.fetch("map")
    GW[0] = mi.map_w
    sky = np.zeros(H, np.uint32)
    for v in range(H):
        g = min((v + 0.5) / (H / 2.0), 1.0)
        rgb = 0
        for ch in range(3):
            rgb |= int(math.floor(SKY_TOP[ch] + (SKY_HOR[ch] - SKY_TOP[ch]) * g + 0.5)) << (8 * ch)
        sky[v] = rgb
    far = 0
    for ch in range(3):
        far |= int(math.floor(float(BASE[0, ch]) * (1.0 - FOG_MAX) + float(FOG[ch]) * FOG_MAX + 0.5)) << (8 * ch)
    seen = set()
    for step in range(6):
        env.step(np.float32([0.1, -0.2, 0.0, 0.3]), np.float32(0.8), 0.0)
        idx = env.fetch("seg_idx")
        x, z, yaw = env.fetch("pos_x"), env.fetch("pos_z"), env.fetch("yaw")
        imgs, deps = env.fetch("img"), env.fetch("depth")
        for i in range(n):
            P = f32(pitch_f + dp[idx[i]])
            lz, kk, dep, pal = env_row_tables(P, H, inv_f, cfg.cam_h, cfg.z_far, inv_cell, sky, far)
            s, c = np_sincos(yaw[i])
            camx = f32(f32(f32(x[i] + f32(f32(cfg.cam_fwd) * s)) - f32(mi.x0)) * inv_cell)
            camz = f32(f32(f32(z[i] + f32(f32(cfg.cam_fwd) * c)) - f32(mi.z0)) * inv_cell)
            want = render(lz, kk, pal, (camx, camz, s, c), class_map, W)
            assert np.array_equal(imgs[i], want), (step, i)
            assert np.array_equal(deps[i], np.repeat(dep[:, None], W, axis=1)), (step, i)
            seen.add(float(P))
    assert len(seen) >= 4                                       # the view pitch really varies over envs and frames
