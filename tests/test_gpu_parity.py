"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Bars (BASELINE.json north_star): integer track indices, flags and image bytes bit-exact; pose / speed /
cte within 1e-5 (the spec is written so that they are in fact bit-identical).
"""
import numpy as np
import pytest

from conftest import load_golden, track_points

pytestmark = pytest.mark.gpu

FLOATS = ("pos_x", "pos_y", "pos_z", "speed", "cte", "yaw", "vel", "ep_return", "last_return", "steer_filt")
INTS = ("seg_idx", "done", "ep_len")
TOL = 1e-5


def assert_state_equal(a, b, where=""):
    for name in INTS:
        x, y = a.fetch(name), b.fetch(name)
        assert np.array_equal(x, y), f"{name} differs {where}: {np.flatnonzero(x != y)[:8]}"
    for name in FLOATS:
        x, y = a.fetch(name), b.fetch(name)
        err = float(np.max(np.abs(x.astype(np.float64) - y.astype(np.float64))))
        assert err <= TOL, f"{name} differs by {err} {where}"


# ---------------------------------------------------------------------------------------------- tables

@pytest.mark.parametrize("track", ["generated", "mountain"])
def test_tables_bit_exact(make_env, track):
    """Map, camera rows, palette, tangents built by the product (C++) == oracle (C), bit for bit."""
    pts = track_points(track)
    g, o = make_env("hip", n_envs=2, track=pts), make_env("oracle", n_envs=2, track=pts)
    for f in ("map_w", "map_h", "map_words", "cell", "x0", "z0", "n_points"):
        assert getattr(g.map_info, f) == getattr(o.map_info, f), f
    for name in ("map", "rowtab", "palette", "tangent"):
        assert np.array_equal(g.fetch(name).view(np.uint32), o.fetch(name).view(np.uint32)), name
    assert g.map_info.lds_bytes <= 160 * 1024


# ---------------------------------------------------------------------------------------------- a8: LocationTracker

@pytest.mark.parametrize("track", ["generated", "mountain"])
def test_locate_matches_reference_golden(make_env, track):
    """G1: integer indices produced by the reference's LocationTracker itself (6,994 / 2,597 queries incl.
    duplicates, far points -> 0, near ties, points around L1 = 100)."""
    g1 = load_golden(f"locate_{track}.json")
    env = make_env("hip", n_envs=1, track=track_points(track), render=False)
    idx = env.locate(g1["queries"])
    assert np.array_equal(idx, np.asarray(g1["idx"], dtype=np.int32))
    seg = env.segment(idx)
    assert np.array_equal(seg, np.asarray(g1["segment"], dtype=np.float64))   # 'loc/segment' floats, exact


def test_locate_edge_cases(make_env, oracle_api):
    pts = track_points("generated")
    env, ora = make_env("hip", n_envs=1, track=pts, render=False), make_env("oracle", n_envs=1, track=pts, render=False)
    assert env.locate(np.zeros((0, 3))).shape == (0,)
    rng = np.random.default_rng(7)
    q = np.concatenate([
        pts,                                                # every raw point (duplicates -> first of the pair)
        pts[rng.integers(0, len(pts), 20000)] + rng.normal(0, 2.0, (20000, 3)),
        rng.uniform(-500, 500, (3000, 3)),                  # mostly "lost" -> 0
    ])
    assert np.array_equal(env.locate(q), ora.locate(q))
    big = pts[rng.integers(0, len(pts), 300000)] + rng.normal(0, 1.0, (300000, 3))   # multi-wave, grid-stride path
    assert np.array_equal(env.locate(big), ora.locate(big))


# ---------------------------------------------------------------------------------------------- config 2: physics

def test_physics_256_envs_checkpoints(make_env):
    """BASELINE config 2: 256 envs, physics only; parity of all state at steps {1, 10, 100, 1000}."""
    g = make_env("hip", n_envs=256, render=False, auto_reset=True)
    o = make_env("oracle", n_envs=256, render=False, auto_reset=True)
    done = 0
    for target in (1, 10, 100, 1000):
        for env in (g, o):
            env.step_synthetic(target - done, 1)
        done = target
        assert_state_equal(g, o, f"at step {target}")
    assert int(o.fetch("ep_len").max()) < 1000, "auto-reset never fired: the reset path was not exercised"


def test_multi_step_launch_equals_single_steps(make_env):
    a = make_env("hip", n_envs=300, render=False, auto_reset=True)     # 300: ragged last workgroup
    b = make_env("hip", n_envs=300, render=False, auto_reset=True)
    a.step_synthetic(257, 1)
    b.step_synthetic(257, 32)                                           # 8 launches of 32 + 1 of 1
    assert_state_equal(a, b, "K-step launch vs single-step launches")


def test_host_controls_and_reset_semantics(make_env):
    n = 96
    rng = np.random.default_rng(3)
    g, o = make_env("hip", n_envs=n, render=False), make_env("oracle", n_envs=n, render=False)
    for k in range(60):
        st = rng.uniform(-1.3, 1.3, n).astype(np.float32)               # beyond [-1,1]: clamp path
        th = rng.uniform(-1.2, 1.2, n).astype(np.float32)
        br = None if k % 3 == 0 else rng.uniform(-0.2, 1.2, n).astype(np.float32)   # None = 'breaking is None'
        rs = None if k % 7 else (rng.random(n) < 0.2)
        for env in (g, o):
            env.step(st, th, br, reset=rs, n_steps=1 + (k % 2))
        assert_state_equal(g, o, f"host controls step {k}")
    mask = (rng.random(n) < 0.5).astype(np.uint8)
    for env in (g, o):
        env.reset(mask)
        env.step(0.0, 0.5)
    assert_state_equal(g, o, "after masked reset")


def test_set_pose_offtrack_and_lost(make_env):
    """Cars placed off the road / farther than L1 = 100 from every point: done flag, index 0, penalty."""
    n = 8
    pts = track_points("generated")
    x = np.array([pts[5, 0], pts[5, 0] + 2.9, pts[5, 0] + 3.5, 1000.0, -400.0, pts[600, 0], pts[600, 0] - 3.2, pts[0, 0]], np.float32)
    z = np.array([pts[5, 2], pts[5, 2], pts[5, 2], 1000.0, 90.0, pts[600, 2], pts[600, 2], pts[0, 2]], np.float32)
    g, o = make_env("hip", n_envs=n, render=False), make_env("oracle", n_envs=n, render=False)
    for env in (g, o):
        env.set_pose(x=x, y=np.full(n, 0.56, np.float32), z=z, yaw=np.linspace(-3, 3, n).astype(np.float32), v=np.zeros(n, np.float32))
        env.step(0.0, 0.0)
    assert_state_equal(g, o, "off-track placement")
    done = g.fetch("done")
    assert done[3] == 1 and done[4] == 1 and g.fetch("seg_idx")[3] == 0


# ---------------------------------------------------------------------------------------------- config 3: camera

def test_render_pixel_exact_64_envs_8_steps(make_env):
    """BASELINE config 3 sample: 64 envs x 8 steps, every byte of every frame."""
    g = make_env("hip", n_envs=64, auto_reset=True)
    o = make_env("oracle", n_envs=64, auto_reset=True)
    for k in range(8):
        for env in (g, o):
            env.step_synthetic(1, 1)
        a, b = g.fetch("img"), o.fetch("img")
        assert a.shape == (64, 120, 160, 3) and a.dtype == np.uint8
        assert np.array_equal(a, b), f"frame {k}: {int((a != b).sum())} bytes differ"
    assert_state_equal(g, o, "after rendering")
    assert len(np.unique(a.reshape(-1, 3), axis=0)) > 20, "image is degenerate"


def test_render_after_long_run_and_other_sizes(make_env):
    for (h, w, n) in ((120, 160, 33), (240, 320, 5), (64, 64, 7)):
        g = make_env("hip", n_envs=n, img_h=h, img_w=w, auto_reset=True)
        o = make_env("oracle", n_envs=n, img_h=h, img_w=w, auto_reset=True)
        for env in (g, o):
            env.step_synthetic(150, 50)
        assert np.array_equal(g.fetch("img"), o.fetch("img")), (h, w, n)
        assert_state_equal(g, o, f"{h}x{w}")


def test_render_mountain_track(make_env):
    pts = track_points("mountain")
    g = make_env("hip", n_envs=17, track=pts, auto_reset=True)
    o = make_env("oracle", n_envs=17, track=pts, auto_reset=True)
    for env in (g, o):
        env.step_synthetic(40, 8)
    assert np.array_equal(g.fetch("img"), o.fetch("img"))
    assert_state_equal(g, o, "mountain")


def test_double_buffered_image(make_env):
    """The frame of step t stays intact while step t+1 renders into the other buffer."""
    g = make_env("hip", n_envs=8, auto_reset=True)
    g.step_synthetic(5, 1)
    sv = g.state_view()
    first_ptr, first = sv.img, g.fetch("img").copy()
    g.step_synthetic(1, 1)
    assert g.state_view().img != first_ptr
    g.step_synthetic(1, 1)
    assert g.state_view().img == first_ptr
    assert not np.array_equal(g.fetch("img"), first)


# ---------------------------------------------------------------------------------------------- full size: properties

def test_full_size_1024_envs_properties(make_env):
    """BASELINE config 3 at full size: sharding invariance (RNG and start pose keyed by GLOBAL env id) and
    run-to-run determinism — size-independent properties, plus a strided oracle sample."""
    full = make_env("hip", n_envs=1024, auto_reset=True)
    full.step_synthetic(64, 1)
    img_full = full.fetch("img")
    again = make_env("hip", n_envs=1024, auto_reset=True)
    again.step_synthetic(64, 16)
    assert np.array_equal(img_full, again.fetch("img")), "non-deterministic / K-launch mismatch"
    # 8 shards of 128 = what 8 GPUs would each own
    for s in (0, 3, 7):
        shard = make_env("hip", n_envs=128, env_id_base=128 * s, auto_reset=True)
        shard.step_synthetic(64, 1)
        assert np.array_equal(shard.fetch("img"), img_full[128 * s:128 * (s + 1)]), f"shard {s}"
        assert np.array_equal(shard.fetch("ep_return"), full.fetch("ep_return")[128 * s:128 * (s + 1)])
        shard.close()
    o = make_env("oracle", n_envs=1024, auto_reset=True)
    o.step_synthetic(64, 1)
    assert np.array_equal(img_full, o.fetch("img"))
    assert_state_equal(full, o, "1024 envs x 64 steps")


# ---------------------------------------------------------------------------------------------- config 5 frame format

def test_depth_channel_and_240x320(make_env):
    """BASELINE config 5 frame format: 240x320 RGB + binary32 z-depth, bit-exact; depth is constant along a row."""
    for (h, w, n) in ((240, 320, 9), (120, 160, 21)):
        g = make_env("hip", n_envs=n, img_h=h, img_w=w, depth=True, auto_reset=True)
        o = make_env("oracle", n_envs=n, img_h=h, img_w=w, depth=True, auto_reset=True)
        assert np.array_equal(g.fetch("rowdepth"), o.fetch("rowdepth"))
        for env in (g, o):
            env.step_synthetic(37, 8)
        assert np.array_equal(g.fetch("img"), o.fetch("img"))
        dg, do = g.fetch("depth"), o.fetch("depth")
        assert dg.shape == (n, h, w) and dg.dtype == np.float32
        assert np.array_equal(dg.view(np.uint32), do.view(np.uint32))
        assert (dg == dg[:, :, :1]).all() and dg[0, 0, 0] == 40.0 and 0.7 < dg[0, -1, 0] < 1.0   # sky = z_far, nearest row ~0.84 units ahead
        assert g.state_view().depth and not make_env("hip", n_envs=2).state_view().depth


# ---------------------------------------------------------------------------------------------- device-pointer boundary

def test_device_pointer_controls_from_torch(make_env):
    """trs_step with DEVICE arrays (torch CUDA tensors) == the host-array path == the oracle; outputs read zero-copy."""
    torch = pytest.importorskip("torch")
    n = 200
    rng = np.random.default_rng(11)
    g = make_env("hip", n_envs=n, auto_reset=True)
    h = make_env("hip", n_envs=n, auto_reset=True)
    o = make_env("oracle", n_envs=n, auto_reset=True)
    for k in range(12):
        st = rng.uniform(-1, 1, n).astype(np.float32)
        th = rng.uniform(0, 1, n).astype(np.float32)
        br = rng.uniform(0, 0.3, n).astype(np.float32)
        rs = (rng.random(n) < 0.05).astype(np.uint8)
        d_st, d_th, d_br, d_rs = (torch.from_numpy(a).cuda() for a in (st, th, br, rs))
        torch.cuda.synchronize()                                         # the handle runs on its own stream
        g.step_device(d_st.data_ptr(), d_th.data_ptr(), d_br.data_ptr(), d_rs.data_ptr(), n_steps=3)
        g.sync()
        for env in (h, o):
            env.step(st, th, br, reset=rs, n_steps=3)
    assert_state_equal(g, o, "device-pointer controls")
    assert_state_equal(g, h, "device vs host controls")
    frames = torch.as_tensor(g.device_array("img"), device="cuda")
    assert frames.shape == (n, 120, 160, 3) and frames.dtype == torch.uint8
    assert np.array_equal(frames.cpu().numpy(), o.fetch("img"))
    ret = torch.as_tensor(g.device_array("ep_return"), device="cuda")
    assert np.array_equal(ret.cpu().numpy(), o.fetch("ep_return"))


@pytest.mark.gpu
def test_fetch_outputs_equals_field_copies(make_env):
    """trs_fetch_outputs: the whole GymInterface tuple in one synchronisation == the per-field copies, HIP == oracle."""
    g = make_env("hip", n_envs=33, auto_reset=True)
    o = make_env("oracle", n_envs=33, auto_reset=True)
    for env in (g, o):
        env.step_synthetic(9, 1)
    got, want = g.fetch_outputs(), o.fetch_outputs()
    names = ["img", "pos_x", "pos_y", "pos_z", "speed", "cte", "seg_idx", "done"]
    for name, a, b in zip(names, got, want):
        assert np.array_equal(a, g.fetch(name)), name
        if a.dtype == np.float32:
            assert np.max(np.abs(a - b)) <= 1e-5, name
        else:
            assert np.array_equal(a, b), name
    assert g.fetch_outputs(image=False)[0] is None


VARIANTS = [
    dict(cam_pitch_deg=-35.0, cam_h=2.5),                                   # looking down: (almost) no sky rows
    dict(cam_pitch_deg=12.0),                                               # looking up: mostly sky
    dict(z_far=6.0),                                                        # near far plane: many rows beyond it
    dict(fov_v_deg=35.0, cam_fwd=1.5), dict(fov_v_deg=110.0),
    dict(dt=0.02, max_steer=0.6, accel_max=9.0, drag_lin=0.1, v_max=30.0),
    dict(road_half=1.2, edge_half=0.3, centre_half=0.08, dash_period=1.5, dash_on=0.5, offtrack_cte=1.0),
    dict(seed=12345, offtrack_penalty=25.0, brake_max=3.0, roll_res=0.4, v_rev_max=1.0),
]


@pytest.mark.gpu
@pytest.mark.parametrize("kw", VARIANTS)
def test_non_default_parameters_match_oracle(make_env, kw):
    """Physics, camera and surface parameters away from the defaults: same tables, same trajectories, same pixels.
    The camera variants move the boundary of the rows whose four class colours are equal (the step kernel writes those
    without a map lookup) from 'none' to 'nearly all'."""
    n = 37
    g = make_env("hip", n_envs=n, auto_reset=True, depth=True, **kw)
    o = make_env("oracle", n_envs=n, auto_reset=True, depth=True, **kw)
    for name in ("map", "rowtab", "palette", "rowdepth", "tangent"):
        assert np.array_equal(g.fetch(name), o.fetch(name)), (kw, name)
    for env in (g, o):
        env.step_synthetic(1, 1)
        env.step_synthetic(14, 1)
        env.step_synthetic(9, 3)
    for name in ("seg_idx", "done", "ep_len", "img", "depth"):
        assert np.array_equal(g.fetch(name), o.fetch(name)), (kw, name)
    for name in ("pos_x", "pos_y", "pos_z", "speed", "cte", "yaw", "ep_return"):
        assert np.max(np.abs(g.fetch(name) - o.fetch(name))) <= 1e-5, (kw, name)
    pal = o.fetch("palette").reshape(-1, 4)
    uni = int(np.argmin((pal[:, 0] == pal[:, 1]) & (pal[:, 1] == pal[:, 2]) & (pal[:, 2] == pal[:, 3]))) if not np.all(pal[:, :1] == pal) else len(pal)
    assert 0 <= uni <= 120


@pytest.mark.gpu
@pytest.mark.parametrize("render", [True, False])
def test_step_sequence_equals_single_steps(make_env, render):
    """trs_step_sequence: a different control set per step inside multi-step launches == the same controls fed one
    trs_step at a time (HIP) == the oracle."""
    n, k = 70, 13
    rng = np.random.default_rng(5)
    st = rng.uniform(-1, 1, (k, n)).astype(np.float32)
    th = rng.uniform(-0.2, 1, (k, n)).astype(np.float32)
    br = (rng.uniform(0, 1, (k, n)) * (rng.uniform(0, 1, (k, n)) < 0.2)).astype(np.float32)
    seq = make_env("hip", n_envs=n, render=render, auto_reset=True)
    one = make_env("hip", n_envs=n, render=render, auto_reset=True)
    ora = make_env("oracle", n_envs=n, render=render, auto_reset=True)
    for env in (seq, one, ora):
        env.step_synthetic(3, 1)
    seq.step_sequence(st, th, br, steps_per_launch=5)
    ora.step_sequence(st, th, br, steps_per_launch=5)
    for t in range(k):
        one.step(st[t], th[t], br[t])
    names = ["seg_idx", "done", "ep_len"] + (["img"] if render else [])
    for name in names:
        assert np.array_equal(seq.fetch(name), one.fetch(name)), name
        assert np.array_equal(seq.fetch(name), ora.fetch(name)), name
    for name in ("pos_x", "pos_z", "speed", "cte", "yaw", "ep_return"):
        assert np.array_equal(seq.fetch(name), one.fetch(name)), name
        assert np.max(np.abs(seq.fetch(name) - ora.fetch(name))) <= 1e-5, name
