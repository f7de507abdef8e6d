"""Edge cases and limits of the boundary: single env, ragged env counts, odd image sizes, configuration errors — the same
behaviour on the oracle (CPU) and, under -m gpu, on the HIP library."""
import numpy as np
import pytest

from conftest import track_points


def check_config_errors(make_env, kind):
    for bad in (dict(n_envs=0), dict(img_w=158), dict(img_h=1), dict(env_id_base=-1)):
        with pytest.raises(RuntimeError, match="bad n_envs / image size"):
            make_env(kind, **bad)
    with pytest.raises(TypeError, match="unknown env parameter"):
        make_env(kind, n_envs=1, warp_drive=1)
    with pytest.raises(ValueError):
        make_env(kind, n_envs=1, track=np.zeros((1, 3)))
    env = make_env(kind, n_envs=3, track=None)
    with pytest.raises(RuntimeError, match="no track loaded"):
        env.step(0.0, 0.0)
    with pytest.raises(RuntimeError, match="no track loaded"):
        env.locate([[0, 0, 0]])
    env.load_track(track_points("generated")[:2])                     # the smallest legal track: two distinct points
    env.step(0.0, 0.5)
    with pytest.raises(RuntimeError, match="n_steps"):
        env.step(0.0, 0.0, n_steps=0)
    with pytest.raises(RuntimeError, match="byte count mismatch"):
        import ctypes
        buf = np.zeros(2, np.float32)
        env.api.check(env.api.copy_to_host(env._h, 1, buf.ctypes.data, buf.nbytes), "copy_to_host")
    two_same = np.tile(track_points("generated")[:1], (3, 1))
    with pytest.raises(RuntimeError, match="no two distinct points"):
        env.load_track(two_same)


def test_config_errors_oracle(make_env):
    check_config_errors(make_env, "oracle")


def test_reload_track_resets_the_env(make_env):
    env = make_env("oracle", n_envs=5, auto_reset=True)
    env.step_synthetic(30, 1)
    env.load_track(track_points("mountain"))
    assert env.state_view().step_count == 0 and env.n_points == 2664
    assert np.array_equal(env.fetch("seg_idx"), (37 * np.arange(5)) % 2664)
    env.step_synthetic(5, 1)
    assert (env.fetch("ep_len") == 4).all()


@pytest.mark.gpu
def test_config_errors_gpu(make_env):
    check_config_errors(make_env, "hip")
    with pytest.raises(RuntimeError, match="device index out of range"):
        make_env("hip", n_envs=1, device=99)
    with pytest.raises(RuntimeError, match="img_w too large"):
        make_env("hip", n_envs=1, img_w=2564)                          # more 4-pixel groups per row than raster threads


@pytest.mark.gpu
@pytest.mark.parametrize("n,h,w", [(1, 120, 160), (2, 2, 4), (255, 120, 160), (257, 30, 44), (1281, 16, 24), (3, 120, 636)])
def test_ragged_sizes_match_oracle(make_env, n, h, w):
    """Env counts around the 256-CU / 5-physics-wave boundaries (1 env, 255, 257, > 1280 = several envs per physics wave)
    and image sizes that leave raster lanes idle or a single row pass."""
    g = make_env("hip", n_envs=n, img_h=h, img_w=w, auto_reset=True)
    o = make_env("oracle", n_envs=n, img_h=h, img_w=w, auto_reset=True)
    for env in (g, o):
        env.step_synthetic(23, 1)
        env.step_synthetic(14, 5)
        env.step(0.3, 0.8, n_steps=3)
    assert np.array_equal(g.fetch("img"), o.fetch("img"))
    for name in ("pos_x", "pos_z", "yaw", "ep_return", "cte"):
        assert np.array_equal(g.fetch(name), o.fetch(name)), name
    for name in ("seg_idx", "done", "ep_len"):
        assert np.array_equal(g.fetch(name), o.fetch(name)), name
    assert np.array_equal(g.fetch("stats")[:2], o.fetch("stats")[:2])   # off-track events and resets, counted on the device


@pytest.mark.gpu
def test_reload_track_on_gpu(make_env):
    g = make_env("hip", n_envs=40, auto_reset=True)
    o = make_env("oracle", n_envs=40, auto_reset=True)
    for env in (g, o):
        env.step_synthetic(10, 4)
        env.load_track(track_points("mountain"))
        env.step_synthetic(33, 8)
    assert np.array_equal(g.fetch("img"), o.fetch("img")) and np.array_equal(g.fetch("seg_idx"), o.fetch("seg_idx"))
