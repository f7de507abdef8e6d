"""Tracks with elevation on the GPU (include/trsim_spec.h, "tracks with elevation"; round 5): the rasteriser evaluates a frame's row tables per env from the
slope ahead of the car's track point.  HIP == oracle (frames, depth frames, indices bit-exact; pose within 1e-5) on the reference's mountain track
(car_templates/track_data/mountain_track.json) in every step path; tests/test_hills_spec.py pins the oracle to an independent numpy restatement on the CPU.
Flat tracks are untouched: every other GPU test runs on the generated track through the same kernels."""
import numpy as np
import pytest

from conftest import track_points
from test_gpu_parity import assert_state_equal

pytestmark = pytest.mark.gpu


def frames_equal(g, o, where, depth=False):
    a, b = g.fetch("img"), o.fetch("img")
    bad = np.flatnonzero((a != b).reshape(a.shape[0], -1).any(axis=1))
    assert bad.size == 0, f"{where}: frames of envs {bad[:8]} differ"
    if depth:
        assert np.array_equal(g.fetch("depth").view(np.uint32), o.fetch("depth").view(np.uint32)), f"{where}: depth frames differ"


def test_dpitch_table_and_hilliness(make_env):
    pts = track_points("mountain")
    g, o = make_env("hip", n_envs=2, track=pts), make_env("oracle", n_envs=2, track=pts)
    assert np.array_equal(g.fetch("dpitch"), o.fetch("dpitch"))
    assert np.abs(g.fetch("dpitch")).max() > 0.02
    flat = make_env("hip", n_envs=2)
    assert not flat.fetch("dpitch").any()


@pytest.mark.parametrize("n,h,w,depth", [(64, 120, 160, True), (300, 120, 160, False), (9, 240, 320, True), (5, 60, 80, False)])
def test_hilly_track_launch_mode_equals_oracle(make_env, n, h, w, depth):
    """One launch per step (the raster team waits for that step's pose and pitch), multi-step launches (the pitch rides through the LDS ring and, across launches,
    through the global ring beside the camera parameters), host-array controls with resets."""
    pts = track_points("mountain")
    g = make_env("hip", n_envs=n, track=pts, img_h=h, img_w=w, depth=depth, auto_reset=True)
    o = make_env("oracle", n_envs=n, track=pts, img_h=h, img_w=w, depth=depth, auto_reset=True)
    for env in (g, o):
        env.step_synthetic(7, 1)
    assert_state_equal(g, o, "7 single-step launches")
    frames_equal(g, o, "7 single-step launches", depth)
    for env in (g, o):
        env.step_synthetic(13, 4)
    assert_state_equal(g, o, "13 steps, 4 per launch")
    frames_equal(g, o, "13 steps, 4 per launch", depth)
    rng = np.random.default_rng(5)
    for k in range(5):
        st, th = rng.uniform(-1, 1, n).astype(np.float32), rng.uniform(0.2, 1, n).astype(np.float32)
        rs = (rng.uniform(0, 1, n) < 0.1) if k == 2 else None
        for env in (g, o):
            env.step(st, th, 0.0, reset=rs)
    assert_state_equal(g, o, "host controls")
    frames_equal(g, o, "host controls", depth)
    # the view really changes with the slope: the row where the ground begins differs between envs
    if depth:
        d = g.fetch("depth")[:, :, 0]
        first_ground = np.array([int(np.argmax(row < np.float32(40.0))) for row in d])
        assert first_ground.max() - first_ground.min() >= 1, first_ground


@pytest.mark.parametrize("n,depth", [(256, False), (1024, True), (37, True)])
def test_hilly_track_resident_mode_equals_oracle(make_env, n, depth):
    pts = track_points("mountain")
    g = make_env("hip", n_envs=n, track=pts, depth=depth, auto_reset=True)
    o = make_env("oracle", n_envs=n, track=pts, depth=depth, auto_reset=True)
    g.set_step_mode(True)
    for env in (g, o):
        env.step_synthetic(21, 1)                              # queued posts
    assert_state_equal(g, o, "21 posted steps")
    frames_equal(g, o, "21 posted steps", depth)
    rng = np.random.default_rng(6)
    for k in range(6):                                         # lock step with host controls
        st, th = rng.uniform(-1, 1, n).astype(np.float32), rng.uniform(0.2, 1, n).astype(np.float32)
        g.step(st, th, 0.0); g.sync()
        o.step(st, th, 0.0)
    assert_state_equal(g, o, "lock step")
    frames_equal(g, o, "lock step", depth)
    g.load_track(track_points("generated")); o.load_track(track_points("generated"))      # back to a flat track on the same handle
    for env in (g, o):
        env.step_synthetic(9, 1)
    assert_state_equal(g, o, "flat track after a hilly one")
    frames_equal(g, o, "flat track after a hilly one", depth)


def test_frame_filters_on_a_hilly_track(make_env):
    """The static frame filter behind the rasteriser (trs_set_frame_filter: trim + HSV masks, components/img_preprocessing.py:37-74,92-99) on a track with elevation: a row's
    ground colours are blended per env and frame inside the kernel, so the filter of one colour runs there on each of them (hill_filter_colour) — against the oracle, which
    renders the raw frame and filters every pixel.  The dynamic-brightness variant is refused there with the reason (its palette and the per-env row tables are two
    instantiations of the kernels that do not combine yet); trs_preprocess on the rendered frames works on any track."""
    pts = track_points("mountain")
    cfgs = ({"preprocessing_contrast_enhancement_ratio": 1.25, "preprocessing_contrast_enhancement_offset": 100.0},
            {"preprocessing_contrast_enhancement_ratio": 1.2, "preprocessing_color_filter_enabled": True})
    for cfg in cfgs:
        g, o = make_env("hip", n_envs=40, track=pts, auto_reset=True), make_env("oracle", n_envs=40, track=pts, auto_reset=True)
        for env in (g, o):
            env.set_frame_filter(cfg)
            env.step_synthetic(9, 1)
        frames_equal(g, o, f"static filter {cfg}, launches")
        g.set_step_mode(True)
        for env in (g, o):
            env.step_synthetic(11, 1)
        frames_equal(g, o, f"static filter {cfg}, resident")
        for env in (g, o):
            env.set_frame_filter(enabled=False)
            env.step_synthetic(3, 1)
        frames_equal(g, o, "raw frames again")
    g = make_env("hip", n_envs=4, track=pts)
    with pytest.raises(RuntimeError, match="elevation"):
        g.set_frame_filter({"preprocessing_contrast_enhancement_ratio": 1.2, "preprocessing_dynamic_brightness_enabled": True})
    flat = make_env("hip", n_envs=4)
    flat.set_frame_filter({"preprocessing_contrast_enhancement_ratio": 1.2, "preprocessing_dynamic_brightness_enabled": True})
    with pytest.raises(RuntimeError, match="elevation"):
        flat.load_track(pts)                                   # the dynamic filter is removed, the track is loaded
    o = make_env("oracle", n_envs=4, track=pts)
    for env in (flat, o):
        env.step_synthetic(3, 1)
    frames_equal(flat, o, "after the refused dynamic filter")
    cfg = {"preprocessing_contrast_enhancement_ratio": 1.2, "preprocessing_color_filter_enabled": True, "preprocessing_dynamic_brightness_enabled": True}
    assert np.array_equal(flat.preprocess_host(flat.fetch("img"), cfg), o.preprocess_host(o.fetch("img"), cfg))
