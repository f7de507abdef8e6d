"""Image path.  The numpy lines of the reference's filter are restated here WITH NUMPY
ITSELF (img_preprocessing.py:88-99 and keras_pilot.py:49-50 are numpy one-liners), which pins the oracle's
binary32 arithmetic; the OpenCV parts (cv2.mean as exact integer mean, 8-bit RGB->HSV, inRange) are
cross-checked against an independent float HSV.  GPU tests compare the HIP kernels with the oracle, bit for bit."""
import colorsys

import numpy as np
import pytest

from triton_racer_sim_amd.components import HipImgPreprocessing

CFGS = [
    {},                                                                                     # reference defaults: identity trim
    {"preprocessing_dynamic_brightness_enabled": True, "preprocessing_brightness_baseline": 550},
    {"preprocessing_contrast_enhancement_ratio": 1.37, "preprocessing_contrast_enhancement_offset": 125},
    {"preprocessing_dynamic_brightness_enabled": True, "preprocessing_brightness_baseline": 300,
     "preprocessing_contrast_enhancement_ratio": 0.6, "preprocessing_contrast_enhancement_offset": 90},
]


def frames(n=5, seed=0, h=120, w=160):
    rng = np.random.default_rng(seed)
    out = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    if n < 3:
        return out
    out[0] = 0
    out[1] = 255
    yy, xx = np.mgrid[0:h, 0:w]
    out[2] = np.stack([(xx * 255 // (w - 1)), (yy * 255 // (h - 1)), ((xx + yy) % 256)], -1).astype(np.uint8)
    return out


def numpy_trim(img, cfg):
    """img_preprocessing.py:84-99 with numpy; cv2.mean(img[40:119]) = per-channel mean in binary64 (+ a 0.0 4th entry)."""
    contrast = cfg.get("preprocessing_contrast_enhancement_ratio", 1.0)
    offset = cfg.get("preprocessing_contrast_enhancement_offset", 125)
    baseline = cfg.get("preprocessing_brightness_baseline", 550)
    roi = img[40:119, :, :]
    mean = [float(roi[:, :, c].astype(np.uint64).sum()) / float(roi.shape[0] * roi.shape[1]) for c in range(3)] + [0.0]
    current_brightness = sum(list(mean))
    delta = (baseline - current_brightness) / 3
    img_arr = img.astype(np.float32)
    if cfg.get("preprocessing_dynamic_brightness_enabled", False):
        img_arr += delta
    img_arr -= offset
    img_arr *= contrast
    img_arr += offset
    img_arr = np.clip(img_arr, 0, 255)
    return img_arr.astype(np.uint8)


@pytest.mark.parametrize("cfg", CFGS)
def test_oracle_trim_equals_numpy(make_env, cfg):
    env = make_env("oracle", n_envs=1, track=None, render=False)
    src = frames()
    got = env.preprocess_host(src, cfg)
    for i in range(len(src)):
        assert np.array_equal(got[i], numpy_trim(src[i], cfg)), (cfg, i)


def test_oracle_normalize_equals_numpy(make_env):
    env = make_env("oracle", n_envs=1, track=None, render=False)
    src = frames(3, seed=2)
    want = np.asarray(src, dtype=np.float32)
    want /= 255                                                           # keras_pilot.py:49-50
    assert np.array_equal(env.normalize_host(src), want)


def test_oracle_colour_masks_against_float_hsv(make_env):
    """OpenCV's 8-bit HSV is a fixed-point approximation of (H/2, S*255, V*255); away from the range bounds the
    in-range decision must agree with a float HSV (colorsys).  Default filters: white -> R, yellow -> G."""
    env = make_env("oracle", n_envs=1, track=None, render=False)
    cfg = {"preprocessing_color_filter_enabled": True}
    src = frames(4, seed=5)
    got = env.preprocess_host(src, cfg)
    bounds = [((0, 0, 130), (180, 64, 255)), ((25, 180, 155), (43, 255, 255))]
    rng = np.random.default_rng(1)
    checked = 0
    for _ in range(6000):
        i, y, x = rng.integers(0, 4), rng.integers(0, 120), rng.integers(0, 160)
        r, g, b = (int(c) for c in src[i, y, x])
        h, s, v = colorsys.rgb_to_hsv(r / 255, g / 255, b / 255)
        hsv = (h * 180, s * 255, v * 255)
        for f, (lo, hi) in enumerate(bounds):
            margin = min(min(abs(hsv[k] - lo[k]), abs(hsv[k] - hi[k])) for k in range(3))
            if margin < 1.5:
                continue                                                  # too close to a bound for a float check
            inside = all(lo[k] <= hsv[k] <= hi[k] for k in range(3))
            assert got[i, y, x, f] == (255 if inside else 0), (r, g, b, hsv, f)
            checked += 1
        assert got[i, y, x, 2] == src[i, y, x, 2]                         # blue channel untouched (identity trim)
    assert checked > 5000
    assert set(np.unique(got[..., :2])) <= {0, 255}


def test_oracle_rejects_bad_channels(make_env):
    env = make_env("oracle", n_envs=1, track=None, render=False)
    with pytest.raises(RuntimeError, match="dst_channel"):
        env.preprocess_host(frames(1), {"preprocessing_color_filter_enabled": True, "preprocessing_color_filter_hsvs": [((0, 0, 0), (180, 255, 255))],
                                        "preprocessing_color_filter_destination_channels": [3]})
    with pytest.raises(RuntimeError, match="edge_dst_channel"):
        env.preprocess_host(frames(1), {"preprocessing_edge_detection_enabled": True, "preprocessing_edge_detection_destination_channel": 5})


EDGE = {"preprocessing_edge_detection_enabled": True}


def test_oracle_canny_properties(make_env):
    """cv2 is absent, so the Canny restatement is checked through properties of the algorithm it states."""
    env = make_env("oracle", n_envs=1, track=None, render=False)
    img = np.zeros((1, 120, 160, 3), np.uint8)
    img[0, :, 80:] = 200                                                # one vertical step edge: |dx| = 4*200, dy = 0
    out = env.preprocess_host(img, EDGE)[0]
    cols = np.flatnonzero(out[:, :, 2].any(0))
    assert out[:, :, 2].max() == 255 and len(cols) == 1 and cols[0] in (79, 80)   # thinned to ONE column by the NMS
    assert (out[:, cols[0], 2] == 255).all()                            # replicated borders: the edge runs to both image borders
    assert np.array_equal(out[..., :2], img[0, ..., :2])                # the other channels keep the (identity-)trimmed image
    # thresholds are put in order; a gradient between them survives only when connected to a strong one
    weak = np.zeros((1, 120, 160, 3), np.uint8)
    weak[0, :, 80:] = 16                                                # |dx| = 4 * 16 = 64: between the thresholds 60 and 100
    for a, b in ((60, 100), (100, 60)):
        cfg = dict(EDGE, preprocessing_edge_detection_threshold_a=a, preprocessing_edge_detection_threshold_b=b)
        assert env.preprocess_host(weak, cfg)[0, :, :, 2].max() == 0    # weak only: dropped
    weak[0, :60, 80:] = 40                                              # upper half strong (|dx| = 160): the weak half hangs on it
    e = env.preprocess_host(weak, EDGE)[0, :, :, 2]
    assert e[5:50, 78:82].any(1).all() and e[70:115, 78:82].any(1).all()
    weak[0, 56:64] = 0                                                  # cut the connection; the band's lower corner stays weak (L1 = 48 + 48 < 100)
    e = env.preprocess_host(weak, EDGE)[0, :, :, 2]
    assert e[5:50, 78:82].any(1).all() and not e[66:].any()
    # merge order (img_preprocessing.py:43-53): the edge layer is merged after the colour masks
    both = dict(EDGE, preprocessing_color_filter_enabled=True, preprocessing_color_filter_hsvs=[((0, 0, 0), (180, 255, 255))],
                preprocessing_color_filter_destination_channels=[2])
    assert np.array_equal(env.preprocess_host(img, both)[0, :, :, 2], out[:, :, 2])


def test_component_handoff_semantics(oracle_api):
    """step() returns the frame processed from the PREVIOUS deposit; None first (img_preprocessing.py:18-21)."""
    part = HipImgPreprocessing({"preprocessing_contrast_enhancement_ratio": 1.2}, _api=oracle_api)
    assert part.step_inputs == ["cam/img"] and part.step_outputs == ["cam/processed_img"] and part.getName() == "Image Preprocessing"
    a, b = frames(2, seed=9)
    assert part.step(None) == (None,)
    assert part.step(a) == (None,)
    out = part.step(b)[0]
    assert np.array_equal(out, numpy_trim(a, part.cfg))
    assert np.array_equal(part.step(None)[0], numpy_trim(b, part.cfg))   # no new frame: the last result stays
    part.onShutdown()


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(120, 160), (240, 320), (64, 64)])
def test_gpu_preprocess_and_normalize_equal_oracle(make_env, size):
    h, w = size
    g = make_env("hip", n_envs=1, track=None, render=False, img_h=h, img_w=w)
    o = make_env("oracle", n_envs=1, track=None, render=False, img_h=h, img_w=w)
    src = frames(7, seed=3, h=h, w=w)
    for cfg in CFGS + [{"preprocessing_color_filter_enabled": True},
                       {"preprocessing_color_filter_enabled": True, "preprocessing_dynamic_brightness_enabled": True,
                        "preprocessing_color_filter_hsvs": [((0, 0, 100), (90, 255, 255)), ((90, 30, 0), (180, 255, 200)), ((10, 10, 10), (170, 200, 240))],
                        "preprocessing_color_filter_destination_channels": [2, 0, 2]}]:
        assert np.array_equal(g.preprocess_host(src, cfg), o.preprocess_host(src, cfg)), (size, cfg)
    assert np.array_equal(g.normalize_host(src), o.normalize_host(src))
    edge_cfgs = [EDGE, dict(EDGE, preprocessing_color_filter_enabled=True, preprocessing_dynamic_brightness_enabled=True,
                            preprocessing_edge_detection_threshold_a=30, preprocessing_edge_detection_threshold_b=20,
                            preprocessing_edge_detection_destination_channel=0)]
    for cfg in edge_cfgs:                                                   # 240x320 runs the global-scratch instantiation of the Canny kernel
        assert np.array_equal(g.preprocess_host(src, cfg), o.preprocess_host(src, cfg)), (size, cfg)


@pytest.mark.gpu
def test_gpu_preprocess_latest_frames_on_device(make_env):
    """Batched path: the env's own frames are filtered without leaving the device."""
    import ctypes
    g = make_env("hip", n_envs=48, auto_reset=True)
    o = make_env("oracle", n_envs=48, auto_reset=True)
    for env in (g, o):
        env.step_synthetic(20, 1)
    cfg = {"preprocessing_color_filter_enabled": True, "preprocessing_dynamic_brightness_enabled": True,
           "preprocessing_edge_detection_enabled": True}
    want = o.preprocess_host(o.fetch("img"), cfg)
    handle = g.preprocess_latest(cfg)
    got = g.preprocess_host(g.fetch("img"), cfg)                          # same frames through the host path
    assert np.array_equal(got, want)
    torch = pytest.importorskip("torch")
    g.sync()
    dev = torch.as_tensor(handle, device="cuda").cpu().numpy()
    assert np.array_equal(dev, want)


DYNAMIC = [
    {"preprocessing_dynamic_brightness_enabled": True},
    {"preprocessing_dynamic_brightness_enabled": True, "preprocessing_brightness_baseline": 420, "preprocessing_contrast_enhancement_ratio": 1.25,
     "preprocessing_color_filter_enabled": True},
]

FUSED = [
    {"preprocessing_contrast_enhancement_ratio": 1.37, "preprocessing_contrast_enhancement_offset": 110},
    {"preprocessing_color_filter_enabled": True},
    {"preprocessing_color_filter_enabled": True, "preprocessing_contrast_enhancement_ratio": 0.8,
     "preprocessing_color_filter_hsvs": [((0, 0, 100), (90, 255, 255)), ((35, 30, 0), (180, 255, 200)), ((10, 10, 10), (170, 200, 240))],
     "preprocessing_color_filter_destination_channels": [2, 0, 2]},
]


@pytest.mark.parametrize("cfg", FUSED + DYNAMIC)
def test_oracle_frame_filter_is_render_then_filter(make_env, cfg):
    """Definition of the fused mode: frames equal the raw frames pushed through ImgPreprocessing.__process."""
    raw = make_env("oracle", n_envs=24, auto_reset=True)
    fil = make_env("oracle", n_envs=24, auto_reset=True)
    fil.set_frame_filter(cfg)
    for env in (raw, fil):
        env.step_synthetic(9, 1)
    want = raw.preprocess_host(raw.fetch("img"), cfg)
    assert np.array_equal(fil.fetch("img"), want) and not np.array_equal(want, raw.fetch("img"))
    assert np.array_equal(fil.fetch("pos_x"), raw.fetch("pos_x"))           # the filter touches pixels only
    fil.set_frame_filter(enabled=False)
    for env in (raw, fil):
        env.step_synthetic(1, 1)
    assert np.array_equal(fil.fetch("img"), raw.fetch("img"))
    with pytest.raises(RuntimeError, match="not a palette filter"):
        fil.set_frame_filter({"preprocessing_edge_detection_enabled": True})


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", FUSED)
def test_gpu_fused_frame_filter_equals_oracle(make_env, cfg):
    """The product filters the PALETTE (no extra pass); the oracle renders and then filters every pixel."""
    g = make_env("hip", n_envs=96, auto_reset=True)
    o = make_env("oracle", n_envs=96, auto_reset=True)
    for env in (g, o):
        env.set_frame_filter(cfg)
        env.step_synthetic(7, 1)
    assert np.array_equal(g.fetch("img"), o.fetch("img"))
    for env in (g, o):                                                     # multi-step launches and a reload keep the filter
        env.step_synthetic(8, 4)
    assert np.array_equal(g.fetch("img"), o.fetch("img"))
    for env in (g, o):
        env.set_frame_filter(enabled=False)
        env.step_synthetic(2, 1)
    assert np.array_equal(g.fetch("img"), o.fetch("img"))
    with pytest.raises(RuntimeError, match="not a palette filter"):
        g.set_frame_filter({"preprocessing_edge_detection_enabled": True})


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", DYNAMIC)
@pytest.mark.parametrize("shape", [(120, 160, 101, False), (240, 320, 37, True), (64, 64, 300, False), (120, 160, 7, False)])   # (7: a last batch of three envs)
def test_gpu_fused_dynamic_brightness_equals_oracle(make_env, cfg, shape):
    """Dynamic brightness inside the step kernel (class histogram of rows 40..118 -> per-env palette) against the
    oracle's render-then-filter, bit for bit; env counts that leave the last workgroup partly filled; single-step calls,
    pipelined calls and multi-step launches."""
    h, w, n, depth = shape
    g = make_env("hip", n_envs=n, auto_reset=True, img_h=h, img_w=w, depth=depth)
    o = make_env("oracle", n_envs=n, auto_reset=True, img_h=h, img_w=w, depth=depth)
    for env in (g, o):
        env.set_frame_filter(cfg)
        env.step_synthetic(1, 1)
    assert np.array_equal(g.fetch("img"), o.fetch("img"))
    for env in (g, o):
        env.step_synthetic(5, 1)
    assert np.array_equal(g.fetch("img"), o.fetch("img"))
    for env in (g, o):
        env.step_synthetic(6, 3)
    assert np.array_equal(g.fetch("img"), o.fetch("img"))
    if depth:
        assert np.array_equal(g.fetch("depth"), o.fetch("depth"))
    assert np.array_equal(g.fetch("seg_idx"), o.fetch("seg_idx"))
    for env in (g, o):                                                     # and back to a static palette filter, then raw
        env.set_frame_filter(FUSED[1]); env.step_synthetic(2, 1)
    assert np.array_equal(g.fetch("img"), o.fetch("img"))
    for env in (g, o):
        env.set_frame_filter(enabled=False); env.step_synthetic(2, 1)
    assert np.array_equal(g.fetch("img"), o.fetch("img"))


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", DYNAMIC)
@pytest.mark.parametrize("shape", [(120, 160, 101, False), (240, 320, 37, True), (120, 160, 1024, False), (64, 64, 300, False), (120, 160, 6, True)])   # (6: a last batch of two envs, with depth)
def test_gpu_dynamic_brightness_in_resident_mode_equals_oracle(make_env, cfg, shape):
    """Round 3: the dynamic-brightness frame filter has its own instantiation of the resident worker (until round 2 resident-mode
    calls silently fell back to launches, include/trsim.h).  Every step posted on its own, lock-step calls, a step sequence, a filter
    change while the worker is resident, host-array controls: frames and state bit for bit the oracle's render-then-filter."""
    h, w, n, depth = shape
    g = make_env("hip", n_envs=n, auto_reset=True, img_h=h, img_w=w, depth=depth)
    o = make_env("oracle", n_envs=n, auto_reset=True, img_h=h, img_w=w, depth=depth)
    g.set_step_mode(True)
    before = int(g.fetch("stats")[2])
    for env in (g, o):
        env.set_frame_filter(cfg)
        env.step_synthetic(1, 1)
    assert np.array_equal(g.fetch("img"), o.fetch("img"))
    for env in (g, o):
        env.step_synthetic(23, 1)                                       # a queue of posts: the physics team runs ahead, arrivals lag
    assert np.array_equal(g.fetch("img"), o.fetch("img"))
    for k in range(4):                                                  # lock step: post, wait for the frame
        for env in (g, o):
            env.step_synthetic(1, 1)
        assert np.array_equal(g.fetch("img"), o.fetch("img")), k
    steer = np.linspace(-1, 1, n).astype(np.float32)
    for env in (g, o):
        env.step(steer, 0.6, 0.0)                                       # host arrays through the pinned staging ring
        env.step_synthetic(3, 1)
    assert np.array_equal(g.fetch("img"), o.fetch("img"))
    if depth:
        assert np.array_equal(g.fetch("depth"), o.fetch("depth"))
    for name in ("seg_idx", "done", "ep_len"):
        assert np.array_equal(g.fetch(name), o.fetch(name)), name
    for env in (g, o):                                                  # the filter changes under a resident worker: it leaves, the next post starts the other instantiation
        env.set_frame_filter(FUSED[1]); env.step_synthetic(3, 1)
    assert np.array_equal(g.fetch("img"), o.fetch("img"))
    for env in (g, o):
        env.set_frame_filter(cfg); env.step_synthetic(2, 1)
    assert np.array_equal(g.fetch("img"), o.fetch("img"))
    assert int(g.fetch("stats")[2]) == before                           # no layout fault
