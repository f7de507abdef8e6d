"""The other two model types the reference's KerasPilot serves (components/keras_pilot.py:67-76,97-118):
``cnn_2d_speed_as_feature`` = Keras_2D_CNN.get_model(num_feature_vectors=1) (components/keras_train.py:127-174,390-392) and
``cnn_2d_full_house`` = Keras_2D_FULL_HOUSE.get_model (keras_train.py:184-245).  The checker is a plain PyTorch fp32 restatement
of those architectures with seeded random weights (no model file ships; TensorFlow is absent): conv stack as in
tests/test_pilot.py, the small dense branches and both heads in fp32.  Tolerances as for cnn_2d_speed_control: the two outputs
within 1e-3 of fp32-everywhere (the fp16 effect of the conv stack; 5e-2 in the bfloat16 rounds), and within 5e-4 of a reference whose conv stack is the
kernel's own conv7 activation (everything behind it is fp32 on both sides)."""
import math

import numpy as np
import pytest

from test_pilot import SPEC, make_weights, pilot_postprocess, torch_layer

pytestmark = pytest.mark.gpu


def dense_init(rng, a, b):
    lim = math.sqrt(6.0 / (a + b))
    return rng.uniform(-lim, lim, (a, b)).astype(np.float32), rng.uniform(-0.05, 0.05, b).astype(np.float32)


def make_named_weights(h, w, kind, seed=0):
    """{layer name: (kernel, bias)} in Keras layouts for the two architectures."""
    rng = np.random.default_rng(seed)
    base = make_weights(h, w, seed=seed + 100)
    named = {f"conv{i + 1}": (base[2 * i], base[2 * i + 1]) for i in range(7)}
    ih, iw = h, w
    for k, s, _, _ in SPEC:
        ih, iw = (ih - k) // s + 1, (iw - k) // s + 1
    F = ih * iw * 128
    if kind == "cnn_2d_speed_as_feature":
        f = (4, 8, 16)
        named["feature1"], named["feature2"], named["feature3"] = dense_init(rng, 1, f[0]), dense_init(rng, f[0], f[1]), dense_init(rng, f[1], f[2])
        named["dense1"] = dense_init(rng, F + f[2], 100)
        named["dense2"], named["dense3"], named["output_layer"] = dense_init(rng, 100, 50), dense_init(rng, 50, 25), dense_init(rng, 25, 2)
    else:
        f = (16, 32, 64)
        named["feature1"], named["feature2"], named["feature3"] = dense_init(rng, 1, f[0]), dense_init(rng, f[0], f[1]), dense_init(rng, f[1], f[2])
        named["current_spd_1"], named["current_spd_2"], named["current_spd_3"] = dense_init(rng, 1, f[0]), dense_init(rng, f[0], f[1]), dense_init(rng, f[1], f[2])
        named["dense1"] = dense_init(rng, F + 64, 100)
        named["dense2"], named["dense3"], named["output_speed"] = dense_init(rng, 100, 50), dense_init(rng, 50, 25), dense_init(rng, 25, 1)
        named["dense4"] = dense_init(rng, F + 128, 100)
        named["dense5"], named["dense6"], named["out_steering"] = dense_init(rng, 100, 50), dense_init(rng, 50, 25), dense_init(rng, 25, 1)
    # give the extra rows of dense1 / dense4 weight: the branches must matter for the test to see them
    for nm in ("dense1", "dense4"):
        if nm in named:
            k, b = named[nm]
            k[F:] *= 20.0
    return named, F


def torch_heads(x_flat, speed, segment, named, kind, h16_dense=False):
    """Everything behind the flatten, fp32 (with ``h16_dense`` the flatten rows of dense1 / dense4 and x are rounded to fp16, as the
    matrix-core path does)."""
    import torch
    import torch.nn.functional as Fn
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    lin = lambda z, nm, act=True: (Fn.relu(z @ t(named[nm][0]) + t(named[nm][1])) if act else z @ t(named[nm][0]) + t(named[nm][1]))
    x = t(x_flat)
    F = x.shape[1]

    def big(z_extra, nm):
        k, b = t(named[nm][0]), t(named[nm][1])
        kx = k[:F].half().float() if h16_dense else k[:F]
        xx = x.half().float() if h16_dense else x
        return Fn.relu(xx @ kx + z_extra @ k[F:] + b)

    spd = t(speed).reshape(-1, 1) / 20.0
    if kind == "cnn_2d_speed_as_feature":
        y = lin(lin(lin(spd, "feature1"), "feature2"), "feature3")
        z = big(y, "dense1")
        return lin(lin(lin(z, "dense2"), "dense3"), "output_layer", act=False).numpy()
    y = lin(lin(lin(t(segment).reshape(-1, 1), "feature1"), "feature2"), "feature3")
    s = lin(lin(lin(spd, "current_spd_1"), "current_spd_2"), "current_spd_3")
    out_speed = lin(lin(lin(big(y, "dense1"), "dense2"), "dense3"), "output_speed", act=False)
    out_steer = lin(lin(lin(big(torch.cat([y, s], 1), "dense4"), "dense5"), "dense6"), "out_steering", act=False)
    return torch.cat([out_steer, out_speed], 1).numpy()


def conv_stack_pure(frames, named):
    ws = [a for i in range(7) for a in named[f"conv{i + 1}"]]
    x = frames
    for i in range(7):
        x = torch_layer(i, x, ws, mirror=False)
    return x.reshape(x.shape[0], -1)


@pytest.mark.parametrize("kind", ["cnn_2d_speed_as_feature", "cnn_2d_full_house"])
def test_forward_matches_torch_fp32(make_env, kind):
    h, w, n = 120, 160, 6
    env = make_env("hip", n_envs=n, img_h=h, img_w=w, auto_reset=True)
    named, F = make_named_weights(h, w, kind, seed=3)
    env.pilot_load(named)
    env.step_synthetic(12, 1)
    rng = np.random.default_rng(1)
    frames = np.concatenate([env.fetch("img")[:4], rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)])
    speed = rng.uniform(0, 20, n).astype(np.float32)
    segment = rng.uniform(0, 10, n).astype(np.float32)
    out = env.pilot_forward_host(frames, speed=speed, segment=segment if kind == "cnn_2d_full_house" else None)
    pure = torch_heads(conv_stack_pure(frames, named), speed, segment, named, kind)
    assert np.max(np.abs(out - pure)) <= 1e-3, float(np.max(np.abs(out - pure)))
    oh, ow = h, w
    for k, s_, _, _ in SPEC:
        oh, ow = (oh - k) // s_ + 1, (ow - k) // s_ + 1
    shape7 = (n, oh, ow, 128)
    x7 = env.pilot_layer(6, shape7).reshape(n, -1)                    # the kernel's own conv7 activation (fp16 values as fp32)
    assert x7.shape[1] == F
    mirror = torch_heads(x7, speed, segment, named, kind, h16_dense=True)
    assert np.max(np.abs(out - mirror)) <= 5e-4, float(np.max(np.abs(out - mirror)))
    # the extra inputs are really read: other speeds / segments, other outputs
    out2 = env.pilot_forward_host(frames, speed=speed[::-1].copy(), segment=(segment[::-1].copy() if kind == "cnn_2d_full_house" else None))
    assert np.max(np.abs(out2 - out)) > 1e-3
    again = env.pilot_forward_host(frames, speed=speed, segment=segment if kind == "cnn_2d_full_house" else None)
    assert np.array_equal(again, out)                                 # fixed summation orders: bit-identical reruns


@pytest.mark.parametrize("kind", ["cnn_2d_speed_as_feature", "cnn_2d_full_house"])
def test_closed_loop_equals_the_loop_done_by_hand(make_env, kind):
    """trs_step_pilot with the env's own speed / tracker index == [forward(previous frame, speed, segment); KerasPilot's
    post-processing; env.step] by hand (keras_pilot.py:67-76 / 97-118)."""
    n = 16
    named, _ = make_named_weights(120, 160, kind, seed=8)
    out_name = "output_layer" if kind == "cnn_2d_speed_as_feature" else "output_speed"
    k, b = named[out_name]
    named[out_name] = (k, b + np.float32(0.5))                        # bias towards driving
    cfg = {"model_type": kind, "spd_ctl_threshold": 1.1, "spd_ctl_reverse_multiplier": 1.0}
    a, bb = make_env("hip", n_envs=n), make_env("hip", n_envs=n)
    a.pilot_load(named); bb.pilot_load(named)
    a.step_pilot(6, cfg)
    bb.step(0.0, 0.0, 0.0)
    for _ in range(5):
        spd = bb.fetch("speed")
        seg = bb.segment(bb.fetch("seg_idx")).astype(np.float32)      # 'loc/segment' = idx / n_points * 10 (track_data_process.py:106-107)
        out = bb.pilot_forward_host(bb.fetch("img"), speed=spd, segment=seg if kind == "cnn_2d_full_house" else None)
        if kind == "cnn_2d_speed_as_feature":
            ctl = np.stack([np.clip(out[:, 0], -1, 1), np.clip(out[:, 1], -1, 1), np.zeros(n, np.float32)], 1).astype(np.float32)
        else:
            ctl = np.array([pilot_postprocess(out[i], float(spd[i]), cfg) for i in range(n)], dtype=np.float32)
        bb.step(ctl[:, 0], ctl[:, 1], ctl[:, 2])
    for name in ("pos_x", "pos_z", "yaw", "speed"):
        assert np.max(np.abs(a.fetch(name) - bb.fetch(name))) <= 1e-4, name
    assert np.array_equal(a.fetch("seg_idx"), bb.fetch("seg_idx")) and np.array_equal(a.fetch("img"), bb.fetch("img"))
    assert a.fetch("speed").max() > 0.05
    with pytest.raises(RuntimeError, match="model_type"):
        a.step_pilot(1, {"model_type": "cnn_2d_speed_control"})       # the loaded weights are another architecture's
