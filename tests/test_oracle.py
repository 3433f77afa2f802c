"""CPU tests that pin the oracle (oracle/w2xc_oracle.c).

The reference ships no golden vectors (src/test.cpp asserts nothing), so the pins are:
  (1) bit-equality with oracle/_ref -- the reference's OWN modelHandler.cpp/convertRoutine.cpp
      compiled against the OpenCV shim (prebuilt; travels with the repo);
  (2) the committed fixtures in tests/golden/ (generated from oracle/_ref by tests/golden/make_golden.py);
  (3) known-answer models: identity kernel, 9 single-tap shift kernels (pins correlation-not-convolution
      and the (r,c) order), bias-only (pins LeakyReLU on the last layer, Q1);
  (4) an independent torch fp64 conv2d restatement;
  (5) SURVEY invariants I1 (pad/crop == valid conv) and I2 (block split == unsplit).
"""
import json
import os

import numpy as np
import pytest

from conftest import ramp_plane, rand_plane, small_layers
from tools import gen_model
from oracle import oracle as orc

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def one_layer(w, b):
    w = np.ascontiguousarray(w, dtype=np.float32)
    return (w.shape[1], w.shape[0], w, np.ascontiguousarray(b, dtype=np.float64))


def test_identity_kernel(oracle_built):
    k = np.zeros((1, 1, 3, 3), np.float32)
    k[0, 0, 1, 1] = 1.0
    o = orc.Oracle([one_layer(k, [0.0])])
    x = rand_plane(19, 23, 0) + 0.5   # positive: leaky is the identity
    assert np.array_equal(o.convert(x), x)
    assert np.array_equal(o.filter(0, x[None])[0], x)


@pytest.mark.parametrize("r,c", [(r, c) for r in range(3) for c in range(3)])
def test_single_tap_is_correlation(oracle_built, r, c):
    """filter2D is correlation: tap (r,c) reads src(y+r-1, x+c-1), clamped (modelHandler.cpp:141-142)."""
    k = np.zeros((1, 1, 3, 3), np.float32)
    k[0, 0, r, c] = 1.0
    o = orc.Oracle([one_layer(k, [0.0])])
    x = ramp_plane(11, 14) + 1.0
    h, w = x.shape
    ys = np.clip(np.arange(h) + r - 1, 0, h - 1)
    xs = np.clip(np.arange(w) + c - 1, 0, w - 1)
    want = x[ys][:, xs]
    assert np.array_equal(o.filter(0, x[None])[0], want)
    assert np.array_equal(o.convert(x), want)   # pad-1/crop-1 with replicate == same-size replicate


def test_bias_only_leaky_on_last_layer(oracle_built):
    """Q1: LeakyReLU(0.1) is applied on every layer including the last (modelHandler.cpp:147-152)."""
    k = np.zeros((2, 1, 3, 3), np.float32)
    b = np.array([0.25, -0.3])
    o = orc.Oracle([one_layer(k, b)])
    out = o.filter(0, np.zeros((1, 5, 6), np.float32))
    assert np.all(out[0] == np.float32(0.25))
    assert np.all(out[1] == np.float32(-0.3) * np.float32(0.1))


def test_plane_count_mismatch(oracle_built):
    o = orc.Oracle(small_layers([2, 3], 5))
    assert o.filter(0, np.zeros((3, 4, 4), np.float32)) is None   # Model::filter returns false (:29-35)


@pytest.mark.parametrize("h,w", [(1, 1), (7, 9), (40, 52), (31, 64)])
def test_bit_equal_to_reference_build_unsplit(oracle_built, models_dir, h, w):
    p = os.path.join(models_dir, "noise1_model.json")
    o, ref = orc.Oracle.from_json(p), orc.RefBuild(p)
    x = rand_plane(h, w, 3)
    assert np.array_equal(o.convert(x, block_splitting=False), ref.convert(x, block_splitting=False))


@pytest.mark.parametrize("h,w,blk", [(70, 45, 32), (36, 37, 32), (18, 19, 32), (50, 50, 40)])
def test_bit_equal_to_reference_build_blocksplit(oracle_built, tmp_path, h, w, blk):
    """block walk of convertRoutine.cpp:84-169 incl. exact multiples (stride blk-2n) and 1-px remainders"""
    layers = small_layers([1, 4, 6, 1], 9)
    p = gen_model.write_json(layers, str(tmp_path / "m.json"))
    o, ref = orc.Oracle.from_json(p), orc.RefBuild(p)
    x = rand_plane(h, w, 4)
    a = o.convert(x, block=(blk, blk))
    b = ref.convert(x, block=(blk, blk))
    assert np.array_equal(a, b)
    assert np.array_equal(a, o.convert(x, block_splitting=False))   # invariant I2


@pytest.mark.parametrize("njob", [1, 3, 4, 7])
def test_filter_thread_partition_matches_reference(oracle_built, tmp_path, njob):
    layers = small_layers([3, 10], 11)
    p = gen_model.write_json(layers, str(tmp_path / "m.json"))
    o, ref = orc.Oracle.from_json(p), orc.RefBuild(p)
    x = np.random.default_rng(2).standard_normal((3, 9, 13)).astype(np.float32)
    assert np.array_equal(o.filter(0, x, njob=njob), ref.filter(0, x, njob=njob))


def test_json_roundtrip_is_exact(tmp_path):
    layers = small_layers([1, 5, 2], 13)
    p = gen_model.write_json(layers, str(tmp_path / "m.json"))
    back = orc.load_model_json(p)
    for a, b in zip(layers, back):
        assert a[0] == b[0] and a[1] == b[1]
        assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    assert json.load(open(p))[0]["kW"] == 3


def test_against_torch_fp64(oracle_built):
    """independent restatement: valid conv2d (cross-correlation) on the replicate-padded plane (I1)"""
    torch = pytest.importorskip("torch")
    import torch.nn.functional as F
    layers = small_layers([1, 8, 8, 1], 17)
    o = orc.Oracle(layers)
    x = rand_plane(21, 26, 6)
    t = torch.from_numpy(x.astype(np.float64))[None, None]
    t = F.pad(t, (3, 3, 3, 3), mode="replicate")
    for nin, nout, w, b in layers:
        t = F.conv2d(t, torch.from_numpy(w.astype(np.float64)), torch.from_numpy(b))
        t = torch.where(t > 0, t, 0.1 * t)
    want = t[0, 0].numpy()
    got = o.convert(x)
    assert np.abs(got - want).max() < 2e-6
    assert np.abs(o.convert_f64(x) - want).max() < 1e-12


def test_golden_fixtures(oracle_built):
    """tests/golden/*.npz were produced by the reference's own code (oracle/_ref) -- see make_golden.py"""
    files = sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz"))
    assert files, "no golden fixtures committed"
    for f in files:
        g = np.load(os.path.join(GOLDEN, f))
        planes = [int(v) for v in g["planes"]]
        layers = gen_model.synth_layers(planes, int(g["seed"]), init=str(g["init"]) if "init" in g else "he_leaky")
        o = orc.Oracle(layers)
        got = o.convert(g["input"], block=(int(g["block"]), int(g["block"])))
        assert np.array_equal(got, g["output"]), f


def test_color_oracle_sanity(oracle_built):
    """N2 restatement (oracle/w2xc_oracle_color.c): u8 -> YUV -> u8 is the identity on every byte triple sampled,
    the cubic kernel agrees with an independent implementation (torch bicubic, A = -0.75, half-pixel centres),
    nearest 2x is a pixel repeat."""
    torch = pytest.importorskip("torch")
    import torch.nn.functional as F
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    y, u, v = orc.u8_to_yuv(img)
    assert np.array_equal(orc.yuv_to_u8(y, u, v), img)
    assert np.allclose(y, (0.299 * img[..., 0] + 0.587 * img[..., 1] + 0.114 * img[..., 2]) / 255.0, atol=1e-6)
    x = rng.random((9, 11), dtype=np.float32)
    t = F.interpolate(torch.from_numpy(x)[None, None], scale_factor=2, mode="bicubic", align_corners=False)[0, 0].numpy()
    assert np.abs(orc.resize2x_cubic(x) - t).max() < 1e-6
    assert np.array_equal(orc.resize2x_nearest(x), np.repeat(np.repeat(x, 2, 0), 2, 1))
    const = np.full((5, 6), 0.37, np.float32)
    assert np.abs(orc.resize2x_cubic(const) - 0.37).max() < 1e-6       # partition of unity incl. clipped borders
