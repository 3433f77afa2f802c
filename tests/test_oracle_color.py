"""Pins for the N2 / N4 restatement (oracle/w2xc_oracle_color.c): every OpenCV primitive the CLI's colour front/back end
uses (main.cpp:74-76,136,144,158-167,171-172) against an INDEPENDENT implementation -- numpy fp64 matrices for the colour
transforms and rounding, torch's own bicubic / bilinear for the resizes.  OpenCV itself is absent; what these
checks cannot settle is listed at the end of the C file's header."""
import numpy as np
import pytest

from oracle import oracle as orc

torch = pytest.importorskip("torch")
import torch.nn.functional as F  # noqa: E402


def test_convert_to_float(oracle_built):
    """convertTo(CV_32F, 1/255): (float)u * (float)(1/255.0), for all 256 byte values"""
    img = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, axis=2)
    y, u, v = orc.u8_to_yuv(img)
    c = img[..., 0].astype(np.float32) * np.float32(1.0 / 255.0)
    # Y of a grey pixel = c*(0.299+0.587+0.114) evaluated as three float products summed left to right
    want = (c * np.float32(0.299) + c * np.float32(0.587)) + c * np.float32(0.114)
    assert np.array_equal(y, want.astype(np.float32))
    assert np.abs(u - 0.5).max() < 1e-6 and np.abs(v - 0.5).max() < 1e-6      # grey has no chroma


def test_yuv_matrix(oracle_built):
    """RGB2YUV / YUV2RGB against the documented equations in fp64 matrix form (independent of the C evaluation order)"""
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (40, 50, 3), dtype=np.uint8)
    y, u, v = orc.u8_to_yuv(img)
    c = img.astype(np.float64) / 255.0
    Y = c @ np.array([0.299, 0.587, 0.114])
    U = 0.492 * (c[..., 2] - Y) + 0.5
    V = 0.877 * (c[..., 0] - Y) + 0.5
    assert np.abs(y - Y).max() < 2e-7 and np.abs(u - U).max() < 2e-7 and np.abs(v - V).max() < 2e-7
    # inverse: channel 2 = Y + 2.032 (U - .5); channel 1 = Y - 0.395 (U - .5) - 0.581 (V - .5); channel 0 = Y + 1.140 (V - .5)
    yy, uu, vv = (rng.random((30, 20)).astype(np.float32) for _ in range(3))
    M = np.array([[1.0, 0.0, 1.140], [1.0, -0.395, -0.581], [1.0, 2.032, 0.0]])
    rgb = np.stack([yy, uu - 0.5, vv - 0.5], -1).astype(np.float64) @ M.T
    want = np.clip(np.rint(rgb * 255.0), 0, 255)
    got = orc.yuv_to_u8(yy, uu, vv).astype(np.float64)
    # fp32 evaluation may land on the other side of a .5 boundary on a handful of pixels: never more than 1 LSB, < 0.1 %
    diff = np.abs(got - want)
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3
    # the forward and inverse matrices are (to 3 decimals) inverses: u8 -> YUV -> u8 is the identity on every byte
    assert np.array_equal(orc.yuv_to_u8(y, u, v), img)


def test_round_saturate(oracle_built):
    """convertTo(CV_8U, 255): cvRound = round half to EVEN, saturate_cast clips (the pipeline's only clip, Q2)"""
    # grey pixels (U = V = 0.5): channel value = Y exactly, so byte = cvRound(Y * 255.f)
    ks = np.arange(0, 255)
    cand = ((ks + 0.5) / 255.0).astype(np.float32)
    prod = cand * np.float32(255.0)
    exact = prod == (ks + 0.5).astype(np.float32)          # products that land EXACTLY on k + 0.5 in fp32
    assert exact.sum() >= 20
    yy = cand[exact].reshape(1, -1)
    half = np.full_like(yy, 0.5)
    got = orc.yuv_to_u8(yy, half, half)[0, :, 0]
    want = np.rint(ks[exact] + 0.5).astype(np.uint8)       # numpy rint: half to even
    assert np.array_equal(got, want)
    assert ((want % 2) == 0).all()
    # saturation
    yy = np.array([[-0.3, -1e-9, 0.0, 1.0, 1.0 + 1e-3, 7.5]], np.float32)
    half = np.full_like(yy, 0.5)
    assert orc.yuv_to_u8(yy, half, half)[0, :, 1].tolist() == [0, 0, 0, 255, 255, 255]


@pytest.mark.parametrize("h,w", [(9, 11), (1, 7), (6, 1), (2, 2), (33, 17)])
def test_cubic(oracle_built, h, w):
    """resize(2x, INTER_CUBIC) against torch's independent bicubic (A = -0.75, half-pixel centres, clamped taps)"""
    x = np.random.default_rng(h * 31 + w).random((h, w), dtype=np.float32)
    t = F.interpolate(torch.from_numpy(x)[None, None], scale_factor=2, mode="bicubic", align_corners=False)[0, 0].numpy()
    assert np.abs(orc.resize2x_cubic(x) - t).max() < 5e-6
    assert np.array_equal(orc.resize2x_nearest(x), np.repeat(np.repeat(x, 2, 0), 2, 1))


def _keys_matrix(n, a=-0.75):
    """(2n x n) fp64 matrix of a 2x cubic-convolution upscale along one axis: destination d samples the source at (d + 0.5) / 2 - 0.5 with Keys'
    kernel (a = -0.75, OpenCV's INTER_CUBIC) on the four taps floor(s) - 1 .. floor(s) + 2, tap indices CLAMPED to the plane (BORDER_REPLICATE)"""
    def k(t):
        t = abs(t)
        return (a + 2) * t ** 3 - (a + 3) * t ** 2 + 1 if t <= 1 else a * t ** 3 - 5 * a * t ** 2 + 8 * a * t - 4 * a if t < 2 else 0.0
    m = np.zeros((2 * n, n))
    for d in range(2 * n):
        sx = (d + 0.5) / 2 - 0.5
        f = int(np.floor(sx))
        for tap in range(f - 1, f + 3):
            m[d, min(max(tap, 0), n - 1)] += k(sx - tap)
    return m


@pytest.mark.parametrize("h,w", [(1, 1), (1, 2), (2, 1), (2, 2), (3, 5), (1, 7), (12, 9)])
def test_cubic_border_rule_second_implementation(oracle_built, h, w):
    """the border handling of resize(2x, INTER_CUBIC) (main.cpp:144) pinned a second time, on one- and two-pixel planes where EVERY tap is a clamped
    one: a dense fp64 matrix form of Keys' cubic convolution (separable: M_h x M_w^T), written from the kernel's definition -- independent of the C
    loops and of torch's implementation.  (OpenCV itself is absent: what stays unpinned is its fp32 evaluation ORDER, worth ~1e-7, see the C header.)"""
    x = np.random.default_rng(100 * h + w).random((h, w), dtype=np.float32)
    want = _keys_matrix(h) @ x.astype(np.float64) @ _keys_matrix(w).T
    got = orc.resize2x_cubic(x)
    assert got.shape == (2 * h, 2 * w) and np.abs(got - want).max() < 2e-6
    if h == 1 and w == 1:
        assert np.allclose(got, x[0, 0], atol=1e-6)   # one pixel: every tap clamps onto it, the weights sum to 1


@pytest.mark.parametrize("sh,sw,dh,dw", [(40, 60, 30, 45), (64, 64, 40, 40), (50, 30, 30, 18), (17, 23, 34, 46), (8, 8, 5, 3), (1, 9, 1, 5)])
def test_linear(oracle_built, sh, sw, dh, dw):
    """resize(INTER_LINEAR) -- the CLI's final shrink (main.cpp:158-167: ratios 0.75, 0.625, 0.6, ...) and an enlargement --
    against torch's independent bilinear with half-pixel centres (align_corners=False, no antialias): same source-coordinate
    formula fx = (dx + 0.5) * sw/dw - 0.5, negative coordinates clamp to 0, the last column/row replicates"""
    x = np.random.default_rng(sh + 7 * sw + dh).random((sh, sw), dtype=np.float32)
    t = F.interpolate(torch.from_numpy(x)[None, None], size=(dh, dw), mode="bilinear", align_corners=False, antialias=False)[0, 0].numpy()
    # (torch evaluates the source coordinate in float32, the restatement in double like OpenCV: weights differ by ~1e-6;
    #  a wrong coordinate formula or border rule shows up at 1e-2)
    assert np.abs(orc.resize_linear(x, dw, dh) - t).max() < 1e-5
    const = np.full((sh, sw), 0.37, np.float32)
    assert np.abs(orc.resize_linear(const, dw, dh) - 0.37).max() < 1e-6
