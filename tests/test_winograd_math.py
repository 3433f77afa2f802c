"""The algebra conv3x3_wino (waifu2x-converter-cpp_amd/csrc/w2xc_wino.hip) rests on, checked on the CPU in float64:
Winograd F(2x2, 3x3) with the matrices quoted in that file reproduces the reference's CORRELATION (cv::filter2D does not flip the
kernel, /root/reference/src/modelHandler.cpp:141) on a 4x4 patch, and the position order / plane order of the packed weight image
w2xc_wino_pack documents is the one the kernel's lanes read."""
import numpy as np

G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=np.float64)
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def test_f2x2_3x3_is_the_reference_correlation():
    rng = np.random.default_rng(3)
    for _ in range(50):
        g, d = rng.standard_normal((3, 3)), rng.standard_normal((4, 4))
        y = AT @ ((G @ g @ G.T) * (BT @ d @ BT.T)) @ AT.T
        want = np.array([[(g * d[i:i + 3, j:j + 3]).sum() for j in range(2)] for i in range(2)])   # out(y,x) = sum K[r][c] * in(y+r, x+c)
        assert np.allclose(y, want, rtol=1e-12, atol=1e-12)


def test_transform_coefficients_are_exact_in_fp32():
    """every coefficient is 0, +-1 or +-1/2 (and 1/4 in G g G^T): the input and output transforms are additions, nothing to round"""
    for m in (G, BT, AT):
        assert set(np.abs(m).ravel().tolist()) <= {0.0, 0.5, 1.0}
    assert np.array_equal(BT.astype(np.float32).astype(np.float64), BT)


def test_column_then_row_order_of_the_kernel():
    """the kernel transforms the patch columns first (t = B^T d: t0 = d0 - d2, t1 = d1 + d2, t2 = d2 - d1, t3 = d1 - d3 on ROWS of d)
    and then applies the same four combinations along each row of t; xi = 4 i + j indexes V[i][j]"""
    rng = np.random.default_rng(4)
    d = rng.standard_normal((4, 4))
    t = np.stack([d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]])
    v = np.stack([t[:, 0] - t[:, 2], t[:, 1] + t[:, 2], t[:, 2] - t[:, 1], t[:, 1] - t[:, 3]], axis=1)
    assert np.allclose(v, BT @ d @ BT.T)
    # output transform as the epilogue does it: tm = A^T M (rows), then the same along the columns
    m = rng.standard_normal((4, 4))
    tm = np.stack([m[0] + m[1] + m[2], m[1] - m[2] - m[3]])
    y = np.stack([tm[:, 0] + tm[:, 1] + tm[:, 2], tm[:, 1] - tm[:, 2] - tm[:, 3]], axis=1)
    assert np.allclose(y, AT @ m @ AT.T)


def test_fp32_error_of_the_transform_domain_sum():
    """fp32 Winograd over 128 input planes stays at the error level of an fp32 direct sum (the 1e-4 gate has 2 orders of margin)"""
    rng = np.random.default_rng(5)
    cin = 128
    g = (rng.standard_normal((cin, 3, 3)) * np.sqrt(2.0 / (1.01 * 9 * cin))).astype(np.float32)
    d = rng.random((cin, 4, 4)).astype(np.float32)
    want = sum(np.array([[(g[c].astype(np.float64) * d[c, i:i + 3, j:j + 3]).sum() for j in range(2)] for i in range(2)]) for c in range(cin))
    u = np.stack([(G @ g[c].astype(np.float64) @ G.T).astype(np.float32) for c in range(cin)])          # rounded once, like w2xc_wino_pack
    v = np.stack([(BT.astype(np.float32) @ d[c] @ BT.T.astype(np.float32)) for c in range(cin)])       # additions in fp32
    acc = np.zeros((4, 4), np.float32)
    for c in range(cin):
        acc = acc + u[c] * v[c]                                                                         # fp32 multiply-accumulate per position
    y = AT.astype(np.float32) @ acc @ AT.T.astype(np.float32)
    assert np.abs(y - want).max() <= 1e-5 * max(np.abs(want).max(), 1e-3)
