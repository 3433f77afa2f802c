"""CPU checks of the Winograd F(4x4,3x3) algebra behind conv3x3_wino4 (no GPU: the library's HOST weight transform + numpy).

The kernel computes Y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A on the interpolation points 0, +-3/4, +-3/2, inf
(waifu2x-converter-cpp_amd/csrc/w2xc_wino4.hip: bt6 / at6 / w2xc_wino4_pack).  Here: the transformed weights are taken from the
library's own w2xc_wino4_pack (a host function; an internal symbol, not part of the C ABI), un-permuted by the documented fragment layout,
and combined with B^T and A^T restated from bt6 / at6 -- the result has to be the 3x3 correlation of modelHandler.cpp:127-145
(filter2D per input plane, summed) on every 6x6 patch, to fp32 rounding of U.  A mismatch between the three matrices, or a weight image
whose layout is not the one the kernel's fragment reads assume, fails here without a GPU.
"""
import ctypes

import numpy as np
import pytest


# y = B^T x as bt6 computes it (w2xc_wino4.hip: bt6)
BT = np.array([
    [1.265625, 0.0, -2.8125, 0.0, 1.0, 0.0],
    [0.0, -1.6875, -2.25, 0.75, 1.0, 0.0],
    [0.0, 1.6875, -2.25, -0.75, 1.0, 0.0],
    [0.0, -0.84375, -0.5625, 1.5, 1.0, 0.0],
    [0.0, 0.84375, -0.5625, -1.5, 1.0, 0.0],
    [0.0, 1.265625, 0.0, -2.8125, 0.0, 1.0],
])
# y = A^T m as at6 computes it (w2xc_wino4.hip: at6)
AT = np.array([
    [1.0, 1.0, 1.0, 1.0, 1.0, 0.0],
    [0.0, 0.75, -0.75, 1.5, -1.5, 0.0],
    [0.0, 0.5625, 0.5625, 2.25, 2.25, 0.0],
    [0.0, 0.421875, -0.421875, 3.375, -3.375, 1.0],
])


def _pack(w2xc, cin, cout, w):
    lib = ctypes.CDLL(w2xc.LIB_PATH)
    fn = getattr(lib, "_Z15w2xc_wino4_packiiPKfPf")
    fn.restype = None
    fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    dst = np.zeros(36 * cin * cout, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    fn(cin, cout, w.ctypes.data, dst.ctypes.data)
    return dst


def _xi_of(i, j):
    """Position (i, j) of the transformed domain in the fragment order (w2xc_wino4.hip: xi_of): the column halves j < 3 / j >= 3 as two xi ranges."""
    return 3 * i + j if j < 3 else 18 + 3 * i + (j - 3)


def _unpack(cin, cout, img):
    """U[6 i + j][plane][channel] from [ob][s][xi / 4][pt][lane = 16 k + o][xi % 4], xi = xi_of(i, j) (w2xc_wino4_pack's comment)."""
    nst, nob = cin // 4, cout // 64
    a = img.reshape(nob, nst, 9, 4, 4, 16, 4)          # ob, s, xi >> 2, pt, k, o, xi & 3
    a = a.transpose(2, 6, 0, 3, 5, 1, 4)                # xi >> 2, xi & 3, ob, pt, o, s, k
    a = a.reshape(36, cout, cin)
    return a[[_xi_of(i, j) for i in range(6) for j in range(6)]]


@pytest.mark.parametrize("cin,cout", [(32, 64), (64, 128), (128, 128)])
def test_wino4_weight_image_and_matrices_reproduce_the_3x3_correlation(w2xc, cin, cout):
    rng = np.random.default_rng(cin * 1000 + cout)
    w = (rng.standard_normal((cout, cin, 3, 3)) * 0.1).astype(np.float32)
    U = _unpack(cin, cout, _pack(w2xc, cin, cout, w)).astype(np.float64)
    assert np.isfinite(U).all() and np.abs(U).max() > 0
    d = rng.standard_normal((cin, 6, 6)).astype(np.float32).astype(np.float64)
    V = np.einsum("ia,cab,jb->cij", BT, d, BT).reshape(cin, 36)           # B^T d B per channel
    M = np.einsum("xpc,cx->px", U, V).reshape(cout, 6, 6)                 # the 36 xi GEMMs
    Y = np.einsum("ia,pab,jb->pij", AT, M, AT)                            # A^T M A: 4x4 outputs per plane
    ref = np.zeros((cout, 4, 4))
    for ky in range(3):
        for kx in range(3):
            ref += np.einsum("pc,cyx->pyx", w[:, :, ky, kx].astype(np.float64), d[:, ky:ky + 4, kx:kx + 4])
    scale = np.abs(ref).max()
    # U is rounded to fp32 once (2^-24 relative per term, cin * 36 terms, transform gains of a few units): 1e-5 of the range is ~30x that
    assert np.abs(Y - ref).max() <= 1e-5 * scale, (np.abs(Y - ref).max(), scale)


def test_wino4_matrices_are_the_cook_toom_matrices_of_their_points(w2xc):
    """B^T, A^T and the G implied by the weight image belong to the points 0, +-3/4, +-3/2, inf: F(4,3) in one dimension, exactly (float64)."""
    rng = np.random.default_rng(7)
    g = rng.standard_normal(3)
    x = rng.standard_normal(6)
    pts = [0.0, 0.75, -0.75, 1.5, -1.5]
    # G g = g evaluated at the points, scaled by 1 / prod_{k != i} (p_i - p_k); last row = leading coefficient
    G = np.zeros((6, 3))
    for i, p in enumerate(pts):
        den = np.prod([p - q for k, q in enumerate(pts) if k != i])
        G[i] = np.array([1.0, p, p * p]) / den
    G[5] = [0.0, 0.0, 1.0]
    y = AT @ ((G @ g) * (BT @ x))
    ref = np.array([np.dot(g, x[k:k + 3]) for k in range(4)])
    assert np.abs(y - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    # and the library's image uses that G (up to the rounding of U): one plane block, channel 0 carries g (x) g
    w = np.zeros((64, 32, 3, 3), np.float32)
    g32 = g.astype(np.float32)
    w[0, 0] = np.outer(g32, g32)
    U = _unpack(32, 64, _pack(w2xc, 32, 64, w))[:, 0, 0].reshape(6, 6).astype(np.float64)
    expect = G @ w[0, 0].astype(np.float64) @ G.T
    assert np.abs(U - expect).max() <= 2.0 ** -23 * np.abs(expect).max()


# ---- conv3x3_first2_wino4 (w2xc_first2_wino4.hip): layer 2's register-stationary weight image, and layers 1 + 2 together ----
def _pack_first2(w2xc, w):
    lib = ctypes.CDLL(w2xc.LIB_PATH)
    fn = getattr(lib, "_Z22w2xc_first2_wino4_packPKfPf")
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    dst = np.zeros(36 * 32 * 32, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    fn(w.ctypes.data, dst.ctypes.data)
    return dst


def _unpack_first2(img):
    """U[xi = 6 i + j][plane][channel] from [wave g][xi - 9 g][pt][ks][lane = 16 k + o] = U_xi[16 pt + o][4 ks + k] (w2xc_first2_wino4_pack's comment)."""
    a = img.reshape(4, 9, 2, 8, 4, 16)                  # g, xl, pt, ks, k, o
    a = a.transpose(0, 1, 2, 5, 3, 4)                   # g, xl, pt, o, ks, k
    return a.reshape(36, 32, 32)


def test_first2_wino4_weight_image_and_both_layers(w2xc):
    """the fused first two layers as the kernel evaluates them -- layer 1 as a bias-first sum over its nine taps + LeakyReLU on every 6x6 patch value, layer 2 as
    Y = A^T [ sum_c U (.) (B^T d B) ] A from the library's own weight image -- against the two 3x3 correlations of modelHandler.cpp:127-152, in float64."""
    rng = np.random.default_rng(2025)
    w1 = (rng.standard_normal((32, 1, 3, 3)) * 0.5).astype(np.float32)
    b1 = rng.uniform(-0.05, 0.05, 32)
    w2 = (rng.standard_normal((32, 32, 3, 3)) * 0.1).astype(np.float32)
    U = _unpack_first2(_pack_first2(w2xc, w2)).astype(np.float64)
    assert np.isfinite(U).all() and np.abs(U).max() > 0
    src = rng.random((8, 8))                                                # the 8 x 8 source window of one 4x4 output block
    leaky = lambda v: np.maximum(v, 0.1 * v)
    d = np.zeros((32, 6, 6))
    for ky in range(3):
        for kx in range(3):
            d += w1[:, 0, ky, kx].astype(np.float64)[:, None, None] * src[None, ky:ky + 6, kx:kx + 6]
    d = leaky(d + b1[:, None, None])                                        # layer 1 on the patch
    V = np.einsum("ia,cab,jb->cij", BT, d, BT).reshape(32, 36)
    M = np.einsum("xpc,cx->px", U, V).reshape(32, 6, 6)
    Y = np.einsum("ia,pab,jb->pij", AT, M, AT)
    ref = np.zeros((32, 4, 4))
    for ky in range(3):
        for kx in range(3):
            ref += np.einsum("pc,cyx->pyx", w2[:, :, ky, kx].astype(np.float64), d[:, ky:ky + 4, kx:kx + 4])
    assert np.abs(Y - ref).max() <= 1e-5 * np.abs(ref).max(), (np.abs(Y - ref).max(), np.abs(ref).max())
    # the image is a permutation of G g G^T: one plane, one channel
    w = np.zeros((32, 32, 3, 3), np.float32)
    g = rng.standard_normal(3).astype(np.float32)
    w[17, 5] = np.outer(g, g)
    Ug = _unpack_first2(_pack_first2(w2xc, w))
    assert np.count_nonzero(Ug) == np.count_nonzero(Ug[:, 17, 5]) > 0          # nothing lands on another (plane, channel)


# ---- the fp32 error of F(4x4,3x3) as the kernel orders it, emulated in numpy float32 ----
def _bt6_f32(x0, x1, x2, x3, x4, x5):
    """bt6 of w2xc_wino4.hip on float32 arrays (numpy rounds the product and the sum separately where the kernel's fma rounds once: an upper bound)."""
    f = np.float32
    y0 = f(-2.8125) * x2 + (f(1.265625) * x0 + x4)
    p, q = f(-2.25) * x2 + x4, f(-1.6875) * x1 + f(0.75) * x3
    u, v = f(-0.5625) * x2 + x4, f(-0.84375) * x1 + f(1.5) * x3
    y5 = f(-2.8125) * x3 + (f(1.265625) * x1 + x5)
    return y0, p + q, p - q, u + v, u - v, y5


def _at6_f32(m0, m1, m2, m3, m4, m5):
    f = np.float32
    s1, d1, s2, d2 = m1 + m2, m1 - m2, m3 + m4, m3 - m4
    return m0 + s1 + s2, f(1.5) * d2 + f(0.75) * d1, f(2.25) * s2 + f(0.5625) * s1, f(3.375) * d2 + (f(0.421875) * d1 + m5)


@pytest.mark.parametrize("data", ["image", "normal"])
def test_wino4_fp32_error_meets_the_elementwise_gate(w2xc, data):
    """128 -> 128 planes, weights drawn like the upstream initialisation, on 36 output tiles: F(4x4,3x3) in float32 -- the library's weight image, the
    column-then-row input transform, an fp32 accumulation over the channels in stage order, the row-pair output transform, bias, LeakyReLU -- against the
    float64 correlation of modelHandler.cpp:127-154, at the gate the GPU parity tests use (|got - want| <= 1e-5 + 1e-4 |want|).  `image`: inputs in [0, 1)
    like a luma plane / an activation; `normal`: zero-mean unit-variance inputs, the cancellation-heavy case the interpolation points were chosen on
    (tools/winograd_points.py: the Lavin-Gray points 0, +-1, +-2 fail it by 3x)."""
    cin = cout = 128
    rng = np.random.default_rng(11 if data == "image" else 12)
    w = (rng.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (1.01 * 9 * cin))).astype(np.float32)
    bias = (rng.standard_normal(cout) * 0.01).astype(np.float32)
    th = tw = 6                                                        # 6 x 6 tiles of 4 x 4 outputs: a 26 x 26 input
    x = (rng.random((cin, 4 * th + 2, 4 * tw + 2)) if data == "image" else rng.standard_normal((cin, 4 * th + 2, 4 * tw + 2))).astype(np.float32)
    U = _unpack(cin, cout, _pack(w2xc, cin, cout, w))                  # [36][plane][channel], float32
    d = np.stack([x[:, 4 * ty:4 * ty + 6, 4 * tx:4 * tx + 6] for ty in range(th) for tx in range(tw)])            # [tile][channel][6][6]
    t = np.stack(_bt6_f32(*[d[:, :, i, :] for i in range(6)]), axis=2)                                            # columns first: over the row index
    V = np.stack(_bt6_f32(*[t[:, :, :, j] for j in range(6)]), axis=3).reshape(len(d), cin, 36)                   # then along each row
    acc = np.zeros((len(d), cout, 36), np.float32)
    for c in range(cin):                                               # channel order = stage order (4 per MFMA, k ascending)
        acc = acc + U[:, :, c].T[None, :, :] * V[:, c, None, :]
    M = acc.reshape(len(d), cout, 6, 6)
    tm = np.stack(_at6_f32(*[M[:, :, i, :] for i in range(6)]), axis=2)                                           # A^T M: 4 x 6
    Y = np.stack(_at6_f32(*[tm[:, :, :, j] for j in range(6)]), axis=3)                                           # (A^T M) A: 4 x 4
    Y = Y + bias[None, :, None, None]
    got = np.maximum(Y, np.float32(0.1) * Y)
    want = np.zeros((cout, 4 * th, 4 * tw))
    for ky in range(3):
        for kx in range(3):
            want += np.einsum("pc,cyx->pyx", w[:, :, ky, kx].astype(np.float64), x[:, ky:ky + 4 * th, kx:kx + 4 * tw].astype(np.float64))
    want += bias[:, None, None]
    want = np.maximum(want, 0.1 * want)
    got_img = got.reshape(th, tw, cout, 4, 4).transpose(2, 0, 3, 1, 4).reshape(cout, 4 * th, 4 * tw)
    err = np.abs(got_img - want)
    gate = 1e-5 + 1e-4 * np.abs(want)
    worst = (err / gate).max()
    assert worst <= 1.0, "worst |error| / gate = %.2f" % worst
