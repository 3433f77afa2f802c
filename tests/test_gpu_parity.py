"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against the CPU oracle on the same seeded inputs.

Bars:  conv3x3_direct (reference summation order, unfused mul/add) -- BIT-EXACT;
       MFMA kernels (fp32, exact fma chains in a different order)  -- rtol 1e-4 (+atol 1e-5) per
       BASELINE.json's north_star, and max-norm relative error <= 1e-4.
"""
import os

import numpy as np
import pytest

from conftest import ATOL, RTOL, assert_close, ramp_plane, rand_plane, small_layers
from tools import gen_model
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gpu(w2xc):
    assert w2xc.device_count() >= 1, "no HIP device visible: libw2xc_hip has no CPU fallback, -m gpu tests need an MI355X"
    return w2xc


def direct(w2xc):
    return w2xc.make_opts(kernel=w2xc.KERNEL_DIRECT)


# ---- Model::filter boundary, one layer at a time (each kernel kind in isolation) -----------------
LAYER_SHAPES = [(1, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (128, 1),
                (3, 128), (128, 3), (64, 32), (128, 64), (32, 128), (3, 32), (64, 1), (32, 3),
                (5, 7), (2, 33)]


@pytest.mark.parametrize("cin,cout", LAYER_SHAPES)
def test_layer_filter_direct_bit_exact(gpu, cin, cout):
    layers = small_layers([cin, cout], 100 + cin * 7 + cout)
    ms = gpu._ModelSet.from_layers(layers)
    x = np.random.default_rng(cin + cout).standard_normal((cin, 21, 37)).astype(np.float32)
    want = orc.Oracle(layers).filter(0, x)
    got = ms.filter(0, x, direct(gpu))
    assert ms.kernel_name(0, direct(gpu)) == "conv3x3_direct"
    assert np.array_equal(got, want), "max abs diff %g" % np.abs(got - want).max()


@pytest.mark.parametrize("cin,cout", [s for s in LAYER_SHAPES if s not in [(5, 7), (2, 33)]])
@pytest.mark.parametrize("h,w", [(21, 37), (8, 32), (9, 33), (1, 1), (40, 70)])
def test_layer_filter_fast_kernels(gpu, cin, cout, h, w):
    layers = small_layers([cin, cout], 200 + cin * 7 + cout)
    ms = gpu._ModelSet.from_layers(layers)
    assert ms.kernel_name(0) != "conv3x3_direct"
    x = np.random.default_rng(cin * 3 + cout + h).standard_normal((cin, h, w)).astype(np.float32)
    want = orc.Oracle(layers).filter(0, x)
    got = ms.filter(0, x)
    assert_close(got, want, "%s %d->%d %dx%d" % (ms.kernel_name(0), cin, cout, h, w))


@pytest.mark.parametrize("cin,cout", [(32, 32), (128, 128), (1, 32), (128, 1)])
def test_fast_kernels_orientation(gpu, cin, cout):
    """transpose / tap-order detector: a single asymmetric tap and asymmetric plane<->plane map,
    on a non-symmetric ramp (a symmetric kernel or image would hide a swapped row/col)."""
    w = np.zeros((cout, cin, 3, 3), np.float32)
    rng = np.random.default_rng(5)
    for o in range(cout):
        w[o, (o * 5 + 1) % cin, rng.integers(0, 3), rng.integers(0, 3)] = 1.0 + o / 64.0
    b = np.linspace(-0.5, 0.5, cout)
    layers = [(cin, cout, w, b.astype(np.float64))]
    ms = gpu._ModelSet.from_layers(layers)
    x = np.stack([ramp_plane(19, 45) * (1 + 0.1 * i) - 0.3 * i for i in range(cin)])
    want = orc.Oracle(layers).filter(0, x)
    got = ms.filter(0, x)
    # one product per output: the MFMA result is exact up to the bias add
    assert np.abs(got - want).max() <= 1e-6 * max(1.0, np.abs(want).max())


def test_filter_plane_count_mismatch(gpu, noise1_layers):
    ms = gpu._ModelSet.from_layers(noise1_layers)
    models = [gpu.Model(ms, i) for i in range(7)]
    outs = ["stale"]
    assert models[1].filter([gpu.Mat(rand_plane(8, 8, i)) for i in range(3)], outs) is False
    assert outs == ["stale"]
    assert models[0].filter([gpu.Mat(rand_plane(8, 8, 0))], outs) is True
    assert len(outs) == 32 and outs[0].array.shape == (8, 8)


# ---- convertWithModels ----------------------------------------------------------------------------
def test_cfg1_noise1_256(gpu, noise1_layers):
    """BASELINE.json configs[0]: noise1 topology on a 256x256 luma plane, vs the CPU convertRoutine."""
    ms = gpu._ModelSet.from_layers(noise1_layers)
    x = rand_plane(256, 256, 1)
    o = orc.Oracle(noise1_layers)
    want = o.convert(x, njob=8)
    got = ms.convert(x)
    assert_close(got, want, "cfg1")
    # error budget against the fp64 truth: GPU error should be of the same class as the oracle's own
    truth = o.convert_f64(x)
    e_gpu, e_cpu = np.abs(got - truth).max(), np.abs(want - truth).max()
    assert e_gpu <= max(4 * e_cpu, 2e-6), (e_gpu, e_cpu)


def test_direct_path_is_bit_exact_end_to_end(gpu, noise1_layers):
    ms = gpu._ModelSet.from_layers(noise1_layers)
    x = rand_plane(70, 45, 2)
    want = orc.Oracle(noise1_layers).convert(x)
    got = ms.convert(x, opts=direct(gpu))
    assert np.array_equal(got, want)


@pytest.mark.parametrize("h,w", [(1, 1), (7, 9), (33, 65), (100, 31), (64, 200)])
def test_odd_sizes(gpu, scale_layers, h, w):
    ms = gpu._ModelSet.from_layers(scale_layers)
    x = rand_plane(h, w, h * 1000 + w)
    assert_close(ms.convert(x), orc.Oracle(scale_layers).convert(x), "%dx%d" % (h, w))


@pytest.mark.parametrize("band", [1, 5, 32, 64])
def test_banding_does_not_change_results(gpu, scale_layers, band):
    """any tiling with an n-px halo gives the same math per output pixel -- bit-exact
    between band sizes on the GPU (same kernels, same per-pixel summation order)."""
    ms = gpu._ModelSet.from_layers(scale_layers)
    x = rand_plane(90, 75, 7)
    whole = ms.convert(x, opts=gpu.make_opts(band_rows=90))
    banded = ms.convert(x, opts=gpu.make_opts(band_rows=band))
    assert np.array_equal(whole, banded)
    tiny_ws = ms.convert(x, opts=gpu.make_opts(workspace_mb=1))   # workspace-derived banding
    assert np.array_equal(whole, tiny_ws)


def test_strided_roi_in_and_out(gpu, noise1_layers):
    """cv::Mat ROIs: honour `step` on both sides"""
    ms = gpu._ModelSet.from_layers(noise1_layers)
    big = rand_plane(80, 120, 8)
    roi = big[10:60, 20:90]
    assert not roi.flags["C_CONTIGUOUS"]
    want = orc.Oracle(noise1_layers).convert(np.ascontiguousarray(roi))
    assert_close(ms.convert(roi), want, "strided in")
    lib = gpu.lib()
    outbig = np.full((60, 100), -7.0, np.float32)
    outroi = outbig[5:55, 10:80]
    rc = lib.w2xc_convert_plane(ms.handle, roi.ctypes.data, roi.strides[0], 70, 50, outroi.ctypes.data,
                                outroi.strides[0], 1, None)
    assert rc == 0, gpu.last_error()
    assert_close(outroi, want, "strided out")
    mask = np.ones_like(outbig, bool)
    mask[5:55, 10:80] = False
    assert np.all(outbig[mask] == -7.0), "wrote outside the output ROI"


def test_reference_api_mirror(gpu, models_dir):
    """reads like the reference's main.cpp:83-98"""
    models = []
    assert gpu.modelUtility.generateModelFromJSON(os.path.join(models_dir, "noise1_model.json"), models)
    gpu.modelUtility.getInstance().setNumberOfJobs(4)
    y = rand_plane(48, 64, 9)
    out = gpu.Mat()
    assert gpu.convertWithModels(gpu.Mat(y), out, models)
    want = orc.Oracle.from_json(os.path.join(models_dir, "noise1_model.json")).convert(y)
    assert_close(out.array, want, "api mirror")
    assert gpu.convertWithModels(gpu.Mat(y), out, models, False)      # blockSplitting=false: same result (I2)
    assert_close(out.array, want, "api mirror unsplit")
    # a sub-list of layers is a valid model vector too (test.cpp drives Model::filter directly)
    out2 = gpu.Mat()
    assert gpu.convertWithModels(gpu.Mat(y), out2, models[:1]) is True   # 1->32: returns outputPlanes[0]
    sub = orc.Oracle(orc.load_model_json(os.path.join(models_dir, "noise1_model.json"))[:1])
    assert_close(out2.array, sub.convert(y), "single-layer vector")


def test_golden_fixtures_on_gpu(gpu):
    """fixtures produced by the reference's own code (oracle/_ref), see tests/golden/make_golden.py"""
    for f in sorted(os.listdir(GOLDEN)):
        if not f.endswith(".npz"):
            continue
        g = np.load(os.path.join(GOLDEN, f))
        layers = gen_model.synth_layers([int(v) for v in g["planes"]], int(g["seed"]), init=str(g["init"]) if "init" in g else "he_leaky")
        ms = gpu._ModelSet.from_layers(layers)
        assert_close(ms.convert(g["input"]), g["output"], f)
        assert np.array_equal(ms.convert(g["input"], opts=direct(gpu)), g["output"]), f


def test_device_pointer_entry_point(gpu, scale_layers):
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    ms = gpu._ModelSet.from_layers(scale_layers)
    x = rand_plane(120, 200, 11)
    want = orc.Oracle(scale_layers).convert(x)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.zeros_like(d_in)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    o = gpu.make_opts(device=0, profile=1)
    ms.profile_reset(0)
    ms.convert_device(d_in.data_ptr(), 200 * 4, 200, 120, d_out.data_ptr(), 200 * 4, stream=side.cuda_stream, opts=o)
    side.synchronize()
    assert_close(d_out.cpu().numpy(), want, "device entry")
    ms_t, launches = ms.profile_read(0)
    # (layers 1 + 2 are one launch under the default kernels: layer 1 has none of its own)
    assert launches == [0] + [1] * 6 and ms_t[0] == 0 and all(t > 0 for t in ms_t[1:]) and ms.kernel_name(1) == "conv3x3_first2_wino4"
    ms.profile_reset(0)
    ms.convert_device(d_in.data_ptr(), 200 * 4, 200, 120, d_out.data_ptr(), 200 * 4, stream=side.cuda_stream, opts=gpu.make_opts(device=0, profile=1, fusion=gpu.FUSION_OFF))
    side.synchronize()
    assert ms.profile_read(0)[1] == [1] * 7


@pytest.mark.parametrize("precision", [0, 3, 4])   # fp32, BF16X3, FP16X2
@pytest.mark.parametrize("parts", [2, 3])
def test_row_band_entry_point(gpu, scale_layers, parts, precision):
    """w2xc_convert_rows_device: shards of one plane are independent and stitch bit-exactly
    (the multi-GPU decomposition, exercised here on one device) -- in every precision."""
    torch = pytest.importorskip("torch")
    ms = gpu._ModelSet.from_layers(scale_layers)
    h, w = 150, 90
    x = rand_plane(h, w, 13)
    whole = ms.convert(x, opts=gpu.make_opts(precision=precision))
    out = torch.zeros((h, w), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream()
    o = gpu.make_opts(device=0, precision=precision)
    for p in range(parts):
        ra, rb = gpu.shard_rows(h, parts, p)
        y0, y1 = gpu.shard_view(h, ra, rb, 4 * ms.n_layers)   # the wide halo: bit-identity with the whole-plane call under the default F(4x4) kernel
        view = torch.from_numpy(np.ascontiguousarray(x[y0:y1])).cuda()
        ms.convert_rows_device(view.data_ptr(), w * 4, y1 - y0, y0, w, h, ra, rb, out[ra:].data_ptr(), w * 4,
                               stream=st.cuda_stream, opts=o)
    st.synchronize()
    assert np.array_equal(out.cpu().numpy(), whole)
    with pytest.raises(gpu.W2xcError) as e:     # a view that lacks its halo rows is rejected
        v = torch.zeros((10, w), device="cuda")
        ms.convert_rows_device(v.data_ptr(), w * 4, 10, 50, w, h, 50, 60, out.data_ptr(), w * 4, opts=o)
    assert e.value.code == gpu.ERR_ARG


def test_row_bands_on_minimum_halo_views(gpu, scale_layers):
    """The default F(4x4) mid-layer kernel needs 4 n halo rows in a band's view for its banding-invariant geometry.  On the MINIMUM
    view ([ra - n, rb + n)) W2XC_KERNEL_AUTO is REFUSED (W2XC_ERR_ARG: no silent change of kernel and rounding with the view's halo); a caller who
    chooses the F(2x2) kernels explicitly (W2XC_KERNEL_WINOGRAD32) gets bands that stitch bit-identically with the whole-plane run of THOSE
    kernels and sit within the usual tolerance of the oracle and of the default whole-plane run."""
    torch = pytest.importorskip("torch")
    ms = gpu._ModelSet.from_layers(scale_layers)
    h, w, parts = 150, 90, 3
    x = rand_plane(h, w, 13)
    out = torch.zeros((h, w), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream()
    o = gpu.make_opts(device=0, kernel=gpu.KERNEL_WINOGRAD32)
    for p in range(parts):
        ra, rb = gpu.shard_rows(h, parts, p)
        y0, y1 = gpu.shard_view(h, ra, rb, ms.n_layers)
        view = torch.from_numpy(np.ascontiguousarray(x[y0:y1])).cuda()
        with pytest.raises(gpu.W2xcError) as e:
            ms.convert_rows_device(view.data_ptr(), w * 4, y1 - y0, y0, w, h, ra, rb, out[ra:].data_ptr(), w * 4, stream=st.cuda_stream, opts=gpu.make_opts(device=0))
        assert e.value.code == gpu.ERR_ARG and "halo rows" in str(e.value)
        ms.convert_rows_device(view.data_ptr(), w * 4, y1 - y0, y0, w, h, ra, rb, out[ra:].data_ptr(), w * 4, stream=st.cuda_stream, opts=o)
    st.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(got, ms.convert(x, opts=gpu.make_opts(kernel=gpu.KERNEL_WINOGRAD32)))
    assert_close(got, orc.Oracle(scale_layers).convert(x, njob=8), "minimum-halo row bands")
    whole = ms.convert(x)
    assert np.abs(got - whole).max() <= 1e-5 * np.abs(whole).max()


@pytest.mark.parametrize("h,w", [(37, 53), (1, 1), (128, 160)])
def test_nn2x_fused_into_layer1(gpu, scale_layers, h, w):
    """N1: cv::resize(INTER_NEAREST, 2x) + convertWithModels (main.cpp:132-148) as one call equals
    the conversion of the explicitly upscaled plane -- bit-exact on the GPU (same kernels downstream),
    bit-exact against the oracle with the direct kernel, rtol 1e-4 with the MFMA kernels."""
    ms = gpu._ModelSet.from_layers(scale_layers)
    x = rand_plane(h, w, 31 + h)
    up = np.repeat(np.repeat(x, 2, axis=0), 2, axis=1)
    want = orc.Oracle(scale_layers).convert(up)
    fused = ms.convert_nn2x(x)
    assert fused.shape == (2 * h, 2 * w)
    assert np.array_equal(fused, ms.convert(up))
    assert_close(fused, want, "nn2x %dx%d" % (h, w))
    assert np.array_equal(ms.convert_nn2x(x, direct(gpu)), want)
    assert np.array_equal(ms.convert_nn2x(x, gpu.make_opts(band_rows=16)), fused)   # banded: odd/even row origins


def test_wide_model_cfg5_boundary(gpu):
    """BASELINE.json configs[4] shape (3->128->...->3) goes through Model::filter (convertWithModels
    can only push one plane, convertRoutine.cpp:63-64)."""
    layers = gen_model.synth_layers(gen_model.TOPOLOGY_WIDE, gen_model.SEEDS["wide"])
    ms = gpu._ModelSet.from_layers(layers)
    o = orc.Oracle(layers)
    x = np.random.default_rng(12).random((3, 24, 40), dtype=np.float32)
    a, b = x, x
    for l in range(7):
        a = ms.filter(l, a)
        b = o.filter(l, b, njob=8)
    assert_close(a, b, "wide chain")
    with pytest.raises(gpu.W2xcError) as e:
        ms.convert(x[0])
    assert e.value.code == gpu.ERR_PLANES


# ---- full-size properties (BASELINE.json configs[1]: 1920x1080 -> CNN plane 2160x3840) --------------
def test_full_size_patches_and_band_invariance(gpu, scale_layers):
    ms = gpu._ModelSet.from_layers(scale_layers)
    rng = np.random.default_rng(2)
    small = rng.random((1080, 1920), dtype=np.float32)
    plane = np.repeat(np.repeat(small, 2, axis=0), 2, axis=1)       # nearest 2x (main.cpp:132-140)
    got = ms.convert(plane)
    assert got.shape == (2160, 3840) and np.isfinite(got).all()
    o = orc.Oracle(scale_layers)
    H, W = plane.shape
    # oracle on patches: corners (replicate border), edges, a band seam, interior
    spots = [(0, 0), (0, W - 48), (H - 48, 0), (H - 48, W - 48), (1000, 2000), (0, 1900), (1056, 0), (517, 3001)]
    for (y, x) in spots:
        ph, pw = 48, 48
        y0, y1, x0, x1 = max(0, y - 7), min(H, y + ph + 7), max(0, x - 7), min(W, x + pw + 7)
        # a crop that touches the plane border keeps replicate semantics there; interior crop edges are
        # discarded (7-px rim) -- exactly the reference's block-split argument (convertRoutine.cpp:84-169)
        sub = o.convert(np.ascontiguousarray(plane[y0:y1, x0:x1]), block_splitting=False, njob=8)
        want = sub[y - y0:y - y0 + ph, x - x0:x - x0 + pw]
        assert_close(got[y:y + ph, x:x + pw], want, "patch (%d,%d)" % (y, x))
    banded = ms.convert(plane, opts=gpu.make_opts(band_rows=500))
    assert np.array_equal(got, banded)


@pytest.mark.parametrize("precision", [0, 3, 4])   # fp32, BF16X3 (20.8 GB term buffers), FP16X2
def test_large_plane_beyond_4gib_buffers(gpu, scale_layers, precision):
    """maximum sizes: a 3000 x 9000 plane in ONE band makes the 128-plane activation buffers 13.9 GB each,
    so every kernel addresses well past 2^32 bytes; patches at the far end (highest addresses), at a band
    seam of a second, banded run and at the borders are checked against the oracle."""
    ms = gpu._ModelSet.from_layers(scale_layers)
    H, W = 3000, 9000
    rng = np.random.default_rng(77)
    plane = rng.random((H, W), dtype=np.float32)
    got = ms.convert(plane, opts=gpu.make_opts(workspace_mb=60000, precision=precision))
    assert np.isfinite(got).all()
    o = orc.Oracle(scale_layers)
    for (y, x) in [(H - 40, W - 40), (H - 40, 0), (0, W - 40), (1499, 4500), (2990, 8000), (2000, 8960)]:
        ph = pw = 32
        y, x = min(y, H - ph), min(x, W - pw)
        y0, y1, x0, x1 = max(0, y - 7), min(H, y + ph + 7), max(0, x - 7), min(W, x + pw + 7)
        sub = o.convert(np.ascontiguousarray(plane[y0:y1, x0:x1]), block_splitting=False, njob=8)
        assert_close(got[y:y + ph, x:x + pw], sub[y - y0:y - y0 + ph, x - x0:x - x0 + pw], "patch (%d,%d)" % (y, x))
    banded = ms.convert(plane, opts=gpu.make_opts(workspace_mb=6000, precision=precision))
    assert np.array_equal(got, banded)


# ---- W2XC_PRECISION_BF16 (BASELINE.json configs[3]) ------------------------------------------------
def psnr(a, b, peak=1.0):
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return 10 * np.log10(peak * peak / max(mse, 1e-30))


@pytest.mark.parametrize("planes", [[1, 32, 1], [1, 32, 32, 1], [1, 64, 128, 1], [1, 128, 64, 32, 1], [1, 32, 32, 64, 64, 128, 128, 1]])
@pytest.mark.parametrize("h,w", [(45, 77), (8, 32)])
def test_bf16_path_matches_its_emulation(gpu, planes, h, w):
    """bf16 activations between layers, bf16 weights in the middle layers, fp32 accumulate: checked
    against a float64-accumulate emulation of the same dataflow.  Tolerance: 1e-2 of the output range
    (one bf16 ulp of a mid-layer activation is 2^-8 relative; accumulation-order differences can flip
    individual roundings), plus a tight mean-error bound that a layout bug could not meet."""
    import bf16_ref
    layers = small_layers(planes, 400 + len(planes))
    ms = gpu._ModelSet.from_layers(layers)
    x = rand_plane(h, w, 5 + h)
    got = ms.convert(x, opts=gpu.make_opts(precision=gpu.PRECISION_BF16))
    want = bf16_ref.convert_bf16_emulated(layers, x)
    scale = float(np.abs(want).max())
    assert np.abs(got - want).max() <= 1e-2 * scale, (np.abs(got - want).max(), scale)
    assert np.abs(got - want).mean() <= 1e-3 * scale


def test_bf16_vs_fp32_oracle_accuracy_statement(gpu, scale_layers):
    """configs[3] tolerance check vs the CPU convertRoutine (fp32): stated, not silently chosen --
    max |bf16 - oracle32| <= 2e-2 and PSNR >= 45 dB on [0,1] data for the 7-layer scale2.0x topology."""
    ms = gpu._ModelSet.from_layers(scale_layers)
    x = rand_plane(96, 128, 3)
    want = orc.Oracle(scale_layers).convert(x)
    got = ms.convert(x, opts=gpu.make_opts(precision=gpu.PRECISION_BF16))
    err = float(np.abs(got - want).max())
    p = psnr(got, want)
    print("bf16 vs oracle32: max abs err %.4g, PSNR %.1f dB, max|want| %.3f" % (err, p, np.abs(want).max()))
    assert err <= 2e-2 and p >= 45.0, (err, p)
    # banding / nn2x compose with bf16 exactly as with fp32
    assert np.array_equal(got, ms.convert(x, opts=gpu.make_opts(precision=gpu.PRECISION_BF16, band_rows=17)))
    half = np.ascontiguousarray(x[::2, ::2])
    up = np.repeat(np.repeat(half, 2, 0), 2, 1)
    assert np.array_equal(ms.convert_nn2x(half, gpu.make_opts(precision=gpu.PRECISION_BF16)),
                          ms.convert(up, opts=gpu.make_opts(precision=gpu.PRECISION_BF16)))


def test_bf16_unsupported_shapes_are_rejected(gpu):
    ms = gpu._ModelSet.from_layers(small_layers([1, 5, 1], 2))
    with pytest.raises(gpu.W2xcError) as e:
        ms.convert(rand_plane(8, 8, 0), opts=gpu.make_opts(precision=gpu.PRECISION_BF16))
    assert e.value.code == gpu.ERR_UNSUPPORTED
    ms2 = gpu._ModelSet.from_layers(small_layers([1, 32, 1], 2))
    with pytest.raises(gpu.W2xcError) as e:
        ms2.filter(0, rand_plane(8, 8, 0)[None], gpu.make_opts(precision=gpu.PRECISION_BF16))
    assert e.value.code == gpu.ERR_UNSUPPORTED


# ---- W2XC_PRECISION_BF16X2 / BF16X3 / FP16X2: split products on the 16-bit MFMAs (w2xc_split.hip) --------------
SPLIT_MODES = ["bf16x2", "bf16x3", "fp16x2"]


def _prec(gpu, mode):
    return {"bf16x2": gpu.PRECISION_BF16X2, "bf16x3": gpu.PRECISION_BF16X3, "fp16x2": gpu.PRECISION_FP16X2}[mode]


def _emulated(layers, x, mode, n_in=1):
    import bf16_ref
    return bf16_ref.convert_split_emulated(layers, x, 3 if mode == "bf16x3" else 2, n_in=n_in, fp16=(mode == "fp16x2"))


@pytest.mark.parametrize("mode", SPLIT_MODES)
@pytest.mark.parametrize("planes", [[1, 32, 32, 1], [1, 64, 128, 1], [1, 128, 64, 32, 1], [1, 32, 128, 128, 64, 1],
                                    [1, 128, 32, 64, 1], [1, 32, 32, 64, 64, 128, 128, 1]])
@pytest.mark.parametrize("h,w", [(45, 77), (8, 32), (70, 130)])
def test_split_path_matches_its_emulation(gpu, mode, planes, h, w):
    """Layers 2..n-1 carry every fp32 activation / weight as 2 or 3 16-bit terms and sum 3 (two terms) or 6
    (three) term products in the fp32 accumulator of the bf16 / fp16 MFMA.  Checked against a float64-accumulate
    emulation of exactly that dataflow (incl. the fp16 mode's power-of-two weight scale and clamp): only the
    fp32 accumulation order differs, so the bound is the fp32 path's own (2e-5 of the output range) for the 22-
    and 24-bit modes; BF16X2 re-rounds every activation to 16 bits, so a last-bit difference of the fp32 sums
    moves results at its own 2^-17 level (bound 1e-4).  A dropped / duplicated product or a layout bug misses
    these by orders of magnitude."""
    layers = small_layers(planes, 700 + len(planes) + len(mode))
    ms = gpu._ModelSet.from_layers(layers)
    x = rand_plane(h, w, 9 + h)
    got = ms.convert(x, opts=gpu.make_opts(precision=_prec(gpu, mode)))
    want = _emulated(layers, x, mode)[0]
    scale = float(np.abs(want).max())
    err = float(np.abs(got - want).max())
    print("%s %s %dx%d: max err / range vs emulation %.3g" % (mode, planes, h, w, err / scale))
    assert err <= (1e-4 if mode == "bf16x2" else 2e-5) * scale, (err, scale)


@pytest.mark.parametrize("mode,bound", [("bf16x2", 2e-4), ("bf16x3", 2e-5), ("fp16x2", 2e-5)])
def test_split_vs_fp32_oracle_accuracy_statement(gpu, scale_layers, mode, bound):
    """Stated accuracy vs the CPU convertRoutine (fp32) on the 7-layer scale2.0x topology, [0,1] input: BF16X3 and
    FP16X2 are inside the north-star fp32 tolerance (rtol 1e-4 / atol 1e-5, and <= 2e-5 of the output range: the
    level at which two fp32 summation orders differ); BF16X2 is <= 2e-4 of the range.  Banding and the fused
    nearest-neighbour 2x compose bit-identically, as on the fp32 path."""
    ms = gpu._ModelSet.from_layers(scale_layers)
    x = rand_plane(96, 128, 3)
    want = orc.Oracle(scale_layers).convert(x)
    o = gpu.make_opts(precision=_prec(gpu, mode))
    got = ms.convert(x, opts=o)
    scale = float(np.abs(want).max())
    err = float(np.abs(got - want).max())
    print("%s vs oracle32: max err / range %.3g, PSNR %.1f dB" % (mode, err / scale, psnr(got, want)))
    assert err <= bound * scale, (err, scale)
    if mode != "bf16x2":
        assert_close(got, want, mode + " vs oracle")
    assert np.array_equal(got, ms.convert(x, opts=gpu.make_opts(precision=_prec(gpu, mode), band_rows=17)))
    half = np.ascontiguousarray(x[::2, ::2])
    up = np.repeat(np.repeat(half, 2, 0), 2, 1)
    assert np.array_equal(ms.convert_nn2x(half, o), ms.convert(up, opts=o))


@pytest.mark.parametrize("amp,bound", [(1.0, 2e-5), (100.0, 2e-5), (1e-3, 3e-4), (1e7, None)])
def test_fp16x2_domain(gpu, scale_layers, amp, bound):
    """FP16X2's stated domain: fp16 has 5 exponent bits.  Weights are rescaled per layer (exact), activations are
    not: planes of ordinary amplitude (image data in [0,1], or 100x that) keep the 2e-5 bound; a 1000x dimmer
    plane falls back to BF16X2-like accuracy (the low terms go subnormal: 2^-25 absolute); absurd amplitudes
    saturate at +-65504 in the hidden layers -- finite output, no NaN/inf."""
    ms = gpu._ModelSet.from_layers(scale_layers)
    x = (rand_plane(64, 80, 5) * np.float32(amp)).astype(np.float32)
    got = ms.convert(x, opts=gpu.make_opts(precision=gpu.PRECISION_FP16X2))
    assert np.isfinite(got).all()
    if bound is not None:
        want = orc.Oracle(scale_layers).convert(x)
        err = float(np.abs(got - want).max()) / float(np.abs(want).max())
        print("fp16x2 amplitude %g: max err / range %.3g" % (amp, err))
        assert err <= bound, err


def test_split_unsupported_shapes_are_rejected(gpu):
    for planes in ([1, 5, 1], [32, 32, 1], [1, 32, 7, 1]):
        ms = gpu._ModelSet.from_layers(small_layers(planes, 2))
        x = rand_plane(8, 8, 0)
        for prec in (gpu.PRECISION_BF16X3, gpu.PRECISION_FP16X2):
            with pytest.raises(gpu.W2xcError) as e:
                ms.convert(x, opts=gpu.make_opts(precision=prec))
            assert e.value.code in (gpu.ERR_UNSUPPORTED, gpu.ERR_PLANES)
    ms2 = gpu._ModelSet.from_layers(small_layers([1, 32, 1], 2))
    with pytest.raises(gpu.W2xcError) as e:
        ms2.filter(0, rand_plane(8, 8, 0)[None], gpu.make_opts(precision=gpu.PRECISION_BF16X3))
    assert e.value.code == gpu.ERR_UNSUPPORTED
    # first -> last only: nothing to split, runs as fp32
    x = rand_plane(20, 20, 1)
    assert np.array_equal(ms2.convert(x, opts=gpu.make_opts(precision=gpu.PRECISION_BF16X3)), ms2.convert(x))


@pytest.mark.parametrize("mode", SPLIT_MODES)
def test_split_multi_plane_and_image_pipeline(gpu, scale_layers, mode):
    torch = pytest.importorskip("torch")
    planes = [3, 128, 128, 3]
    layers = small_layers(planes, 31)
    ms = gpu._ModelSet.from_layers(layers)
    h, w = 30, 44
    x = np.random.default_rng(8).random((3, h, w), dtype=np.float32)
    want = _emulated(layers, x, mode, n_in=3)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.zeros((3, h, w), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream()
    ms.convert_planes_device(3, d_in.data_ptr(), h * w * 4, w * 4, w, h, d_out.data_ptr(), h * w * 4, w * 4,
                             stream=st.cuda_stream, opts=gpu.make_opts(device=0, precision=_prec(gpu, mode)))
    st.synchronize()
    assert np.abs(d_out.cpu().numpy() - want).max() <= (1e-4 if mode == "bf16x2" else 2e-5) * np.abs(want).max()
    # the uint8 image pipeline: the split precisions change at most the odd rounding tie
    img = np.random.default_rng(4).integers(0, 256, size=(40, 56, 3), dtype=np.uint8)
    mscale = gpu._ModelSet.from_layers(scale_layers)
    a = mscale.scale2x_image_u8(img, opts=gpu.make_opts(precision=_prec(gpu, mode)))
    b = mscale.scale2x_image_u8(img)
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() <= (2e-2 if mode == "bf16x2" else 2e-3), (d.max(), (d != 0).mean())


@pytest.mark.parametrize("planes", [[3, 128, 128, 3], [3, 32, 64, 2], [2, 5, 4], [1, 32, 32]])
def test_multi_plane_wrapper(gpu, planes):
    """w2xc_convert_planes_device (configs[4] boundary): pad n / layers / crop on several planes equals the
    reference's only route to such a model -- chaining Model::filter on the replicate-padded planes."""
    torch = pytest.importorskip("torch")
    layers = small_layers(planes, 900 + sum(planes))
    n = len(layers)
    ms = gpu._ModelSet.from_layers(layers)
    h, w = 30, 44
    x = np.random.default_rng(8).random((planes[0], h, w), dtype=np.float32)
    o = orc.Oracle(layers)
    t = np.pad(x, ((0, 0), (n, n), (n, n)), mode="edge")
    for l in range(n):
        t = o.filter(l, t, njob=4)
    want = t[:, n:n + h, n:n + w]
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.zeros((planes[-1], h, w), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream()
    ms.convert_planes_device(planes[0], d_in.data_ptr(), h * w * 4, w * 4, w, h, d_out.data_ptr(), h * w * 4, w * 4,
                             stream=st.cuda_stream, opts=gpu.make_opts(device=0))
    st.synchronize()
    assert_close(d_out.cpu().numpy(), want, "planes %s" % planes)


# ---- N2: colour front/back end + bicubic U/V of the CLI scale loop (main.cpp:74-76,136-156,171-172) ----------
def test_color_building_blocks_bit_exact(gpu):
    torch = pytest.importorskip("torch")
    lib = gpu.lib()
    rng = np.random.default_rng(21)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    d_img = torch.from_numpy(img).cuda()
    planes = torch.empty((3, 37, 53), dtype=torch.float32, device="cuda")
    assert lib.w2xc_u8_to_yuv_device(d_img.data_ptr(), 53 * 3, 53, 37, planes[0].data_ptr(), planes[1].data_ptr(),
                                     planes[2].data_ptr(), None) == 0
    torch.cuda.synchronize()
    y, u, v = orc.u8_to_yuv(img)
    got = planes.cpu().numpy()
    assert np.array_equal(got[0], y) and np.array_equal(got[1], u) and np.array_equal(got[2], v)
    for (h, w) in [(37, 53), (1, 1), (2, 3)]:
        src = rng.standard_normal((h, w)).astype(np.float32)
        d_src = torch.from_numpy(src).cuda()
        d_dst = torch.empty((2 * h, 2 * w), dtype=torch.float32, device="cuda")
        assert lib.w2xc_resize2x_cubic_device(d_src.data_ptr(), w, h, d_dst.data_ptr(), None) == 0
        torch.cuda.synchronize()
        assert np.array_equal(d_dst.cpu().numpy(), orc.resize2x_cubic(src)), (h, w)
    yuv = rng.random((3, 20, 31), dtype=np.float32) * 1.4 - 0.2      # exercises both saturation ends
    d_yuv = torch.from_numpy(yuv).cuda()
    d_out = torch.empty((20, 31, 3), dtype=torch.uint8, device="cuda")
    assert lib.w2xc_yuv_to_u8_device(d_yuv[0].data_ptr(), d_yuv[1].data_ptr(), d_yuv[2].data_ptr(), 31, 20, d_out.data_ptr(), 31 * 3, None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), orc.yuv_to_u8(yuv[0], yuv[1], yuv[2]))


@pytest.mark.parametrize("iterations", [1, 2])
def test_scale2x_image_u8_pipeline(gpu, scale_layers, iterations):
    """whole scale phase on a uint8 image vs the CPU restatement of main.cpp's loop: identical bytes with the
    reference-ordered direct kernel; with the MFMA kernels the luma differs by ~1e-6, which may move a value
    across a rounding boundary: at most 1 LSB, on well under 1 % of the bytes."""
    ms = gpu._ModelSet.from_layers(scale_layers)
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (24, 36, 3), dtype=np.uint8)
    want = orc.scale2x_image_u8(orc.Oracle(scale_layers), img, iterations)
    exact = ms.scale2x_image_u8(img, iterations, direct(gpu))
    assert exact.shape == want.shape == (24 << iterations, 36 << iterations, 3)
    assert np.array_equal(exact, want)
    fast = ms.scale2x_image_u8(img, iterations)
    diff = np.abs(fast.astype(np.int16) - want.astype(np.int16))
    assert diff.max() <= 1 and (diff != 0).mean() < 0.01, (diff.max(), (diff != 0).mean())
    # strided input rows (an ROI of a wider image)
    wide = rng.integers(0, 256, (24, 50, 3), dtype=np.uint8)
    roi = wide[:, 5:41]
    out = np.empty_like(want)
    rc = gpu.lib().w2xc_scale2x_image_u8(ms.handle, roi.ctypes.data, wide.strides[0], 36, 24, out.ctypes.data, out.strides[0],
                                         iterations, gpu.make_opts(kernel=gpu.KERNEL_DIRECT))
    assert rc == 0, gpu.last_error()
    assert np.array_equal(out, orc.scale2x_image_u8(orc.Oracle(scale_layers), np.ascontiguousarray(roi), iterations))


@pytest.mark.parametrize("mode", ["noise", "noise_scale"])
def test_process_image_modes(gpu, noise1_layers, scale_layers, mode):
    """-m noise and -m noise_scale of the CLI (main.cpp:83-98 then :126-156) on a uint8 image"""
    mn, msc = gpu._ModelSet.from_layers(noise1_layers), gpu._ModelSet.from_layers(scale_layers)
    img = np.random.default_rng(9).integers(0, 256, (20, 28, 3), dtype=np.uint8)
    it = 1 if mode == "noise_scale" else 0
    want = orc.process_image_u8(img, orc.Oracle(noise1_layers), orc.Oracle(scale_layers) if it else None, it)
    got = gpu.process_image_u8(img, mn, msc if it else None, it, direct(gpu))
    assert np.array_equal(got, want)
    fast = gpu.process_image_u8(img, mn, msc if it else None, it)
    assert np.abs(fast.astype(np.int16) - want.astype(np.int16)).max() <= 1
    with pytest.raises(gpu.W2xcError) as e:
        gpu.process_image_u8(img, None, None, 0)
    assert e.value.code == gpu.ERR_ARG


@pytest.mark.parametrize("ratio", [1.5, 3.0])
def test_process_image_shrink(gpu, scale_layers, ratio):
    """--scale_ratio that is not a power of two: iter = ceil(log2 r) 2x steps, then INTER_LINEAR shrink by
    r / 2^iter (main.cpp:107-114,158-167)"""
    import math
    msc = gpu._ModelSet.from_layers(scale_layers)
    img = np.random.default_rng(3).integers(0, 256, (18, 26, 3), dtype=np.uint8)
    it = int(math.ceil(math.log2(ratio)))
    shrink = ratio / 2.0 ** it
    want = orc.process_image_u8(img, None, orc.Oracle(scale_layers), it, shrink)
    got = gpu.process_image_u8(img, None, msc, it, direct(gpu), shrink)
    assert got.shape == want.shape == (int(float((18 << it) * shrink)), int(float((26 << it) * shrink)), 3)
    assert np.array_equal(got, want)


def test_cli_shell_end_to_end(gpu, models_dir, tmp_path):
    """N4: tools/w2xc_cli.py -m noise_scale --scale_ratio 1.5 on a PNG vs the CPU restatement of main.cpp"""
    import subprocess, sys
    from PIL import Image
    from conftest import ROOT
    rgb = np.random.default_rng(12).integers(0, 256, (20, 30, 3), dtype=np.uint8)
    src = tmp_path / "in.png"
    Image.fromarray(rgb).save(src)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "w2xc_cli.py"), "-i", str(src), "-m", "noise_scale",
                        "--noise_level", "2", "--scale_ratio", "1.5", "--model_dir", models_dir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out_path = tmp_path / "in(noise_scale)(Level2)(x1.500000).png"
    assert out_path.exists() and "process successfully done!" in r.stdout
    got = np.asarray(Image.open(out_path))
    no = orc.Oracle.from_json(os.path.join(models_dir, "noise2_model.json"))
    so = orc.Oracle.from_json(os.path.join(models_dir, "scale2.0x_model.json"))
    want = orc.process_image_u8(np.ascontiguousarray(rgb[:, :, ::-1]), no, so, 1, 0.75)[:, :, ::-1]
    assert got.shape == want.shape == (30, 45, 3)
    assert np.abs(got.astype(np.int16) - want.astype(np.int16)).max() <= 1


@pytest.mark.parametrize("precision", ["fp32", "bf16x2", "bf16x3", "fp16x2"])
def test_mfma2_race_screen(gpu, scale_layers, precision):
    """conv3x3_mfma2 (and conv3x3_split, same protocol) orders its LDS-DMA transfers with hand-counted vmcnt + raw barriers.  It is deterministic by
    construction, so a protocol error (a transfer landing late, a ring slot overwritten early) would surface as
    run-to-run differences, most likely under memory load: repeat on the same input with and without a
    competing HBM stream, demand bit-identical planes (tools/stress_determinism.py is the long form)."""
    torch = pytest.importorskip("torch")
    ms = gpu._ModelSet.from_layers(scale_layers)
    o = gpu.make_opts(device=0, precision={"fp32": gpu.PRECISION_FP32, "bf16x2": gpu.PRECISION_BF16X2, "bf16x3": gpu.PRECISION_BF16X3, "fp16x2": gpu.PRECISION_FP16X2}[precision])
    st = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    junk = torch.rand(4096, 4096, device="cuda")
    for (h, w) in [(700, 1900), (257, 1025)]:
        x = torch.rand(h, w, device="cuda")
        ref = torch.empty_like(x)
        ms.convert_device(x.data_ptr(), w * 4, w, h, ref.data_ptr(), w * 4, stream=st.cuda_stream, opts=o)
        torch.cuda.synchronize()
        for it in range(6):
            y = torch.empty_like(x)
            if it % 2:
                with torch.cuda.stream(side):
                    for _ in range(10):
                        junk = junk * 1.0001 + 0.1
            ms.convert_device(x.data_ptr(), w * 4, w, h, y.data_ptr(), w * 4, stream=st.cuda_stream, opts=o)
            torch.cuda.synchronize()
            assert torch.equal(y, ref), "run %d of %dx%d differs" % (it, h, w)


def test_host_multi_band_path(gpu, scale_layers):
    """the in-process multi-device path of w2xc_convert_plane / _nn2x (one host thread per unit: upload rows +
    halo rows, convert, download rows) -- on a 1-GPU box w2xc_opts.host_units = 3 runs its three units on the same
    device.  Result must be bit-identical to the one-unit run."""
    ms = gpu._ModelSet.from_layers(scale_layers)
    x = np.random.default_rng(4).random((101, 77), dtype=np.float32)
    o3 = gpu.make_opts(host_units=3)
    assert np.array_equal(ms.convert(x), ms.convert(x, opts=o3))
    assert np.array_equal(ms.convert_nn2x(x), ms.convert_nn2x(x, opts=o3))


def test_default_precision_from_environment(gpu, scale_layers, monkeypatch):
    """callers that pass no w2xc_opts (the C++ adapter behind the unmodified CLI) get the process defaults: W2XC_PRECISION from the
    environment (read where the defaults are formed -- w2xc_set_default_opts(NULL) re-reads it), or whatever w2xc_set_default_opts was given.
    The bf16x3 default must equal an explicit BF16X3 call bit for bit, and differ from fp32."""
    ms = gpu._ModelSet.from_layers(scale_layers)
    x = np.random.default_rng(4).random((64, 96), dtype=np.float32)
    x3 = ms.convert(x, opts=gpu.make_opts(precision=gpu.PRECISION_BF16X3))
    fp32 = ms.convert(x, opts=gpu.make_opts(precision=gpu.PRECISION_FP32))
    assert not np.array_equal(x3, fp32)
    try:
        monkeypatch.setenv("W2XC_PRECISION", "bf16x3")
        assert gpu.lib().w2xc_set_default_opts(None) == 0
        assert np.array_equal(ms.convert(x), x3)
        monkeypatch.delenv("W2XC_PRECISION")
        assert gpu.lib().w2xc_set_default_opts(None) == 0
        assert np.array_equal(ms.convert(x), fp32)
        o = gpu.make_opts(precision=gpu.PRECISION_BF16X3)
        assert gpu.lib().w2xc_set_default_opts(o) == 0
        assert np.array_equal(ms.convert(x), x3)
    finally:
        monkeypatch.delenv("W2XC_PRECISION", raising=False)
        gpu.lib().w2xc_set_default_opts(None)
    assert np.array_equal(ms.convert(x), fp32)


def test_cli_shell_precision_flag(gpu, models_dir, tmp_path):
    """tools/w2xc_cli.py --precision bf16x3: same picture as the fp32 run up to single LSBs"""
    import subprocess, sys
    from PIL import Image
    from conftest import ROOT
    rgb = np.random.default_rng(13).integers(0, 256, (24, 28, 3), dtype=np.uint8)
    src = tmp_path / "in.png"
    Image.fromarray(rgb).save(src)
    outs = []
    for prec in ("fp32", "bf16x3"):
        o = tmp_path / ("out_%s.png" % prec)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "w2xc_cli.py"), "-i", str(src), "-o", str(o), "-m", "scale",
                            "--model_dir", models_dir, "--precision", prec], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        outs.append(np.asarray(Image.open(o)).astype(np.int16))
    assert outs[0].shape == (48, 56, 3) and np.abs(outs[0] - outs[1]).max() <= 1


def test_fused_last_layer_vs_unfused(gpu):
    """the 16-bit modes compute the one-plane last layer inside the epilogue of the layer before it (conv3x3_split
    out_terms = 9 + conv3x3_last_gather).  w2xc_opts.fusion = W2XC_FUSION_FIRST restores the separate fp32 conv3x3_last: the two
    must agree to the fp32-order level on odd sizes, borders, banding and the nearest-2x entry point."""
    res = []
    for fusion in (gpu.FUSION_AUTO, gpu.FUSION_FIRST):
        outs, flags = [], []
        for planes, seed in (([1, 32, 32, 64, 64, 128, 128, 1], 102), ([1, 32, 64, 1], 7), ([1, 64, 32, 1], 8)):
            ms = gpu._ModelSet.from_layers(gen_model.synth_layers(planes, seed))
            for prec in (gpu.PRECISION_FP16X2, gpu.PRECISION_BF16X2, gpu.PRECISION_BF16X3):
                for (h, wd) in ((37, 61), (8, 32), (130, 70)):
                    x = np.random.default_rng(h).random((h, wd), dtype=np.float32)
                    outs.append(ms.convert(x, opts=gpu.make_opts(precision=prec, fusion=fusion)).ravel())
                    outs.append(ms.convert(x, opts=gpu.make_opts(precision=prec, fusion=fusion, band_rows=11)).ravel())
                    outs.append(ms.convert_nn2x(x, gpu.make_opts(precision=prec, fusion=fusion)).ravel())
            flags.append(ms.kernel_name(len(planes) - 2, gpu.make_opts(precision=gpu.PRECISION_FP16X2, fusion=fusion)) == 'conv3x3_last_gather')
        res.append((np.concatenate(outs), flags))
    (a, fa), (b, fb) = res
    assert a.shape == b.shape
    assert all(fa) and not any(fb)                                  # the fused path really ran (and really did not)
    scale = np.abs(b).max()
    assert np.abs(a - b).max() <= 1e-4 * scale                      # BF16X2 bound; FP16X2 is ~1e-6


def test_fused_first_two_layers_vs_unfused(gpu):
    """the 16-bit modes run layers 1 (1 -> 32) and 2 (32 -> C) as ONE kernel (conv3x3_first2_split: layer 1's terms stay in
    LDS).  It uses the same arithmetic in the same order as conv3x3_first_split + conv3x3_split, so with
    w2xc_opts.fusion = W2XC_FUSION_LAST the results must be BIT-IDENTICAL -- odd sizes, borders, banding, nearest-2x, every precision."""
    res = []
    for fusion in (gpu.FUSION_AUTO, gpu.FUSION_LAST):
        outs, flags = [], []
        for planes, seed in (([1, 32, 32, 64, 64, 128, 128, 1], 102), ([1, 32, 64, 32, 1], 7), ([1, 32, 128, 64, 1], 8), ([1, 32, 32, 3], 9)):
            ms = gpu._ModelSet.from_layers(gen_model.synth_layers(planes, seed))
            for prec in (gpu.PRECISION_FP16X2, gpu.PRECISION_BF16X2, gpu.PRECISION_BF16X3, gpu.PRECISION_BF16):
                if planes[-1] != 1:
                    continue
                for (h, wd) in ((37, 61), (8, 32), (130, 70)):
                    x = np.random.default_rng(h).random((h, wd), dtype=np.float32)
                    outs.append(ms.convert(x, opts=gpu.make_opts(precision=prec, fusion=fusion)).ravel())
                    outs.append(ms.convert(x, opts=gpu.make_opts(precision=prec, fusion=fusion, band_rows=11)).ravel())
                    outs.append(ms.convert_nn2x(x, gpu.make_opts(precision=prec, fusion=fusion)).ravel())
            flags.append(ms.kernel_name(1, gpu.make_opts(precision=gpu.PRECISION_FP16X2, fusion=fusion)) == 'conv3x3_first2_split')
        res.append((np.concatenate(outs), flags))
    (a, fa), (b, fb) = res
    assert a.shape == b.shape
    assert all(fa) and not any(fb)                                  # the fused kernel really ran (and really did not)
    assert np.array_equal(a, b)
