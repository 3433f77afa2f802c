"""The Winograd kernels of the fp32 mid layers -- conv3x3_wino (csrc/w2xc_wino.hip, F(2x2,3x3) on v_mfma_f32_32x32x2_f32, W2XC_KERNEL_WINOGRAD32 and its
alias W2XC_KERNEL_WINOGRAD) and conv3x3_wino4 (csrc/w2xc_wino4.hip, F(4x4,3x3), the default) -- against the direct fp32 MFMA kernel (conv3x3_mfma2) and the
CPU oracle of Model::filterWorker (/root/reference/src/modelHandler.cpp:117-159).  The kernel is chosen per call through
w2xc_opts.kernel, so all of them run in this one process.  (The round-3 F(2x2) kernel on 16x16x4 tiles, conv3x3_wino16, was retired in round 5.)"""
import numpy as np
import pytest

from conftest import assert_close, rand_plane, small_layers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(w2xc):
    assert w2xc.device_count() >= 1, "no HIP device visible: libw2xc_hip has no CPU fallback, -m gpu tests need an MI355X"
    return w2xc


KERNELS = {"conv3x3_wino": "KERNEL_WINOGRAD32", "conv3x3_mfma": "KERNEL_MFMA"}


def _opts(w, name, **kw):
    return w.make_opts(kernel=getattr(w, KERNELS[name]), **kw)


def _run(w, name):
    """every instantiated shape (32/64/128 -> 32/64/128), odd sizes, planes smaller than a work item, banding, nearest-2x, Model::filter"""
    from tools import gen_model
    outs, names = [], []
    for planes, seed in (([1, 32, 64, 64, 128, 128, 1], 31), ([1, 64, 128, 64, 64, 1], 32), ([1, 32, 64, 128, 128, 64, 1], 33),
                         ([1, 32, 32, 128, 32, 1], 34), ([1, 64, 32, 1], 35)):
        layers = gen_model.synth_layers(planes, seed)
        ms = w._ModelSet.from_layers(layers)
        names += [ms.kernel_name(l, _opts(w, name)) for l in range(len(planes) - 1)]
        for (h, wd) in ((37, 61), (8, 32), (130, 70), (16, 33), (1, 1)):
            x = np.random.default_rng(h * 7 + wd).random((h, wd), dtype=np.float32)
            a = ms.convert(x, opts=_opts(w, name))
            outs.append(a.ravel())
            for band in (1, 5, 16):   # odd and even band origins: the 2x2 blocks stay on even rows of the plane
                assert np.array_equal(a, ms.convert(x, opts=_opts(w, name, band_rows=band))), ("banding", name, planes, h, wd, band)
            outs.append(ms.convert_nn2x(x, opts=_opts(w, name)).ravel())
        l = next(i for i in range(len(planes) - 1) if planes[i] >= 32 and planes[i + 1] >= 32)
        outs.append(ms.filter(l, np.random.default_rng(6).random((planes[l], 21, 45), dtype=np.float32), opts=_opts(w, name)).ravel())   # same-size conv
    return np.concatenate(outs), names


@pytest.fixture(scope="module")
def runs(gpu):
    return {name: _run(gpu, name) for name in KERNELS}


@pytest.mark.parametrize("name", ["conv3x3_wino"])
def test_winograd_vs_direct_mfma(gpu, runs, name):
    """same fp32 arithmetic type, other summation order: every output within 4e-6 of the output range of the direct MFMA kernel's
    (two fp32 summation orders differ by about that much: conv3x3_mfma2 vs the oracle is 3-4e-6 too); banding from odd and even rows
    is bit-identical inside each kernel's run."""
    a, names_w = runs[name]
    b, names_d = runs["conv3x3_mfma"]
    assert name in names_w and name not in names_d
    # every mid layer (32 / 64 / 128 planes in and out) takes the Winograd kernel
    assert names_w.count("conv3x3_wino") == names_d.count("conv3x3_mfma")
    assert a.shape == b.shape and np.isfinite(a).all()
    err = np.abs(a - b).max() / np.abs(b).max()
    print("%s vs conv3x3_mfma2: max err %.2e of the output range" % (name, err))
    assert err <= 4e-6, err


def test_kernel_choice_per_call(gpu):
    import os
    from tools import gen_model
    ms = gpu._ModelSet.from_layers(gen_model.synth_layers([1, 32, 64, 1], 5))
    assert ms.kernel_name(1) == ms.kernel_name(1, gpu.make_opts()) == "conv3x3_wino4"   # W2XC_KERNEL_AUTO
    assert ms.kernel_name(1, gpu.make_opts(kernel=gpu.KERNEL_MFMA)) == "conv3x3_mfma"
    assert ms.kernel_name(1, gpu.make_opts(kernel=gpu.KERNEL_WINOGRAD32)) == "conv3x3_wino"
    assert ms.kernel_name(1, gpu.make_opts(kernel=gpu.KERNEL_WINOGRAD)) == "conv3x3_wino"       # (alias since conv3x3_wino16 was retired)
    assert ms.kernel_name(1, gpu.make_opts(kernel=gpu.KERNEL_WINOGRAD4)) == "conv3x3_wino4"
    assert ms.kernel_name(1, gpu.make_opts(kernel=gpu.KERNEL_DIRECT)) == "conv3x3_direct"


@pytest.mark.parametrize("name", ["conv3x3_wino"])
@pytest.mark.parametrize("planes", [[1, 64, 64, 1], [1, 32, 128, 128, 1], [1, 64, 128, 64, 1], [1, 32, 32, 32, 1]])
def test_winograd_vs_oracle(gpu, planes, name):
    """against the CPU oracle, rtol 1e-4 + atol 1e-5 (north_star) and max-norm 1e-5, on a plane that is not a multiple of the work item"""
    from oracle import oracle as orc
    from tools import gen_model
    layers = gen_model.synth_layers(planes, 900 + len(planes))
    ms = gpu._ModelSet.from_layers(layers)
    assert name in [ms.kernel_name(l, _opts(gpu, name)) for l in range(len(planes) - 1)]
    x = np.random.default_rng(11).random((75, 101), dtype=np.float32)
    got, want = ms.convert(x, opts=_opts(gpu, name)), orc.Oracle(layers).convert(x)
    assert np.allclose(got, want, rtol=1e-4, atol=1e-5)
    assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()


@pytest.mark.parametrize("mid", ["conv3x3_wino4"])
@pytest.mark.parametrize("planes", [[1, 32, 32, 64, 64, 128, 128, 1], [1, 32, 64, 1], [1, 64, 128, 64, 1], [1, 32, 128, 1]])
def test_fused_last_layer_fp32_vs_unfused(gpu, planes, mid):
    """N3 on the fp32 path: the one-plane last layer inside the epilogue of the DEFAULT kernel conv3x3_wino4 (taps-as-rows MFMAs on the activations the
    epilogue has just produced, the four plane tiles of a 64-plane block summed on chip: Cout / 64 x 9 partial tap planes + conv3x3_last_gather)
    against the separate conv3x3_last launch
    (w2xc_opts.fusion = W2XC_FUSION_ON / _OFF) and against the CPU oracle.  Same fp32 arithmetic type, the last layer's channel sum
    split in 32-plane partials: the two runs agree to the level two fp32 summation orders do.  Odd sizes, planes smaller than a
    work item, banding (bit-identical inside the fused run), the nearest-2x entry and the host pipeline's chunked path (>= 128 rows)."""
    from oracle import oracle as orc
    from tools import gen_model
    layers = gen_model.synth_layers(planes, 77 + len(planes))
    ms = gpu._ModelSet.from_layers(layers)
    n = len(planes) - 1
    kern = gpu.KERNEL_WINOGRAD4
    on, off = gpu.make_opts(fusion=gpu.FUSION_ON, kernel=kern), gpu.make_opts(fusion=gpu.FUSION_OFF, kernel=kern)
    # (the launch of layer n - 1 also FINISHES the last layer where conv3x3_wino4 PROG has an instantiation -- planar 64 / 128-plane inputs; a gather launch follows otherwise)
    # (the host entry points let the launch of layer n - 1 FINISH the last layer where conv3x3_wino4 PROG has an instantiation -- planar 64 / 128-plane inputs --,
    #  the device entry points -- which kernel_name describes -- only with W2XC_FUSION_PROG; a gather launch follows otherwise.  Bit-identical either way.)
    assert ms.kernel_name(n - 1, on) == "conv3x3_last_gather" and ms.kernel_name(n - 2, on) == mid
    assert ms.kernel_name(n - 1, off) == "conv3x3_last"
    if mid == "conv3x3_wino4":   # ... and it is what the default options run
        assert ms.kernel_name(n - 1) == "conv3x3_last_gather" and ms.kernel_name(n - 2) == mid
    sep = gpu.make_opts(fusion=gpu.FUSION_GATHER_LAUNCH, kernel=kern)   # the gather launch in the host entry point too
    assert ms.kernel_name(n - 1, sep) == "conv3x3_last_gather"
    has_prog = planes[-3] in (64, 128) and len(planes) > 4
    assert ms.kernel_name(n - 1, gpu.make_opts(fusion=gpu.FUSION_PROG, kernel=kern)) == ("(in_previous_layer)" if has_prog else "conv3x3_last_gather")
    gate = 4e-5   # (the max-norm gate of the F(4x4) kernel on short models, test_wino4_f4x4_kernel)
    o = orc.Oracle(layers)
    worst = 0.0
    for (h, wd) in ((37, 61), (8, 32), (300, 170), (1, 1), (16, 33)):
        x = np.random.default_rng(h * 11 + wd).random((h, wd), dtype=np.float32)
        a, b = ms.convert(x, opts=on), ms.convert(x, opts=off)
        assert np.array_equal(a, ms.convert(x, opts=sep)), ("gather in the launch vs gather launch", planes, h, wd)
        worst = max(worst, float(np.abs(a - b).max() / np.abs(b).max()))
        for band in (1, 7, 64):
            assert np.array_equal(a, ms.convert(x, opts=gpu.make_opts(fusion=gpu.FUSION_ON, kernel=kern, band_rows=band))), ("banding", planes, h, wd, band)
        want = o.convert(x, njob=8)
        assert np.allclose(a, want, rtol=1e-4, atol=1e-5) and np.abs(a - want).max() <= gate * np.abs(want).max()
        a2, b2 = ms.convert_nn2x(x, opts=on), ms.convert_nn2x(x, opts=off)
        worst = max(worst, float(np.abs(a2 - b2).max() / np.abs(b2).max()))
    print("fused vs unfused last layer, %s: max err %.2e of the output range" % (planes, worst))
    assert worst <= 4e-6, worst


@pytest.mark.parametrize("planes", [[1, 32, 32, 64, 64, 128, 128, 1], [1, 32, 32, 64, 32, 1], [1, 32, 32, 128, 64, 1]])
def test_fused_first_layers_fp32_vs_unfused(gpu, planes):
    """N3 on the fp32 path, the front end: layers 1 (1 -> 32) and 2 (32 -> 32) in ONE launch (conv3x3_first2_wino4: layer 1 on the fly per Winograd patch,
    layer 2 as F(4x4,3x3) with register-stationary weights; convertRoutine.cpp:66-76's loop collapsed by one more iteration) against the separate launches
    conv3x3_first + conv3x3_wino (w2xc_opts.fusion = W2XC_FUSION_OFF), against the CPU oracle, across bandings (bit-identical), through the nearest-2x entry
    (the upscale folded into the fused kernel's tile fill), through the host pipeline (chunked under the upload: bit-identical to resident) and as
    row shards (bit-identical).  The two forms differ like two fp32 summation orders do (layer 2 runs F(4x4) fused, F(2x2) alone)."""
    from oracle import oracle as orc
    from tools import gen_model
    torch = pytest.importorskip("torch")
    layers = gen_model.synth_layers(planes, 500 + len(planes))
    ms = gpu._ModelSet.from_layers(layers)
    on, off = gpu.make_opts(), gpu.make_opts(fusion=gpu.FUSION_OFF)
    assert ms.kernel_name(0, on) == "(in_next_layer)" and ms.kernel_name(1, on) == "conv3x3_first2_wino4"
    assert ms.kernel_name(0, off) == "conv3x3_first" and ms.kernel_name(1, off) == "conv3x3_wino"
    o = orc.Oracle(layers)
    worst = 0.0
    for (h, wd) in ((37, 61), (8, 32), (300, 170), (1, 1), (16, 33), (129, 257)):
        x = np.random.default_rng(h * 13 + wd).random((h, wd), dtype=np.float32)
        a, b = ms.convert(x, opts=on), ms.convert(x, opts=off)
        worst = max(worst, float(np.abs(a - b).max() / np.abs(b).max()))
        for band in (1, 7, 64):
            assert np.array_equal(a, ms.convert(x, opts=gpu.make_opts(band_rows=band))), ("banding", planes, h, wd, band)
        want = o.convert(x, njob=8)
        assert np.allclose(a, want, rtol=1e-4, atol=1e-5) and np.abs(a - want).max() <= 4e-5 * np.abs(want).max()
        a2, b2 = ms.convert_nn2x(x, opts=on), ms.convert_nn2x(x, opts=off)
        worst = max(worst, float(np.abs(a2 - b2).max() / np.abs(b2).max()))
        assert np.array_equal(a2, ms.convert(np.repeat(np.repeat(x, 2, 0), 2, 1), opts=on))   # the folded nearest-2x == the explicit one
        # resident (device pointers) == host pipeline, bit for bit
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.empty_like(d_in)
        ms.convert_device(d_in.data_ptr(), wd * 4, wd, h, d_out.data_ptr(), wd * 4, opts=gpu.make_opts(device=0))
        torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy(), a)
    # row shards with the wide halo stitch to the whole plane
    h, wd = 150, 90
    x = np.random.default_rng(3).random((h, wd), dtype=np.float32)
    whole = ms.convert(x)
    n = len(planes) - 1
    out = torch.zeros((h, wd), dtype=torch.float32, device="cuda")
    for p in range(3):
        ra, rb = gpu.shard_rows(h, 3, p)
        y0, y1 = gpu.shard_view(h, ra, rb, 4 * n)
        view = torch.from_numpy(np.ascontiguousarray(x[y0:y1])).cuda()
        ms.convert_rows_device(view.data_ptr(), wd * 4, y1 - y0, y0, wd, h, ra, rb, out[ra:].data_ptr(), wd * 4, opts=gpu.make_opts(device=0))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), whole)
    print("fused vs unfused first two layers, %s: max err %.2e of the output range" % (planes, worst))
    assert worst <= 6e-6, worst


def test_wino4_f4x4_kernel(gpu):
    """conv3x3_wino4 (csrc/w2xc_wino4.hip): Winograd F(4x4,3x3) on the fp32 MFMA, the default mid-layer kernel since round 3 (layers with >= 64 output
    planes; the others take the F(2x2) kernels), here asked for through w2xc_opts.kernel = W2XC_KERNEL_WINOGRAD4.  Same fp32 arithmetic type and the same
    north_star gate against the CPU oracle (rtol 1e-4 + atol 1e-5); its larger transform (interpolation points 0, +-3/4, +-3/2) costs ~1.3x the rounding error
    of F(2x2): measured 3.1e-6 of the output range on the 7-layer scale2.0x topology, up to 1.4e-5 against the oracle on the short test models (stated max-norm
    gate: 4e-5 against the oracle and the direct MFMA kernel).  An output of a 4x4 block depends -- at rounding level -- on all 36 patch values, so the engine
    runs these calls on band regions that end on block rows (four rows of halo per layer, run_rows): banded runs are BIT-identical to the unbanded one."""
    from oracle import oracle as orc
    from tools import gen_model
    w = gpu
    o4 = lambda **kw: w.make_opts(kernel=w.KERNEL_WINOGRAD4, **kw)
    worst_o = worst_d = 0.0
    for planes, seed in (([1, 32, 64, 64, 128, 128, 1], 31), ([1, 64, 128, 64, 64, 1], 32), ([1, 32, 64, 128, 128, 64, 1], 33), ([1, 32, 32, 128, 32, 1], 34),
                         ([1, 32, 32, 64, 64, 128, 128, 1], 102)):
        layers = gen_model.synth_layers(planes, seed)
        ms = w._ModelSet.from_layers(layers)
        names = [ms.kernel_name(l, o4()) for l in range(len(planes) - 1)]
        for l in range(len(planes) - 1):
            if planes[l] >= 32 and planes[l + 1] >= 64:
                assert names[l] == "conv3x3_wino4", (planes, l, names)
        oracle = orc.Oracle(layers)
        for (h, wd) in ((37, 61), (16, 32), (130, 70), (17, 33), (1, 1), (64, 200)):
            x = np.random.default_rng(h * 7 + wd).random((h, wd), dtype=np.float32)
            a = ms.convert(x, opts=o4())
            want = oracle.convert(x, njob=8)
            d = ms.convert(x, opts=w.make_opts(kernel=w.KERNEL_MFMA))
            assert np.allclose(a, want, rtol=1e-4, atol=1e-5), (planes, h, wd)
            rng = float(np.abs(want).max())
            worst_o = max(worst_o, float(np.abs(a - want).max()) / rng)
            worst_d = max(worst_d, float(np.abs(a - d).max()) / rng)
            for band in (1, 5, 16, 64):
                assert np.array_equal(a, ms.convert(x, opts=o4(band_rows=band))), ("banding", planes, h, wd, band)
            a2 = ms.convert_nn2x(x, opts=o4())
            worst_d = max(worst_d, float(np.abs(a2 - ms.convert_nn2x(x, opts=w.make_opts(kernel=w.KERNEL_MFMA))).max()) / float(np.abs(a2).max()))
        l = next(i for i in range(len(planes) - 1) if planes[i] >= 32 and planes[i + 1] >= 64)
        xin = np.random.default_rng(6).random((planes[l], 21, 45), dtype=np.float32)
        f4, fd = ms.filter(l, xin, opts=o4()), ms.filter(l, xin, opts=w.make_opts(kernel=w.KERNEL_MFMA))   # Model::filter: same-size conv
        worst_d = max(worst_d, float(np.abs(f4 - fd).max() / np.abs(fd).max()))
    print("conv3x3_wino4: max err %.2e of the output range vs the oracle, %.2e vs conv3x3_mfma2; bandings bit-identical" % (worst_o, worst_d))
    assert worst_o <= 4e-5 and worst_d <= 4e-5, (worst_o, worst_d)


def test_wino4_whole_frame_vs_direct_mfma(gpu):
    """every pixel of the 2160x3840 plane of BASELINE configs[1] (strip walk, XCD chunks, all tile borders): F(4x4) vs the direct MFMA kernel"""
    from tools import gen_model
    ms = gpu._ModelSet.from_layers(gen_model.synth_layers(seed=102))
    x = np.random.default_rng(2).random((2160, 3840), dtype=np.float32)
    a = ms.convert(x, opts=gpu.make_opts(kernel=gpu.KERNEL_WINOGRAD4))
    for _ in range(3):   # the counted-vmcnt / barrier protocol is deterministic: a transfer landing late or a buffer overwritten early shows up as differing tiles
        assert np.array_equal(a, ms.convert(x, opts=gpu.make_opts(kernel=gpu.KERNEL_WINOGRAD4)))
    d = ms.convert(x, opts=gpu.make_opts(kernel=gpu.KERNEL_MFMA))
    err = float(np.abs(a - d).max() / np.abs(d).max())
    print("conv3x3_wino4 whole frame vs conv3x3_mfma2: %.2e of the output range" % err)
    assert np.isfinite(a).all() and err <= 2e-5, err


def device_convert(gpu, ms, x, **okw):
    import torch
    h, wd = x.shape
    d_in = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    d_out = torch.empty_like(d_in)
    st = torch.cuda.current_stream()
    ms.convert_device(d_in.data_ptr(), wd * 4, wd, h, d_out.data_ptr(), wd * 4, stream=st.cuda_stream, opts=gpu.make_opts(device=0, **okw))
    st.synchronize()
    return d_out.cpu().numpy()


@pytest.mark.parametrize("planes", [[1, 32, 32, 64, 64, 128, 128, 1], [1, 32, 32, 64, 64, 1], [1, 32, 64, 128, 64, 1], [1, 32, 128, 128, 1]])
def test_last_layer_finished_inside_the_producing_launch(gpu, planes):
    """conv3x3_wino4 PROG (round 6): the launch of layer n - 1 writes its partial tap planes through to memory, every workgroup counts its arrival on the
    16-row x 256-column gather jobs its tile feeds, and the workgroup whose arrival completes a job sums it -- the last layer is finished inside the launch,
    rows complete top to bottom (what lets the host pipeline ship rows while layer n - 1 is still running; the stitch of convertRoutine.cpp:143-161).
    Same sum in the same order as conv3x3_last_gather (w2xc_opts.fusion = W2XC_FUSION_GATHER_LAUNCH): BIT-identical for every size -- planes smaller than a
    tile, one tile column, widths that are no multiple of 4 / 32 / 256 (ragged quads, ragged job groups, XCD bands of unequal width), more tile rows than a
    dither period, bandings, row shards with the wide halo, the nearest-2x entry -- and repeatable (the job counters restart with every launch)."""
    layers = small_layers(planes, 300 + len(planes))
    ms = gpu._ModelSet.from_layers(layers)
    n = len(planes) - 1
    prog, sep = gpu.make_opts(fusion=gpu.FUSION_PROG), gpu.make_opts(fusion=gpu.FUSION_GATHER_LAUNCH)
    assert ms.kernel_name(n - 1, prog) == "(in_previous_layer)" and ms.kernel_name(n - 1, sep) == "conv3x3_last_gather"
    assert ms.kernel_name(n - 1) == "conv3x3_last_gather"   # (what the device entry points launch by default; ms.convert below is the HOST entry: PROG by default)
    for (h, wd) in ((1, 1), (5, 3), (16, 32), (17, 33), (40, 257), (150, 290), (333, 1000), (131, 2051), (700, 70)):
        x = rand_plane(h, wd, h * 7 + wd)
        want = ms.convert(x, opts=sep)
        for rep in range(2):
            assert np.array_equal(ms.convert(x, opts=prog), want), (planes, h, wd, rep)            # host entry: rows over PCIe, followed by the drainer
            assert np.array_equal(ms.convert(x), want), (planes, h, wd, rep, "default host path")
        assert np.array_equal(device_convert(gpu, ms, x, fusion=gpu.FUSION_PROG), want), (planes, h, wd, "device entry")
        assert np.array_equal(device_convert(gpu, ms, x), want), (planes, h, wd, "device entry, gather launch")
        for band in (16, 100):
            if band < h:
                assert np.array_equal(ms.convert(x, opts=gpu.make_opts(band_rows=band)), want), (planes, h, wd, band)
                assert np.array_equal(device_convert(gpu, ms, x, fusion=gpu.FUSION_PROG, band_rows=band), want), (planes, h, wd, band, "device entry")
        assert np.array_equal(ms.convert_nn2x(x[:60, :70], opts=prog), ms.convert_nn2x(x[:60, :70], opts=sep))
    # against the oracle on one size, and row shards with the wide halo through the device entry point
    import torch
    from oracle import oracle as orc
    x = rand_plane(300, 417, 9)
    want = ms.convert(x, opts=sep)
    assert_close(want, orc.Oracle(layers).convert(x, njob=8), "gather launch vs oracle")
    h, wd = x.shape
    out = np.empty_like(x)
    for p in range(3):
        ra, rb = gpu.shard_rows(h, 3, p)
        y0, y1 = gpu.shard_view(h, ra, rb, 4 * n)
        d_view = torch.from_numpy(np.ascontiguousarray(x[y0:y1])).cuda()
        d_out = torch.empty((rb - ra, wd), dtype=torch.float32, device="cuda")
        st = torch.cuda.current_stream()
        ms.convert_rows_device(d_view.data_ptr(), wd * 4, y1 - y0, y0, wd, h, ra, rb, d_out.data_ptr(), wd * 4, stream=st.cuda_stream, opts=gpu.make_opts(device=0, fusion=gpu.FUSION_PROG))
        st.synchronize()
        out[ra:rb] = d_out.cpu().numpy()
    assert np.array_equal(out, want)
