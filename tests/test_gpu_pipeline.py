"""GPU tests of the host side of the path (run with -m gpu on an MI355X): the host->host tile farm behind
w2xc_convert_plane / _nn2x / _rows (pinned staging rings, H2D || layers || D2H + stitch, chunked last layer), the
units of the multi-GPU farm, Model::filter's persistent buffers and device-resident chain, and bench.py's JSON lines
at N = 1 and N = 2 (two ranks on one GPU over gloo).  Everything goes through the C ABI; the checker is the oracle."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, assert_close, rand_plane, small_layers
from tools import gen_model
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(w2xc):
    assert w2xc.device_count() >= 1, "no HIP device visible: libw2xc_hip has no CPU fallback"
    return w2xc


def device_result(gpu, ms, x, nn2x=False, **okw):
    """the same conversion through the device-pointer entry point (one band, one launch per layer)"""
    import torch
    h, w = x.shape
    up = 2 if nn2x else 1
    d_in = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    d_out = torch.empty((h * up, w * up), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream()
    o = gpu.make_opts(device=0, **okw)
    if nn2x:
        ms.convert_nn2x_device(d_in.data_ptr(), w * 4, w, h, d_out.data_ptr(), w * up * 4, stream=st.cuda_stream, opts=o)
    else:
        ms.convert_device(d_in.data_ptr(), w * 4, w, h, d_out.data_ptr(), w * 4, stream=st.cuda_stream, opts=o)
    st.synchronize()
    return d_out.cpu().numpy()


# ---- the host -> host pipeline -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("chunk_kb", ["", "16", "100000"])
@pytest.mark.parametrize("nn2x", [False, True])
def test_host_pipeline_bit_identical_to_resident_path(gpu, scale_layers, monkeypatch, chunk_kb, nn2x):
    """pageable planes through the staging rings, last layer in row chunks (16 KiB chunks: ~70 chunks, every slot of both
    rings reused many times; huge chunks: none) == the device-pointer entry point, bit for bit, and close to the oracle"""
    ho = gpu.make_opts(host_chunk_kb=int(chunk_kb or 0))
    ms = gpu._ModelSet.from_layers(scale_layers)
    x = rand_plane(301, 423, 5)
    want = device_result(gpu, ms, x, nn2x)
    got = ms.convert_nn2x(x, opts=ho) if nn2x else ms.convert(x, opts=ho)
    assert np.array_equal(got, want)
    ref = orc.Oracle(scale_layers).convert(np.repeat(np.repeat(x, 2, 0), 2, 1) if nn2x else x, njob=8)
    assert_close(got, ref, "host pipeline")
    # again on the same model: the persistent pipe (streams, buffers, rings) is reused, with a larger and a smaller plane
    for (h, w) in ((350, 500), (40, 64)):
        y = rand_plane(h, w, h)
        assert np.array_equal(ms.convert(y, opts=ho), device_result(gpu, ms, y))


@pytest.mark.parametrize("planes", [[1, 32, 32, 64, 32, 1], [1, 32, 32, 64, 64, 1], [1, 32, 32, 64, 64, 128, 128, 1]])
def test_chunked_first_launch_waits_for_the_whole_block_it_touches(gpu, planes):
    """the fused first launch (layers 1 + 2) runs in row chunks under the upload of the source plane.  A chunk ends on a multiple of 8 local rows,
    which is a 4x4-block edge only when the region starts on one ((n - 2) & 3 == 0 ...): the block straddling the end reads source rows
    beyond the chunk's last row + 4, and the launch must wait for THEM -- otherwise its stored rows depend on what the device copy of the
    source held before (here: NaN from the previous call, which no cancellation removes).  host == resident bit for bit, small chunks."""
    ms = gpu._ModelSet.from_layers(small_layers(planes, 31 + len(planes)))
    assert ms.kernel_name(1) == "conv3x3_first2_wino4"
    poison = np.full((301, 423), np.nan, np.float32)
    for it in range(4):
        x = rand_plane(301, 423, 50 + it)
        want = device_result(gpu, ms, x)
        assert np.isfinite(want).all()
        for kb in (16, 48):
            ms.convert(poison, opts=gpu.make_opts(host_chunk_kb=kb))            # the pipe's device rows now hold NaN everywhere
            got = ms.convert(x, opts=gpu.make_opts(host_chunk_kb=kb))
            assert np.array_equal(got, want), (planes, it, kb, int(np.isnan(got).sum()))
            got = ms.convert_nn2x(x[:150, :211], opts=gpu.make_opts(host_chunk_kb=kb))
            assert np.array_equal(got, device_result(gpu, ms, x[:150, :211], True))


@pytest.mark.parametrize("precision", ["fp32", "fp16x2", "bf16", "direct"])
def test_host_pipeline_multi_band_and_every_last_layer_kernel(gpu, scale_layers, monkeypatch, precision):
    """several workspace bands (upload of band k+1 under band k) x chunked last layer, for every kernel the last layer can be:
    conv3x3_last, conv3x3_last_gather (16-bit modes: last layer fused into layer 6), conv3x3_direct"""
    ms = gpu._ModelSet.from_layers(scale_layers)
    kw = {"kernel": gpu.KERNEL_DIRECT} if precision == "direct" else \
         {"precision": {"fp32": gpu.PRECISION_FP32, "fp16x2": gpu.PRECISION_FP16X2, "bf16": gpu.PRECISION_BF16}[precision]}
    x = rand_plane(333, 260, 8)
    one = device_result(gpu, ms, x, **kw)
    # one band through the host pipeline: chunked last layer (fp32 / direct) or layer 6 + gather chunked together (16-bit modes)
    assert np.array_equal(ms.convert(x, opts=gpu.make_opts(host_chunk_kb=64, **kw)), one)
    assert np.array_equal(ms.convert_nn2x(x[:170, :131], opts=gpu.make_opts(host_chunk_kb=64, **kw)), device_result(gpu, ms, x[:170, :131], True, **kw))
    banded = ms.convert(x, opts=gpu.make_opts(band_rows=100, host_chunk_kb=64, **kw))
    assert np.array_equal(banded, device_result(gpu, ms, x, band_rows=100, **kw))
    if precision in ("fp32", "direct"):
        assert np.array_equal(banded, one)      # banding never changes fp32 results
    if precision == "direct":
        assert np.array_equal(banded, orc.Oracle(scale_layers).convert(x, njob=8))


def test_host_pipeline_pinned_and_strided_planes(gpu, noise1_layers):
    """planes that are already page-locked are DMA'd in place (no staging); strided ROIs of pageable and pinned planes"""
    import torch
    ms = gpu._ModelSet.from_layers(noise1_layers)
    h, w = 190, 333
    x = rand_plane(h, w, 3)
    want = ms.convert(x)
    pin_in = torch.from_numpy(x).pin_memory()
    out = np.empty((h, w), np.float32)
    lib = gpu.lib()
    import ctypes as C
    # pinned in, pageable out
    assert lib.w2xc_convert_plane(ms.handle, pin_in.data_ptr(), w * 4, w, h, out.ctypes.data, w * 4, 1, None) == 0
    assert np.array_equal(out, want)
    # pinned in, pinned out, both strided ROIs of larger pinned planes
    big_in = torch.zeros((h + 6, w + 10)).pin_memory()
    big_in[3:3 + h, 5:5 + w] = torch.from_numpy(x)
    big_out = torch.full((h + 4, w + 8), -7.0).pin_memory()
    roi_in, roi_out = big_in[3:3 + h, 5:5 + w], big_out[2:2 + h, 4:4 + w]
    assert lib.w2xc_convert_plane(ms.handle, roi_in.data_ptr(), (w + 10) * 4, w, h, roi_out.data_ptr(), (w + 8) * 4, 1, None) == 0
    assert np.array_equal(roi_out.numpy(), want)
    assert float(big_out[0, 0]) == -7.0 and float(big_out[-1, -1]) == -7.0 and float(big_out[2, 3]) == -7.0   # nothing outside the ROI
    # pageable strided ROI in and out
    pg_in = np.zeros((h + 6, w + 10), np.float32)
    pg_in[3:3 + h, 5:5 + w] = x
    pg_out = np.full((h + 4, w + 8), -7.0, np.float32)
    ri, ro = pg_in[3:3 + h, 5:5 + w], pg_out[2:2 + h, 4:4 + w]
    assert lib.w2xc_convert_plane(ms.handle, ri.ctypes.data, ri.strides[0], w, h, ro.ctypes.data, ro.strides[0], 1, None) == 0
    assert np.array_equal(ro, want) and pg_out[1].max() == -7.0 and pg_out[:, :4].max() == -7.0


def test_in_place_conversion_and_trim(gpu, scale_layers):
    """in == out (the reference never does it, main.cpp:94-96 copies first; a library must survive it): with several workspace
    bands the drainer writes band b's rows while band b+1's source rows have yet to be read -- unless every source row is staged
    first, which is what overlapping planes get.  Then w2xc_model_trim: buffers released, the next call re-grows them."""
    ms = gpu._ModelSet.from_layers(scale_layers)
    x = rand_plane(400, 300, 41)
    want = ms.convert(x)
    lib = gpu.lib()
    for band in (0, 64):
        buf = x.copy()
        o = gpu.make_opts(band_rows=band)
        import ctypes as C
        assert lib.w2xc_convert_plane(ms.handle, buf.ctypes.data, buf.strides[0], 300, 400, buf.ctypes.data, buf.strides[0], 1, C.byref(o)) == 0
        assert np.array_equal(buf, want), "band_rows=%d" % band
    # partially overlapping: output plane shifted by 5 rows inside the same allocation
    big = np.zeros((405, 300), np.float32)
    big[5:] = x
    o = gpu.make_opts(band_rows=64)
    assert lib.w2xc_convert_plane(ms.handle, big[5:].ctypes.data, big.strides[0], 300, 400, big.ctypes.data, big.strides[0], 1, C.byref(o)) == 0
    assert np.array_equal(big[:400], want)
    # in place with MORE THAN ONE unit (w2xc_opts.host_units: three units on this device, as three devices would run them): unit t's
    # output rows are the halo source rows of its neighbours -- the source rows are snapshotted before the units fan out
    o3 = gpu.make_opts(host_units=3)
    assert np.array_equal(ms.convert(x, opts=o3), want)           # (the units stitch to the one-unit result)
    buf = x.copy()
    assert lib.w2xc_convert_plane(ms.handle, buf.ctypes.data, buf.strides[0], 300, 400, buf.ctypes.data, buf.strides[0], 1, C.byref(o3)) == 0
    assert np.array_equal(buf, want), "in place, 3 units"
    big = np.zeros((405, 300), np.float32)
    big[5:] = x
    assert lib.w2xc_convert_plane(ms.handle, big[5:].ctypes.data, big.strides[0], 300, 400, big.ctypes.data, big.strides[0], 1, C.byref(o3)) == 0
    assert np.array_equal(big[:400], want), "overlapping, 3 units"
    src = x[::2].copy()                                  # nearest-2x entry in place is impossible (sizes differ): overlapping allocation
    both = np.zeros((500, 300), np.float32)              # source rows 0..99 of the view ARE output rows 300..399
    both[300:, :150] = src[:, :150]
    want2 = ms.convert_nn2x(np.ascontiguousarray(src[:, :150]))
    assert lib.w2xc_convert_plane_nn2x(ms.handle, both[300:].ctypes.data, both.strides[0], 150, 200, both.ctypes.data, both.strides[0], C.byref(o3)) == 0
    assert np.array_equal(both[:400], want2), "nn2x into an overlapping allocation, 3 units"
    ms.trim()
    assert np.array_equal(ms.convert(x), want)
    ms.trim()
    assert np.array_equal(ms.filter(0, x[None])[3], ms.filter(0, x[None], opts=gpu.make_opts(kernel=gpu.KERNEL_AUTO))[3])


@pytest.mark.parametrize("nn2x", [0, 1])
@pytest.mark.parametrize("parts", [2, 5])
def test_farm_units_from_host_memory(gpu, scale_layers, parts, nn2x):
    """w2xc_convert_plane_rows: every unit gets only ITS source rows (+ halo) and returns only its output rows; the units
    stitch to the whole conversion bit-exactly (this is what N ranks do, one unit each)"""
    ms = gpu._ModelSet.from_layers(scale_layers)
    h, w = 97, 140
    x = rand_plane(h, w, 17)
    whole = ms.convert_nn2x(x) if nn2x else ms.convert(x)
    H = h << nn2x
    n = ms.n_layers
    out = np.zeros_like(whole)
    for p in range(parts):
        ra, rb = gpu.shard_rows(H, parts, p)
        sy0, sy1 = max(0, ra - 4 * n) >> nn2x, (min(H, rb + 4 * n) + nn2x) >> nn2x   # the wide halo (4 n): bit-identity under the default F(4x4) kernel
        view = np.ascontiguousarray(x[sy0:sy1])          # the unit does not even see the other rows
        out[ra:rb] = ms.convert_rows(view, sy0, h, ra, rb, nn2x=nn2x)
    assert np.array_equal(out, whole)
    with pytest.raises(gpu.W2xcError) as e:               # a view without its halo rows is rejected
        ms.convert_rows(np.ascontiguousarray(x[50:60]), 50, h, 50 << nn2x, 60 << nn2x, nn2x=nn2x)
    assert e.value.code == gpu.ERR_ARG
    with pytest.raises(gpu.W2xcError):
        ms.convert_rows(x, 0, h, 10, 10, nn2x=nn2x)      # empty row range


def test_device_mask_over_every_visible_device(gpu, scale_layers):
    """w2xc_convert_plane with device_mask = all devices (one host thread + pipe per device, contiguous row shares, host
    gather); on a 1-GPU box this is the single-device path -- the N-device arithmetic is exercised by w2xc_opts.host_units
    (test_host_multi_band_path) and by the 2-rank bench launch below."""
    ms = gpu._ModelSet.from_layers(scale_layers)
    nd = gpu.device_count()
    x = rand_plane(257, 190, 23)
    got = ms.convert(x, opts=gpu.make_opts(device_mask=(1 << nd) - 1))
    assert_close(got, orc.Oracle(scale_layers).convert(x, njob=8), "device_mask all (%d devices)" % nd)
    assert np.array_equal(got, ms.convert(x, opts=gpu.make_opts(device_mask=1)))
    with pytest.raises(gpu.W2xcError) as e:
        ms.convert(x, opts=gpu.make_opts(device_mask=1 << 31))
    assert e.value.code == gpu.ERR_ARG


def test_jobs_are_the_staging_threads(gpu, scale_layers):
    """modelUtility::setNumberOfJobs maps to the host staging threads: any count gives the same planes"""
    ms = gpu._ModelSet.from_layers(scale_layers)
    x = rand_plane(600, 900, 29)       # 2 MB rows * 600: large enough for the copy pool to split the copies
    util = gpu.modelUtility.getInstance()
    prev = util.getNumberOfJobs()
    try:
        outs = []
        for j in (1, 3, 16):
            assert util.setNumberOfJobs(j)
            outs.append(ms.convert(x))
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    finally:
        util.setNumberOfJobs(prev)


# ---- Model::filter: persistent buffers, device-resident chain ----------------------------------------------------------
def test_filter_chain_resident_and_not(gpu):
    """the reference's test.cpp:72-85 pattern (filter() chained by hand).  filter_resident = 1 reuses the previous call's
    output still on the device when the SAME planes come back; without the flag every call uploads what it is given, so
    a caller that edits the planes in between is honoured.  Both equal the oracle chain."""
    layers = small_layers([3, 32, 64, 32, 2], 77)
    ms = gpu._ModelSet.from_layers(layers)
    o = orc.Oracle(layers)
    x = np.random.default_rng(12).random((3, 61, 83), dtype=np.float32)
    res = gpu.make_opts(filter_resident=1)
    a, b, c = x, x, x
    for l in range(4):
        a = ms.filter(l, a)                          # default: upload every time
        b = ms.filter(l, list(b), opts=res)          # the very arrays the previous call returned
        c = o.filter(l, c, njob=4)
        assert np.array_equal(a, b), "layer %d" % l
    assert_close(a, c, "filter chain")
    # the promise is about the planes handed back: different arrays (copies) are uploaded even with the flag set
    a1 = ms.filter(0, x, opts=res)
    edited = a1.copy()
    edited[0, 5, 5] += 1.0
    want = ms.filter(1, edited)
    assert np.array_equal(ms.filter(1, edited, opts=res), want)
    # and without the flag an in-place edit of the returned planes is seen
    a1 = ms.filter(0, x)
    a1[0, 5, 5] += 1.0
    assert np.array_equal(ms.filter(1, a1), want)
    # a second model between two calls does not confuse the first one's cache
    other = gpu._ModelSet.from_layers(small_layers([3, 32, 32], 5))
    p = ms.filter(0, x, opts=res)
    other.filter(0, x, opts=res)
    assert np.array_equal(ms.filter(1, list(p), opts=res), ms.filter(1, p.copy()))


def test_filter_device_nhwc_chain(gpu):
    """w2xc_layer_filter_device: planar device planes in / out (what Model::filter has), and an NHWC chain that hands the
    MFMA kernels their own layout from layer to layer (no repack) -- same numbers"""
    import torch
    layers = small_layers([3, 32, 64, 64, 3], 91)
    ms = gpu._ModelSet.from_layers(layers)
    h, w = 45, 70
    x = np.random.default_rng(3).random((3, h, w), dtype=np.float32)
    want = x
    for l in range(4):
        want = ms.filter(l, want)
    st = torch.cuda.current_stream()
    o = gpu.make_opts(device=0)
    # planar chain
    cur = torch.from_numpy(x).cuda()
    for l in range(4):
        nin, nout = ms.planes(l)
        nxt = torch.empty((nout, h, w), device="cuda")
        ms.filter_device(l, nin, cur.data_ptr(), (h * w, w, 1), w, h, nxt.data_ptr(), (h * w, w, 1), stream=st.cuda_stream, opts=o)
        cur = nxt
    st.synchronize()
    assert np.array_equal(cur.cpu().numpy(), want)
    # NHWC chain: layer 1 reads planar, everything in between is NHWC, the last layer writes planar
    cur, strides = torch.from_numpy(x).cuda(), (h * w, w, 1)
    for l in range(4):
        nin, nout = ms.planes(l)
        if l < 3:
            nxt, ostr = torch.empty((h, w, nout), device="cuda"), (1, w * nout, nout)
        else:
            nxt, ostr = torch.empty((nout, h, w), device="cuda"), (h * w, w, 1)
        ms.filter_device(l, nin, cur.data_ptr(), strides, w, h, nxt.data_ptr(), ostr, stream=st.cuda_stream, opts=o)
        cur, strides = nxt, ostr
    st.synchronize()
    assert np.array_equal(cur.cpu().numpy(), want)
    with pytest.raises(gpu.W2xcError) as e:
        ms.filter_device(1, 3, cur.data_ptr(), (h * w, w, 1), w, h, cur.data_ptr(), (h * w, w, 1), opts=o)
    assert e.value.code == gpu.ERR_PLANES


@pytest.mark.parametrize("w", [70, 69, 71, 72])
def test_filter_device_roi_of_a_wider_tensor_keeps_the_neighbours(gpu, w):
    """w2xc_layer_filter_device on an ROI: the output planes are a w-column sub-view of wider rows (row stride a multiple of 4, aligned base --
    everything conv3x3_wino4's planar epilogue could store straight into).  Its stores are whole 16-byte pixel quads, so with w % 4 != 0 the last
    quad of a row reaches into columns [w, roundup4(w)): those belong to the caller and must come back untouched (the engine takes the direct
    path only when w % 4 == 0), and the ROI itself equals the contiguous call bit for bit."""
    import torch
    layers = small_layers([64, 64], 17)
    ms = gpu._ModelSet.from_layers(layers)
    assert ms.kernel_name(0) == "conv3x3_wino4"
    h, W = 37, 96
    x = np.random.default_rng(5).random((64, h, w), dtype=np.float32)
    want = ms.filter(0, x)
    st = torch.cuda.current_stream()
    o = gpu.make_opts(device=0)
    d_in = torch.from_numpy(x).cuda()
    wide = torch.full((64, h, W), -7.25, device="cuda")
    ms.filter_device(0, 64, d_in.data_ptr(), (h * w, w, 1), w, h, wide.data_ptr(), (h * W, W, 1), stream=st.cuda_stream, opts=o)
    st.synchronize()
    got = wide.cpu().numpy()
    assert np.array_equal(got[:, :, :w], want)
    assert np.all(got[:, :, w:] == np.float32(-7.25)), "columns outside the ROI were written"


# ---- bench.py: the lines the driver records ------------------------------------------------------------------------------
def run_bench(args, env=None, nproc=1, timeout=600):
    e = dict(os.environ, **(env or {}))
    if nproc == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", "29531", os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + args
    r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_single_gpu(gpu):
    """N = 1: `value` = resident planes (bench contract), host->host beside it with its ratio, roofline from the profiled pass"""
    j = run_bench(["--height", "270", "--width", "480", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras"])
    assert j["n_gpus"] == 1 and j["scaling"] == "weak" and j["dtype"] == "f32" and j["output_finite"]
    assert j["value"] == j["value_resident"] > 0 and j["value_host_to_host"] > 0
    h = j["host_to_host"]
    assert h["max_abs_diff_vs_resident_output"] == 0.0
    for k in ("pageable", "pinned"):
        assert 0 < h[k]["ratio_vs_resident"] < 1.2
    r = j["roofline"]
    # `achieved` / `frac` = the FLOPs the kernel ISSUES over time (a fraction of the MFMA peak, < 1); the algorithmic rate (SURVEY 8d's
    # FLOPs over the same time) sits beside it and may pass the peak with a Winograd kernel on the dominant layer
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0 < r["frac"] < 1 and abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-3
    assert r["algorithmic_tflops"] >= r["achieved"] - 1e-6 and abs(r["algorithmic_tflops"] / r["peak"] - r["algorithmic_speedup_vs_direct_roofline"]) < 1e-3
    assert abs((r["flops_per_launch"] - r["fused_last_layer_flops_per_launch"]) / r["algorithmic_flops_per_launch"] -
               (36 / 144 if "conv3x3_wino4" in r["kernel"] else 16 / 36 if "wino" in r["kernel"] else 1.0)) < 1e-9   # F(4x4,3x3) / F(2x2,3x3) / direct
    assert r["fused_last_layer_flops_per_launch"] == (0 if j["layers"][6]["kernel"] == "conv3x3_last" else r["fused_last_layer_flops_per_launch"]) >= 0
    assert all(0 < l["frac_of_peak"] < 1 for l in j["layers"] if l["frac_of_peak"] is not None) and sum(l["frac_of_peak"] is None for l in j["layers"]) <= 2   # (layer 1 inside layer 2's launch, the last layer inside layer 6's)
    assert ("conv3x3_wino" in r["kernel"] or "conv3x3_mfma" in r["kernel"]) and "128->128" in r["kernel"]
    assert len(j["layers"]) == 7 and "workload" in j["config"]
    assert abs(r["frac"] - j["layers"][5]["frac_of_peak"]) < 2e-4   # one accounting for the dominant launch in both places
    assert j["parity_patch_max_rel_err"] is None   # (--no-cpu-baseline: the oracle is not touched)


def test_bench_rccl_process_group_one_rank(gpu):
    """the N > 1 code of bench.py on the REAL backend ("nccl" = RCCL): a one-rank process group under torch.distributed.run, so that the
    RCCL initialisation (dmabuf IPC: HSA_ENABLE_IPC_MODE_LEGACY=0), the barrier, the MAX all_reduce and the all_gather of the per-rank
    figures have run on this image before a multi-GPU node sees them (two ranks cannot share a device under RCCL: that is the gloo test)"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "plane", "--height", "96", "--width", "128", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-extras"]   # (the row-sharded workload of N > 1, here with one shard)
    r = subprocess.run(cmd, env=dict(os.environ, W2XC_BENCH_FORCE_PG="1", HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["ranks_seen"] == 1 and j["backend"] == "nccl" and len(j["rank_ms_per_step"]) == 1 and j["value"] > 0 and j["output_finite"]
    assert j["scaling"] == "strong" and "row-sharded" in j["config"]["workload"] and j["host_to_host"]["max_abs_diff_vs_resident_output"] == 0.0


def test_bench_two_ranks_shard_one_plane_on_the_hip_path(gpu, tmp_path):
    """N = 2 (two ranks on this GPU, gloo for the barrier): the default workload is ONE frame row-sharded over the ranks on
    the HIP path -- strong scaling, the weak figure beside it -- and the rows the ranks produced stitch to the oracle's plane"""
    from bench import synth_luma, nn2x
    d = str(tmp_path / "dump")
    j = run_bench(["--height", "96", "--width", "128", "--steps", "2", "--warmup", "1", "--dump-out", d],
                  env={"W2XC_BENCH_BACKEND": "gloo"}, nproc=2)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["frames_per_step"] == 1
    assert j["ranks_seen"] == 2 and len(j["rank_ms_per_step"]) == 2 and len(j["rank_devices"]) == 2 and all(v > 0 for v in j["rank_ms_per_step"])
    assert "row-sharded" in j["config"]["workload"] and j["value"] > 0 and j["value_host_to_host"] > 0
    assert j["host_to_host"]["max_abs_diff_vs_resident_output"] == 0.0
    assert j["weak"]["scaling"] == "weak" and j["weak"]["value"] > 0
    parts = [np.load(os.path.join(d, "rank%d.npz" % r)) for r in range(2)]
    assert int(parts[0]["ra"]) == 0 and int(parts[0]["rb"]) == int(parts[1]["ra"]) == 96 and int(parts[1]["rb"]) == 192
    got = np.concatenate([p["rows"] for p in parts])
    layers = gen_model.synth_layers(seed=gen_model.SEEDS["scale2.0x"])
    want = orc.Oracle(layers).convert(nn2x(synth_luma(2, 96, 128)), njob=8)
    assert_close(got, want, "2-rank sharded plane")


def test_bench_plain_invocation_gpus2(gpu):
    """`python bench.py --gpus 2` WITHOUT torch.distributed.run (the shape of the driver's N = 1 command with another N): bench.py re-executes
    itself under the launcher, one rank per GPU -- on this one-GPU box the two ranks share the device, so it picks gloo for the barrier by
    itself -- and prints the one N = 2 line.  A first multi-GPU run cannot fail for a launcher reason."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "W2XC_BENCH_BACKEND")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--height", "96", "--width", "128", "--steps", "2", "--warmup", "1", "--no-extras"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["scaling"] == "strong" and len(j["rank_ms_per_step"]) == 2 and j["value"] > 0
    import torch
    assert j["backend"] == ("nccl" if torch.cuda.device_count() >= 2 else "gloo")
    assert j["host_to_host"]["max_abs_diff_vs_resident_output"] == 0.0 and j["output_finite"]
    # the timed plane against the oracle, in the line itself
    assert j["parity_patch_within_gate"] is True and 0 <= j["parity_patch_max_rel_err"] <= 1e-4


def test_bench_eight_ranks_shard_one_plane_bit_identical_to_one_rank(gpu, tmp_path):
    """N = 8 without an 8-GPU node: eight ranks (gloo for the barrier) share this GPU and row-shard ONE plane exactly as `bench.py --gpus 8` does on a
    node -- every rank its rows plus the 28-row halo conv3x3_wino4's band geometry wants (4 rows per layer) -- and the eight shards, stitched, are
    BIT-identical to the plane one rank computes alone.  The record's self-checks (ranks_seen, per-rank times, devices) hold for N = 8."""
    d8, d1 = str(tmp_path / "d8"), str(tmp_path / "d1")
    args = ["--workload", "plane", "--height", "1024", "--width", "768", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extras"]
    j8 = run_bench(args + ["--dump-out", d8], env={"W2XC_BENCH_BACKEND": "gloo"}, nproc=8, timeout=1500)
    assert j8["n_gpus"] == 8 and j8["ranks_seen"] == 8 and j8["scaling"] == "strong" and j8["config"]["frames_per_step"] == 1
    assert len(j8["rank_ms_per_step"]) == 8 and all(v > 0 for v in j8["rank_ms_per_step"]) and len(j8["rank_devices"]) == 8
    assert j8["host_to_host"]["max_abs_diff_vs_resident_output"] == 0.0 and j8["output_finite"]
    parts = [np.load(os.path.join(d8, "rank%d.npz" % r)) for r in range(8)]
    assert int(parts[0]["ra"]) == 0 and int(parts[7]["rb"]) == 2048 and all(int(parts[r]["rb"]) == int(parts[r + 1]["ra"]) for r in range(7))
    got = np.concatenate([p["rows"] for p in parts])
    j1 = run_bench(args + ["--dump-out", d1])
    one = np.load(os.path.join(d1, "rank0.npz"))
    assert int(one["ra"]) == 0 and int(one["rb"]) == 2048 and j1["ranks_seen"] == 1
    assert np.array_equal(got, one["rows"]), "8 row shards != the plane of one rank (max |diff| %g)" % float(np.abs(got - one["rows"]).max())
    h = j8["host_to_host"]
    print("8 ranks on one device, 2048x1536 plane: resident %.2f ms per step, host gather %s" % (j8["ms_per_step"], {k: v for k, v in h.items() if isinstance(v, dict)}))
