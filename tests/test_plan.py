"""CPU unit tests of the host arithmetic that decides WHERE the band loop cuts a plane: w2xc_plan_rows / w2xc_plan_region
(csrc/w2xc_select.cpp: plan_rows, RowPlan::region, the workspace solver) and the shard split of the multi-GPU farm
(shard_rows / shard_view).  No device is needed: the plan is what w2xc_convert_rows_device would run with.

The block walk being replaced: /root/reference/src/convertRoutine.cpp:84-169 (512-pixel blocks at stride 512 - 2 nModel,
each padded by nModel); the invariants here are the ones that make a band split invisible in the result."""
import ctypes as C

import numpy as np
import pytest

from tools import gen_model

SCALE = [1, 32, 32, 64, 64, 128, 128, 1]


@pytest.fixture(scope="module")
def ms(w2xc):
    return w2xc._ModelSet.from_layers(gen_model.synth_layers(SCALE, 102))


def bands_of(plan, ra, rb):
    return [(y0, min(rb, y0 + plan.band_rows)) for y0 in range(ra, rb, plan.band_rows)]


def test_whole_frame_is_one_band_with_the_wide_geometry(w2xc, ms):
    p = ms.plan_rows(3840, 2160)
    assert (p.n_layers, p.halo_rows_per_layer, p.band_rows, p.n_bands) == (7, 4, 2160, 1)
    assert p.fused_first == 1 and p.fused_last == 1
    # layer 1's activations never reach the workspace (fused first layers); layer 6 writes 2 x 9 tap planes
    assert p.workspace_bytes[0] > 0 and p.workspace_bytes[1] > 0
    assert p.workspace_bytes[0] + p.workspace_bytes[1] < 16384 << 20


def test_fusion_switches_show_in_the_plan(w2xc, ms):
    for fusion, want in ((w2xc.FUSION_AUTO, (1, 1)), (w2xc.FUSION_ON, (1, 1)), (w2xc.FUSION_OFF, (0, 0)),
                         (w2xc.FUSION_FIRST, (1, 0)), (w2xc.FUSION_LAST, (0, 1))):
        for prec in (w2xc.PRECISION_FP32, w2xc.PRECISION_BF16X3, w2xc.PRECISION_BF16):
            p = ms.plan_rows(640, 480, opts=w2xc.make_opts(fusion=fusion, precision=prec))
            assert (p.fused_first, p.fused_last) == want, (fusion, prec)
    with pytest.raises(w2xc.W2xcError) as e:
        ms.plan_rows(640, 480, opts=w2xc.make_opts(fusion=9))
    assert e.value.code == w2xc.ERR_ARG
    # W2XC_KERNEL_DIRECT runs every layer alone whatever `fusion` says; the F(2x2) kernels fuse nothing on the fp32 path
    for k in (w2xc.KERNEL_DIRECT, w2xc.KERNEL_WINOGRAD32, w2xc.KERNEL_MFMA):
        p = ms.plan_rows(640, 480, opts=w2xc.make_opts(kernel=k))
        assert (p.fused_first, p.fused_last, p.halo_rows_per_layer) == (0, 0, 1)


@pytest.mark.parametrize("budget_mb", [64, 256, 1024])
def test_banded_plan_fits_its_budget_and_tiles_the_rows(w2xc, ms, budget_mb):
    H, W = 4000, 3000
    p = ms.plan_rows(W, H, opts=w2xc.make_opts(workspace_mb=budget_mb))
    assert p.n_bands > 1
    assert p.workspace_bytes[0] + p.workspace_bytes[1] <= budget_mb << 20
    assert p.band_rows % 4 == 0                                    # band edges on 4x4-block rows: nothing is rounded out twice
    bands = bands_of(p, 0, H)
    assert len(bands) == p.n_bands and bands[0][0] == 0 and bands[-1][1] == H
    assert all(a[1] == b[0] for a, b in zip(bands, bands[1:]))     # no gap, no overlap
    # one more row per band would not fit (the solver does not leave half the budget unused) -- within the equalisation of band heights
    q = ms.plan_rows(W, H, opts=w2xc.make_opts(workspace_mb=budget_mb, band_rows=p.band_rows * 2))
    assert q.workspace_bytes[0] + q.workspace_bytes[1] > budget_mb << 20


def test_regions_nest_and_end_on_block_rows(w2xc, ms):
    """layer k + 1 reads one row above and below what it computes: layer k's region must hold them; with the wide geometry every
    region edge that is not the plane's padded edge is a multiple of 4 (a 4x4-block edge of conv3x3_wino4)"""
    H, W = 1000, 200
    n = 7
    p = ms.plan_rows(W, H, opts=w2xc.make_opts(band_rows=96))
    assert p.halo_rows_per_layer == 4
    for (y0, y1) in bands_of(p, 0, H):
        prev = None
        for k in range(1, n + 1):
            t, b = ms.plan_region(p, H, k, y0, y1)
            lo, hi = -(n - k), H + (n - k)                          # the layer's whole extent (the padded plane seen from layer k)
            assert lo <= t < b <= hi
            if k < n:
                assert t == lo or t % 4 == 0
                assert b == hi or b % 4 == 0
            else:
                assert (t, b) == (y0, y1)                           # the last layer computes exactly the band
            if prev is not None:
                assert prev[0] <= t - 1 and prev[1] >= b + 1        # the rows layer k reads exist in layer k - 1's region
            prev = (t, b)
        t1, b1 = ms.plan_region(p, H, 1, y0, y1)
        assert t1 - 1 >= -n and b1 + 1 <= H + n                     # layer 1 reads the padded source (pad = n, convertRoutine.cpp:35)


def test_regions_of_two_bandings_agree_on_block_rows(w2xc, ms):
    """what makes bandings bit-identical: a row of layer k is computed inside the same 4x4 block whichever band computes it -- block rows
    are multiples of 4 of the PLANE in every banding, so two bandings can only differ in which band owns a block, never in its content"""
    H = 777
    p1 = ms.plan_rows(160, H, opts=w2xc.make_opts(band_rows=64))
    p2 = ms.plan_rows(160, H, opts=w2xc.make_opts(band_rows=200))
    for p in (p1, p2):
        for (y0, y1) in bands_of(p, 0, H):
            for k in range(1, 7):
                t, b = ms.plan_region(p, H, k, y0, y1)
                assert (t % 4 == 0 or t == -(7 - k)) and (b % 4 == 0 or b == H + 7 - k)


def test_minimum_view_refuses_auto_and_runs_named_kernels_on_one_row_per_layer(w2xc, ms):
    H, W, n = 600, 128, 7
    ra, rb = w2xc.shard_rows(H, 3, 1)
    y0, y1 = w2xc.shard_view(H, ra, rb, n)
    with pytest.raises(w2xc.W2xcError) as e:
        ms.plan_rows(W, H, ra, rb, y0, y1 - y0)
    assert e.value.code == w2xc.ERR_ARG and "halo rows" in str(e.value)
    p = ms.plan_rows(W, H, ra, rb, y0, y1 - y0, opts=w2xc.make_opts(kernel=w2xc.KERNEL_WINOGRAD32))
    assert p.halo_rows_per_layer == 1
    for k in range(1, n + 1):
        assert ms.plan_region(p, H, k, ra, rb) == (ra - (n - k), rb + (n - k))
    wy0, wy1 = w2xc.shard_view(H, ra, rb, 4 * n)
    assert ms.plan_rows(W, H, ra, rb, wy0, wy1 - wy0).halo_rows_per_layer == 4
    with pytest.raises(w2xc.W2xcError):                            # a view that does not even hold the minimum halo
        ms.plan_rows(W, H, ra, rb, ra, rb - ra)


@pytest.mark.parametrize("H,parts", [(2160, 8), (16384, 8), (1081, 7), (17, 5), (8, 8)])
def test_shard_rows_partition_and_views(w2xc, H, parts):
    """the farm's split (convertRoutine.cpp:114-165 made parallel): contiguous, exhaustive, sizes within one row of each other;
    a shard's view is its rows + halo clipped to the plane"""
    cuts = [w2xc.shard_rows(H, parts, p) for p in range(parts)]
    assert cuts[0][0] == 0 and cuts[-1][1] == H
    assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
    sizes = [b - a for a, b in cuts]
    assert max(sizes) - min(sizes) <= 1 and min(sizes) >= 1
    for (a, b) in cuts:
        for halo in (7, 28):
            y0, y1 = w2xc.shard_view(H, a, b, halo)
            assert y0 == max(0, a - halo) and y1 == min(H, b + halo)


def test_plan_of_every_shard_of_the_bench_plane(w2xc, ms):
    """bench.py --gpus N cuts the 16384-row plane of configs[2] into N row shards with the wide halo: every shard plans with the
    banding-invariant geometry and bands that tile its rows"""
    H, W, n = 16384, 16384, 7
    for parts in (2, 4, 8):
        for part in range(parts):
            ra, rb = w2xc.shard_rows(H, parts, part)
            y0, y1 = w2xc.shard_view(H, ra, rb, 4 * n)
            p = ms.plan_rows(W, H, ra, rb, y0, y1 - y0)
            assert p.halo_rows_per_layer == 4 and p.fused_first == 1 and p.fused_last == 1
            bands = bands_of(p, ra, rb)
            assert len(bands) == p.n_bands and bands[-1][1] == rb
            assert p.workspace_bytes[0] + p.workspace_bytes[1] <= 16384 << 20


def test_plane_count_mismatch_and_bad_ranges(w2xc, ms):
    three = w2xc._ModelSet.from_layers(gen_model.synth_layers([3, 32, 3], 5))
    p = three.plan_rows(64, 64)
    assert p.n_layers == 2
    broken = w2xc._ModelSet.from_layers(gen_model.synth_layers([1, 32], 5) + gen_model.synth_layers([64, 1], 6))
    with pytest.raises(w2xc.W2xcError) as e:
        broken.plan_rows(64, 64)
    assert e.value.code == w2xc.ERR_PLANES
    for bad in ((0, 0), (10, 5), (-1, 4), (0, 65)):
        with pytest.raises(w2xc.W2xcError):
            ms.plan_rows(64, 64, bad[0], bad[1])


def test_opts_init_sized_writes_only_the_callers_prefix(w2xc):
    """a binary compiled against an older, shorter w2xc_opts (44 bytes in rounds 3 / 4) must not have bytes written behind its struct"""
    lib = w2xc.lib()
    buf = (C.c_ubyte * 96)(*([0xEE] * 96))
    lib.w2xc_opts_init_sized(C.cast(buf, C.POINTER(w2xc.Opts)), 44)
    assert bytes(buf[44:]) == b"\xee" * 52
    o = C.cast(buf, C.POINTER(w2xc.Opts)).contents
    assert (o.struct_size, o.precision, o.kernel, o.device, o.fusion) == (44, 0, 0, -1, 0)
    lib.w2xc_opts_init_sized(C.cast(buf, C.POINTER(w2xc.Opts)), 4096)           # a future, larger caller: the library's own size
    assert C.cast(buf, C.POINTER(w2xc.Opts)).contents.struct_size == C.sizeof(w2xc.Opts)
    assert bytes(buf[C.sizeof(w2xc.Opts):]) == b"\xee" * (96 - C.sizeof(w2xc.Opts))
