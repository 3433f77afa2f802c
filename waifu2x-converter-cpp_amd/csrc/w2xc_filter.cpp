// w2xc_filter.cpp -- Model::filter (src/modelHandler.cpp:26-72,117-159) at the host and the device boundary: ONE layer,
// same-size conv with BORDER_REPLICATE, bias, LeakyReLU, on the planes the caller hands over.
#include "w2xc_engine.hpp"
#include <iostream>

#include "w2xc_copy_pool.hpp"

namespace w2xc_eng {

namespace {

int grow(float **buf, size_t *have, size_t want)
{
    if (*have >= want) return W2XC_OK;
    if (*buf) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(*buf)); *buf = nullptr; *have = 0; }
    if (hipMalloc((void **)buf, want * sizeof(float)) != hipSuccess) return fail(W2XC_ERR_NOMEM, "hipMalloc(%zu MiB) for Model::filter planes failed", (want * 4) >> 20);
    *have = want;
    return W2XC_OK;
}

bool nhwc_ok(const float *p, long long cs, long long rs, long long ps, int planes)
{
    return cs == 1 && ps == planes && (rs & 3) == 0 && (((size_t)p) & 15) == 0;
}

// One Model::filter layer (same-size conv, BORDER_REPLICATE, bias, LeakyReLU; modelHandler.cpp:117-159) on DEVICE data
// with arbitrary element strides (floats): element (plane c, row y, pixel x) at base[c*cs + y*rs + x*ps].  The MFMA
// kernels want NHWC (cs = 1, ps = planes); other layouts are repacked through the context's NHWC buffers
// nhwc[ob ^ 1] (input) / nhwc[ob] (output).  *res_nhwc tells whether an NHWC copy of the result was left in nhwc[ob].
int filter_on_device(w2xc_model *m, DevCtx *c, int layer, const float *in, long long in_cs, long long in_rs, long long in_ps, int w, int h,
                     float *out, long long out_cs, long long out_rs, long long out_ps, hipStream_t st, const w2xc_opts &o_, int ob, bool *res_nhwc)
{
    const HostLayer &hl = m->layers[layer];
    FilterCache &fc = c->fc;
    const size_t px = (size_t)w * h;
    w2xc_opts of = o_;          // Model::filter runs ONE layer: nothing to fuse it with (and no profiling events)
    of.fusion = W2XC_FUSION_OFF;
    of.profile = 0;
    const w2xc_opts &o = of;
    const W2xcKernelKind kind = layer_kind(m, layer, o);
    const bool want_nhwc_in = (kind == W2XC_K_MFMA || kind == W2XC_K_LAST);
    const bool writes_nhwc = (kind == W2XC_K_MFMA || kind == W2XC_K_FIRST);
    W2xcConvDesc d;
    memset(&d, 0, sizeof d);
    d.in_h = d.out_h = h;
    d.in_w = d.out_w = w;
    d.off_y = d.off_x = -1;   // same-size conv, BORDER_REPLICATE via clamped loads (:141-142)
    if (is_wino4_layer(m, layer, o)) {
        // conv3x3_wino4 runs a valid conv on planar planes with 16-byte aligned pixel quads: the replicate border (:141-142) is made explicit in a
        // padded planar copy of the input (one pass over Cin planes; the kernel then reads it with offset 0)
        const long long prs = ((long long)w + 2 + 31) & ~31ll, pcs = prs * (h + 2);
        int rc = grow(&fc.pad, &fc.pad_floats, (size_t)pcs * hl.nin);
        if (rc) return rc;
        HIP_TRY(w2xc_launch_pad_planar(in, in_rs, in_ps, in_cs, fc.pad, prs, pcs, h, w, hl.nin, 1, st));
        d.in = fc.pad; d.in_rs = prs; d.in_ps = 1; d.in_cs = pcs;
        d.in_h = h + 2; d.in_w = w + 2;
        d.off_y = d.off_x = 0;
        const long long ors = ((long long)w + 31) & ~31ll;
        // (the planar epilogue stores whole 16-byte pixel quads: straight into the caller's planes only when w is a multiple of 4 -- with a ragged
        //  last quad it would write up to three floats past column w - 1 of every row, which in a sub-view of a wider tensor are the caller's)
        const bool planar_direct = out_ps == 1 && (w & 3) == 0 && (out_rs & 3) == 0 && (out_cs & 3) == 0 && (((size_t)out) & 15) == 0 && out_rs >= (long long)w;
        const bool nhwc_direct = nhwc_ok(out, out_cs, out_rs, out_ps, hl.nout);
        if (planar_direct || nhwc_direct) {
            d.out = out; d.out_rs = out_rs; d.out_ps = out_ps; d.out_cs = out_cs;
        } else {
            rc = grow(&fc.pout, &fc.pout_floats, (size_t)ors * h * hl.nout);
            if (rc) return rc;
            d.out = fc.pout; d.out_rs = ors; d.out_ps = 1; d.out_cs = ors * h;
        }
        int r = launch_layer(c, m, layer, kind, d, st, of);
        if (r) return r;
        if (!(planar_direct || nhwc_direct)) HIP_TRY(w2xc_launch_repack(d.out, d.out_rs, 1, d.out_cs, out, out_rs, out_ps, out_cs, h, w, hl.nout, st));
        if (res_nhwc) *res_nhwc = false;
        return W2XC_OK;
    }
    if (want_nhwc_in && !nhwc_ok(in, in_cs, in_rs, in_ps, hl.nin)) {
        int rc = grow(&fc.nhwc[ob ^ 1], &fc.nhwc_floats[ob ^ 1], px * hl.nin);
        if (rc) return rc;
        HIP_TRY(w2xc_launch_repack(in, in_rs, in_ps, in_cs, fc.nhwc[ob ^ 1], (long long)w * hl.nin, hl.nin, 1, h, w, hl.nin, st));
        d.in = fc.nhwc[ob ^ 1]; d.in_rs = (long long)w * hl.nin; d.in_ps = hl.nin; d.in_cs = 1;
    } else {
        d.in = in; d.in_rs = in_rs; d.in_ps = in_ps; d.in_cs = in_cs;
    }
    const bool direct_out = !writes_nhwc || nhwc_ok(out, out_cs, out_rs, out_ps, hl.nout);
    if (direct_out) {
        d.out = out; d.out_rs = out_rs; d.out_ps = out_ps; d.out_cs = out_cs;
    } else {
        int rc = grow(&fc.nhwc[ob], &fc.nhwc_floats[ob], px * hl.nout);
        if (rc) return rc;
        d.out = fc.nhwc[ob]; d.out_rs = (long long)w * hl.nout; d.out_ps = hl.nout; d.out_cs = 1;
    }
    int r = launch_layer(c, m, layer, kind, d, st, of);
    if (r) return r;
    if (!direct_out) HIP_TRY(w2xc_launch_repack(d.out, d.out_rs, d.out_ps, 1, out, out_rs, out_ps, out_cs, h, w, hl.nout, st));
    if (res_nhwc) *res_nhwc = !direct_out;
    return W2XC_OK;
}

int filter_check(const w2xc_model *m, int layer, int n_in_planes)
{
    if (!m || layer < 0 || layer >= (int)m->layers.size()) return fail(W2XC_ERR_ARG, "bad model/layer");
    const HostLayer &hl = m->layers[layer];
    if (n_in_planes != hl.nin) {   // modelHandler.cpp:29-35
        std::cerr << "Error : Model-filter : \nnumber of input planes mismatch." << std::endl;
        std::cerr << n_in_planes << "," << hl.nin << std::endl;
        return fail(W2XC_ERR_PLANES, "Error : Model-filter : \nnumber of input planes mismatch.\n%d,%d", n_in_planes, hl.nin);
    }
    return W2XC_OK;
}

}  // namespace

}  // namespace w2xc_eng

using namespace w2xc_eng;

extern "C" {

int w2xc_layer_filter_device(w2xc_model *m, int layer, int n_in_planes, const float *d_in, long long in_plane_stride, long long in_row_stride,
                             long long in_pixel_stride, int w, int h, float *d_out, long long out_plane_stride, long long out_row_stride,
                             long long out_pixel_stride, void *hip_stream, const w2xc_opts *opts)
try {
    int rc = filter_check(m, layer, n_in_planes);
    if (rc) return rc;
    if (!d_in || !d_out || w <= 0 || h <= 0) return fail(W2XC_ERR_ARG, "bad argument");
    const w2xc_opts o = resolve_opts(opts);
    if (o.precision != W2XC_PRECISION_FP32) return fail(W2XC_ERR_UNSUPPORTED, "w2xc_layer_filter* is fp32 only (16-bit activations exist only between layers of w2xc_convert_*)");
    int dev = o.device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    DeviceGuard guard(dev);
    if (!guard.ok) return fail(W2XC_ERR_HIP, "cannot select HIP device %d", dev);
    DevCtx *c = nullptr;
    if ((rc = get_ctx(m, dev, &c))) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    c->fc.res_valid = false;   // the NHWC scratch buffers are about to be reused
    return filter_on_device(m, c, layer, d_in, in_plane_stride, in_row_stride, in_pixel_stride, w, h, d_out, out_plane_stride, out_row_stride,
                            out_pixel_stride, (hipStream_t)hip_stream, o, c->fc.ob, nullptr);
} W2XC_CATCH_ALL

int w2xc_layer_filter(w2xc_model *m, int layer, int n_in_planes, const float *const *in_planes, size_t in_stride_bytes,
                      int w, int h, float *const *out_planes, size_t out_stride_bytes, const w2xc_opts *opts)
try {
    int rc = filter_check(m, layer, n_in_planes);
    if (rc) return rc;
    const HostLayer &hl = m->layers[layer];
    if (!in_planes || !out_planes || w <= 0 || h <= 0) return fail(W2XC_ERR_ARG, "bad argument");
    if (in_stride_bytes < (size_t)w * 4 || out_stride_bytes < (size_t)w * 4) return fail(W2XC_ERR_ARG, "bad stride");
    const w2xc_opts o = resolve_opts(opts);
    if (o.precision != W2XC_PRECISION_FP32)   // Model::filter hands fp32 planes in and out of EVERY layer
        return fail(W2XC_ERR_UNSUPPORTED, "w2xc_layer_filter is fp32 only (bf16 activations exist only between layers of w2xc_convert_*)");
    if (w2xc_device_count() <= 0) return fail(W2XC_ERR_HIP, "no HIP device available (libw2xc_hip has no CPU fallback)");
    int dev = o.device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    DeviceGuard guard(dev);
    if (!guard.ok) return fail(W2XC_ERR_HIP, "cannot select HIP device %d", dev);
    DevCtx *c = nullptr;
    if ((rc = get_ctx(m, dev, &c))) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    FilterCache &fc = c->fc;
    if (!fc.st) {
        HIP_TRY(hipStreamCreateWithFlags(&fc.st, hipStreamNonBlocking));
        for (auto &e : fc.ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const size_t px = (size_t)w * h, row = (size_t)w * 4;
    const size_t want_slot = std::max<size_t>((size_t)8 << 20, row);
    if (fc.slot_bytes < want_slot) {
        HIP_TRY(hipStreamSynchronize(fc.st));
        if (fc.pin) { HIP_TRY(hipHostFree(fc.pin)); fc.pin = nullptr; fc.slot_bytes = 0; }
        if (hipHostMalloc((void **)&fc.pin, want_slot * FilterCache::SLOTS, hipHostMallocDefault) != hipSuccess)
            return fail(W2XC_ERR_NOMEM, "hipHostMalloc of the Model::filter bounce ring failed");
        fc.slot_bytes = want_slot;
    }
    const int copy_threads = std::max(1, std::min(w2xc_get_jobs(), 32));
    const int rows_per_slot = (int)(fc.slot_bytes / row);

    // filter_resident: the planes handed in are exactly the planes the previous filter() call on this model wrote (same
    // pointers, count, size) and the caller has not touched them since -- its result is still on the device
    bool resident = o.filter_resident && fc.res_valid && fc.res_planes == hl.nin && fc.res_w == w && fc.res_h == h &&
                    fc.res_stride == in_stride_bytes && (int)fc.res_host.size() == hl.nin;
    for (int i = 0; resident && i < hl.nin; i++) resident = fc.res_host[i] == in_planes[i];
    const int ob = fc.ob ^ 1;   // this call writes buffers [ob]; the previous result sits in [ob ^ 1]
    fc.res_valid = false;

    // planes x rows as one index space g = plane * h + r, moved in slot-sized runs through the pinned ring
    long seq = 0;
    auto for_runs = [&](int planes, const std::function<int(long, long, char *, int)> &fn) -> int {
        const long total = (long)planes * h;
        for (long g0 = 0; g0 < total; g0 += rows_per_slot, seq++) {
            const long g1 = std::min(total, g0 + rows_per_slot);
            const int si = (int)(seq % FilterCache::SLOTS);
            int r = fn(g0, g1, fc.pin + (size_t)si * fc.slot_bytes, si);
            if (r) return r;
        }
        return W2XC_OK;
    };
    auto host_rows = [&](bool to_slot, char *slot, long g0, long g1, const float *const *src_planes, float *const *dst_planes, size_t stride) {
        for (long g = g0; g < g1;) {   // split the run at plane boundaries
            const int pl = (int)(g / h), r0 = (int)(g % h);
            const int nr = (int)std::min<long>(g1 - g, h - r0);
            char *sp = slot + (size_t)(g - g0) * row;
            if (to_slot) w2xc_host::CopyPool::get().copy_rows(sp, row, (const char *)src_planes[pl] + (size_t)r0 * stride, stride, row, nr, copy_threads);
            else w2xc_host::CopyPool::get().copy_rows((char *)dst_planes[pl] + (size_t)r0 * stride, stride, sp, row, row, nr, copy_threads);
            g += nr;
        }
    };

    rc = grow(&fc.planar[ob], &fc.planar_floats[ob], px * hl.nout);
    if (rc) return rc;
    const float *d_in;
    long long in_cs, in_rs, in_ps;
    if (resident && fc.res_nhwc) {
        d_in = fc.nhwc[ob ^ 1]; in_cs = 1; in_rs = (long long)w * hl.nin; in_ps = hl.nin;
    } else {
        if (!resident) {
            rc = grow(&fc.planar[ob ^ 1], &fc.planar_floats[ob ^ 1], px * hl.nin);
            if (rc) return rc;
            float *dst = fc.planar[ob ^ 1];
            rc = for_runs(hl.nin, [&](long g0, long g1, char *slot, int si) -> int {
                if (seq >= FilterCache::SLOTS) HIP_TRY(hipEventSynchronize(fc.ev[si]));   // the slot's previous DMA is done
                host_rows(true, slot, g0, g1, in_planes, nullptr, in_stride_bytes);
                HIP_TRY(hipMemcpyAsync(dst + (size_t)g0 * w, slot, (size_t)(g1 - g0) * row, hipMemcpyHostToDevice, fc.st));
                HIP_TRY(hipEventRecord(fc.ev[si], fc.st));
                return W2XC_OK;
            });
            if (rc) { hipStreamSynchronize(fc.st); return rc; }
        }
        d_in = fc.planar[ob ^ 1]; in_cs = (long long)px; in_rs = w; in_ps = 1;
    }
    bool res_nhwc = false;
    rc = filter_on_device(m, c, layer, d_in, in_cs, in_rs, in_ps, w, h, fc.planar[ob], (long long)px, w, 1, fc.st, o, ob, &res_nhwc);
    if (rc) { hipStreamSynchronize(fc.st); return rc; }

    // download: D2H of run k+1 overlaps the host copy of run k
    struct Run { long g0, g1; int si; };
    std::vector<Run> inflight;
    auto finish_run = [&](const Run &r) -> int {
        HIP_TRY(hipEventSynchronize(fc.ev[r.si]));
        host_rows(false, fc.pin + (size_t)r.si * fc.slot_bytes, r.g0, r.g1, nullptr, out_planes, out_stride_bytes);
        return W2XC_OK;
    };
    HIP_TRY(hipStreamSynchronize(fc.st));   // uploads done: the ring is free again, the layer has run
    seq = 0;
    rc = for_runs(hl.nout, [&](long g0, long g1, char *slot, int si) -> int {
        if ((int)inflight.size() == FilterCache::SLOTS) {
            int r = finish_run(inflight.front());
            if (r) return r;
            inflight.erase(inflight.begin());
        }
        HIP_TRY(hipMemcpyAsync(slot, fc.planar[ob] + (size_t)g0 * w, (size_t)(g1 - g0) * row, hipMemcpyDeviceToHost, fc.st));
        HIP_TRY(hipEventRecord(fc.ev[si], fc.st));
        inflight.push_back({g0, g1, si});
        return W2XC_OK;
    });
    for (size_t i = 0; !rc && i < inflight.size(); i++) rc = finish_run(inflight[i]);
    if (rc) { hipStreamSynchronize(fc.st); return rc; }

    fc.ob = ob;
    fc.res_valid = true;
    fc.res_nhwc = res_nhwc;
    fc.res_planes = hl.nout; fc.res_w = w; fc.res_h = h;
    fc.res_stride = out_stride_bytes;
    fc.res_host.assign(out_planes, out_planes + hl.nout);
    return W2XC_OK;
} W2XC_CATCH_ALL

}  // extern "C"
