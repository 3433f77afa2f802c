// w2xc_image.cpp -- N2: the CLI's image pipeline around the plane conversion -- uint8 BGR -> float YUV, the noise /
// scale passes on Y through run_rows, bicubic U/V, the final shrink, YUV -> uint8 (main.cpp:74-76,83-98,126-172).
#include "w2xc_engine.hpp"

namespace w2xc_eng {

namespace {

// noise (optional, main.cpp:83-98) then `iterations` 2x scale steps (optional model, main.cpp:126-156).
// `c` is the context that owns the plane buffer (the scale model's when present, else the noise model's);
// cn / cs are the contexts of the two models (locked by the caller).
// final size of the pipeline: (w << iterations) x (h << iterations), then the optional shrink of main.cpp:158-167
void final_size(int w, int h, int iterations, double shrink, int *fw, int *fh)
{
    *fw = w << iterations;
    *fh = h << iterations;
    if (shrink > 0.0) {
        *fw = static_cast<int>(static_cast<double>(*fw * shrink));   // :160-165
        *fh = static_cast<int>(static_cast<double>(*fh * shrink));
    }
}

int process_image_device(w2xc_model *mn, DevCtx *cn, w2xc_model *msc, DevCtx *cs, const unsigned char *d_in, size_t in_stride, int w,
                         int h, unsigned char *d_out, size_t out_stride, int iterations, double shrink, hipStream_t st, const w2xc_opts &o)
{
    DevCtx *c = cs ? cs : cn;
    // planes: level 0 (w x h) twice when a noise pass needs a second Y, then one level per iteration
    size_t need = 4 * (size_t)w * h, lvl = (size_t)w * h;
    for (int i = 1; i <= iterations; i++) { lvl *= 4; need += 3 * lvl; }
    int fw, fh;
    final_size(w, h, iterations, shrink, &fw, &fh);
    if (shrink > 0.0) need += 3 * (size_t)fw * fh;
    if (c->aux_floats < need) {
        if (c->aux) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(c->aux)); c->aux = nullptr; c->aux_floats = 0; }
        hipError_t e = hipMalloc((void **)&c->aux, need * sizeof(float));
        if (e != hipSuccess) return fail(W2XC_ERR_NOMEM, "hipMalloc(%zu MiB) for the image planes failed: %s", (need * 4) >> 20, hipGetErrorString(e));
        c->aux_floats = need;
    }
    float *base = c->aux;
    int cw = w, ch = h;
    float *y = base, *u = y + (size_t)cw * ch, *v = u + (size_t)cw * ch, *yn = v + (size_t)cw * ch;
    base = yn + (size_t)cw * ch;
    HIP_TRY(w2xc_launch_u8_to_yuv(d_in, in_stride, w, h, y, u, v, st));                                   // :75-76
    if (mn) {                                                                                             // :91-98
        int rc = run_rows(mn, cn, y, cw, ch, 0, cw, 0, ch, yn, cw, st, o, 0, 1, 0, 0, nullptr, ch);
        if (rc) return rc;
        y = yn;
    }
    for (int it = 0; it < iterations; it++) {
        const int nw = cw * 2, nh = ch * 2;
        float *y2 = base, *u2 = y2 + (size_t)nw * nh, *v2 = u2 + (size_t)nw * nh;
        base = v2 + (size_t)nw * nh;
        // Y: INTER_NEAREST 2x folded into layer 1 (:136-140) + convertWithModels (:148)
        int rc = run_rows(msc, cs, y, cw, nh, 0, nw, 0, nh, y2, nw, st, o, 1, 1, 0, 0, nullptr, nh);
        if (rc) return rc;
        HIP_TRY(w2xc_launch_resize2x_cubic(u, cw, ch, u2, st));                                            // :144-146
        HIP_TRY(w2xc_launch_resize2x_cubic(v, cw, ch, v2, st));
        y = y2; u = u2; v = v2; cw = nw; ch = nh;
    }
    if (shrink > 0.0) {                                                                                   // :158-167
        float *ys = base, *us = ys + (size_t)fw * fh, *vs = us + (size_t)fw * fh;
        HIP_TRY(w2xc_launch_resize_linear(y, cw, ch, ys, fw, fh, st));
        HIP_TRY(w2xc_launch_resize_linear(u, cw, ch, us, fw, fh, st));
        HIP_TRY(w2xc_launch_resize_linear(v, cw, ch, vs, fw, fh, st));
        y = ys; u = us; v = vs; cw = fw; ch = fh;
    }
    HIP_TRY(w2xc_launch_yuv_to_u8(y, u, v, cw, ch, d_out, out_stride, st));                               // :171-172
    return W2XC_OK;
}

// Both contexts of a noise + scale call, taken TOGETHER (std::lock's deadlock avoidance): two threads that pass the same two models in opposite
// roles -- (A as noise, B as scale) and (B as noise, A as scale) -- would otherwise each hold one mutex and wait for the other.
void lock_contexts(DevCtx *cn, DevCtx *cs, std::unique_lock<std::mutex> &l1, std::unique_lock<std::mutex> &l2)
{
    if (cn && cs && cn != cs) {
        l1 = std::unique_lock<std::mutex>(cn->mu, std::defer_lock);
        l2 = std::unique_lock<std::mutex>(cs->mu, std::defer_lock);
        std::lock(l1, l2);
    } else if (cn) l1 = std::unique_lock<std::mutex>(cn->mu);
    else if (cs) l1 = std::unique_lock<std::mutex>(cs->mu);
}

// resolve device + contexts of the (up to two) models and run the pipeline under their locks
int process_image_locked(w2xc_model *mn, w2xc_model *msc, const unsigned char *d_in, size_t in_stride, int w, int h, unsigned char *d_out,
                         size_t out_stride, int iterations, double shrink, hipStream_t st, const w2xc_opts &o, int dev)
{
    DevCtx *cn = nullptr, *cs = nullptr;
    int rc;
    if (mn && (rc = get_ctx(mn, dev, &cn))) return rc;
    if (msc && (rc = get_ctx(msc, dev, &cs))) return rc;
    std::unique_lock<std::mutex> l1, l2;
    lock_contexts(cn, cs, l1, l2);
    return process_image_device(mn, cn, msc, cs, d_in, in_stride, w, h, d_out, out_stride, iterations, shrink, st, o);
}

int check_process_args(const w2xc_model *mn, const w2xc_model *msc, int iterations)
{
    if (!mn && !msc) return fail(W2XC_ERR_ARG, "need a noise model, a scale model or both");
    if (iterations > 0 && !msc) return fail(W2XC_ERR_ARG, "scale iterations need a scale model");
    if (!mn && iterations == 0) return fail(W2XC_ERR_ARG, "nothing to do (no noise model, 0 iterations)");
    return W2XC_OK;
}

int check_image_args(const w2xc_model *m, const void *in, size_t in_stride, int w, int h, const void *out, size_t out_stride, int iterations,
                     double shrink = 0.0)
{
    if (!m || !in || !out) return fail(W2XC_ERR_ARG, "null argument");
    if (w <= 0 || h <= 0 || iterations < 0 || iterations > 4) return fail(W2XC_ERR_ARG, "bad image size / iteration count");
    if (shrink < 0.0 || shrink >= 1.0) return fail(W2XC_ERR_ARG, "shrink_ratio must be 0 (none) or in (0,1)");
    int fw, fh;
    final_size(w, h, iterations, shrink, &fw, &fh);
    if (fw < 1 || fh < 1) return fail(W2XC_ERR_ARG, "shrink_ratio leaves an empty image");
    if (in_stride < (size_t)w * 3 || out_stride < (size_t)fw * 3) return fail(W2XC_ERR_ARG, "row strides must be >= 3*width bytes");
    return W2XC_OK;
}

}  // namespace

}  // namespace w2xc_eng

using namespace w2xc_eng;

extern "C" {

int w2xc_process_image_u8_ex_device(w2xc_model *noise_model, w2xc_model *scale_model, const unsigned char *d_in, size_t in_stride_bytes,
                                    int w, int h, unsigned char *d_out, size_t out_stride_bytes, int iterations, double shrink_ratio,
                                    void *hip_stream, const w2xc_opts *opts)
try {
    int rc = check_process_args(noise_model, scale_model, iterations);
    if (rc) return rc;
    rc = check_image_args(noise_model ? noise_model : scale_model, d_in, in_stride_bytes, w, h, d_out, out_stride_bytes, iterations, shrink_ratio);
    if (rc) return rc;
    const w2xc_opts o = resolve_opts(opts);
    int dev = o.device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    DeviceGuard guard(dev);
    if (!guard.ok) return fail(W2XC_ERR_HIP, "cannot select HIP device %d", dev);
    return process_image_locked(noise_model, scale_model, d_in, in_stride_bytes, w, h, d_out, out_stride_bytes, iterations, shrink_ratio,
                                (hipStream_t)hip_stream, o, dev);
} W2XC_CATCH_ALL

int w2xc_process_image_u8_device(w2xc_model *noise_model, w2xc_model *scale_model, const unsigned char *d_in, size_t in_stride_bytes,
                                 int w, int h, unsigned char *d_out, size_t out_stride_bytes, int iterations, void *hip_stream,
                                 const w2xc_opts *opts)
{
    return w2xc_process_image_u8_ex_device(noise_model, scale_model, d_in, in_stride_bytes, w, h, d_out, out_stride_bytes, iterations, 0.0,
                                           hip_stream, opts);
}

int w2xc_process_image_u8_ex(w2xc_model *noise_model, w2xc_model *scale_model, const unsigned char *in, size_t in_stride_bytes, int w, int h,
                             unsigned char *out, size_t out_stride_bytes, int iterations, double shrink_ratio, const w2xc_opts *opts)
try {
    int rc = check_process_args(noise_model, scale_model, iterations);
    if (rc) return rc;
    rc = check_image_args(noise_model ? noise_model : scale_model, in, in_stride_bytes, w, h, out, out_stride_bytes, iterations, shrink_ratio);
    if (rc) return rc;
    if (w2xc_device_count() <= 0) return fail(W2XC_ERR_HIP, "no HIP device available (libw2xc_hip has no CPU fallback)");
    const w2xc_opts o = resolve_opts(opts);
    int dev = o.device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    DeviceGuard guard(dev);
    if (!guard.ok) return fail(W2XC_ERR_HIP, "cannot select HIP device %d", dev);
    int W, H;
    final_size(w, h, iterations, shrink_ratio, &W, &H);
    // contexts of the (up to two) models, locked for the whole call: the device copies of the image live in the owning context
    // (the scale model's when present) and are kept between calls -- no hipMalloc / hipFree per image
    DevCtx *cn = nullptr, *cs = nullptr;
    if (noise_model && (rc = get_ctx(noise_model, dev, &cn))) return rc;
    if (scale_model && (rc = get_ctx(scale_model, dev, &cs))) return rc;
    std::unique_lock<std::mutex> l1, l2;
    lock_contexts(cn, cs, l1, l2);
    DevCtx *c = cs ? cs : cn;
    const size_t in_bytes = ((size_t)w * 3 * h + 255) & ~(size_t)255, out_bytes = (size_t)W * 3 * H;
    if (c->img_io_bytes < in_bytes + out_bytes) {
        if (c->img_io) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(c->img_io)); c->img_io = nullptr; c->img_io_bytes = 0; }
        if (hipMalloc((void **)&c->img_io, in_bytes + out_bytes) != hipSuccess)
            return fail(W2XC_ERR_NOMEM, "hipMalloc(%zu MiB) for the image failed", (in_bytes + out_bytes) >> 20);
        c->img_io_bytes = in_bytes + out_bytes;
    }
    unsigned char *d_in = c->img_io, *d_out = c->img_io + in_bytes;
    HIP_TRY(hipMemcpy2D(d_in, (size_t)w * 3, in, in_stride_bytes, (size_t)w * 3, h, hipMemcpyHostToDevice));
    rc = process_image_device(noise_model, cn, scale_model, cs, d_in, (size_t)w * 3, w, h, d_out, (size_t)W * 3, iterations, shrink_ratio, nullptr, o);
    if (rc) { hipDeviceSynchronize(); return rc; }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy2D(out, out_stride_bytes, d_out, (size_t)W * 3, (size_t)W * 3, H, hipMemcpyDeviceToHost));
    return W2XC_OK;
} W2XC_CATCH_ALL

int w2xc_process_image_u8(w2xc_model *noise_model, w2xc_model *scale_model, const unsigned char *in, size_t in_stride_bytes, int w, int h,
                          unsigned char *out, size_t out_stride_bytes, int iterations, const w2xc_opts *opts)
{
    return w2xc_process_image_u8_ex(noise_model, scale_model, in, in_stride_bytes, w, h, out, out_stride_bytes, iterations, 0.0, opts);
}

int w2xc_scale2x_image_u8_device(w2xc_model *m, const unsigned char *d_in, size_t in_stride_bytes, int w, int h, unsigned char *d_out,
                                 size_t out_stride_bytes, int iterations, void *hip_stream, const w2xc_opts *opts)
{
    return w2xc_process_image_u8_device(nullptr, m, d_in, in_stride_bytes, w, h, d_out, out_stride_bytes, iterations, hip_stream, opts);
}

int w2xc_scale2x_image_u8(w2xc_model *m, const unsigned char *in, size_t in_stride_bytes, int w, int h, unsigned char *out,
                          size_t out_stride_bytes, int iterations, const w2xc_opts *opts)
{
    return w2xc_process_image_u8(nullptr, m, in, in_stride_bytes, w, h, out, out_stride_bytes, iterations, opts);
}

int w2xc_resize2x_cubic_device(const float *d_src, int w, int h, float *d_dst, void *hip_stream)
{
    if (!d_src || !d_dst || w <= 0 || h <= 0) return fail(W2XC_ERR_ARG, "bad argument");
    HIP_TRY(w2xc_launch_resize2x_cubic(d_src, w, h, d_dst, (hipStream_t)hip_stream));
    return W2XC_OK;
}

int w2xc_u8_to_yuv_device(const unsigned char *d_in, size_t in_stride_bytes, int w, int h, float *d_y, float *d_u, float *d_v, void *hip_stream)
{
    if (!d_in || !d_y || !d_u || !d_v || w <= 0 || h <= 0 || in_stride_bytes < (size_t)w * 3) return fail(W2XC_ERR_ARG, "bad argument");
    HIP_TRY(w2xc_launch_u8_to_yuv(d_in, in_stride_bytes, w, h, d_y, d_u, d_v, (hipStream_t)hip_stream));
    return W2XC_OK;
}

int w2xc_yuv_to_u8_device(const float *d_y, const float *d_u, const float *d_v, int w, int h, unsigned char *d_out, size_t out_stride_bytes,
                          void *hip_stream)
{
    if (!d_out || !d_y || !d_u || !d_v || w <= 0 || h <= 0 || out_stride_bytes < (size_t)w * 3) return fail(W2XC_ERR_ARG, "bad argument");
    HIP_TRY(w2xc_launch_yuv_to_u8(d_y, d_u, d_v, w, h, d_out, out_stride_bytes, (hipStream_t)hip_stream));
    return W2XC_OK;
}

}  // extern "C"
