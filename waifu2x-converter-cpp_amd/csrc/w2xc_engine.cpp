// w2xc_engine.cpp -- C ABI of libw2xc_hip.so (include/w2xc_hip.h): model container + JSON
// loader (the reference's Model / modelUtility, src/modelHandler.{hpp,cpp}), and the band loop that
// replaces convertWithModels / convertWithModelsBasic / convertWithModelsBlockSplit
// (src/convertRoutine.cpp:21-169) on MI355X.
//
// Data layout in HBM: activations between layers are NHWC fp32 (pixel stride = plane count), two
// ping-pong workspaces per device sized for one band; layer 1 reads the caller's planar plane with
// clamp-to-edge addressing (= copyMakeBorder, convertRoutine.cpp:35,96) and the last layer writes
// the planar output rows in place (= crop + stitch, :40-46,143-161).  A band is `band_rows`
// output rows x full width; layer k of n computes (rows + 2(n-k)) x (w + 2(n-k)) pixels (valid
// conv on the haloed band -- SURVEY invariants I1/I2).
//
// There is no CPU fallback: without a HIP device every compute entry point fails.
#include "../../include/w2xc_hip.h"

#include <hip/hip_runtime.h>
#include <sched.h>
#include <pthread.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "json_min.hpp"
#include "w2xc_copy_pool.hpp"
#include "w2xc_kernels.h"

// ------------------------------------------------------------------------------------------------
namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(W2XC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// no C++ exception (std::bad_alloc from a staging vector, std::system_error from std::thread ...) may cross the C ABI
#define W2XC_CATCH_ALL                                                                                       \
    catch (const std::bad_alloc &) { return fail(W2XC_ERR_NOMEM, "out of host memory"); }                    \
    catch (const std::exception &e_) { return fail(W2XC_ERR_HIP, "internal error: %s", e_.what()); }         \
    catch (...) { return fail(W2XC_ERR_HIP, "internal error"); }

struct HostLayer {
    int nin = 0, nout = 0;
    std::vector<float> w;       // [nout][nin][3][3], index o*nin+i (modelHandler.cpp:102)
    std::vector<double> bias;   // modelHandler.cpp:109-112 keeps doubles
};

struct DevLayer {
    W2xcKernelKind fast = W2XC_K_DIRECT;
    float *w_fast = nullptr;
    float *w_direct = nullptr;
    float *w_wino = nullptr;     // w2xc_wino_pack image (fp32 Winograd path, 32x32x2 kernel), packed on first use
    float *w_first2 = nullptr;  // w2xc_first2_wino4_pack image (layer 2 of the fused first two layers), packed on first use
    float *w_wino4 = nullptr;    // w2xc_wino4_pack image (F(4x4,3x3) kernel), packed on first use
    float *w_split[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // conv3x3_split images, index terms + 3*fmt, packed on first use
    float split_scale[6] = {1, 1, 1, 1, 1, 1};                                   // power-of-two weight scale of each image
    float *w_last_fused[4] = {nullptr, nullptr, nullptr, nullptr};   // w2xc_split_pack_last images: [0] 2 bf16 terms, [1] 2 fp16 terms, [2] 3 bf16 terms, [3] 1 bf16 term
    float last_fused_scale[4] = {1, 1, 1, 1};
    float *w_last_wino4 = nullptr;    // w2xc_wino4_pack_last image (fp32 path: last layer inside conv3x3_wino4's epilogue)
    float *bias = nullptr;
};

struct ProfEvent {
    hipEvent_t a, b;
    int layer;
};

// The host->host tile farm of one (model, device): everything a w2xc_convert_plane call needs beyond the
// kernels, created once and kept (no hipMalloc / hipStreamCreate / hipHostMalloc on the per-call path):
//   three streams   s_h2d: staging ring -> d_in      s_compute: the layer launches of every band
//                   s_d2h: d_out -> staging ring     (ordered by events; copies run on the SDMA engines)
//   d_in / d_out    this device's share of the caller's plane (source rows incl. halo / output rows)
//   pin_in/pin_out  rings of pinned staging slots between the caller's pageable planes and the DMA engines
// This is the parallel replacement of the sequential block walk of convertRoutine.cpp:114-165.
struct HostPipe {
    static constexpr int IN_SLOTS = 3, OUT_SLOTS = 4;
    hipStream_t s_compute = nullptr, s_h2d = nullptr, s_d2h = nullptr;
    float *d_in = nullptr, *d_out = nullptr;
    size_t d_in_bytes = 0, d_out_bytes = 0;
    char *pin_in = nullptr, *pin_out = nullptr;
    size_t in_slot_bytes = 0, out_slot_bytes = 0;
    hipEvent_t ev_in_slot[IN_SLOTS] = {}, ev_out_slot[OUT_SLOTS] = {};   // last DMA that used the slot
    hipEvent_t ev_input = nullptr, ev_chunk = nullptr;                   // band input landed / last-layer chunk computed
    bool ready = false;

    void destroy()
    {
        if (s_compute) hipStreamSynchronize(s_compute);
        if (s_h2d) hipStreamSynchronize(s_h2d);
        if (s_d2h) hipStreamSynchronize(s_d2h);
        for (auto &e : ev_in_slot) if (e) { hipEventDestroy(e); e = nullptr; }
        for (auto &e : ev_out_slot) if (e) { hipEventDestroy(e); e = nullptr; }
        if (ev_input) { hipEventDestroy(ev_input); ev_input = nullptr; }
        if (ev_chunk) { hipEventDestroy(ev_chunk); ev_chunk = nullptr; }
        if (d_in) { hipFree(d_in); d_in = nullptr; d_in_bytes = 0; }
        if (d_out) { hipFree(d_out); d_out = nullptr; d_out_bytes = 0; }
        if (pin_in) { hipHostFree(pin_in); pin_in = nullptr; in_slot_bytes = 0; }
        if (pin_out) { hipHostFree(pin_out); pin_out = nullptr; out_slot_bytes = 0; }
        if (s_compute) { hipStreamDestroy(s_compute); s_compute = nullptr; }
        if (s_h2d) { hipStreamDestroy(s_h2d); s_h2d = nullptr; }
        if (s_d2h) { hipStreamDestroy(s_d2h); s_d2h = nullptr; }
        ready = false;
    }
};

// Model::filter at the host boundary (w2xc_layer_filter): persistent device buffers, a pinned bounce ring and one stream
// per (model, device), and what the previous call left on the device for a caller that chains filter() by hand
// (the reference's test.cpp:72-85 pattern) -- see w2xc_opts.filter_resident.
struct FilterCache {
    static constexpr int SLOTS = 2;
    float *planar[2] = {nullptr, nullptr}, *nhwc[2] = {nullptr, nullptr};   // ping-pong: a call reads [ob ^ 1], writes [ob]
    size_t planar_floats[2] = {0, 0}, nhwc_floats[2] = {0, 0};
    float *pad = nullptr, *pout = nullptr;   // conv3x3_wino4: the replicate-padded planar copy of the input planes / an aligned planar result
    size_t pad_floats = 0, pout_floats = 0;
    char *pin = nullptr;            // SLOTS pinned bounce slots between the caller's pageable planes and the DMA engine
    size_t slot_bytes = 0;
    hipStream_t st = nullptr;
    hipEvent_t ev[SLOTS] = {};
    int ob = 0;                     // buffer index the LAST call wrote
    bool res_valid = false, res_nhwc = false;
    int res_planes = 0, res_w = 0, res_h = 0;
    std::vector<const float *> res_host;   // the host planes the result was downloaded to
    size_t res_stride = 0;
};

struct DevCtx {
    int device = 0;
    std::vector<DevLayer> layers;
    float *ws[2] = {nullptr, nullptr};
    size_t ws_floats[2] = {0, 0};
    HostPipe pipe;
    FilterCache fc;
    float *aux = nullptr;       // N2: Y/U/V planes of the image pipeline
    size_t aux_floats = 0;
    unsigned char *img_io = nullptr;   // N2, host entry points: device copies of the uint8 image in / out (grow-only)
    size_t img_io_bytes = 0;
    std::vector<ProfEvent> pending, pool;
    std::vector<double> layer_ms;
    std::vector<int> layer_launches;
    std::mutex mu;

    ~DevCtx()
    {
        int prev = 0;
        hipGetDevice(&prev);
        hipSetDevice(device);
        for (auto &l : layers) {
            if (l.w_fast) hipFree(l.w_fast);
            if (l.w_direct) hipFree(l.w_direct);
            if (l.w_wino) hipFree(l.w_wino);
            if (l.w_wino4) hipFree(l.w_wino4);
            if (l.w_first2) hipFree(l.w_first2);
            if (l.w_last_wino4) hipFree(l.w_last_wino4);
            for (float *p : l.w_split)
                if (p) hipFree(p);
            for (float *p : l.w_last_fused)
                if (p) hipFree(p);
            if (l.bias) hipFree(l.bias);
        }
        pipe.destroy();
        if (fc.st) hipStreamSynchronize(fc.st);
        for (float *p : fc.planar) if (p) hipFree(p);
        for (float *p : fc.nhwc) if (p) hipFree(p);
        if (fc.pad) hipFree(fc.pad);
        if (fc.pout) hipFree(fc.pout);
        if (fc.pin) hipHostFree(fc.pin);
        for (auto &e : fc.ev) if (e) hipEventDestroy(e);
        if (fc.st) hipStreamDestroy(fc.st);
        for (int i = 0; i < 2; i++)
            if (ws[i]) hipFree(ws[i]);
        if (aux) hipFree(aux);
        if (img_io) hipFree(img_io);
        for (auto &e : pending) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
        for (auto &e : pool) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
        hipSetDevice(prev);
    }
};

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) return;
        ok = (dev == prev) || hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) hipSetDevice(prev);
    }
};

// modelUtility singleton state (modelHandler.hpp:92-100)
std::mutex g_util_mu;
int g_njob = 4;
int g_block_w = 512, g_block_h = 512;

}  // namespace

struct w2xc_model {
    std::vector<HostLayer> layers;
    std::mutex mu;
    std::map<int, std::unique_ptr<DevCtx>> ctx;
};

namespace {

w2xc_opts resolve_opts(const w2xc_opts *o)
{
    w2xc_opts r;
    w2xc_opts_init(&r);
    if (o) {
        size_t n = o->struct_size > 0 && (size_t)o->struct_size < sizeof(w2xc_opts) ? (size_t)o->struct_size : sizeof(w2xc_opts);
        memcpy(&r, o, n);
        r.struct_size = (int)sizeof(w2xc_opts);
    } else {
        // callers that pass no options (the C++ adapter behind the reference's CLI): the default precision can
        // be switched without recompiling -- W2XC_PRECISION = fp32 | bf16x3 | fp16x2 | bf16x2 | bf16
        static const int env_prec = [] {
            const char *e = getenv("W2XC_PRECISION");
            if (!e) return W2XC_PRECISION_FP32;
            if (!strcmp(e, "bf16x3")) return W2XC_PRECISION_BF16X3;
            if (!strcmp(e, "bf16x2")) return W2XC_PRECISION_BF16X2;
            if (!strcmp(e, "fp16x2")) return W2XC_PRECISION_FP16X2;
            if (!strcmp(e, "bf16")) return W2XC_PRECISION_BF16;
            return W2XC_PRECISION_FP32;
        }();
        r.precision = env_prec;
        static const int env_res = [] { const char *e = getenv("W2XC_FILTER_RESIDENT"); return (e && atoi(e) != 0) ? 1 : 0; }();
        r.filter_resident = env_res;
    }
    return r;
}

// 16-bit terms per activation value between the layers of the split pipeline, w2xc_split.hip (0 = not that pipeline)
int split_terms(const w2xc_opts &o)
{
    if (o.precision == W2XC_PRECISION_BF16) return 1;   // plain bf16 = the same pipeline with ONE term
    return (o.precision == W2XC_PRECISION_BF16X2 || o.precision == W2XC_PRECISION_FP16X2) ? 2 : o.precision == W2XC_PRECISION_BF16X3 ? 3 : 0;
}
int split_fmt(const w2xc_opts &o) { return o.precision == W2XC_PRECISION_FP16X2 ? 1 : 0; }

bool fuse_last(const w2xc_model *m, const w2xc_opts &o);
bool fuse_first(const w2xc_model *m, const w2xc_opts &o);
bool fuse_first_fp32(const w2xc_model *m, const w2xc_opts &o);
bool fuse_last_fp32(const w2xc_model *m, const w2xc_opts &o);
bool is_wino4_layer(const w2xc_model *m, int l, const w2xc_opts &o);
int layer_mid_variant(const w2xc_model *m, int l, const w2xc_opts &o);
bool uses_wino4(const w2xc_model *m, const w2xc_opts &o);
bool planar_between(const w2xc_model *m, int l, const w2xc_opts &o);

W2xcKernelKind layer_kind(const w2xc_model *m, int l, const w2xc_opts &o)
{
    if (o.kernel == W2XC_KERNEL_DIRECT) return W2XC_K_DIRECT;
    const W2xcKernelKind k = w2xc_pick_kernel(m->layers[l].nin, m->layers[l].nout);
    const int n = (int)m->layers.size();
    if (split_terms(o) > 0) {
        // term planes live only BETWEEN a first/mid layer and a mid layer; everything that touches the
        // caller's planes or the last layer is fp32.  Shapes without an MFMA kernel are unsupported.
        if (l <= 1 && fuse_first(m, o)) return l == 0 ? W2XC_K_FUSED_AWAY : W2XC_K_FIRST2_SPLIT;
        if (k == W2XC_K_MFMA) return l > 0 ? W2XC_K_MID_SPLIT : W2XC_K_DIRECT;
        if (k == W2XC_K_FIRST && l == 0)
            return (n > 1 && w2xc_pick_kernel(m->layers[1].nin, m->layers[1].nout) == W2XC_K_MFMA) ? W2XC_K_FIRST_SPLIT : W2XC_K_FIRST;
        if (k == W2XC_K_LAST && l == n - 1 && l > 0) return fuse_last(m, o) ? W2XC_K_LAST_GATHER : W2XC_K_LAST;
        return W2XC_K_DIRECT;   // run_rows rejects this
    }
    if (k == W2XC_K_LAST && l == n - 1 && fuse_last_fp32(m, o)) return W2XC_K_LAST_GATHER;
    if (l <= 1 && fuse_first_fp32(m, o)) return l == 0 ? W2XC_K_FUSED_AWAY : W2XC_K_FIRST2_WINO4;
    return k;
}

// 16-bit modes: layers 1 (ONE plane -> 32) and 2 (32 -> {32,64,128}) run as one kernel (conv3x3_first2_split) when
// layer 2 is an ordinary split mid layer.  W2XC_SPLIT_FUSE_FIRST=0 disables.
bool fuse_first(const w2xc_model *m, const w2xc_opts &o)
{
    static const int en = [] { const char *e = getenv("W2XC_SPLIT_FUSE_FIRST"); return (e && atoi(e) == 0) ? 0 : 1; }();   // (thread-safe initialisation)
    const int n = (int)m->layers.size();
    if (!en || o.kernel == W2XC_KERNEL_DIRECT || n < 3 || split_terms(o) == 0) return false;
    if (m->layers[0].nin != 1 || m->layers[0].nout != 32) return false;
    if (w2xc_pick_kernel(m->layers[1].nin, m->layers[1].nout) != W2XC_K_MFMA) return false;
    return !(n == 3 && fuse_last(m, o));   // (layer 2 would be the fused-last producer: keep that fusion instead)
}

// 16-bit modes: the last layer (cin in {32,64,128} -> ONE plane) is computed inside the epilogue of the mid layer
// before it (conv3x3_split, out_terms = 9) and finished by conv3x3_last_gather.  W2XC_SPLIT_FUSE_LAST=0 disables.
bool fuse_last(const w2xc_model *m, const w2xc_opts &o)
{
    static const int en = [] { const char *e = getenv("W2XC_SPLIT_FUSE_LAST"); return (e && atoi(e) == 0) ? 0 : 1; }();
    const int n = (int)m->layers.size();
    if (!en || o.kernel == W2XC_KERNEL_DIRECT || n < 3) return false;
    const int T = split_terms(o);
    if (T < 1 || T > 3) return false;
    return m->layers[n - 1].nout == 1 && w2xc_pick_kernel(m->layers[n - 1].nin, 1) == W2XC_K_LAST &&
           w2xc_pick_kernel(m->layers[n - 2].nin, m->layers[n - 2].nout) == W2XC_K_MFMA && n - 2 > 0;
}

// terms of layer l's OUTPUT in the split pipeline: T when layer l+1 is a split mid layer, else 0 (fp32); 9 = this layer writes
// the partial tap planes of the last layer it computes in its epilogue (16-bit modes: conv3x3_split; fp32: conv3x3_wino4)
int out_terms_of(const w2xc_model *m, int l, const w2xc_opts &o)
{
    const int T = split_terms(o), n = (int)m->layers.size();
    if (T == 0) return (l == n - 2 && fuse_last_fp32(m, o)) ? 9 : 0;
    if (l + 1 >= n) return 0;
    if (l == n - 2 && fuse_last(m, o)) return 9;
    return layer_kind(m, l + 1, o) == W2XC_K_MID_SPLIT ? T : 0;
}

// partial-G planes a fused-last producer writes per tap: wave columns of the split tile shapes, 64-plane blocks of conv3x3_wino4
int fused_halves(int T, int cout) { return T > 0 ? w2xc_split_halves(T, cout) : cout / 64; }

int upload(const std::vector<float> &h, float **d)
{
    HIP_TRY(hipMalloc((void **)d, std::max<size_t>(h.size(), 1) * sizeof(float)));
    HIP_TRY(hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return W2XC_OK;
}

// per-(model, device) context: packed weights + biases resident in HBM; created on first use
int get_ctx(w2xc_model *m, int device, DevCtx **out)
{
    std::lock_guard<std::mutex> lk(m->mu);
    auto it = m->ctx.find(device);
    if (it != m->ctx.end()) { *out = it->second.get(); return W2XC_OK; }
    std::unique_ptr<DevCtx> c(new DevCtx());
    c->device = device;
    c->layers.resize(m->layers.size());
    c->layer_ms.assign(m->layers.size(), 0.0);
    c->layer_launches.assign(m->layers.size(), 0);
    for (size_t l = 0; l < m->layers.size(); l++) {
        const HostLayer &hl = m->layers[l];
        DevLayer &dl = c->layers[l];
        dl.fast = w2xc_pick_kernel(hl.nin, hl.nout);
        std::vector<float> pk(w2xc_packed_weight_floats(W2XC_K_DIRECT, hl.nin, hl.nout));
        w2xc_pack_weights(W2XC_K_DIRECT, hl.nin, hl.nout, hl.w.data(), pk.data());
        int rc = upload(pk, &dl.w_direct);
        if (rc) return rc;
        if (dl.fast != W2XC_K_DIRECT) {
            pk.assign(w2xc_packed_weight_floats(dl.fast, hl.nin, hl.nout), 0.f);
            w2xc_pack_weights(dl.fast, hl.nin, hl.nout, hl.w.data(), pk.data());
            rc = upload(pk, &dl.w_fast);
            if (rc) return rc;
        }
        std::vector<float> bf(hl.nout);
        for (int o = 0; o < hl.nout; o++) bf[o] = (float)hl.bias[o];   // cv::add(UMat, double) narrows to the array depth
        rc = upload(bf, &dl.bias);
        if (rc) return rc;
    }
    *out = c.get();
    m->ctx[device] = std::move(c);
    return W2XC_OK;
}

int ensure_ws(DevCtx *c, int which, size_t floats)
{
    if (c->ws_floats[which] >= floats) return W2XC_OK;
    if (c->ws[which]) {
        HIP_TRY(hipDeviceSynchronize());   // earlier launches may still use the old buffer
        HIP_TRY(hipFree(c->ws[which]));
        c->ws[which] = nullptr;
        c->ws_floats[which] = 0;
    }
    hipError_t e = hipMalloc((void **)&c->ws[which], floats * sizeof(float));
    if (e != hipSuccess) return fail(W2XC_ERR_NOMEM, "hipMalloc(%zu MiB) for the activation workspace failed: %s", (floats * 4) >> 20, hipGetErrorString(e));
    c->ws_floats[which] = floats;
    return W2XC_OK;
}

int prof_begin(DevCtx *c, int layer, hipStream_t st, ProfEvent *ev)
{
    if (!c->pool.empty()) { *ev = c->pool.back(); c->pool.pop_back(); }
    else {
        HIP_TRY(hipEventCreate(&ev->a));
        hipError_t e = hipEventCreate(&ev->b);
        if (e != hipSuccess) {
            hipEventDestroy(ev->a);
            return fail(W2XC_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(e));
        }
    }
    ev->layer = layer;
    HIP_TRY(hipEventRecord(ev->a, st));
    return W2XC_OK;
}

// fp32 path, layers with 32 / 64 / 128 planes in and out (W2XC_K_MFMA): which kernel runs them.
//   MID_MFMA    conv3x3_mfma2: direct implicit GEMM, a k-ordered fp32 fma chain (the closest MFMA analogue of modelHandler.cpp:134-145)
//   MID_WINO32  conv3x3_wino:   Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32, one wave per SIMD (round 2)
// Same arithmetic type (fp32 throughout); Winograd does 2.25x fewer multiplies in another summation order and is held to the same
// rtol 1e-4 gate against the CPU oracle by the same tests.  w2xc_opts.kernel picks per call (W2XC_KERNEL_MFMA / _WINOGRAD /
// _WINOGRAD32 / _WINOGRAD4); W2XC_KERNEL_AUTO = W2XC_KERNEL_WINOGRAD4: conv3x3_wino4 (F(4x4,3x3)) where it applies (>= 64 output planes),
// conv3x3_wino for the rest.  No environment variable takes part in the choice.
//   MID_WINO4   conv3x3_wino4:  Winograd F(4x4,3x3) on v_mfma_f32_16x16x4_f32 (round 3, the default: 1.78x fewer multiplies again, ~1.3x the rounding error of
//               F(2x2); needs the four-rows-per-layer band geometry of run_rows to stay banding-invariant)
enum MidVariant { MID_MFMA = 0, MID_WINO32 = 1, MID_WINO4 = 3 };
int mid_variant(const w2xc_opts &o)
{
    switch (o.kernel) {
    case W2XC_KERNEL_MFMA: return MID_MFMA;
    case W2XC_KERNEL_WINOGRAD:     // (= _WINOGRAD32 since round 5: the round-3 F(2x2) kernel on 16x16x4 tiles, conv3x3_wino16, is retired)
    case W2XC_KERNEL_WINOGRAD32: return MID_WINO32;
    default: return MID_WINO4;   // W2XC_KERNEL_AUTO = W2XC_KERNEL_WINOGRAD4 (no environment switches: the choice is the caller's, per call)
    }
}
// the variant that really runs a (cin, cout) layer: conv3x3_wino4 needs 64-plane output blocks, the F(2x2) kernel takes the rest
int mid_variant_for(int midv, int cin, int cout)
{
    if (midv == MID_WINO4 && !w2xc_wino4_supported(cin, cout)) midv = MID_WINO32;
    if (midv == MID_WINO32 && !w2xc_wino_supported(cin, cout)) midv = MID_MFMA;
    return midv;
}

// the variant mid layer l really runs with these options
int layer_mid_variant(const w2xc_model *m, int l, const w2xc_opts &o)
{
    const HostLayer &p = m->layers[l];
    int midv = mid_variant(o);
    return mid_variant_for(midv, p.nin, p.nout);
}
// does any layer of the fp32 path run conv3x3_wino4 (F(4x4,3x3))?  Its 4x4 blocks make results depend on where a band's per-layer regions end,
// unless they end on block boundaries: run_rows then computes FOUR rows of halo per layer instead of one (and needs 4 n halo rows in its view).
bool uses_wino4(const w2xc_model *m, const w2xc_opts &o)
{
    if (split_terms(o) != 0 || o.precision != W2XC_PRECISION_FP32 || o.kernel == W2XC_KERNEL_DIRECT) return false;
    for (int l = 0; l < (int)m->layers.size(); l++)
        if (w2xc_pick_kernel(m->layers[l].nin, m->layers[l].nout) == W2XC_K_MFMA && layer_mid_variant(m, l, o) == MID_WINO4) return true;
    return false;
}
// conv3x3_wino4 reads PLANAR activations (one plane per channel, rows of roundup32(w) floats: 16-byte aligned pixel quads, tiles on 128-byte lines)
// -- with 32 input planes also the NHWC pixels (one 128-byte line each) that the 32-plane producers conv3x3_first / conv3x3_wino write.
// Layer l's output (l = 0 .. n-2) is planar when its consumer is a conv3x3_wino4 layer with 64 / 128 input planes, when layer l is the fused
// conv3x3_first2_wino4 launch (layers 1 + 2; its consumer conv3x3_wino4<32, .> then reads planar), or when layer l is a conv3x3_wino4 layer and its
// consumer is conv3x3_direct (any strides); everything else stays NHWC (conv3x3_wino4 writes either).  Producers that write planar:
// conv3x3_wino4, conv3x3_first2_wino4, conv3x3_first (3 -> 64 / 128), conv3x3_direct.
bool is_wino4_layer(const w2xc_model *m, int l, const w2xc_opts &o)
{
    if (split_terms(o) != 0 || o.precision != W2XC_PRECISION_FP32 || o.kernel == W2XC_KERNEL_DIRECT) return false;
    return l >= 0 && l < (int)m->layers.size() && w2xc_pick_kernel(m->layers[l].nin, m->layers[l].nout) == W2XC_K_MFMA && layer_mid_variant(m, l, o) == MID_WINO4;
}
bool planar_between(const w2xc_model *m, int l, const w2xc_opts &o)   // layout of layer l's output = layer l + 1's input
{
    const int n = (int)m->layers.size();
    if (l < 0 || l + 1 >= n) return false;
    if (l == 1 && fuse_first_fp32(m, o)) return true;   // (conv3x3_first2_wino4 writes planar planes; conv3x3_wino4<32, .> reads either)
    if (is_wino4_layer(m, l + 1, o)) return m->layers[l + 1].nin != 32;   // (32 input planes: conv3x3_wino4 reads the producer's NHWC pixels, one 128-byte line each)
    if (!is_wino4_layer(m, l, o)) return false;
    return layer_kind(m, l + 1, o) == W2XC_K_DIRECT;   // (conv3x3_last reads NHWC at 5.4 TB/s; its planar variant measured half of that: NHWC out there)
}

// fp32 path: the one-plane last layer inside the epilogue of the layer before it when that layer runs conv3x3_wino4 (Cout 64 / 128):
// the producer writes Cout / 64 x 9 partial tap planes instead of Cout activation planes, conv3x3_last_gather finishes.
// w2xc_opts.fusion = W2XC_FUSION_OFF / _ON decides per call; W2XC_FUSION_AUTO = on.
bool fuse_last_fp32(const w2xc_model *m, const w2xc_opts &o)
{
    const int n = (int)m->layers.size();
    if (split_terms(o) != 0 || o.precision != W2XC_PRECISION_FP32 || o.kernel == W2XC_KERNEL_DIRECT || n < 3) return false;
    if (o.fusion == W2XC_FUSION_OFF) return false;   // (W2XC_FUSION_AUTO = on)
    const HostLayer &p = m->layers[n - 2], &q = m->layers[n - 1];
    if (q.nout != 1 || q.nin != p.nout || w2xc_pick_kernel(q.nin, 1) != W2XC_K_LAST || w2xc_pick_kernel(p.nin, p.nout) != W2XC_K_MFMA) return false;
    const int v = layer_mid_variant(m, n - 2, o);
    return v == MID_WINO4;
}

// fp32 path: layers 1 (ONE plane -> 32) and 2 (32 -> 32) in one launch (conv3x3_first2_wino4: layer 1 on the fly per Winograd patch, layer 2 as F(4x4,3x3)
// with its weights stationary in registers) when the default kernels run the model and layer 3 reads planar planes (a conv3x3_wino4 layer).  Layer 1's 32
// activation planes never reach HBM (convertRoutine.cpp:66-76's loop collapsed by one more launch).  w2xc_opts.fusion = W2XC_FUSION_OFF disables.
bool fuse_first_fp32(const w2xc_model *m, const w2xc_opts &o)
{
    const int n = (int)m->layers.size();
    if (split_terms(o) != 0 || o.precision != W2XC_PRECISION_FP32 || n < 3 || o.fusion == W2XC_FUSION_OFF) return false;
    if (o.kernel != W2XC_KERNEL_AUTO && o.kernel != W2XC_KERNEL_WINOGRAD4) return false;
    const HostLayer &a = m->layers[0], &b = m->layers[1];
    if (!w2xc_first2_wino4_supported(a.nin, a.nout, b.nout) || b.nin != a.nout) return false;
    if (n == 4 && fuse_last_fp32(m, o)) return false;   // (layer 3 would carry the fused last layer: that instantiation reads 32 NHWC planes only)
    return is_wino4_layer(m, 2, o);   // (layer 3 = conv3x3_wino4: it reads layer 2's planar planes)
}

int launch_layer(DevCtx *c, const w2xc_model *m, int l, W2xcKernelKind kind, W2xcConvDesc d, hipStream_t st, const w2xc_opts &o)
{
    DevLayer &dl = c->layers[l];
    d.cin = m->layers[l].nin;
    d.cout = m->layers[l].nout;
    if (kind == W2XC_K_FUSED_AWAY) return W2XC_OK;   // computed by the next layer's W2XC_K_FIRST2_SPLIT / W2XC_K_FIRST2_WINO4 launch
    if (kind == W2XC_K_MID_SPLIT || kind == W2XC_K_FIRST2_SPLIT) {
        if (d.terms < 1 || d.terms > 3 || d.fmt < 0 || d.fmt > 1) return fail(W2XC_ERR_ARG, "bad term count %d / format %d", d.terms, d.fmt);
        const int wi = d.terms + 3 * d.fmt;
        if (!dl.w_split[wi]) {
            std::vector<float> pk((w2xc_split_packed_bytes(d.cin, d.cout, d.terms) + 3) / 4);
            dl.split_scale[wi] = w2xc_split_pack(d.cin, d.cout, d.terms, d.fmt, m->layers[l].w.data(), pk.data());
            int rc = upload(pk, &dl.w_split[wi]);
            if (rc) return rc;
        }
        d.wpk = dl.w_split[wi];
        d.acc_scale = 1.0f / dl.split_scale[wi];
        if (kind == W2XC_K_FIRST2_SPLIT) {
            d.w1pk = c->layers[l - 1].w_fast;
            d.bias1 = c->layers[l - 1].bias;
        }
        if (d.out_terms == 9) {   // the next (last) layer's weights ride along
            DevLayer &nl = c->layers[l + 1];
            const int nin = m->layers[l + 1].nin;
            const int lt = d.terms, li = d.terms == 3 ? 2 : d.terms == 1 ? 3 : d.fmt;   // the fused product uses the mode's own term count
            if (!nl.w_last_fused[li]) {
                std::vector<float> pk((w2xc_split_pack_last_bytes(nin, lt) + 3) / 4);
                nl.last_fused_scale[li] = w2xc_split_pack_last(nin, lt, d.fmt, m->layers[l + 1].w.data(), pk.data());
                int rc = upload(pk, &nl.w_last_fused[li]);
                if (rc) return rc;
            }
            d.w7pk = nl.w_last_fused[li];
            d.g_scale = 1.0f / nl.last_fused_scale[li];
        }
    } else if (kind == W2XC_K_LAST_GATHER) {
        d.wpk = nullptr;
    } else if (kind == W2XC_K_FIRST2_WINO4) {
        if (!dl.w_first2) {
            std::vector<float> pk((size_t)36 * d.cin * d.cout);
            w2xc_first2_wino4_pack(m->layers[l].w.data(), pk.data());
            int rc = upload(pk, &dl.w_first2);
            if (rc) return rc;
        }
        d.wpk = dl.w_first2;
        d.w1pk = c->layers[l - 1].w_fast;
        d.bias1 = c->layers[l - 1].bias;
    } else {
        d.wpk = kind == W2XC_K_DIRECT ? dl.w_direct : dl.w_fast;
    }
    const int midv = kind == W2XC_K_MFMA ? layer_mid_variant(m, l, o) : MID_MFMA;
    const bool wino = midv != MID_MFMA;
    if (wino) {
        float *&img = midv == MID_WINO4 ? dl.w_wino4 : dl.w_wino;
        if (!img) {
            std::vector<float> pk(midv == MID_WINO4 ? (size_t)36 * d.cin * d.cout : w2xc_wino_packed_floats(d.cin, d.cout));
            if (midv == MID_WINO4) w2xc_wino4_pack(d.cin, d.cout, m->layers[l].w.data(), pk.data());
            else w2xc_wino_pack(d.cin, d.cout, m->layers[l].w.data(), pk.data());
            int rc = upload(pk, &img);
            if (rc) return rc;
        }
        d.wpk = img;
        if (d.out_terms == 9) {   // the next (last) layer's weights ride along (fuse_last_fp32)
            DevLayer &nl = c->layers[l + 1];
            if (!nl.w_last_wino4) {
                std::vector<float> pk(w2xc_wino4_pack_last_floats(m->layers[l + 1].nin));
                w2xc_wino4_pack_last(m->layers[l + 1].nin, m->layers[l + 1].w.data(), pk.data());
                int rc = upload(pk, &nl.w_last_wino4);
                if (rc) return rc;
            }
            d.w7pk = nl.w_last_wino4;
        }
    }
    d.bias = dl.bias;
    ProfEvent ev;
    const bool profile = o.profile != 0;
    if (profile) { int rc = prof_begin(c, l, st, &ev); if (rc) return rc; }
    hipError_t e = kind == W2XC_K_MID_SPLIT     ? w2xc_launch_split_mid(d, st)
                   : kind == W2XC_K_FIRST_SPLIT ? w2xc_launch_split_first(d, st)
                   : kind == W2XC_K_LAST_GATHER ? w2xc_launch_last_gather(d, st)
                   : kind == W2XC_K_FIRST2_SPLIT ? w2xc_launch_first2_split(d, st)
                   : kind == W2XC_K_FIRST2_WINO4 ? w2xc_launch_first2_wino4(d, st)
                   : wino                        ? (midv == MID_WINO4 ? w2xc_launch_wino4(d, st) : w2xc_launch_wino(d, st))
                                                : w2xc_launch_conv(kind, d, st);
    if (e != hipSuccess) return fail(W2XC_ERR_HIP, "launch of %s (layer %d, %d->%d) failed: %s", w2xc_kernel_name(kind, d.cin, d.cout), l, d.cin, d.cout, hipGetErrorString(e));
    if (profile) {
        HIP_TRY(hipEventRecord(ev.b, st));
        c->pending.push_back(ev);
    }
    return W2XC_OK;
}

// Hooks of the host->host tile farm into the band loop (all optional; enqueue-only, never synchronise the device):
struct BandHooks {
    int out_chunk_rows = 0;                              // > 0: the last layer of a band is launched in row chunks of at most this size,
    int out_chunk_min = 0;                               //      tapering to this size at the end of the band (the exposed D2H tail)
    std::function<int(int, int)> input_needed;           // before layer 1 of band [y0, y1): make the launch stream wait for its input rows
    // layer 1 in row chunks while the band's input is still arriving: in_chunk(y0, y1) > 0 = output rows of layer 1 per chunk
    // (0: the band's rows are already staged / one launch); input_upto(v) = make the launch stream wait for view rows <= v
    std::function<int(int, int)> in_chunk;
    std::function<int(int)> input_upto;
    std::function<int(int, int)> prefetch;               // layers 1..n-1 of the current band are enqueued; [y0, y1) = the NEXT band
    std::function<int(int, int)> output_ready;           // output rows [r0, r1) have been enqueued on the launch stream
};

// Output rows [ra, rb) of convertWithModels on an h-row plane of which `d_in` holds rows
// [vy0, vy0+vh) -- every row in [ra-n, rb+n) clipped to the plane must be inside the view.
// `up` = 1 folds a nearest-neighbour 2x (main.cpp:132-140) into layer 1: vh, vy0, w, ra, rb are then in
// UPSCALED coordinates while d_in holds the (vh/2) x (w/2) source rows starting at source row vy0/2.
// Multi-plane form (w2xc_convert_planes_*): n_in planar input planes `in_cs` floats apart, ALL planes of the
// last layer written planar `out_cs` floats apart.  n_in == 1 && out_cs == 0 is convertWithModels proper,
// which returns only outputPlanes[0] (convertRoutine.cpp:78).
// plane_h = rows of the whole plane (the units of vh / vy0 / ra / rb), 0 = unknown.  With it, and a view that holds 4 n halo rows, the
// layers run on the banding-invariant geometry conv3x3_wino4 needs (below); without, W2XC_KERNEL_AUTO is refused (W2XC_ERR_ARG).
int run_rows(w2xc_model *m, DevCtx *c, const float *d_in, size_t in_stride_f, int vh, int vy0, int w, int ra, int rb,
             float *d_out, size_t out_stride_f, hipStream_t st, const w2xc_opts &o_in, int up = 0, int n_in = 1,
             long long in_cs = 0, long long out_cs = 0, const BandHooks *hk = nullptr, int plane_h = 0)
{
    const int n = (int)m->layers.size();
    if (n == 0) return fail(W2XC_ERR_ARG, "model has no layers");
    // conv3x3_wino4 (F(4x4,3x3)): an output of a 4x4 block depends, at rounding level, on all 36 patch values, so a block cut by the edge of a
    // band's region (clamped rows instead of the plane's) would make results depend on the banding.  HL = 4: every layer k < n computes the rows
    //     [floor4(y0) - 4 (n - k), ceil4(y1) + 4 (n - k))  clipped to the layer's plane extent [-(n - k), H + (n - k))
    // of a band [y0, y1) instead of [y0 - (n - k), y1 + (n - k)): every region edge that is not a plane edge is a block edge (blocks sit on rows
    // = 0 mod 4 of the plane), and layer k + 1 finds the rows it reads (one more each side) inside.  Costs up to 3 + 3 (n - k) more rows per side.
    w2xc_opts o = o_in;
    int HL = 1;
    if (uses_wino4(m, o)) {
        const int hs = 4 * n;
        if (plane_h > 0 && vy0 <= std::max(0, ra - hs) && vy0 + vh >= std::min(plane_h, rb + hs)) HL = 4;
        else if (o.kernel == W2XC_KERNEL_AUTO)   // a view with n halo rows only: no silent change of kernel (and rounding) -- the caller decides
            return fail(W2XC_ERR_ARG, "row-band view [%d,%d) of rows [%d,%d): the default F(4x4) kernel needs %d halo rows (4 per layer) for banding-invariant results; "
                                      "pass the wide view or choose w2xc_opts.kernel explicitly (W2XC_KERNEL_WINOGRAD32: F(2x2), banding-invariant on the minimum view)",
                        vy0, vy0 + vh, ra, rb, hs);
        // (an explicit W2XC_KERNEL_WINOGRAD4 on a narrow view runs as asked: results then depend on the banding at rounding level)
    }
    auto region = [&](int k, int y0, int y1, int &T, int &B) {   // plane rows [T, B) layer k computes for the band [y0, y1)
        if (HL == 1 || k == n) { T = y0 - (n - k); B = y1 + (n - k); return; }
        T = std::max(-(n - k), (y0 & ~3) - 4 * (n - k));
        B = std::min(plane_h + (n - k), ((y1 + 3) & ~3) + 4 * (n - k));
    };
    if (m->layers[0].nin != n_in)   // convertWithModelsBasic pushes exactly one plane (convertRoutine.cpp:63-64)
        return fail(W2XC_ERR_PLANES, "Error : Model-filter : \nnumber of input planes mismatch.\n%d,%d", n_in, m->layers[0].nin);
    const bool all_out = out_cs != 0;   // multi-plane output
    for (int l = 1; l < n; l++)
        if (m->layers[l].nin != m->layers[l - 1].nout)
            return fail(W2XC_ERR_PLANES, "Error : Model-filter : \nnumber of input planes mismatch.\n%d,%d", m->layers[l - 1].nout, m->layers[l].nin);
    const int T = split_terms(o);
    if (o.precision != W2XC_PRECISION_FP32 && T == 0) return fail(W2XC_ERR_ARG, "unknown precision %d", o.precision);
    if (T > 0)
        for (int l = 0; l < n; l++)
            if (layer_kind(m, l, o) == W2XC_K_DIRECT)
                return fail(W2XC_ERR_UNSUPPORTED, "16-bit precision modes: layer %d (%d->%d) has no kernel ({1,3}->{32,64,128} first, {32,64,128}->{32,64,128}, ->{1,3} last only)",
                            l + 1, m->layers[l].nin, m->layers[l].nout);
    // bytes per activation element of layer k's output (k = 1..n) in the workspace
    auto out_bpe = [&](int k) -> size_t {
        const int ot = out_terms_of(m, k - 1, o);
        return (ot >= 1 && ot <= 3) ? 2 * (size_t)ot : 4;
    };

    // the last layer stores straight into the caller's planar plane(s) when its kernel can address planar
    // output (conv3x3_last / conv3x3_direct); otherwise it goes through the NHWC workspace + a repack
    const W2xcKernelKind last_kind = layer_kind(m, n - 1, o);
    const bool last_direct = (m->layers[n - 1].nout == 1 || all_out) &&
                             (last_kind == W2XC_K_LAST || last_kind == W2XC_K_LAST_GATHER || last_kind == W2XC_K_DIRECT ||
                              (m->layers[n - 1].nout == 1 && last_kind != W2XC_K_MFMA && last_kind != W2XC_K_FIRST));
    // BYTES per band for the two ping-pong buffers (layer k output goes to ws[(k-1)&1])
    auto ws_need = [&](int rows, size_t need[2]) {
        need[0] = need[1] = 0;
        for (int k = 1; k <= n; k++) {
            if (k == n && last_direct) break;   // written straight to d_out
            if (k == 1 && layer_kind(m, 0, o) == W2XC_K_FUSED_AWAY) continue;   // layer 1's activations stay on chip
            const size_t hk = (size_t)rows + ((HL == 1 || k == n) ? 2 * (n - k) : 6 + 8 * (n - k)), wk = (size_t)w + 2 * (n - k);
            const bool fused = out_terms_of(m, k - 1, o) == 9;   // partial G planes of the fused last layer
            const size_t px_bytes = fused ? (size_t)fused_halves(T, m->layers[k - 1].nout) * 9 * 4 : m->layers[k - 1].nout * out_bpe(k);
            const size_t wk_mem = planar_between(m, k - 1, o) ? ((wk + 31) & ~(size_t)31) : wk;   // planar rows start on 128-byte lines
            need[(k - 1) & 1] = std::max(need[(k - 1) & 1], hk * wk_mem * px_bytes);
        }
    };
    int band = o.band_rows;
    const int total = rb - ra;
    if (band <= 0) {
        const size_t budget = (size_t)(o.workspace_mb > 0 ? o.workspace_mb : 16384) << 20;
        size_t need[2];
        ws_need(total, need);
        if (need[0] + need[1] <= budget) band = total;
        else {
            // bytes grow linearly in rows: solve on two probes
            size_t n1[2], n2[2];
            ws_need(1, n1);
            ws_need(2, n2);
            const double per_row = (double)((n2[0] + n2[1]) - (n1[0] + n1[1]));
            const double base = (double)(n1[0] + n1[1]) - per_row;
            band = (int)std::floor(((double)budget - base) / per_row);
            if (band < 1) band = 1;
            if (band > total) band = total;
            // each buffer's need is a MAX over layers, so the slope measured at 1..2 rows is that of the wide-halo,
            // few-plane layers and under-estimates large bands: re-evaluate the real need and shrink until it fits
            for (int it = 0; it < 64 && band > 1; it++) {
                ws_need(band, need);
                if (need[0] + need[1] <= budget) break;
                const int nb2 = (int)((double)band * (double)budget / (double)(need[0] + need[1]));
                band = std::max(1, std::min(band - 1, nb2));
            }
            const int nb = (total + band - 1) / band;
            band = (total + nb - 1) / nb;   // equalise (never larger than the band that was just checked)
        }
    }
    if (fuse_first_fp32(m, o)) {
        // conv3x3_first2_wino4 addresses its 32 output planes with 32-bit lane offsets (12 plane strides + a row): planes of at most 64 Mi floats
        const size_t wk = ((size_t)w + 2 * (n - 2) + 31) & ~(size_t)31;
        const size_t max_rows = ((size_t)64 << 20) / wk;
        const size_t halo = HL == 1 ? 2 * (size_t)(n - 2) : 6 + 8 * (size_t)(n - 2);
        if (max_rows < halo + 8) return fail(W2XC_ERR_UNSUPPORTED, "plane too wide (%d pixels) for the fused first layers; use w2xc_opts.fusion = W2XC_FUSION_OFF", w);
        if ((size_t)band + halo > max_rows) band = (int)(max_rows - halo);
    }
    if (HL > 1 && band < total) band = std::max(4, band & ~3);   // (band edges on block rows: no rounding-out rows)
    band = std::min(band, total);
    {
        size_t need[2];
        ws_need(band, need);
        for (int i = 0; i < 2; i++)
            if (need[i]) { int rc = ensure_ws(c, i, (need[i] + 3) / 4); if (rc) return rc; }
    }

    for (int y0 = ra; y0 < rb; y0 += band) {
        const int y1 = std::min(rb, y0 + band);
        // layer 1 of this band in row chunks (each waits only for the rows it reads) or in one launch behind the whole upload
        const W2xcKernelKind kind1 = layer_kind(m, 0, o);
        const bool first2_fp32 = kind1 == W2XC_K_FUSED_AWAY && n > 1 && layer_kind(m, 1, o) == W2XC_K_FIRST2_WINO4;   // (layers 1 + 2 in one launch: chunked like layer 1)
        const int in_chunk = (hk && hk->in_chunk && hk->input_upto && n > 1 && (kind1 == W2XC_K_FIRST || kind1 == W2XC_K_DIRECT || first2_fp32)) ? hk->in_chunk(y0, y1) : 0;
        if (hk && hk->input_needed && in_chunk <= 0) { int rc = hk->input_needed(y0, y1); if (rc) return rc; }
        const float *src = d_in;
        long long src_rs = (long long)in_stride_f, src_ps = 1, src_cs = in_cs, src_ts = 0, src_gs = 0;
        int src_halves = 0;
        W2xcConvDesc first_d;
        memset(&first_d, 0, sizeof first_d);
        int src_h = vh, src_w = w;
        int Tprev = vy0;   // first plane row held by the buffer layer k reads (the source view for k = 1)
        for (int k = 1; k <= n; k++) {
            if (o.verbose & 1) std::cout << "Iteration #" << k << "..." << std::endl;   // convertRoutine.cpp:67
            const HostLayer &hl = m->layers[k - 1];
            W2xcConvDesc d;
            memset(&d, 0, sizeof d);
            d.in = src; d.in_rs = src_rs; d.in_ps = src_ps; d.in_cs = src_cs;
            d.in_h = src_h; d.in_w = src_w;
            int Tk, Bk;
            region(k, y0, y1, Tk, Bk);
            d.out_h = Bk - Tk;
            d.out_w = w + 2 * (n - k);
            d.off_y = Tk - 1 - Tprev;   // (k = 1: y0 - n - vy0; k > 1: 0 on the one-row-per-layer geometry)
            Tprev = Tk;
            d.off_x = k == 1 ? -n : 0;
            // this launch's first output row in the coordinates of the whole plane, modulo the Winograd block height (2; conv3x3_wino4: 4)
            const W2xcKernelKind kind = layer_kind(m, k - 1, o);
            d.wino_py = Tk & (((w2xc_pick_kernel(hl.nin, hl.nout) == W2XC_K_MFMA && layer_mid_variant(m, k - 1, o) == MID_WINO4) || kind == W2XC_K_FIRST2_WINO4) ? 3 : 1);
            d.in_shift = k == 1 ? up : 0;
            if (kind == W2XC_K_FUSED_AWAY) {   // layer 1 inside layer 2's kernel: keep its input description for that launch
                first_d = d;
                continue;
            }
            if (kind == W2XC_K_FIRST2_SPLIT) {
                d.in = first_d.in; d.in_rs = first_d.in_rs; d.in_ps = first_d.in_ps; d.in_cs = first_d.in_cs;
                d.in_h = first_d.in_h; d.in_w = first_d.in_w;
                d.off_y = first_d.off_y; d.off_x = first_d.off_x; d.in_shift = first_d.in_shift;
            }
            if (kind == W2XC_K_FIRST2_WINO4) {   // layer 1's input view; a source row of layer 2's output row y, taps r' and r: y + r' + r + (both offsets)
                d.in = first_d.in; d.in_rs = first_d.in_rs; d.in_ps = first_d.in_ps; d.in_cs = first_d.in_cs;
                d.in_h = first_d.in_h; d.in_w = first_d.in_w;
                d.off_y += first_d.off_y; d.off_x += first_d.off_x; d.in_shift = first_d.in_shift;
            }
            int split_grp = 0;
            if (T == 0) {   // fp32: only the fused last layer uses the term fields
                d.out_terms = out_terms_of(m, k - 1, o);
                if (kind == W2XC_K_LAST_GATHER) {
                    d.halves = src_halves; d.in_ts = src_ts; d.in_gs = src_gs;
                    d.in += (long long)d.off_y * d.in_rs;   // (no offsets in that kernel; off_y > 0 on the four-rows-per-layer geometry only)
                    d.in_h -= d.off_y;
                    d.off_y = 0;
                }
            }
            if (T > 0) {
                d.terms = (kind == W2XC_K_MID_SPLIT || kind == W2XC_K_FIRST2_SPLIT) ? T : 0;
                if (kind == W2XC_K_LAST_GATHER) d.halves = src_halves;
                d.fmt = split_fmt(o);
                d.in_ts = src_ts;
                d.out_terms = out_terms_of(m, k - 1, o);
                d.out_ts = (long long)d.out_h * d.out_w * hl.nout;
                d.in_gs = src_gs;
                split_grp = 16;                                         // channel-group size of the blocked term planes
                d.out_gs = (long long)d.out_h * d.out_w * split_grp;
            }
            const bool direct_out = (k == n && last_direct);
            if (direct_out) {
                d.out = d_out + (size_t)(y0 - ra) * out_stride_f;
                d.out_rs = (long long)out_stride_f; d.out_ps = 1; d.out_cs = out_cs;
            } else {
                d.out = c->ws[(k - 1) & 1];
                d.out_rs = (long long)d.out_w * hl.nout; d.out_ps = hl.nout; d.out_cs = 1;
                if (planar_between(m, k - 1, o)) {
                    // planes of out_h rows of roundup32(out_w) floats: conv3x3_wino4 reads 16-byte pixel quads, and a tile's 32-pixel row segment
                    // (tiles start at multiples of 32 pixels) is then ONE 128-byte line -- with rows of roundup4(w) floats every segment straddled two
                    // lines, each written in two pieces by different workgroups (the 32 -> 32 layer in front: 2.0 ms instead of 0.8, measured)
                    d.out_rs = (d.out_w + 31) & ~31; d.out_ps = 1; d.out_cs = d.out_rs * (long long)d.out_h;
                }
                if (T > 0 && d.out_terms >= 1 && d.out_terms <= 3) { d.out_rs = (long long)d.out_w * split_grp; d.out_ps = split_grp; }
                if (d.out_terms == 9) {   // G[half][tap][y][x]
                    d.out_rs = d.out_w; d.out_ps = 1;
                    d.out_gs = (long long)d.out_h * d.out_w;
                    d.out_ts = 9 * d.out_gs;
                    d.halves = fused_halves(T, hl.nout);   // (fp32: conv3x3_wino4 writes planar partial planes G[64-plane block][tap][y][x]: its epilogue sums the four plane tiles of a block on chip)
                }
            }
            // 16-bit modes, host pipeline: the last layer lives in layer n-1's epilogue + a 0.2 ms gather, too short to hide the
            // band's download behind.  So layer n-1 and the gather run TOGETHER in row chunks (quarters of the band, whole 16-row
            // tiles): chunk j's rows leave for the host under layer n-1 of chunk j+1.  The producer chunks tile the G rows without
            // overlap (chunk j computes G rows up to r1 + 2, the next one continues there): no recompute.
            // (16-bit producers only: conv3x3_wino4's fused epilogue wants chunks on whole 16-row tiles of ITS block grid -- the tail path below)
            if (hk && HL == 1 && k == n - 1 && n >= 3 && kind == W2XC_K_MID_SPLIT && d.out_terms == 9 && last_direct && hk->out_chunk_rows > 0 &&
                hk->output_ready && (y1 - y0) >= 128) {
                if (hk->prefetch && y1 < rb) { int rc = hk->prefetch(y1, std::min(rb, y1 + band)); if (rc) return rc; }
                const int R = y1 - y0;
                const int cr = std::max(64, ((R / 4) + 15) & ~15);
                int g_done = 0;
                for (int r0 = 0; r0 < R;) {
                    int r1 = std::min(R, r0 + cr);
                    if (R - r1 < 32) r1 = R;
                    const int g1 = r1 + 2;                       // the gather of rows [r0, r1) reads G rows [r0, r1 + 2)
                    W2xcConvDesc dd = d;
                    dd.out_h = g1 - g_done;
                    dd.off_y = d.off_y + g_done;
                    dd.out = d.out + (size_t)g_done * d.out_rs;  // (plane / half strides stay those of the whole band)
                    int rc = launch_layer(c, m, k - 1, kind, dd, st, o);
                    if (rc) return rc;
                    g_done = g1;
                    // the gather of the chunk's rows; the LAST chunk's gather in pieces of ~128 rows, each handed to the download as soon as it
                    // is enqueued: what nothing can hide is then the download + stitch of the last ~2 MB piece, not of the whole last chunk
                    const int piece = (r1 == R && r1 - r0 > 192) ? 128 : r1 - r0;
                    for (int a = r0; a < r1;) {
                        int b = std::min(r1, a + piece);
                        if (r1 - b < 64) b = r1;
                        W2xcConvDesc dg;
                        memset(&dg, 0, sizeof dg);
                        dg.in = d.out + (size_t)a * d.out_rs; dg.in_rs = d.out_rs; dg.in_ps = d.out_ps; dg.in_cs = d.out_cs;
                        dg.in_ts = d.out_ts; dg.in_gs = d.out_gs; dg.halves = d.halves; dg.fmt = d.fmt;
                        dg.in_h = b - a + 2; dg.in_w = d.out_w;
                        dg.out_h = b - a; dg.out_w = w;
                        dg.out = d_out + (size_t)(y0 - ra + a) * out_stride_f;
                        dg.out_rs = (long long)out_stride_f; dg.out_ps = 1; dg.out_cs = out_cs;
                        rc = launch_layer(c, m, n - 1, W2XC_K_LAST_GATHER, dg, st, o);
                        if (rc) return rc;
                        rc = hk->output_ready(y0 + a, y0 + b);
                        if (rc) return rc;
                        a = b;
                    }
                    r0 = r1;
                }
                break;
            }
            // fp32, host pipeline, last layer NOT fused: the last layer (0.8 ms on the 2160x3840 frame) is too short to hide the band's 33 MB download +
            // stitch behind.  So layer n-1 and the last layer run TOGETHER in row chunks: chunk j's output rows leave for the host under layer n-1 of
            // chunk j+1.  The producer chunks are whole 16-row tiles of the SAME tile grid as the unchunked launch (bit-identical results, nothing is
            // computed twice); the last layer follows two rows behind (it reads rows y .. y + 2 of the producer's region).
            const bool tail_unfused = d.out_terms == 0 && last_kind == W2XC_K_LAST;
            const bool tail_fused4 = d.out_terms == 9 && last_kind == W2XC_K_LAST_GATHER && is_wino4_layer(m, k - 1, o);   // (conv3x3_wino4's fused epilogue + gather)
            if (hk && T == 0 && k == n - 1 && n >= 2 && kind == W2XC_K_MFMA && (tail_unfused || tail_fused4) && last_direct &&
                hk->out_chunk_rows > 0 && hk->output_ready && (y1 - y0) >= 256) {
                if (hk->prefetch && y1 < rb) { int rc = hk->prefetch(y1, std::min(rb, y1 + band)); if (rc) return rc; }
                W2xcConvDesc dl;
                memset(&dl, 0, sizeof dl);
                dl.in = d.out; dl.in_rs = d.out_rs; dl.in_ps = d.out_ps; dl.in_cs = d.out_cs;
                dl.in_h = d.out_h; dl.in_w = d.out_w;
                dl.out_w = w;
                dl.out_rs = (long long)out_stride_f; dl.out_ps = 1; dl.out_cs = out_cs;
                const int off_l = y0 - 1 - Tk;   // rows of the producer's region above the last layer's first input row (0 on the one-row-per-layer geometry)
                const int RL = d.out_h, R = y1 - y0;
                // three producer launches -- 1/2, then 5/16, then the rest -- of whole 16-row tiles: every launch of the persistent kernel has a ramp and a tail
                // (measured: four equal chunks cost layer 6 +0.6 ms on the 2160x3840 frame), while what the LAST chunk writes cannot hide behind compute
                // ... and a launch whose item count is not a multiple of the 256 persistent workgroups ends with a partly filled round: among the tile-row
                // counts within 8 of the wanted one, take the one that wastes the fewest workgroup slots (2160x3840, two 64-plane blocks: 64 + 48 + 24
                // tile rows = 60 + 45 + 22.5 rounds against 63.75 + 40.3 + 23.4 for exact halves)
                const int items_per_row = ((d.out_w + 31) / 32) * std::max(1, hl.nout / 64);
                auto chunk_rows = [&](int want) {
                    int best = std::max(4, (want + 15) / 16), waste = 1 << 30;
                    for (int r = std::max(4, (want + 15) / 16 - 8); r <= (want + 15) / 16 + 8; r++) {
                        const int items = items_per_row * r, w_ = ((items + 255) / 256) * 256 - items;
                        if (w_ < waste || (w_ == waste && std::abs(r * 16 - want) < std::abs(best * 16 - want))) { waste = w_; best = r; }
                    }
                    return best * 16;
                };
                for (int p0 = 0, o0 = 0, ci = 0; p0 < RL; ci++) {
                    const int want = ci == 0 ? RL / 2 : ci == 1 ? (RL * 5) / 16 : RL;
                    int p1 = ci < 2 ? std::min(RL, p0 + chunk_rows(want)) : RL;
                    if (RL - p1 < 64) p1 = RL;
                    W2xcConvDesc dd = d;
                    dd.out_h = p1 - p0;
                    dd.off_y = d.off_y + p0;
                    dd.out = d.out + (size_t)p0 * d.out_rs;
                    int rc = launch_layer(c, m, k - 1, kind, dd, st, o);
                    if (rc) return rc;
                    const int o1 = p1 == RL ? R : std::min(R, std::max(o0, p1 - off_l - 2));   // output rows whose three input rows exist
                    // the LAST chunk's rows in pieces of ~128, its last 128 in pieces of 64: what nothing can hide is then the download + stitch of the last piece only
                    const int piece = (p1 == RL && o1 - o0 > 192) ? 128 : std::max(o1 - o0, 1);
                    for (int a = o0; a < o1;) {
                        int b = std::min(o1, a + ((p1 == RL && o1 - a <= 160 && o1 - a > 96) ? 64 : piece));
                        if (o1 - b < 48) b = o1;
                        W2xcConvDesc dg = dl;
                        dg.out_h = b - a;
                        dg.off_y = off_l + a;
                        dg.out = d_out + (size_t)(y0 - ra + a) * out_stride_f;
                        if (tail_fused4) {   // the gather has no offsets: its input view starts at the partial planes' row off_l + a
                            dg.in = d.out + (size_t)(off_l + a) * d.out_rs;
                            dg.in_ts = d.out_ts; dg.in_gs = d.out_gs; dg.halves = d.halves;
                            dg.in_h = b - a + 2;
                            dg.off_y = 0;
                        }
                        rc = launch_layer(c, m, n - 1, last_kind, dg, st, o);
                        if (rc) return rc;
                        rc = hk->output_ready(y0 + a, y0 + b);
                        if (rc) return rc;
                        a = b;
                    }
                    p0 = p1;
                    o0 = o1;
                }
                break;
            }
            if (hk && k == n && hk->prefetch && y1 < rb) {   // stage the next band's input while this one computes
                int rc = hk->prefetch(y1, std::min(rb, y1 + band));
                if (rc) return rc;
            }
            if ((k == 1 || kind == W2XC_K_FIRST2_WINO4) && in_chunk > 0) {
                // the upload of rows [c0 + 2 + off_y ...] and layer 1 of the rows before them overlap: what stays exposed of the
                // input side is the first slice and the last chunk, not upload + layer 1 back to back.  (Layers 1 + 2 in one launch: the same,
                // two rows deeper; chunks of whole 8-row tiles keep the 4x4 blocks where the unchunked launch has them.)
                const int reach = kind == W2XC_K_FIRST2_WINO4 ? 4 : 2;
                for (int c0 = 0; c0 < d.out_h; c0 += in_chunk) {
                    W2xcConvDesc dd = d;
                    dd.out_h = std::min(in_chunk, d.out_h - c0);
                    dd.out = d.out + (size_t)c0 * d.out_rs;
                    dd.off_y = d.off_y + c0;
                    const int vlast = std::min(std::max(c0 + dd.out_h - 1 + reach + d.off_y, 0), d.in_h - 1);   // last view row this chunk reads
                    int rc = hk->input_upto(vlast);
                    if (rc) return rc;
                    rc = launch_layer(c, m, k - 1, kind, dd, st, o);
                    if (rc) return rc;
                }
                src = d.out; src_rs = d.out_rs; src_ps = d.out_ps; src_cs = d.out_cs; src_ts = d.out_ts; src_gs = d.out_gs; src_halves = d.halves;
                src_h = d.out_h; src_w = d.out_w;
                continue;
            }
            const bool chunked = hk && k == n && direct_out && hk->out_chunk_rows > 0 && d.out_h > std::max(hk->out_chunk_min, 8) &&
                                 (kind == W2XC_K_LAST || kind == W2XC_K_LAST_GATHER || kind == W2XC_K_DIRECT);
            if (chunked) {
                // the last layer in row chunks: chunk j's rows leave for the host while chunk j+1 is computed
                for (int c0 = 0, cr = 0; c0 < d.out_h; c0 += cr) {
                    // a third of what is left, within [min, max], in whole 8-row tiles: big chunks while there is compute
                    // left to hide their D2H behind, small ones at the end where the D2H is exposed
                    const int left = d.out_h - c0;
                    cr = std::min(hk->out_chunk_rows, std::max(std::max(hk->out_chunk_min, 8), ((left / 3) + 7) & ~7));
                    if (left - cr < std::max(hk->out_chunk_min, 8)) cr = left;
                    W2xcConvDesc dd = d;
                    dd.out_h = cr;
                    dd.out = d.out + (size_t)c0 * d.out_rs;
                    if (kind == W2XC_K_LAST_GATHER) dd.in = d.in + (size_t)c0 * d.in_rs;   // no offsets in that kernel
                    else dd.off_y = d.off_y + c0;
                    int rc = launch_layer(c, m, k - 1, kind, dd, st, o);
                    if (rc) return rc;
                    if (hk->output_ready) { rc = hk->output_ready(y0 + c0, y0 + c0 + dd.out_h); if (rc) return rc; }
                }
                break;
            }
            int rc = launch_layer(c, m, k - 1, kind, d, st, o);
            if (rc) return rc;
            if (hk && k == n && direct_out && hk->output_ready) { rc = hk->output_ready(y0, y1); if (rc) return rc; }
            if (k == n && !direct_out) {
                // outputPlanes[0] of a multi-plane last layer (convertRoutine.cpp:78)
                hipError_t e = w2xc_launch_repack(d.out, d.out_rs, d.out_ps, 1, d_out + (size_t)(y0 - ra) * out_stride_f,
                                                  (long long)out_stride_f, 1, out_cs, d.out_h, d.out_w, all_out ? hl.nout : 1, st);
                if (e != hipSuccess) return fail(W2XC_ERR_HIP, "repack launch failed: %s", hipGetErrorString(e));
                if (hk && hk->output_ready) { rc = hk->output_ready(y0, y1); if (rc) return rc; }
            }
            src = d.out; src_rs = d.out_rs; src_ps = d.out_ps; src_cs = d.out_cs; src_ts = d.out_ts; src_gs = d.out_gs; src_halves = d.halves;
            src_h = d.out_h; src_w = d.out_w;
        }
    }
    return W2XC_OK;
}

int check_plane_args(const w2xc_model *m, const void *in, size_t in_stride, int w, int h, const void *out, size_t out_stride)
{
    if (!m || !in || !out) return fail(W2XC_ERR_ARG, "null argument");
    if (w <= 0 || h <= 0) return fail(W2XC_ERR_ARG, "plane size must be positive (got %dx%d)", w, h);
    if (in_stride < (size_t)w * 4 || out_stride < (size_t)w * 4 || (in_stride & 3) || (out_stride & 3))
        return fail(W2XC_ERR_ARG, "row strides must be multiples of 4 bytes and >= 4*w");
    return W2XC_OK;
}

}  // namespace

// ================================================================================================
extern "C" {

void w2xc_opts_init(w2xc_opts *o)
{
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->struct_size = (int)sizeof(w2xc_opts);
    o->precision = W2XC_PRECISION_FP32;
    o->kernel = W2XC_KERNEL_AUTO;
    o->device = -1;
}

const char *w2xc_last_error(void) { return g_last_error.c_str(); }
const char *w2xc_version(void) { return "w2xc_hip 0.1 (gfx950)"; }

int w2xc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- model container ---------------------------------------------------------------------------
int w2xc_model_from_arrays(int n_layers, const int *nin, const int *nout, const float *const *weight,
                           const double *const *bias, w2xc_model **out)
try {
    if (!out || n_layers <= 0 || !nin || !nout || !weight || !bias) return fail(W2XC_ERR_ARG, "bad argument");
    std::unique_ptr<w2xc_model> m(new w2xc_model());
    m->layers.resize(n_layers);
    for (int l = 0; l < n_layers; l++) {
        if (nin[l] <= 0 || nout[l] <= 0 || !weight[l] || !bias[l]) return fail(W2XC_ERR_ARG, "bad layer %d", l);
        HostLayer &hl = m->layers[l];
        hl.nin = nin[l];
        hl.nout = nout[l];
        hl.w.assign(weight[l], weight[l] + (size_t)nin[l] * nout[l] * 9);
        hl.bias.assign(bias[l], bias[l] + nout[l]);
    }
    *out = m.release();
    return W2XC_OK;
} catch (const std::bad_alloc &) {
    return fail(W2XC_ERR_NOMEM, "out of memory while copying the model");
}

static int model_load_json_impl(const char *path, w2xc_model **out)
{
    if (!path || !out) return fail(W2XC_ERR_ARG, "null argument");
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) {
        std::cerr << "Error : couldn't open " << path << std::endl;   // modelHandler.cpp:176-178
        return fail(W2XC_ERR_IO, "Error : couldn't open %s", path);
    }
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string text = ss.str();   // c_str() is NUL-terminated: strtod cannot overrun
    jsonmin::Value root;
    std::string err;
    jsonmin::Parser p(text.c_str(), text.c_str() + text.size());
    if (!p.parse(root, err)) {
        std::cerr << "Error : JSON Error : " << err << std::endl;      // modelHandler.cpp:183-186
        return fail(W2XC_ERR_JSON, "Error : JSON Error : %s", err.c_str());
    }
    if (!root.is_array() || root.arr.empty()) return fail(W2XC_ERR_JSON, "model file is not a non-empty JSON array of layers");
    std::unique_ptr<w2xc_model> m(new w2xc_model());
    for (size_t l = 0; l < root.arr.size(); l++) {   // one Model per element (:189-194)
        const jsonmin::Value &o = root.arr[l];
        if (!o.is_object()) return fail(W2XC_ERR_JSON, "layer %zu is not an object", l);
        const jsonmin::Value *nip = o.find("nInputPlane"), *nop = o.find("nOutputPlane"), *kw = o.find("kW"),
                             *kh = o.find("kH"), *wv = o.find("weight"), *bv = o.find("bias");
        if (!nip || !nop || !kw || !kh || !wv || !bv || !nip->is_number() || !nop->is_number() || !kw->is_number() ||
            !kh->is_number() || !wv->is_array() || !bv->is_array())
            return fail(W2XC_ERR_JSON, "layer %zu lacks one of nInputPlane/nOutputPlane/kW/kH/weight/bias", l);
        HostLayer hl;
        // the reference casts these doubles straight to int (modelHandler.hpp:50-51); a library first makes sure the cast
        // is defined and the sizes are sane (NaN / 1e300 / a hostile plane count must not reach resize())
        auto small_int = [](double v) { return std::isfinite(v) && v >= 0.0 && v <= 65536.0; };
        if (!small_int(nip->num) || !small_int(nop->num) || !small_int(kw->num) || !small_int(kh->num))
            return fail(W2XC_ERR_JSON, "layer %zu: nInputPlane / nOutputPlane / kW / kH out of range", l);
        hl.nin = (int)nip->num;     // static_cast<int>(double), modelHandler.hpp:50-51
        hl.nout = (int)nop->num;
        const int ks = (int)kw->num;
        if (ks != (int)kh->num) {   // the reference exit(-1)s here (hpp:52-58); a library reports it
            std::cerr << "Error : Model-Constructor : \nkernel in model is not square.\nstop." << std::endl;
            return fail(W2XC_ERR_UNSUPPORTED, "kernel in model is not square");
        }
        if (ks != 3) return fail(W2XC_ERR_UNSUPPORTED, "layer %zu: kernel size %d; only 3x3 is supported (convertWithModels pads by the layer count, which assumes 3x3)", l, ks);
        if (hl.nin <= 0 || hl.nout <= 0 || hl.nin > 4096 || hl.nout > 4096) return fail(W2XC_ERR_JSON, "layer %zu: bad plane counts (%d, %d)", l, hl.nin, hl.nout);
        if ((int)wv->arr.size() != hl.nout || (int)bv->arr.size() < hl.nout)
            return fail(W2XC_ERR_JSON, "layer %zu: weight/bias outer size does not match nOutputPlane", l);
        hl.w.resize((size_t)hl.nout * hl.nin * 9);
        hl.bias.resize(hl.nout);
        for (int oo = 0; oo < hl.nout; oo++) {
            const jsonmin::Value &wi = wv->arr[oo];
            if (!wi.is_array() || (int)wi.arr.size() != hl.nin) return fail(W2XC_ERR_JSON, "layer %zu: weight[%d] size != nInputPlane", l, oo);
            for (int i = 0; i < hl.nin; i++) {
                const jsonmin::Value &km = wi.arr[i];
                if (!km.is_array() || (int)km.arr.size() < ks) return fail(W2XC_ERR_JSON, "layer %zu: weight[%d][%d] is not a %dx%d matrix", l, oo, i, ks, ks);
                for (int r = 0; r < ks; r++) {
                    const jsonmin::Value &row = km.arr[r];
                    if (!row.is_array() || (int)row.arr.size() < ks) return fail(W2XC_ERR_JSON, "layer %zu: weight[%d][%d][%d] too short", l, oo, i, r);
                    for (int cidx = 0; cidx < ks; cidx++) {
                        if (!row.arr[cidx].is_number()) return fail(W2XC_ERR_JSON, "layer %zu: non-numeric weight", l);
                        hl.w[((size_t)oo * hl.nin + i) * 9 + r * 3 + cidx] = (float)row.arr[cidx].num;   // double -> float, :95-97
                    }
                }
            }
            if (!bv->arr[oo].is_number()) return fail(W2XC_ERR_JSON, "layer %zu: non-numeric bias", l);
            hl.bias[oo] = bv->arr[oo].num;   // stays double, :109-112
        }
        m->layers.push_back(std::move(hl));
    }
    *out = m.release();
    return W2XC_OK;
}

// no C++ exception may cross the C ABI: a hostile / truncated model file or an allocation failure becomes an error code
int w2xc_model_load_json(const char *path, w2xc_model **out)
{
    try {
        return model_load_json_impl(path, out);
    } catch (const std::bad_alloc &) {
        return fail(W2XC_ERR_NOMEM, "out of memory while loading %s", path ? path : "(null)");
    } catch (const std::exception &e) {
        return fail(W2XC_ERR_JSON, "Error : JSON Error : %s", e.what());
    } catch (...) {
        return fail(W2XC_ERR_JSON, "unknown error while loading the model");
    }
}

void w2xc_model_free(w2xc_model *m) { delete m; }

int w2xc_model_trim(w2xc_model *m)
{
    if (!m) return fail(W2XC_ERR_ARG, "null model");
    std::lock_guard<std::mutex> lk(m->mu);
    int prev = 0;
    hipGetDevice(&prev);
    for (auto &kv : m->ctx) {
        DevCtx *c = kv.second.get();
        std::lock_guard<std::mutex> lk2(c->mu);
        hipSetDevice(c->device);
        hipDeviceSynchronize();
        for (int i = 0; i < 2; i++) {
            if (c->ws[i]) { hipFree(c->ws[i]); c->ws[i] = nullptr; c->ws_floats[i] = 0; }
            if (c->fc.planar[i]) { hipFree(c->fc.planar[i]); c->fc.planar[i] = nullptr; c->fc.planar_floats[i] = 0; }
            if (c->fc.nhwc[i]) { hipFree(c->fc.nhwc[i]); c->fc.nhwc[i] = nullptr; c->fc.nhwc_floats[i] = 0; }
            if (i == 0 && c->fc.pad) { hipFree(c->fc.pad); c->fc.pad = nullptr; c->fc.pad_floats = 0; }
            if (i == 0 && c->fc.pout) { hipFree(c->fc.pout); c->fc.pout = nullptr; c->fc.pout_floats = 0; }
        }
        c->fc.res_valid = false;
        if (c->aux) { hipFree(c->aux); c->aux = nullptr; c->aux_floats = 0; }
        if (c->img_io) { hipFree(c->img_io); c->img_io = nullptr; c->img_io_bytes = 0; }
        HostPipe &p = c->pipe;
        if (p.d_in) { hipFree(p.d_in); p.d_in = nullptr; p.d_in_bytes = 0; }
        if (p.d_out) { hipFree(p.d_out); p.d_out = nullptr; p.d_out_bytes = 0; }
        if (p.pin_in) { hipHostFree(p.pin_in); p.pin_in = nullptr; p.in_slot_bytes = 0; }
        if (p.pin_out) { hipHostFree(p.pin_out); p.pin_out = nullptr; p.out_slot_bytes = 0; }
    }
    hipSetDevice(prev);
    return W2XC_OK;
}
int w2xc_model_layers(const w2xc_model *m) { return m ? (int)m->layers.size() : 0; }
int w2xc_model_nin(const w2xc_model *m, int l) { return (m && l >= 0 && l < (int)m->layers.size()) ? m->layers[l].nin : -1; }
int w2xc_model_nout(const w2xc_model *m, int l) { return (m && l >= 0 && l < (int)m->layers.size()) ? m->layers[l].nout : -1; }

int w2xc_model_get_layer(const w2xc_model *m, int l, float *weight, double *bias)
{
    if (!m || l < 0 || l >= (int)m->layers.size()) return fail(W2XC_ERR_ARG, "bad layer index");
    const HostLayer &hl = m->layers[l];
    if (weight) memcpy(weight, hl.w.data(), hl.w.size() * sizeof(float));
    if (bias) memcpy(bias, hl.bias.data(), hl.bias.size() * sizeof(double));
    return W2XC_OK;
}

// ---- modelUtility knobs --------------------------------------------------------------------------
int w2xc_set_jobs(int n)
{
    if (n < 1) return W2XC_ERR_ARG;   // modelHandler.cpp:200
    std::lock_guard<std::mutex> lk(g_util_mu);
    g_njob = n;
    return W2XC_OK;
}
int w2xc_get_jobs(void) { std::lock_guard<std::mutex> lk(g_util_mu); return g_njob; }
int w2xc_set_block_size(int w, int h)
{
    if (w < 0 || h < 0) return W2XC_ERR_ARG;   // :210
    std::lock_guard<std::mutex> lk(g_util_mu);
    g_block_w = w; g_block_h = h;
    return W2XC_OK;
}
int w2xc_set_block_size_exp2(int exp)
{
    if (exp < 0 || exp > 30) return W2XC_ERR_ARG;   // :216
    std::lock_guard<std::mutex> lk(g_util_mu);
    g_block_w = g_block_h = 1 << exp;
    return W2XC_OK;
}
void w2xc_get_block_size(int *w, int *h)
{
    std::lock_guard<std::mutex> lk(g_util_mu);
    if (w) *w = g_block_w;
    if (h) *h = g_block_h;
}

// ---- hot path -------------------------------------------------------------------------------------
int w2xc_convert_plane_device(w2xc_model *m, const float *d_in, size_t in_stride_bytes, int w, int h, float *d_out,
                              size_t out_stride_bytes, void *hip_stream, const w2xc_opts *opts)
try {
    int rc = check_plane_args(m, d_in, in_stride_bytes, w, h, d_out, out_stride_bytes);
    if (rc) return rc;
    const w2xc_opts o = resolve_opts(opts);
    int dev = o.device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    DeviceGuard guard(dev);
    if (!guard.ok) return fail(W2XC_ERR_HIP, "cannot select HIP device %d", dev);
    DevCtx *c = nullptr;
    rc = get_ctx(m, dev, &c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    return run_rows(m, c, d_in, in_stride_bytes / 4, h, 0, w, 0, h, d_out, out_stride_bytes / 4, (hipStream_t)hip_stream, o, 0, 1, 0, 0, nullptr, h);
} W2XC_CATCH_ALL

int w2xc_convert_rows_device(w2xc_model *m, const float *d_view, size_t view_stride_bytes, int view_h, int view_y0, int w,
                             int plane_h, int row_begin, int row_end, float *d_out, size_t out_stride_bytes,
                             void *hip_stream, const w2xc_opts *opts)
try {
    int rc = check_plane_args(m, d_view, view_stride_bytes, w, view_h, d_out, out_stride_bytes);
    if (rc) return rc;
    const int n = (int)m->layers.size();
    if (plane_h <= 0 || row_begin < 0 || row_end > plane_h || row_begin >= row_end)
        return fail(W2XC_ERR_ARG, "bad row range [%d,%d) for a %d-row plane", row_begin, row_end, plane_h);
    if (view_y0 < 0 || view_y0 + view_h > plane_h || view_y0 > std::max(0, row_begin - n) ||
        view_y0 + view_h < std::min(plane_h, row_end + n))
        return fail(W2XC_ERR_ARG, "view rows [%d,%d) do not cover [%d,%d) +- %d halo rows", view_y0, view_y0 + view_h,
                    row_begin, row_end, n);
    const w2xc_opts o = resolve_opts(opts);
    int dev = o.device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    DeviceGuard guard(dev);
    if (!guard.ok) return fail(W2XC_ERR_HIP, "cannot select HIP device %d", dev);
    DevCtx *c = nullptr;
    rc = get_ctx(m, dev, &c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    // a view that starts/ends inside the plane has artificial edges, but every row within n of
    // them lies outside [row_begin, row_end), so clamping there never reaches a kept output row
    // (conv3x3_wino4, the F(4x4) kernel: a view with 4 n halo rows gets its banding-invariant geometry; on a narrower one
    //  W2XC_KERNEL_AUTO runs the F(2x2) kernels: run_rows)
    return run_rows(m, c, d_view, view_stride_bytes / 4, view_h, view_y0, w, row_begin, row_end, d_out,
                    out_stride_bytes / 4, (hipStream_t)hip_stream, o, 0, 1, 0, 0, nullptr, plane_h);
} W2XC_CATCH_ALL

}  // extern "C"

namespace {

// true when [p, p + bytes) is page-locked memory the DMA engines can address directly (hipHostMalloc /
// hipHostRegister, e.g. a pinned torch tensor): such planes skip the staging rings
bool host_range_pinned(const void *p, size_t bytes)
{
    if (!p || bytes == 0) return false;
    // both ends must be page-locked AND belong to ONE allocation / registration that spans the whole range: two registered
    // regions with a pageable (or unmapped) gap between them would pass a probe of the end points alone
    const void *base[2] = {nullptr, nullptr};
    int i = 0;
    for (const char *q : {(const char *)p, (const char *)p + bytes - 1}) {
        hipPointerAttribute_t at;
        memset(&at, 0, sizeof at);
        if (hipPointerGetAttributes(&at, q) != hipSuccess) {
            (void)hipGetLastError();   // an unregistered pointer is not an error of ours
            return false;
        }
        if (at.type != hipMemoryTypeHost) return false;
        hipDeviceptr_t b = nullptr;
        size_t sz = 0;
        if (hipMemGetAddressRange(&b, &sz, (hipDeviceptr_t)at.devicePointer) == hipSuccess && b && sz) {
            const char *hb = (const char *)at.hostPointer - ((const char *)at.devicePointer - (const char *)b);   // host address of the allocation's start
            if ((const char *)p < hb || (const char *)p + bytes > hb + sz) return false;
            base[i] = hb;
        } else {
            (void)hipGetLastError();
            base[i] = nullptr;   // range unknown for this kind of registration: fall back to comparing what we have
        }
        i++;
    }
    return base[0] == base[1];
}

int pipe_init(HostPipe &p)
{
    if (p.ready) return W2XC_OK;
    HIP_TRY(hipStreamCreateWithFlags(&p.s_compute, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&p.s_h2d, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&p.s_d2h, hipStreamNonBlocking));
    for (auto &e : p.ev_in_slot) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : p.ev_out_slot) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&p.ev_input, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&p.ev_chunk, hipEventDisableTiming));
    p.ready = true;
    return W2XC_OK;
}

// ---- NUMA placement of a device's host pipeline (multi-socket hosts: the pinned rings and the threads that fill / drain them belong
// on the CPU node the GPU hangs off, or every staged byte crosses the inter-socket link twice).  W2XC_NUMA=0 disables. ----
int numa_node_of_device(int dev)
{
    static const bool enabled = [] { const char *e = getenv("W2XC_NUMA"); return !(e && atoi(e) == 0); }();
    if (!enabled) return -1;
    int node = -1;
    if (hipDeviceGetAttribute(&node, hipDeviceAttributeHostNumaId, dev) == hipSuccess && node >= 0) return node;
    (void)hipGetLastError();
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, dev) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char *q = bus; *q; q++) *q = (char)tolower(*q);
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}

// the CPUs of a node ("0-63,128-191" in /sys/devices/system/node/nodeN/cpulist); false when unknown or the node has none
bool cpus_of_node(int node, cpu_set_t *set)
{
    if (node < 0) return false;
    char path[96];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return false;
    char buf[4096] = {0};
    const size_t got = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[got] = 0;
    CPU_ZERO(set);
    int count = 0;
    for (const char *q = buf; *q;) {
        char *end = nullptr;
        const long a = strtol(q, &end, 10);
        if (end == q) break;
        long b = a;
        q = end;
        if (*q == '-') { b = strtol(q + 1, &end, 10); q = end; }
        for (long cpu = a; cpu <= b && cpu < CPU_SETSIZE; cpu++) { CPU_SET((int)cpu, set); count++; }
        while (*q == ',' || *q == '\n' || *q == ' ') q++;
    }
    return count > 0;
}

// Binds the calling thread to a device's CPU node for its lifetime and restores the previous affinity afterwards
struct NodeCpus { int node = -1; bool have = false; cpu_set_t set; };
const NodeCpus &node_cpus_of_device(int dev)   // looked up once per device: no /sys read or attribute query on the per-call path
{
    static std::mutex mu;
    static std::map<int, NodeCpus> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(dev);
    if (it != cache.end()) return it->second;
    NodeCpus nc;
    nc.node = numa_node_of_device(dev);
    nc.have = cpus_of_node(nc.node, &nc.set);
    return cache.emplace(dev, nc).first->second;
}
struct NodeAffinity {
    cpu_set_t prev;
    bool bound = false;
    int node = -1;
    explicit NodeAffinity(int dev)
    {
        const NodeCpus &nc = node_cpus_of_device(dev);
        node = nc.node;
        if (!nc.have) return;
        const cpu_set_t want = nc.set;
        if (pthread_getaffinity_np(pthread_self(), sizeof prev, &prev) != 0) return;
        cpu_set_t both;
        CPU_AND(&both, &prev, &want);          // never leave the set the caller (or a cgroup) already confined us to
        if (CPU_COUNT(&both) == 0) return;
        bound = pthread_setaffinity_np(pthread_self(), sizeof both, &both) == 0;
    }
    ~NodeAffinity() { if (bound) pthread_setaffinity_np(pthread_self(), sizeof prev, &prev); }
};

// grow-only device / pinned buffers; growing drains the pipe first (earlier calls may still use the old ones)
int pipe_reserve(HostPipe &p, size_t in_bytes, size_t out_bytes, size_t in_slot, size_t out_slot)
{
    auto drain = [&]() -> int {
        HIP_TRY(hipStreamSynchronize(p.s_compute));
        HIP_TRY(hipStreamSynchronize(p.s_h2d));
        HIP_TRY(hipStreamSynchronize(p.s_d2h));
        return W2XC_OK;
    };
    if (p.d_in_bytes < in_bytes) {
        int rc = drain(); if (rc) return rc;
        if (p.d_in) { HIP_TRY(hipFree(p.d_in)); p.d_in = nullptr; p.d_in_bytes = 0; }
        if (hipMalloc((void **)&p.d_in, in_bytes) != hipSuccess) return fail(W2XC_ERR_NOMEM, "hipMalloc(%zu MiB) for the input rows failed", in_bytes >> 20);
        p.d_in_bytes = in_bytes;
    }
    if (p.d_out_bytes < out_bytes) {
        int rc = drain(); if (rc) return rc;
        if (p.d_out) { HIP_TRY(hipFree(p.d_out)); p.d_out = nullptr; p.d_out_bytes = 0; }
        if (hipMalloc((void **)&p.d_out, out_bytes) != hipSuccess) return fail(W2XC_ERR_NOMEM, "hipMalloc(%zu MiB) for the output rows failed", out_bytes >> 20);
        p.d_out_bytes = out_bytes;
    }
    if (in_slot && p.in_slot_bytes < in_slot) {
        int rc = drain(); if (rc) return rc;
        if (p.pin_in) { HIP_TRY(hipHostFree(p.pin_in)); p.pin_in = nullptr; p.in_slot_bytes = 0; }
        if (hipHostMalloc((void **)&p.pin_in, in_slot * HostPipe::IN_SLOTS, hipHostMallocDefault) != hipSuccess)
            return fail(W2XC_ERR_NOMEM, "hipHostMalloc of the input staging ring failed");
        p.in_slot_bytes = in_slot;   // (default policy: ROCm places pinned host memory near the allocating device)
    }
    if (out_slot && p.out_slot_bytes < out_slot) {
        int rc = drain(); if (rc) return rc;
        if (p.pin_out) { HIP_TRY(hipHostFree(p.pin_out)); p.pin_out = nullptr; p.out_slot_bytes = 0; }
        if (hipHostMalloc((void **)&p.pin_out, out_slot * HostPipe::OUT_SLOTS, hipHostMallocDefault) != hipSuccess)
            return fail(W2XC_ERR_NOMEM, "hipHostMalloc of the output staging ring failed");
        p.out_slot_bytes = out_slot;
    }
    return W2XC_OK;
}

// Output rows [ra, rb) of the (w << up) x (h << up) conversion of the HOST plane `in` (h rows of w floats) on device
// `dev`, written to rows [ra, rb) of the HOST plane `out`.  One unit of the tile farm: the caller runs one of these
// per device (threads) or per rank (processes); units never exchange data.
//
//   feeder (this thread)   stages the band's source rows (pageable -> pinned slot -> s_h2d), enqueues the band's
//                          layers on s_compute, stages the NEXT band's rows while it computes, then launches the
//                          last layer in row chunks and queues each chunk's D2H on s_d2h into a pinned slot
//   drainer (one thread)   waits for each chunk's D2H and copies it into the caller's plane (the "stitch" of
//                          convertRoutine.cpp:143-161), freeing the slot
// so H2D(band k+1) || layers(band k) || D2H + stitch(band k-1 / earlier chunks).  Planes that are already pinned
// are DMA'd in place without staging.
// hs = halo rows of the source view per side (convert_plane_host decides: n, or 4 n for the banding-invariant geometry of conv3x3_wino4)
int host_rows_on_device(w2xc_model *m, int dev, const float *in_, size_t in_stride, int w, int h, int up, int ra, int rb,
                        float *out_, size_t out_stride, const w2xc_opts &o, int copy_threads, int in_row0, int out_row0, int hs)
{
    // `in_` points at source row in_row0, `out_` at output row out_row0: rebase both to row 0 (only rows that exist are touched)
    const float *in = (const float *)((const char *)in_ - (ptrdiff_t)in_row0 * (ptrdiff_t)in_stride);
    float *out = (float *)((char *)out_ - (ptrdiff_t)out_row0 * (ptrdiff_t)out_stride);
    HIP_TRY(hipSetDevice(dev));
    // this thread is the unit's feeder: it (and the drainer it starts, which inherits the affinity) runs on the device's CPU node, and
    // the pinned rings it allocates land there; the caller's affinity is restored on return
    NodeAffinity node_guard(dev);
    DevCtx *c = nullptr;
    int rc = get_ctx(m, dev, &c);
    if (rc) return rc;
    // the context (its workspace and pipe) stays locked for the whole call: calls that share a device serialise
    std::lock_guard<std::mutex> lk(c->mu);
    HostPipe &p = c->pipe;
    if ((rc = pipe_init(p))) return rc;

    const int W = w << up, H = h << up;
    // source rows that cover output rows [ra - hs, rb + hs) (clipped), in source coordinates
    const int sy0 = std::max(0, ra - hs) >> up, sy1 = (std::min(H, rb + hs) + up) >> up;
    const int svh = sy1 - sy0;
    const size_t in_row = (size_t)w * 4, out_row = (size_t)W * 4;
    const bool in_pinned = host_range_pinned((const char *)in + (size_t)sy0 * in_stride, (size_t)(svh - 1) * in_stride + in_row);
    const bool out_pinned = host_range_pinned((const char *)out + (size_t)ra * out_stride, (size_t)(rb - ra - 1) * out_stride + out_row);
    // staging granularity, whole rows: input slices of ~2 MiB; output chunks of at most ~8 MiB tapering to 1/16 of that
    // (multiples of the 8-row tiles of the last-layer kernels).  w2xc_opts.host_chunk_kb overrides the maximum (test aid).
    const size_t chunk_max = o.host_chunk_kb > 0 ? (size_t)o.host_chunk_kb << 10 : (size_t)8 << 20;
    const int in_chunk_rows = (int)std::max<size_t>(1, std::min<size_t>(chunk_max, (size_t)2 << 20) / in_row);
    const int out_chunk_rows = (int)std::max<size_t>(8, (chunk_max / out_row) & ~(size_t)7);
    const int out_chunk_min = (int)std::max<size_t>(8, (chunk_max / 16 / out_row) & ~(size_t)7);
    rc = pipe_reserve(p, (size_t)svh * in_row, (size_t)(rb - ra) * out_row, in_pinned ? 0 : (size_t)in_chunk_rows * in_row,
                      out_pinned ? 0 : (size_t)out_chunk_rows * out_row);
    if (rc) return rc;

    const bool trace = (o.verbose & 2) != 0;   // (debug aid) phase timestamps of one unit on stderr
    const auto t0 = std::chrono::steady_clock::now();
    auto ms_since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    double t_in_done = -1, t_first_out = -1;

    // ---- input side: source rows [0, svh) of this unit's view, uploaded in order up to a high-water mark ----
    int uploaded = 0;        // view rows already queued on s_h2d
    long in_seq = 0;         // staging slots used so far
    auto upload_to = [&](int s_end) -> int {
        s_end = std::min(s_end, svh);
        struct Stamp { double &t; bool on; int &up; int all; std::function<double()> now; ~Stamp() { if (on && t < 0 && up >= all) t = now(); } }
            stamp{t_in_done, trace, uploaded, svh, [&] { return ms_since(t0); }};
        while (uploaded < s_end) {
            if (in_pinned) {   // DMA straight from the caller's plane
                const int rows = s_end - uploaded;
                const char *src = (const char *)in + (size_t)(sy0 + uploaded) * in_stride;
                if (in_stride == in_row) HIP_TRY(hipMemcpyAsync(p.d_in + (size_t)uploaded * w, src, (size_t)rows * in_row, hipMemcpyHostToDevice, p.s_h2d));
                else HIP_TRY(hipMemcpy2DAsync(p.d_in + (size_t)uploaded * w, in_row, src, in_stride, in_row, rows, hipMemcpyHostToDevice, p.s_h2d));
                uploaded = s_end;
                break;
            }
            const int rows = std::min(in_chunk_rows, s_end - uploaded);
            const int slot = (int)(in_seq % HostPipe::IN_SLOTS);
            if (in_seq >= HostPipe::IN_SLOTS) HIP_TRY(hipEventSynchronize(p.ev_in_slot[slot]));   // its last DMA has read it
            char *stage = p.pin_in + (size_t)slot * p.in_slot_bytes;
            w2xc_host::CopyPool::get().copy_rows(stage, in_row, (const char *)in + (size_t)(sy0 + uploaded) * in_stride, in_stride, in_row, rows, copy_threads);
            HIP_TRY(hipMemcpyAsync(p.d_in + (size_t)uploaded * w, stage, (size_t)rows * in_row, hipMemcpyHostToDevice, p.s_h2d));
            HIP_TRY(hipEventRecord(p.ev_in_slot[slot], p.s_h2d));
            in_seq++;
            uploaded += rows;
        }
        return W2XC_OK;
    };
    // in-place / overlapping planes (the reference never does this, main.cpp:94-96 copies first; a library must survive it):
    // the drainer writes band b's rows while later bands still read theirs, so every source row is staged before any output exists
    const char *in_lo = (const char *)in + (size_t)sy0 * in_stride, *in_hi = (const char *)in + (size_t)(sy1 - 1) * in_stride + in_row;
    const char *out_lo = (const char *)out + (size_t)ra * out_stride, *out_hi = (const char *)out + (size_t)(rb - 1) * out_stride + out_row;
    const bool overlap = in_lo < out_hi && out_lo < in_hi;
    // view rows (source coordinates, relative to sy0) a band of output rows [y0, y1) reads
    auto band_src_end = [&](int y1) { return overlap ? svh : ((std::min(H, y1 + hs) + up) >> up) - sy0; };

    // ---- output side ----
    struct Chunk { int r0, r1, slot; };
    std::mutex qmu;
    std::condition_variable qcv;
    std::deque<Chunk> pending;     // D2H queued, not yet stitched (drainer consumes in order)
    long queued = 0, drained = 0;  // chunk counters (slots are used round-robin)
    bool feeder_done = false;
    std::atomic<int> drain_rc{W2XC_OK};
    std::string drain_err;
    std::thread drainer;
    if (!out_pinned) {
        drainer = std::thread([&] {
            hipSetDevice(dev);
            for (;;) {
                Chunk ch;
                {
                    std::unique_lock<std::mutex> ql(qmu);
                    qcv.wait(ql, [&] { return !pending.empty() || feeder_done; });
                    if (pending.empty()) return;
                    ch = pending.front();
                    pending.pop_front();
                }
                if (drain_rc.load() == W2XC_OK) {
                    hipError_t e = hipEventSynchronize(p.ev_out_slot[ch.slot]);
                    if (e != hipSuccess) {
                        drain_err = std::string("hipEventSynchronize(D2H chunk) failed: ") + hipGetErrorString(e);
                        drain_rc.store(W2XC_ERR_HIP);
                    } else {
                        try {   // (a std::bad_alloc / std::system_error on this thread would be std::terminate, not an error code)
                            w2xc_host::CopyPool::get().copy_rows((char *)out + (size_t)ch.r0 * out_stride, out_stride,
                                                                 p.pin_out + (size_t)ch.slot * p.out_slot_bytes, out_row, out_row, ch.r1 - ch.r0, copy_threads);
                        } catch (const std::exception &ex) {
                            drain_err = std::string("host copy of a downloaded chunk failed: ") + ex.what();
                            drain_rc.store(W2XC_ERR_NOMEM);
                        } catch (...) {
                            drain_err = "host copy of a downloaded chunk failed";
                            drain_rc.store(W2XC_ERR_NOMEM);
                        }
                    }
                }
                {
                    std::lock_guard<std::mutex> ql(qmu);
                    drained++;
                }
                qcv.notify_all();
            }
        });
    }
    auto finish_drainer = [&] {
        if (drainer.joinable()) {
            { std::lock_guard<std::mutex> ql(qmu); feeder_done = true; }
            qcv.notify_all();
            drainer.join();
        }
    };
    struct AtExit {   // an exception below (std::bad_alloc in a queue) must not unwind past a joinable thread
        std::function<void()> f;
        ~AtExit() { f(); }
    } join_guard{finish_drainer};

    BandHooks hk;
    hk.out_chunk_rows = out_chunk_rows;
    hk.out_chunk_min = out_chunk_min;
    hk.input_needed = [&](int, int y1) -> int {
        int r = upload_to(band_src_end(y1));
        if (r) return r;
        HIP_TRY(hipEventRecord(p.ev_input, p.s_h2d));
        HIP_TRY(hipStreamWaitEvent(p.s_compute, p.ev_input, 0));
        return W2XC_OK;
    };
    hk.prefetch = [&](int, int y1n) -> int { return upload_to(band_src_end(y1n)); };
    // a band whose rows were prefetched under the previous band is launched whole; otherwise layer 1 follows the upload slice by slice
    hk.in_chunk = [&](int, int y1) -> int {
        if (overlap || uploaded >= std::min(band_src_end(y1), svh)) return 0;
        return std::max(8, ((in_chunk_rows << up) + 7) & ~7);
    };
    hk.input_upto = [&](int vlast) -> int {
        int r = upload_to((vlast >> up) + 1);
        if (r) return r;
        HIP_TRY(hipEventRecord(p.ev_input, p.s_h2d));
        HIP_TRY(hipStreamWaitEvent(p.s_compute, p.ev_input, 0));
        return W2XC_OK;
    };
    hk.output_ready = [&](int r0, int r1) -> int {
        if (trace && t_first_out < 0) t_first_out = ms_since(t0);
        HIP_TRY(hipEventRecord(p.ev_chunk, p.s_compute));
        HIP_TRY(hipStreamWaitEvent(p.s_d2h, p.ev_chunk, 0));
        if (out_pinned) {
            char *dst = (char *)out + (size_t)r0 * out_stride;
            const float *src = p.d_out + (size_t)(r0 - ra) * W;
            if (out_stride == out_row) HIP_TRY(hipMemcpyAsync(dst, src, (size_t)(r1 - r0) * out_row, hipMemcpyDeviceToHost, p.s_d2h));
            else HIP_TRY(hipMemcpy2DAsync(dst, out_stride, src, out_row, out_row, r1 - r0, hipMemcpyDeviceToHost, p.s_d2h));
            return W2XC_OK;
        }
        for (int a = r0; a < r1; a += out_chunk_rows) {   // (an unchunked last layer reports the whole band at once)
            const int b2 = std::min(r1, a + out_chunk_rows);
            int slot;
            {
                std::unique_lock<std::mutex> ql(qmu);
                qcv.wait(ql, [&] { return queued - drained < HostPipe::OUT_SLOTS; });   // a free staging slot
                slot = (int)(queued % HostPipe::OUT_SLOTS);
            }
            if (drain_rc.load()) return drain_rc.load();
            HIP_TRY(hipMemcpyAsync(p.pin_out + (size_t)slot * p.out_slot_bytes, p.d_out + (size_t)(a - ra) * W, (size_t)(b2 - a) * out_row,
                                   hipMemcpyDeviceToHost, p.s_d2h));
            HIP_TRY(hipEventRecord(p.ev_out_slot[slot], p.s_d2h));
            {
                std::lock_guard<std::mutex> ql(qmu);
                pending.push_back({a, b2, slot});
                queued++;
            }
            qcv.notify_all();
        }
        return W2XC_OK;
    };

    rc = run_rows(m, c, p.d_in, w, svh << up, sy0 << up, W, ra, rb, p.d_out, W, p.s_compute, o, up, 1, 0, 0, &hk, H);
    const double t_enq = ms_since(t0);
    double t_comp = 0;
    if (trace) { hipStreamSynchronize(p.s_compute); t_comp = ms_since(t0); }
    std::string err = g_last_error;
    finish_drainer();
    if (trace) fprintf(stderr, "[w2xc host] device %d (cpu node %d%s) rows %d..%d: input queued %.3f ms, first output chunk enqueued %.3f ms, enqueued %.3f ms, layers done %.3f ms, stitched %.3f ms (in %s, out %s, %d copy threads)\n",
                       dev, node_guard.node, node_guard.bound ? ", threads bound" : "", ra, rb, t_in_done, t_first_out, t_enq, t_comp,
                       ms_since(t0), in_pinned ? "pinned" : "pageable", out_pinned ? "pinned" : "pageable", copy_threads);
    // leave nothing in flight, whatever happened: the pipe and the caller's planes are reused by the next call
    hipError_t e1 = hipStreamSynchronize(p.s_h2d), e2 = hipStreamSynchronize(p.s_compute), e3 = hipStreamSynchronize(p.s_d2h);
    if (rc) { g_last_error = err; return rc; }
    if (drain_rc.load()) return fail(drain_rc.load(), "%s", drain_err.c_str());
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess)
        return fail(W2XC_ERR_HIP, "stream synchronisation failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2 != hipSuccess ? e2 : e3));
    return W2XC_OK;
}

// host-pointer path shared by w2xc_convert_plane (up = 0), w2xc_convert_plane_nn2x (up = 1) and w2xc_convert_plane_rows.
// (w, h) is the SOURCE plane; the output is (w << up) x (h << up), of which rows [row_begin, row_end) are produced.
int convert_plane_host(w2xc_model *m, const float *in, size_t in_stride_bytes, int w, int h, float *out,
                       size_t out_stride_bytes, const w2xc_opts *opts, int up, int row_begin = 0, int row_end = -1,
                       int in_row0 = 0, int in_rows = -1)
{
    if (!m || !in || !out) return fail(W2XC_ERR_ARG, "null argument");
    if (w <= 0 || h <= 0) return fail(W2XC_ERR_ARG, "plane size must be positive (got %dx%d)", w, h);
    const int W = w << up, H = h << up;
    if (row_end < 0) row_end = H;
    if (row_begin < 0 || row_end > H || row_begin >= row_end) return fail(W2XC_ERR_ARG, "bad row range [%d,%d) for a %d-row plane", row_begin, row_end, H);
    if (in_stride_bytes < (size_t)w * 4 || out_stride_bytes < (size_t)W * 4 || (in_stride_bytes & 3) || (out_stride_bytes & 3))
        return fail(W2XC_ERR_ARG, "row strides must be multiples of 4 bytes and >= 4*width");
    if (m->layers.empty()) return fail(W2XC_ERR_ARG, "model has no layers");
    {   // the source rows handed over must cover the rows [row_begin - n, row_end + n) reads (clipped to the plane)
        const int n = (int)m->layers.size();
        const int need0 = std::max(0, row_begin - n) >> up, need1 = (std::min(H, row_end + n) + up) >> up;
        if (in_rows < 0) in_rows = h - in_row0;
        if (in_row0 < 0 || in_row0 > need0 || in_row0 + in_rows < need1 || in_row0 + in_rows > h)
            return fail(W2XC_ERR_ARG, "source rows [%d,%d) do not cover the rows [%d,%d) this row range reads", in_row0, in_row0 + in_rows, need0, need1);
    }
    w2xc_opts o = resolve_opts(opts);
    // halo rows of the units' source views: n, or 4 n when conv3x3_wino4 runs (its banding-invariant geometry, run_rows) -- if the rows handed
    // over hold that much around [row_begin, row_end); otherwise W2XC_KERNEL_AUTO means the F(2x2) kernels for this call
    int hs = (int)m->layers.size();
    if (uses_wino4(m, o)) {
        const int h4 = 4 * hs;
        const int need0 = std::max(0, row_begin - h4) >> up, need1 = (std::min(H, row_end + h4) + up) >> up;
        if (in_row0 <= need0 && in_row0 + in_rows >= need1) hs = h4;
        else if (o.kernel == W2XC_KERNEL_AUTO)   // (as run_rows: no silent change of kernel and rounding with the view's halo)
            return fail(W2XC_ERR_ARG, "source rows [%d,%d) hold the minimum halo only: the default F(4x4) kernel needs rows [%d,%d) (4 halo rows per layer) for "
                                      "banding-invariant results; pass them or choose w2xc_opts.kernel explicitly (W2XC_KERNEL_WINOGRAD32: F(2x2))",
                        in_row0, in_row0 + in_rows, need0, need1);
    }
    const int ndev_all = w2xc_device_count();
    if (ndev_all <= 0) return fail(W2XC_ERR_HIP, "no HIP device available (libw2xc_hip has no CPU fallback)");
    std::vector<int> devs;
    for (int d = 0; d < ndev_all && d < 32; d++)
        if (o.device_mask == 0 || (o.device_mask >> d) & 1u) devs.push_back(d);
    if (devs.empty()) return fail(W2XC_ERR_ARG, "device_mask 0x%x selects no available device (%d present)", o.device_mask, ndev_all);
    int nd = (int)devs.size();
    // w2xc_opts.host_units = k (test aid): cut the rows into k units, round-robin over the selected devices,
    // so the multi-device arithmetic below can be exercised on a single-GPU box
    {
        const int k = o.host_units;
        if (k > nd) {
            const size_t have = devs.size();
            for (int i = (int)have; i < k && i < 64; i++) devs.push_back(devs[i % have]);
            nd = (int)devs.size();
        }
    }
    const int R = row_end - row_begin;
    if (nd > R) nd = R;
    // In-place / overlapping planes with MORE THAN ONE unit: unit t writes output rows that are the halo source rows of units t-1 and
    // t+1 (each unit only protects its own rows, host_rows_on_device), on one device in a deterministic wrong order, on several
    // devices as a race.  The reference survives convertWithModels(img, img, ...) through its copyMakeBorder temporary
    // (convertRoutine.cpp:35,96); here the source rows are snapshotted once before the units fan out.
    std::vector<float> snapshot;
    if (nd > 1) {
        const int s0 = std::max(0, row_begin - hs) >> up, s1 = (std::min(H, row_end + hs) + up) >> up;
        const char *in_lo = (const char *)in + (ptrdiff_t)(s0 - in_row0) * (ptrdiff_t)in_stride_bytes;
        const char *in_hi = (const char *)in + (ptrdiff_t)(s1 - 1 - in_row0) * (ptrdiff_t)in_stride_bytes + (size_t)w * 4;
        const char *out_lo = (const char *)out, *out_hi = (const char *)out + (size_t)(R - 1) * out_stride_bytes + (size_t)W * 4;
        if (in_lo < out_hi && out_lo < in_hi) {
            snapshot.resize((size_t)(s1 - s0) * w);
            w2xc_host::CopyPool::get().copy_rows((char *)snapshot.data(), (size_t)w * 4, in_lo, in_stride_bytes, (size_t)w * 4, s1 - s0,
                                                 std::max(1, std::min(w2xc_get_jobs(), 32)));
            in = snapshot.data();
            in_stride_bytes = (size_t)w * 4;
            in_row0 = s0;
        }
    }
    // modelUtility's nJob (modelHandler.hpp:99; the CLI's -j) = host threads that move rows in and out of the staging rings, shared by the units
    const int copy_threads = std::max(1, std::min(w2xc_get_jobs(), 32) / nd);   // nJob bounds the total: never more than nJob staging threads over all units
    // the pool's workers are created HERE, on the caller's (unbound) thread: a unit's feeder binds itself to its device's NUMA node, and workers created
    // lazily from there would keep that node's mask while serving every device
    w2xc_host::CopyPool::get().reserve(std::min(w2xc_get_jobs(), 32) - 1);

    int prev = 0;
    hipGetDevice(&prev);
    std::vector<int> rcs(nd, W2XC_OK);
    std::vector<std::string> errs(nd);
    auto worker = [&](int t) {
        // contiguous share [ra, rb) of the OUTPUT rows for unit t: independent, no exchange
        const int ra = row_begin + (int)((long long)R * t / nd), rb = row_begin + (int)((long long)R * (t + 1) / nd);
        try {   // no exception may leave a unit's thread (std::terminate) or cross the C ABI
            rcs[t] = host_rows_on_device(m, devs[t], in, in_stride_bytes, w, h, up, ra, rb, out, out_stride_bytes, o, copy_threads, in_row0, row_begin, hs);
            if (rcs[t]) errs[t] = g_last_error;
        } catch (const std::bad_alloc &) {
            rcs[t] = W2XC_ERR_NOMEM;
            errs[t] = "out of host memory in a conversion unit";
        } catch (const std::exception &ex) {
            rcs[t] = W2XC_ERR_HIP;
            errs[t] = std::string("exception in a conversion unit: ") + ex.what();
        } catch (...) {
            rcs[t] = W2XC_ERR_HIP;
            errs[t] = "unknown exception in a conversion unit";
        }
    };
    if (nd == 1) worker(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nd; t++) th.emplace_back(worker, t);
        for (auto &x : th) x.join();
    }
    hipSetDevice(prev);
    for (int t = 0; t < nd; t++)
        if (rcs[t]) { g_last_error = errs[t]; return rcs[t]; }   // (the message is w2xc_last_error(); the C++ adapter prints it, a C-ABI consumer decides itself)
    return W2XC_OK;
}
}  // namespace

extern "C" {

int w2xc_convert_plane(w2xc_model *m, const float *in, size_t in_stride_bytes, int w, int h, float *out,
                       size_t out_stride_bytes, int block_splitting, const w2xc_opts *opts)
try {
    (void)block_splitting;   // results do not depend on the reference's block split
    return convert_plane_host(m, in, in_stride_bytes, w, h, out, out_stride_bytes, opts, 0);
} W2XC_CATCH_ALL

int w2xc_convert_plane_nn2x(w2xc_model *m, const float *in, size_t in_stride_bytes, int w, int h, float *out,
                            size_t out_stride_bytes, const w2xc_opts *opts)
try {
    return convert_plane_host(m, in, in_stride_bytes, w, h, out, out_stride_bytes, opts, 1);
} W2XC_CATCH_ALL

int w2xc_convert_plane_rows(w2xc_model *m, const float *in_view, size_t in_stride_bytes, int view_y0, int view_h, int w, int h, int nn2x,
                            int row_begin, int row_end, float *out, size_t out_stride_bytes, const w2xc_opts *opts)
try {
    if (nn2x != 0 && nn2x != 1) return fail(W2XC_ERR_ARG, "nn2x must be 0 or 1");
    if (view_h <= 0) return fail(W2XC_ERR_ARG, "empty source view");
    return convert_plane_host(m, in_view, in_stride_bytes, w, h, out, out_stride_bytes, opts, nn2x, row_begin, row_end, view_y0, view_h);
} W2XC_CATCH_ALL

int w2xc_convert_planes_device(w2xc_model *m, int n_in_planes, const float *d_in, size_t in_plane_stride_bytes,
                               size_t in_stride_bytes, int w, int h, float *d_out, size_t out_plane_stride_bytes,
                               size_t out_stride_bytes, void *hip_stream, const w2xc_opts *opts)
try {
    int rc = check_plane_args(m, d_in, in_stride_bytes, w, h, d_out, out_stride_bytes);
    if (rc) return rc;
    if (n_in_planes < 1 || (in_plane_stride_bytes & 3) || (out_plane_stride_bytes & 3) ||
        (n_in_planes > 1 && in_plane_stride_bytes < in_stride_bytes * (size_t)h) || out_plane_stride_bytes < out_stride_bytes * (size_t)h)
        return fail(W2XC_ERR_ARG, "bad plane count / plane strides");
    const w2xc_opts o = resolve_opts(opts);
    if (o.precision != W2XC_PRECISION_FP32 && split_terms(o) == 0)
        return fail(W2XC_ERR_UNSUPPORTED, "w2xc_convert_planes_* supports W2XC_PRECISION_FP32 / BF16X2 / BF16X3 / FP16X2");
    int dev = o.device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    DeviceGuard guard(dev);
    if (!guard.ok) return fail(W2XC_ERR_HIP, "cannot select HIP device %d", dev);
    DevCtx *c = nullptr;
    rc = get_ctx(m, dev, &c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    return run_rows(m, c, d_in, in_stride_bytes / 4, h, 0, w, 0, h, d_out, out_stride_bytes / 4, (hipStream_t)hip_stream, o, 0,
                    n_in_planes, (long long)(in_plane_stride_bytes / 4), (long long)(out_plane_stride_bytes / 4), nullptr, h);
} W2XC_CATCH_ALL

int w2xc_convert_plane_nn2x_device(w2xc_model *m, const float *d_in, size_t in_stride_bytes, int w, int h, float *d_out,
                                   size_t out_stride_bytes, void *hip_stream, const w2xc_opts *opts)
try {
    if (!m || !d_in || !d_out) return fail(W2XC_ERR_ARG, "null argument");
    if (w <= 0 || h <= 0) return fail(W2XC_ERR_ARG, "plane size must be positive (got %dx%d)", w, h);
    if (in_stride_bytes < (size_t)w * 4 || out_stride_bytes < (size_t)w * 8 || (in_stride_bytes & 3) || (out_stride_bytes & 3))
        return fail(W2XC_ERR_ARG, "row strides must be multiples of 4 bytes and >= 4*width");
    const w2xc_opts o = resolve_opts(opts);
    int dev = o.device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    DeviceGuard guard(dev);
    if (!guard.ok) return fail(W2XC_ERR_HIP, "cannot select HIP device %d", dev);
    DevCtx *c = nullptr;
    int rc = get_ctx(m, dev, &c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    return run_rows(m, c, d_in, in_stride_bytes / 4, 2 * h, 0, 2 * w, 0, 2 * h, d_out, out_stride_bytes / 4,
                    (hipStream_t)hip_stream, o, 1, 1, 0, 0, nullptr, 2 * h);
} W2XC_CATCH_ALL

}  // extern "C"

extern "C" {

}  // extern "C"

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

// ---- N2: the scale phase of the CLI on one uint8 image (main.cpp:74-76,126-156,171-172) --------------------
}  // extern "C"

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

// resolve device + contexts of the (up to two) models and run the pipeline under their locks
int process_image_locked(w2xc_model *mn, w2xc_model *msc, const unsigned char *d_in, size_t in_stride, int w, int h, unsigned char *d_out,
                         size_t out_stride, int iterations, double shrink, hipStream_t st, const w2xc_opts &o, int dev)
{
    DevCtx *cn = nullptr, *cs = nullptr;
    int rc;
    if (mn && (rc = get_ctx(mn, dev, &cn))) return rc;
    if (msc && (rc = get_ctx(msc, dev, &cs))) return rc;
    std::unique_lock<std::mutex> l1, l2;
    if (cn) l1 = std::unique_lock<std::mutex>(cn->mu);
    if (cs && cs != cn) l2 = std::unique_lock<std::mutex>(cs->mu);
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
    if (cn) l1 = std::unique_lock<std::mutex>(cn->mu);
    if (cs && cs != cn) l2 = std::unique_lock<std::mutex>(cs->mu);
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

// ---- measurement ----------------------------------------------------------------------------------
int w2xc_profile_read(w2xc_model *m, int device, float *layer_ms, int *layer_launches, int n_layers)
{
    if (!m) return fail(W2XC_ERR_ARG, "null model");
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    DevCtx *c = nullptr;
    {
        std::lock_guard<std::mutex> lk(m->mu);
        auto it = m->ctx.find(device);
        if (it == m->ctx.end()) return fail(W2XC_ERR_ARG, "no context for device %d", device);
        c = it->second.get();
    }
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard guard(device);
    for (auto &e : c->pending) {
        HIP_TRY(hipEventSynchronize(e.b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e.a, e.b));
        c->layer_ms[e.layer] += ms;
        c->layer_launches[e.layer] += 1;
        c->pool.push_back(e);
    }
    c->pending.clear();
    for (int l = 0; l < n_layers && l < (int)c->layer_ms.size(); l++) {
        if (layer_ms) layer_ms[l] = (float)c->layer_ms[l];
        if (layer_launches) layer_launches[l] = c->layer_launches[l];
    }
    return W2XC_OK;
}

void w2xc_profile_reset(w2xc_model *m, int device)
{
    if (!m) return;
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return;
    std::lock_guard<std::mutex> lk(m->mu);
    auto it = m->ctx.find(device);
    if (it == m->ctx.end()) return;
    DevCtx *c = it->second.get();
    std::lock_guard<std::mutex> lk2(c->mu);
    for (auto &e : c->pending) c->pool.push_back(e);
    c->pending.clear();
    std::fill(c->layer_ms.begin(), c->layer_ms.end(), 0.0);
    std::fill(c->layer_launches.begin(), c->layer_launches.end(), 0);
}

const char *w2xc_layer_kernel_name(const w2xc_model *m, int layer, const w2xc_opts *opts)
{
    if (!m || layer < 0 || layer >= (int)m->layers.size()) return "";
    const w2xc_opts o = resolve_opts(opts);
    const W2xcKernelKind k = layer_kind(m, layer, o);
    if (k == W2XC_K_MFMA) {
        const int midv = layer_mid_variant(m, layer, o);
        if (midv != MID_MFMA) return midv == MID_WINO4 ? "conv3x3_wino4" : "conv3x3_wino";
    }
    return w2xc_kernel_name(k, m->layers[layer].nin, m->layers[layer].nout);
}

}  // extern "C"
