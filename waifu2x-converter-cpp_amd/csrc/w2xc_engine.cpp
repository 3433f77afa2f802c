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

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
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

struct HostLayer {
    int nin = 0, nout = 0;
    std::vector<float> w;       // [nout][nin][3][3], index o*nin+i (modelHandler.cpp:102)
    std::vector<double> bias;   // modelHandler.cpp:109-112 keeps doubles
};

struct DevLayer {
    W2xcKernelKind fast = W2XC_K_DIRECT;
    float *w_fast = nullptr;
    float *w_direct = nullptr;
    float *w_bf16 = nullptr;    // conv3x3_mfma_bf16 image, packed on first use of W2XC_PRECISION_BF16
    float *w_split[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // conv3x3_split images, index terms + 3*fmt, packed on first use
    float split_scale[6] = {1, 1, 1, 1, 1, 1};                                   // power-of-two weight scale of each image
    float *w_last_fused[3] = {nullptr, nullptr, nullptr};   // w2xc_split_pack_last images: [0] 2 bf16 terms, [1] 2 fp16 terms, [2] 3 bf16 terms
    float last_fused_scale[3] = {1, 1, 1};
    float *bias = nullptr;
};

struct ProfEvent {
    hipEvent_t a, b;
    int layer;
};

struct DevCtx {
    int device = 0;
    std::vector<DevLayer> layers;
    float *ws[2] = {nullptr, nullptr};
    size_t ws_floats[2] = {0, 0};
    float *aux = nullptr;       // N2: Y/U/V planes of the image pipeline
    size_t aux_floats = 0;
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
            if (l.w_bf16) hipFree(l.w_bf16);
            for (float *p : l.w_split)
                if (p) hipFree(p);
            for (float *p : l.w_last_fused)
                if (p) hipFree(p);
            if (l.bias) hipFree(l.bias);
        }
        for (int i = 0; i < 2; i++)
            if (ws[i]) hipFree(ws[i]);
        if (aux) hipFree(aux);
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
    }
    return r;
}

// 16-bit terms per activation value between the layers of the split pipeline, w2xc_split.hip (0 = not that pipeline)
int split_terms(const w2xc_opts &o)
{
    if (o.precision == W2XC_PRECISION_BF16) {   // plain bf16 = the same pipeline with ONE term; W2XC_BF16_PIPE=v1 selects
        static int v = -1;                      // the first-generation kernels (conv3x3_mfma_bf16, NHWC bf16 activations)
        if (v < 0) { const char *e = getenv("W2XC_BF16_PIPE"); v = (e && !strcmp(e, "v1")) ? 0 : 1; }
        return v;
    }
    return (o.precision == W2XC_PRECISION_BF16X2 || o.precision == W2XC_PRECISION_FP16X2) ? 2 : o.precision == W2XC_PRECISION_BF16X3 ? 3 : 0;
}
int split_fmt(const w2xc_opts &o) { return o.precision == W2XC_PRECISION_FP16X2 ? 1 : 0; }

bool fuse_last(const w2xc_model *m, const w2xc_opts &o);
bool fuse_first(const w2xc_model *m, const w2xc_opts &o);

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
    if (o.precision == W2XC_PRECISION_BF16) {
        // bf16 activations live only BETWEEN layers: the first layer reads the caller's fp32 plane, the
        // last one writes it; anything else (or a shape without an MFMA kernel) is unsupported
        if (l == 0 && k == W2XC_K_FIRST && m->layers[l].nin == 1) return W2XC_K_FIRST_BF16OUT;
        if (l == n - 1 && k == W2XC_K_LAST && m->layers[l].nout == 1) return W2XC_K_LAST_BF16IN;
        if (l > 0 && l < n - 1 && k == W2XC_K_MFMA) return W2XC_K_MFMA_BF16;
        return W2XC_K_DIRECT;   // run_rows rejects this for bf16
    }
    return k;
}

// 16-bit modes: layers 1 (ONE plane -> 32) and 2 (32 -> {32,64,128}) run as one kernel (conv3x3_first2_split) when
// layer 2 is an ordinary split mid layer.  W2XC_SPLIT_FUSE_FIRST=0 disables.
bool fuse_first(const w2xc_model *m, const w2xc_opts &o)
{
    static int en = -1;
    if (en < 0) { const char *e = getenv("W2XC_SPLIT_FUSE_FIRST"); en = (e && atoi(e) == 0) ? 0 : 1; }
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
    static int en = -1;
    if (en < 0) { const char *e = getenv("W2XC_SPLIT_FUSE_LAST"); en = (e && atoi(e) == 0) ? 0 : 1; }
    const int n = (int)m->layers.size();
    if (!en || o.kernel == W2XC_KERNEL_DIRECT || n < 3) return false;
    const int T = split_terms(o);
    if (T < 1 || T > 3) return false;
    return m->layers[n - 1].nout == 1 && w2xc_pick_kernel(m->layers[n - 1].nin, 1) == W2XC_K_LAST &&
           w2xc_pick_kernel(m->layers[n - 2].nin, m->layers[n - 2].nout) == W2XC_K_MFMA && n - 2 > 0;
}

// terms of layer l's OUTPUT in the split pipeline: T when layer l+1 is a split mid layer, else 0 (fp32)
int out_terms_of(const w2xc_model *m, int l, const w2xc_opts &o)
{
    const int T = split_terms(o), n = (int)m->layers.size();
    if (T == 0 || l + 1 >= n) return 0;
    if (l == n - 2 && fuse_last(m, o)) return 9;
    return layer_kind(m, l + 1, o) == W2XC_K_MID_SPLIT ? T : 0;
}

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
        HIP_TRY(hipEventCreate(&ev->b));
    }
    ev->layer = layer;
    HIP_TRY(hipEventRecord(ev->a, st));
    return W2XC_OK;
}

int launch_layer(DevCtx *c, const w2xc_model *m, int l, W2xcKernelKind kind, W2xcConvDesc d, hipStream_t st, bool profile)
{
    DevLayer &dl = c->layers[l];
    d.cin = m->layers[l].nin;
    d.cout = m->layers[l].nout;
    if (kind == W2XC_K_MFMA_BF16 && !dl.w_bf16) {
        std::vector<float> pk(w2xc_packed_weight_floats(kind, d.cin, d.cout));
        w2xc_pack_weights(kind, d.cin, d.cout, m->layers[l].w.data(), pk.data());
        int rc = upload(pk, &dl.w_bf16);
        if (rc) return rc;
    }
    if (kind == W2XC_K_FUSED_AWAY) return W2XC_OK;   // computed by the next layer's W2XC_K_FIRST2_SPLIT launch
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
            const int lt = d.terms == 3 ? 3 : 2, li = d.terms == 3 ? 2 : d.fmt;
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
    } else {
        d.wpk = kind == W2XC_K_DIRECT ? dl.w_direct : kind == W2XC_K_MFMA_BF16 ? dl.w_bf16 : dl.w_fast;
    }
    d.bias = dl.bias;
    ProfEvent ev;
    if (profile) { int rc = prof_begin(c, l, st, &ev); if (rc) return rc; }
    hipError_t e = kind == W2XC_K_MID_SPLIT     ? w2xc_launch_split_mid(d, st)
                   : kind == W2XC_K_FIRST_SPLIT ? w2xc_launch_split_first(d, st)
                   : kind == W2XC_K_LAST_GATHER ? w2xc_launch_last_gather(d, st)
                   : kind == W2XC_K_FIRST2_SPLIT ? w2xc_launch_first2_split(d, st)
                                                : w2xc_launch_conv(kind, d, st);
    if (e != hipSuccess) return fail(W2XC_ERR_HIP, "launch of %s (layer %d, %d->%d) failed: %s", w2xc_kernel_name(kind, d.cin, d.cout), l, d.cin, d.cout, hipGetErrorString(e));
    if (profile) {
        HIP_TRY(hipEventRecord(ev.b, st));
        c->pending.push_back(ev);
    }
    return W2XC_OK;
}

// Output rows [ra, rb) of convertWithModels on an h-row plane of which `d_in` holds rows
// [vy0, vy0+vh) -- every row in [ra-n, rb+n) clipped to the plane must be inside the view.
// `up` = 1 folds a nearest-neighbour 2x (main.cpp:132-140) into layer 1: vh, vy0, w, ra, rb are then in
// UPSCALED coordinates while d_in holds the (vh/2) x (w/2) source rows starting at source row vy0/2.
// Multi-plane form (w2xc_convert_planes_*): n_in planar input planes `in_cs` floats apart, ALL planes of the
// last layer written planar `out_cs` floats apart.  n_in == 1 && out_cs == 0 is convertWithModels proper,
// which returns only outputPlanes[0] (convertRoutine.cpp:78).
int run_rows(w2xc_model *m, DevCtx *c, const float *d_in, size_t in_stride_f, int vh, int vy0, int w, int ra, int rb,
             float *d_out, size_t out_stride_f, hipStream_t st, const w2xc_opts &o, int up = 0, int n_in = 1,
             long long in_cs = 0, long long out_cs = 0)
{
    const int n = (int)m->layers.size();
    if (n == 0) return fail(W2XC_ERR_ARG, "model has no layers");
    if (m->layers[0].nin != n_in)   // convertWithModelsBasic pushes exactly one plane (convertRoutine.cpp:63-64)
        return fail(W2XC_ERR_PLANES, "Error : Model-filter : \nnumber of input planes mismatch.\n%d,%d", n_in, m->layers[0].nin);
    const bool all_out = out_cs != 0;   // multi-plane output
    for (int l = 1; l < n; l++)
        if (m->layers[l].nin != m->layers[l - 1].nout)
            return fail(W2XC_ERR_PLANES, "Error : Model-filter : \nnumber of input planes mismatch.\n%d,%d", m->layers[l - 1].nout, m->layers[l].nin);
    const int T = split_terms(o);
    const bool bf16 = (o.precision == W2XC_PRECISION_BF16) && T == 0;
    if (o.precision != W2XC_PRECISION_FP32 && !bf16 && T == 0) return fail(W2XC_ERR_ARG, "unknown precision %d", o.precision);
    if (bf16) {
        if (n < 2 || m->layers[n - 1].nout != 1) return fail(W2XC_ERR_UNSUPPORTED, "W2XC_PRECISION_BF16 needs >= 2 layers ending in one plane");
        for (int l = 0; l < n; l++)
            if (layer_kind(m, l, o) == W2XC_K_DIRECT)
                return fail(W2XC_ERR_UNSUPPORTED, "W2XC_PRECISION_BF16: layer %d (%d->%d) has no bf16 kernel (1->{32,64,128}, {32,64,128}->{32,64,128}, ->1 only)",
                            l + 1, m->layers[l].nin, m->layers[l].nout);
    }
    if (T > 0)
        for (int l = 0; l < n; l++)
            if (layer_kind(m, l, o) == W2XC_K_DIRECT)
                return fail(W2XC_ERR_UNSUPPORTED, "16-bit precision modes: layer %d (%d->%d) has no kernel ({1,3}->{32,64,128} first, {32,64,128}->{32,64,128}, ->{1,3} last only)",
                            l + 1, m->layers[l].nin, m->layers[l].nout);
    // bytes per activation element of layer k's output (k = 1..n) in the workspace
    auto out_bpe = [&](int k) -> size_t {
        if (bf16) return 2;
        const int ot = out_terms_of(m, k - 1, o);
        return (ot >= 1 && ot <= 3) ? 2 * (size_t)ot : 4;
    };

    // the last layer stores straight into the caller's planar plane(s) when its kernel can address planar
    // output (conv3x3_last / conv3x3_direct); otherwise it goes through the NHWC workspace + a repack
    const W2xcKernelKind last_kind = layer_kind(m, n - 1, o);
    const bool last_direct = (m->layers[n - 1].nout == 1 || all_out) &&
                             (last_kind == W2XC_K_LAST || last_kind == W2XC_K_LAST_GATHER || last_kind == W2XC_K_LAST_BF16IN || last_kind == W2XC_K_DIRECT ||
                              (m->layers[n - 1].nout == 1 && last_kind != W2XC_K_MFMA && last_kind != W2XC_K_FIRST));
    // BYTES per band for the two ping-pong buffers (layer k output goes to ws[(k-1)&1])
    auto ws_need = [&](int rows, size_t need[2]) {
        need[0] = need[1] = 0;
        for (int k = 1; k <= n; k++) {
            if (k == n && last_direct) break;   // written straight to d_out
            if (k == 1 && layer_kind(m, 0, o) == W2XC_K_FUSED_AWAY) continue;   // layer 1's activations stay on chip
            const size_t hk = (size_t)rows + 2 * (n - k), wk = (size_t)w + 2 * (n - k);
            const bool fused = T > 0 && out_terms_of(m, k - 1, o) == 9;   // partial G planes of the fused last layer
            const size_t px_bytes = fused ? (size_t)w2xc_split_halves(T, m->layers[k - 1].nout) * 9 * 4 : m->layers[k - 1].nout * out_bpe(k);
            need[(k - 1) & 1] = std::max(need[(k - 1) & 1], hk * wk * px_bytes);
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
            const int nb = (total + band - 1) / band;
            band = (total + nb - 1) / nb;   // equalise
        }
    }
    band = std::min(band, total);
    {
        size_t need[2];
        ws_need(band, need);
        for (int i = 0; i < 2; i++)
            if (need[i]) { int rc = ensure_ws(c, i, (need[i] + 3) / 4); if (rc) return rc; }
    }

    for (int y0 = ra; y0 < rb; y0 += band) {
        const int y1 = std::min(rb, y0 + band);
        const float *src = d_in;
        long long src_rs = (long long)in_stride_f, src_ps = 1, src_cs = in_cs, src_ts = 0, src_gs = 0;
        int src_halves = 0;
        W2xcConvDesc first_d;
        memset(&first_d, 0, sizeof first_d);
        int src_h = vh, src_w = w;
        for (int k = 1; k <= n; k++) {
            if (o.verbose) std::cout << "Iteration #" << k << "..." << std::endl;   // convertRoutine.cpp:67
            const HostLayer &hl = m->layers[k - 1];
            W2xcConvDesc d;
            memset(&d, 0, sizeof d);
            d.in = src; d.in_rs = src_rs; d.in_ps = src_ps; d.in_cs = src_cs;
            d.in_h = src_h; d.in_w = src_w;
            d.out_h = (y1 - y0) + 2 * (n - k);
            d.out_w = w + 2 * (n - k);
            d.off_y = k == 1 ? (y0 - n - vy0) : 0;
            d.off_x = k == 1 ? -n : 0;
            d.in_shift = k == 1 ? up : 0;
            const W2xcKernelKind kind = layer_kind(m, k - 1, o);
            if (kind == W2XC_K_FUSED_AWAY) {   // layer 1 inside layer 2's kernel: keep its input description for that launch
                first_d = d;
                continue;
            }
            if (kind == W2XC_K_FIRST2_SPLIT) {
                d.in = first_d.in; d.in_rs = first_d.in_rs; d.in_ps = first_d.in_ps; d.in_cs = first_d.in_cs;
                d.in_h = first_d.in_h; d.in_w = first_d.in_w;
                d.off_y = first_d.off_y; d.off_x = first_d.off_x; d.in_shift = first_d.in_shift;
            }
            int split_grp = 0;
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
                if (T > 0 && d.out_terms >= 1 && d.out_terms <= 3) { d.out_rs = (long long)d.out_w * split_grp; d.out_ps = split_grp; }
                if (T > 0 && d.out_terms == 9) {   // G[half][tap][y][x]
                    d.out_rs = d.out_w; d.out_ps = 1;
                    d.out_gs = (long long)d.out_h * d.out_w;
                    d.out_ts = 9 * d.out_gs;
                    d.halves = w2xc_split_halves(T, hl.nout);
                }
            }
            int rc = launch_layer(c, m, k - 1, kind, d, st, o.profile != 0);
            if (rc) return rc;
            if (k == n && !direct_out) {
                // outputPlanes[0] of a multi-plane last layer (convertRoutine.cpp:78)
                hipError_t e = w2xc_launch_repack(d.out, d.out_rs, d.out_ps, 1, d_out + (size_t)(y0 - ra) * out_stride_f,
                                                  (long long)out_stride_f, 1, out_cs, d.out_h, d.out_w, all_out ? hl.nout : 1, st);
                if (e != hipSuccess) return fail(W2XC_ERR_HIP, "repack launch failed: %s", hipGetErrorString(e));
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
{
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
}

int w2xc_model_load_json(const char *path, w2xc_model **out)
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
        hl.nin = (int)nip->num;     // static_cast<int>(double), modelHandler.hpp:50-51
        hl.nout = (int)nop->num;
        const int ks = (int)kw->num;
        if (ks != (int)kh->num) {   // the reference exit(-1)s here (hpp:52-58); a library reports it
            std::cerr << "Error : Model-Constructor : \nkernel in model is not square.\nstop." << std::endl;
            return fail(W2XC_ERR_UNSUPPORTED, "kernel in model is not square");
        }
        if (ks != 3) return fail(W2XC_ERR_UNSUPPORTED, "layer %zu: kernel size %d; only 3x3 is supported (convertWithModels pads by the layer count, which assumes 3x3)", l, ks);
        if (hl.nin <= 0 || hl.nout <= 0) return fail(W2XC_ERR_JSON, "layer %zu: bad plane counts", l);
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

void w2xc_model_free(w2xc_model *m) { delete m; }
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
{
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
    return run_rows(m, c, d_in, in_stride_bytes / 4, h, 0, w, 0, h, d_out, out_stride_bytes / 4, (hipStream_t)hip_stream, o);
}

int w2xc_convert_rows_device(w2xc_model *m, const float *d_view, size_t view_stride_bytes, int view_h, int view_y0, int w,
                             int plane_h, int row_begin, int row_end, float *d_out, size_t out_stride_bytes,
                             void *hip_stream, const w2xc_opts *opts)
{
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
    return run_rows(m, c, d_view, view_stride_bytes / 4, view_h, view_y0, w, row_begin, row_end, d_out,
                    out_stride_bytes / 4, (hipStream_t)hip_stream, o);
}

}  // extern "C"

namespace {
// host-pointer path shared by w2xc_convert_plane (up = 0) and w2xc_convert_plane_nn2x (up = 1).
// (w, h) is the SOURCE plane; the output is (w << up) x (h << up).
int convert_plane_host(w2xc_model *m, const float *in, size_t in_stride_bytes, int w, int h, float *out,
                       size_t out_stride_bytes, const w2xc_opts *opts, int up)
{
    if (!m || !in || !out) return fail(W2XC_ERR_ARG, "null argument");
    if (w <= 0 || h <= 0) return fail(W2XC_ERR_ARG, "plane size must be positive (got %dx%d)", w, h);
    const int W = w << up, H = h << up;
    if (in_stride_bytes < (size_t)w * 4 || out_stride_bytes < (size_t)W * 4 || (in_stride_bytes & 3) || (out_stride_bytes & 3))
        return fail(W2XC_ERR_ARG, "row strides must be multiples of 4 bytes and >= 4*width");
    const w2xc_opts o = resolve_opts(opts);
    const int ndev_all = w2xc_device_count();
    if (ndev_all <= 0) return fail(W2XC_ERR_HIP, "no HIP device available (libw2xc_hip has no CPU fallback)");
    std::vector<int> devs;
    for (int d = 0; d < ndev_all && d < 32; d++)
        if (o.device_mask == 0 || (o.device_mask >> d) & 1u) devs.push_back(d);
    if (devs.empty()) return fail(W2XC_ERR_ARG, "device_mask 0x%x selects no available device (%d present)", o.device_mask, ndev_all);
    const int n = (int)m->layers.size();
    int nd = (int)devs.size();
    // W2XC_HOST_BANDS=<k> (test aid): cut the plane into k host bands, round-robin over the selected devices,
    // so the multi-device band arithmetic below can be exercised on a single-GPU box
    if (const char *e = getenv("W2XC_HOST_BANDS")) {
        const int k = atoi(e);
        if (k > nd) {
            const size_t have = devs.size();
            for (int i = (int)have; i < k && i < 64; i++) devs.push_back(devs[i % have]);
            nd = (int)devs.size();
        }
    }
    if (nd > H) nd = H;

    std::vector<int> rcs(nd, W2XC_OK);
    std::vector<std::string> errs(nd);
    auto worker = [&](int t) {
        // contiguous band [ra, rb) of OUTPUT rows for device t: independent, no exchange
        const int ra = (int)((long long)H * t / nd), rb = (int)((long long)H * (t + 1) / nd);
        auto body = [&]() -> int {
            HIP_TRY(hipSetDevice(devs[t]));
            DevCtx *c = nullptr;
            int r = get_ctx(m, devs[t], &c);
            if (r) return r;
            // source rows that cover output rows [ra - n, rb + n) (clipped), in source coordinates
            const int sy0 = std::max(0, ra - n) >> up, sy1 = (std::min(H, rb + n) + up) >> up;
            const int svh = sy1 - sy0;
            float *d_in = nullptr, *d_out = nullptr;
            hipStream_t st = nullptr;
            HIP_TRY(hipStreamCreate(&st));
            HIP_TRY(hipMalloc((void **)&d_in, (size_t)svh * w * sizeof(float)));
            HIP_TRY(hipMalloc((void **)&d_out, (size_t)(rb - ra) * W * sizeof(float)));
            HIP_TRY(hipMemcpy2DAsync(d_in, (size_t)w * 4, (const char *)in + (size_t)sy0 * in_stride_bytes, in_stride_bytes,
                                     (size_t)w * 4, svh, hipMemcpyHostToDevice, st));
            {
                // the context (its activation workspace) stays locked until this band's stream has drained:
                // with one band per device that is free, and it keeps bands that share a device correct
                std::lock_guard<std::mutex> lk(c->mu);
                r = run_rows(m, c, d_in, w, svh << up, sy0 << up, W, ra, rb, d_out, W, st, o, up);
                if (r == W2XC_OK) {
                    HIP_TRY(hipMemcpy2DAsync((char *)out + (size_t)ra * out_stride_bytes, out_stride_bytes, d_out, (size_t)W * 4,
                                             (size_t)W * 4, rb - ra, hipMemcpyDeviceToHost, st));
                    HIP_TRY(hipStreamSynchronize(st));
                } else {
                    hipStreamSynchronize(st);
                }
            }
            hipFree(d_in);
            hipFree(d_out);
            hipStreamDestroy(st);
            return r;
        };
        rcs[t] = body();
        if (rcs[t]) errs[t] = g_last_error;
    };
    if (nd == 1) {
        int prev = 0;
        hipGetDevice(&prev);
        worker(0);
        hipSetDevice(prev);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < nd; t++) th.emplace_back(worker, t);
        for (auto &x : th) x.join();
    }
    for (int t = 0; t < nd; t++)
        if (rcs[t]) { g_last_error = errs[t]; std::cerr << errs[t] << std::endl; return rcs[t]; }
    return W2XC_OK;
}
}  // namespace

extern "C" {

int w2xc_convert_plane(w2xc_model *m, const float *in, size_t in_stride_bytes, int w, int h, float *out,
                       size_t out_stride_bytes, int block_splitting, const w2xc_opts *opts)
{
    (void)block_splitting;   // results do not depend on the reference's block split
    return convert_plane_host(m, in, in_stride_bytes, w, h, out, out_stride_bytes, opts, 0);
}

int w2xc_convert_plane_nn2x(w2xc_model *m, const float *in, size_t in_stride_bytes, int w, int h, float *out,
                            size_t out_stride_bytes, const w2xc_opts *opts)
{
    return convert_plane_host(m, in, in_stride_bytes, w, h, out, out_stride_bytes, opts, 1);
}

int w2xc_convert_planes_device(w2xc_model *m, int n_in_planes, const float *d_in, size_t in_plane_stride_bytes,
                               size_t in_stride_bytes, int w, int h, float *d_out, size_t out_plane_stride_bytes,
                               size_t out_stride_bytes, void *hip_stream, const w2xc_opts *opts)
{
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
                    n_in_planes, (long long)(in_plane_stride_bytes / 4), (long long)(out_plane_stride_bytes / 4));
}

int w2xc_convert_plane_nn2x_device(w2xc_model *m, const float *d_in, size_t in_stride_bytes, int w, int h, float *d_out,
                                   size_t out_stride_bytes, void *hip_stream, const w2xc_opts *opts)
{
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
                    (hipStream_t)hip_stream, o, 1);
}

}  // extern "C"

extern "C" {

int w2xc_layer_filter(w2xc_model *m, int layer, int n_in_planes, const float *const *in_planes, size_t in_stride_bytes,
                      int w, int h, float *const *out_planes, size_t out_stride_bytes, const w2xc_opts *opts)
{
    if (!m || layer < 0 || layer >= (int)m->layers.size()) return fail(W2XC_ERR_ARG, "bad model/layer");
    const HostLayer &hl = m->layers[layer];
    if (n_in_planes != hl.nin) {   // modelHandler.cpp:29-35
        std::cerr << "Error : Model-filter : \nnumber of input planes mismatch." << std::endl;
        std::cerr << n_in_planes << "," << hl.nin << std::endl;
        return fail(W2XC_ERR_PLANES, "Error : Model-filter : \nnumber of input planes mismatch.\n%d,%d", n_in_planes, hl.nin);
    }
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
    int rc = get_ctx(m, dev, &c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);

    const size_t px = (size_t)w * h;
    float *p_in = nullptr, *p_out = nullptr, *n_in = nullptr, *n_out = nullptr;
    auto cleanup = [&]() { hipFree(p_in); hipFree(p_out); hipFree(n_in); hipFree(n_out); };
    auto body = [&]() -> int {
        const W2xcKernelKind kind = layer_kind(m, layer, o);
        const bool nhwc_in = (kind == W2XC_K_MFMA || kind == W2XC_K_LAST);
        const bool nhwc_out = (kind == W2XC_K_MFMA || kind == W2XC_K_FIRST);
        HIP_TRY(hipMalloc((void **)&p_in, px * hl.nin * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&p_out, px * hl.nout * sizeof(float)));
        for (int i = 0; i < hl.nin; i++)
            HIP_TRY(hipMemcpy2D(p_in + px * i, (size_t)w * 4, in_planes[i], in_stride_bytes, (size_t)w * 4, h, hipMemcpyHostToDevice));
        W2xcConvDesc d;
        memset(&d, 0, sizeof d);
        d.in_h = d.out_h = h;
        d.in_w = d.out_w = w;
        d.off_y = d.off_x = -1;   // same-size conv, BORDER_REPLICATE via clamped loads (:141-142)
        if (nhwc_in) {
            HIP_TRY(hipMalloc((void **)&n_in, px * hl.nin * sizeof(float)));
            HIP_TRY(w2xc_launch_repack(p_in, w, 1, (long long)px, n_in, (long long)w * hl.nin, hl.nin, 1, h, w, hl.nin, nullptr));
            d.in = n_in; d.in_rs = (long long)w * hl.nin; d.in_ps = hl.nin; d.in_cs = 1;
        } else {
            d.in = p_in; d.in_rs = w; d.in_ps = 1; d.in_cs = (long long)px;
        }
        if (nhwc_out) {
            HIP_TRY(hipMalloc((void **)&n_out, px * hl.nout * sizeof(float)));
            d.out = n_out; d.out_rs = (long long)w * hl.nout; d.out_ps = hl.nout; d.out_cs = 1;
        } else {
            d.out = p_out; d.out_rs = w; d.out_ps = 1; d.out_cs = (long long)px;
        }
        int r = launch_layer(c, m, layer, kind, d, nullptr, false);
        if (r) return r;
        if (nhwc_out)
            HIP_TRY(w2xc_launch_repack(n_out, (long long)w * hl.nout, hl.nout, 1, p_out, w, 1, (long long)px, h, w, hl.nout, nullptr));
        HIP_TRY(hipDeviceSynchronize());
        for (int oo = 0; oo < hl.nout; oo++)
            HIP_TRY(hipMemcpy2D(out_planes[oo], out_stride_bytes, p_out + px * oo, (size_t)w * 4, (size_t)w * 4, h, hipMemcpyDeviceToHost));
        return W2XC_OK;
    };
    rc = body();
    cleanup();
    return rc;
}

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
        int rc = run_rows(mn, cn, y, cw, ch, 0, cw, 0, ch, yn, cw, st, o, 0);
        if (rc) return rc;
        y = yn;
    }
    for (int it = 0; it < iterations; it++) {
        const int nw = cw * 2, nh = ch * 2;
        float *y2 = base, *u2 = y2 + (size_t)nw * nh, *v2 = u2 + (size_t)nw * nh;
        base = v2 + (size_t)nw * nh;
        // Y: INTER_NEAREST 2x folded into layer 1 (:136-140) + convertWithModels (:148)
        int rc = run_rows(msc, cs, y, cw, nh, 0, nw, 0, nh, y2, nw, st, o, 1);
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
{
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
}

int w2xc_process_image_u8_device(w2xc_model *noise_model, w2xc_model *scale_model, const unsigned char *d_in, size_t in_stride_bytes,
                                 int w, int h, unsigned char *d_out, size_t out_stride_bytes, int iterations, void *hip_stream,
                                 const w2xc_opts *opts)
{
    return w2xc_process_image_u8_ex_device(noise_model, scale_model, d_in, in_stride_bytes, w, h, d_out, out_stride_bytes, iterations, 0.0,
                                           hip_stream, opts);
}

int w2xc_process_image_u8_ex(w2xc_model *noise_model, w2xc_model *scale_model, const unsigned char *in, size_t in_stride_bytes, int w, int h,
                             unsigned char *out, size_t out_stride_bytes, int iterations, double shrink_ratio, const w2xc_opts *opts)
{
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
    unsigned char *d_in = nullptr, *d_out = nullptr;
    auto body = [&]() -> int {
        HIP_TRY(hipMalloc((void **)&d_in, (size_t)w * 3 * h));
        HIP_TRY(hipMalloc((void **)&d_out, (size_t)W * 3 * H));
        HIP_TRY(hipMemcpy2D(d_in, (size_t)w * 3, in, in_stride_bytes, (size_t)w * 3, h, hipMemcpyHostToDevice));
        int r = process_image_locked(noise_model, scale_model, d_in, (size_t)w * 3, w, h, d_out, (size_t)W * 3, iterations, shrink_ratio,
                                     nullptr, o, dev);
        if (r) return r;
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy2D(out, out_stride_bytes, d_out, (size_t)W * 3, (size_t)W * 3, H, hipMemcpyDeviceToHost));
        return W2XC_OK;
    };
    rc = body();
    hipFree(d_in);
    hipFree(d_out);
    return rc;
}

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
    return w2xc_kernel_name(layer_kind(m, layer, o), m->layers[layer].nin, m->layers[layer].nout);
}

}  // extern "C"
