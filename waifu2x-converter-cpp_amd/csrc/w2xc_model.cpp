// w2xc_model.cpp -- error state, the model container + JSON loader (the reference's Model / modelUtility,
// src/modelHandler.{hpp,cpp}), the per-(model, device) contexts that hold packed weights and workspaces, and the
// measurement entry points of include/w2xc_hip.h.
#include "w2xc_engine.hpp"

#include <cmath>
#include <fstream>
#include <iostream>
#include <sstream>

#include "json_min.hpp"

namespace w2xc_eng {

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

// modelUtility singleton state (modelHandler.hpp:92-100)
static std::mutex g_util_mu;
static int g_njob = 4;
static int g_block_w = 512, g_block_h = 512;

int njob() { std::lock_guard<std::mutex> lk(g_util_mu); return g_njob; }

// what `opts == NULL` resolves to (w2xc_set_default_opts): w2xc_opts_init's defaults + the two environment variables the library reads
static std::mutex g_defaults_mu;
static w2xc_opts g_defaults;
static bool g_defaults_set = false;
static w2xc_opts env_defaults()
{
    w2xc_opts r;
    w2xc_opts_init(&r);
    // callers that pass no options (the C++ adapter behind the reference's CLI): the default precision can
    // be switched without recompiling -- W2XC_PRECISION = fp32 | bf16x3 | fp16x2 | bf16x2 | bf16
    if (const char *e = getenv("W2XC_PRECISION")) {
        if (!strcmp(e, "bf16x3")) r.precision = W2XC_PRECISION_BF16X3;
        else if (!strcmp(e, "bf16x2")) r.precision = W2XC_PRECISION_BF16X2;
        else if (!strcmp(e, "fp16x2")) r.precision = W2XC_PRECISION_FP16X2;
        else if (!strcmp(e, "bf16")) r.precision = W2XC_PRECISION_BF16;
    }
    if (const char *e = getenv("W2XC_FILTER_RESIDENT")) r.filter_resident = atoi(e) != 0 ? 1 : 0;
    return r;
}

w2xc_opts resolve_opts(const w2xc_opts *o)
{
    w2xc_opts r;
    if (o) {
        w2xc_opts_init(&r);
        size_t n = o->struct_size > 0 && (size_t)o->struct_size < sizeof(w2xc_opts) ? (size_t)o->struct_size : sizeof(w2xc_opts);
        memcpy(&r, o, n);
        r.struct_size = (int)sizeof(w2xc_opts);
    } else {
        std::lock_guard<std::mutex> lk(g_defaults_mu);
        if (!g_defaults_set) { g_defaults = env_defaults(); g_defaults_set = true; }
        r = g_defaults;
    }
    return r;
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

}  // namespace w2xc_eng

using namespace w2xc_eng;

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

void w2xc_opts_init_sized(w2xc_opts *o, size_t struct_size)
{
    if (!o || struct_size < 4 * sizeof(int)) return;   // (struct_size, precision, kernel, device: every version of the struct has them)
    w2xc_opts full;
    w2xc_opts_init(&full);
    const size_t n = std::min(struct_size, sizeof(w2xc_opts));
    full.struct_size = (int)n;
    memcpy(o, &full, n);
}

int w2xc_set_default_opts(const w2xc_opts *defaults)
{
    w2xc_opts r;
    if (defaults) {
        w2xc_opts_init(&r);
        const size_t n = defaults->struct_size > 0 && (size_t)defaults->struct_size < sizeof(w2xc_opts) ? (size_t)defaults->struct_size : sizeof(w2xc_opts);
        memcpy(&r, defaults, n);
        r.struct_size = (int)sizeof(w2xc_opts);
    } else {
        r = env_defaults();
    }
    std::lock_guard<std::mutex> lk(g_defaults_mu);
    g_defaults = r;
    g_defaults_set = true;
    return W2XC_OK;
}

const char *w2xc_last_error(void) { return g_last_error.c_str(); }
// 0.2: w2xc_opts grew (host_numa; 56 bytes), W2XC_FUSION_FIRST / _LAST, w2xc_opts_init_sized, w2xc_set_default_opts, w2xc_plan_rows; since 0.1 (rounds 3-5)
// also: W2XC_KERNEL_WINOGRAD = _WINOGRAD32, W2XC_KERNEL_AUTO refused on a minimum-halo row view, `verbose` a bit mask (INTEGRATION.md "ABI history")
const char *w2xc_version(void) { return "w2xc_hip 0.2 (gfx950)"; }

int w2xc_plan_rows(const w2xc_model *m, int w, int view_y0, int view_h, int plane_h, int row_begin, int row_end, const w2xc_opts *opts, w2xc_row_plan *plan)
try {
    if (!m || !plan) return fail(W2XC_ERR_ARG, "null argument");
    const int n = (int)m->layers.size();
    if (w <= 0 || plane_h <= 0 || row_begin < 0 || row_end > plane_h || row_begin >= row_end)
        return fail(W2XC_ERR_ARG, "bad row range [%d,%d) for a %d-row plane", row_begin, row_end, plane_h);
    if (view_y0 < 0 || view_h <= 0 || view_y0 + view_h > plane_h || view_y0 > std::max(0, row_begin - n) || view_y0 + view_h < std::min(plane_h, row_end + n))
        return fail(W2XC_ERR_ARG, "view rows [%d,%d) do not cover [%d,%d) +- %d halo rows", view_y0, view_y0 + view_h, row_begin, row_end, n);
    RowPlan P;
    int rc = plan_rows(m, resolve_opts(opts), w, view_h, view_y0, row_begin, row_end, plane_h, m->layers.empty() ? 1 : m->layers[0].nin, false, &P);
    if (rc) return rc;
    w2xc_row_plan r;
    memset(&r, 0, sizeof r);
    r.struct_size = (int)sizeof r;
    r.n_layers = P.n;
    r.halo_rows_per_layer = P.HL;
    r.band_rows = P.band;
    r.n_bands = (row_end - row_begin + P.band - 1) / P.band;
    r.fused_first = (split_terms(P.o) ? fuse_first(m, P.o) : fuse_first_fp32(m, P.o)) ? 1 : 0;
    r.fused_last = (split_terms(P.o) ? fuse_last(m, P.o) : fuse_last_fp32(m, P.o)) ? 1 : 0;
    r.workspace_bytes[0] = P.need[0];
    r.workspace_bytes[1] = P.need[1];
    const size_t nb = plan->struct_size > 0 && (size_t)plan->struct_size < sizeof r ? (size_t)plan->struct_size : sizeof r;
    r.struct_size = (int)nb;
    memcpy(plan, &r, nb);
    return W2XC_OK;
} W2XC_CATCH_ALL

int w2xc_plan_region(const w2xc_row_plan *plan, int plane_h, int layer, int y0, int y1, int *top, int *bottom)
{
    if (!plan || !top || !bottom || layer < 1 || layer > plan->n_layers || y0 >= y1) return fail(W2XC_ERR_ARG, "bad argument");
    RowPlan P;
    P.n = plan->n_layers;
    P.HL = plan->halo_rows_per_layer;
    P.plane_h = plane_h;
    P.region(layer, y0, y1, *top, *bottom);
    return W2XC_OK;
}

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
        for (int i = 0; i < 2; i++)
            if (p.pin_band[i]) { hipHostFree(p.pin_band[i]); p.pin_band[i] = nullptr; p.band_bytes[i] = 0; }
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
    // (what the DEVICE entry points launch: conv3x3_wino4 PROG finishes the last layer itself there only on request; the host entry points always use it)
    if (k == W2XC_K_LAST_GATHER && gather_in_producer(m, o) && o.fusion == W2XC_FUSION_PROG) return "(in_previous_layer)";
    if (k == W2XC_K_MFMA) {
        const int midv = layer_mid_variant(m, layer, o);
        if (midv != MID_MFMA) return midv == MID_WINO4 ? "conv3x3_wino4" : "conv3x3_wino";
    }
    return w2xc_kernel_name(k, m->layers[layer].nin, m->layers[layer].nout);
}

}  // extern "C"
