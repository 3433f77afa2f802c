// w2xc_engine.hpp -- what the translation units of the engine share (internal; the public boundary is include/w2xc_hip.h).
//
//   w2xc_model.cpp          error state, model container + JSON loader (the reference's Model / modelUtility,
//                           src/modelHandler.{hpp,cpp}), per-(model, device) contexts, the measurement entry points
//   w2xc_select.cpp         which kernel runs which layer, which layers fuse, the layouts between layers, and the band
//                           geometry of run_rows (pure host arithmetic: unit-tested on the CPU through w2xc_plan_rows)
//   w2xc_rows.cpp           launch_layer + run_rows: the band loop that replaces convertWithModels / ...Basic / ...BlockSplit
//                           (src/convertRoutine.cpp:21-169), and the device-pointer entry points
//   w2xc_host_pipeline.cpp  host plane in -> host plane out: staging rings, three streams, feeder + drainer, the unit fan-out
//   w2xc_filter.cpp         Model::filter at the host / device boundary (src/modelHandler.cpp:26-72)
//   w2xc_image.cpp          N2: the CLI's image pipeline around the plane conversion (main.cpp:74-172)
#pragma once
#include "../../include/w2xc_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <exception>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "w2xc_kernels.h"

namespace w2xc_eng {

extern thread_local std::string g_last_error;
int fail(int code, const char *fmt, ...);

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
    // conv3x3_wino4 PROG: two page-locked band buffers the gather jobs write the output rows into over PCIe (bands alternate), and their job flags
    char *pin_band[2] = {nullptr, nullptr};
    size_t band_bytes[2] = {0, 0};
    unsigned *pin_flags[2] = {nullptr, nullptr};
    size_t flags_n[2] = {0, 0};
    unsigned flags_epoch[2] = {0, 0};
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
        for (int i = 0; i < 2; i++) {
            if (pin_band[i]) { hipHostFree(pin_band[i]); pin_band[i] = nullptr; band_bytes[i] = 0; }
            if (pin_flags[i]) { hipHostFree(pin_flags[i]); pin_flags[i] = nullptr; flags_n[i] = 0; flags_epoch[i] = 0; }
        }
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
    unsigned *prog_cnt = nullptr;   // conv3x3_wino4 PROG: gather-job counters of the launch in flight (grow-only)
    size_t prog_cnt_n = 0;
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
        if (prog_cnt) hipFree(prog_cnt);
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

}  // namespace w2xc_eng

struct w2xc_model {
    std::vector<w2xc_eng::HostLayer> layers;
    std::mutex mu;
    std::map<int, std::unique_ptr<w2xc_eng::DevCtx>> ctx;
};

namespace w2xc_eng {

// ---- w2xc_model.cpp ----
w2xc_opts resolve_opts(const w2xc_opts *o);
int njob();                                  // modelUtility's nJob (w2xc_get_jobs)
int upload(const std::vector<float> &h, float **d);
int get_ctx(w2xc_model *m, int device, DevCtx **out);
int ensure_ws(DevCtx *c, int which, size_t floats);
int prof_begin(DevCtx *c, int layer, hipStream_t st, ProfEvent *ev);

// ---- w2xc_select.cpp ----
int split_terms(const w2xc_opts &o);
int split_fmt(const w2xc_opts &o);
W2xcKernelKind layer_kind(const w2xc_model *m, int l, const w2xc_opts &o);
bool fuse_last(const w2xc_model *m, const w2xc_opts &o);
bool fuse_first(const w2xc_model *m, const w2xc_opts &o);
bool fuse_first_fp32(const w2xc_model *m, const w2xc_opts &o);
bool fuse_last_fp32(const w2xc_model *m, const w2xc_opts &o);
bool is_wino4_layer(const w2xc_model *m, int l, const w2xc_opts &o);
int layer_mid_variant(const w2xc_model *m, int l, const w2xc_opts &o);
bool uses_wino4(const w2xc_model *m, const w2xc_opts &o);
bool planar_between(const w2xc_model *m, int l, const w2xc_opts &o);
int out_terms_of(const w2xc_model *m, int l, const w2xc_opts &o);
bool gather_in_producer(const w2xc_model *m, const w2xc_opts &o);
int fused_halves(int T, int cout);
enum MidVariant { MID_MFMA = 0, MID_WINO32 = 1, MID_WINO4 = 3 };

// the band geometry of one run_rows call (plan_rows, w2xc_select.cpp): pure host arithmetic
struct RowPlan {
    w2xc_opts o;              // the options the call runs with (W2XC_FUSION_AUTO gives up a fusion whose kernel cannot address the plane)
    int n = 0, w = 0, plane_h = 0;
    int HL = 1;               // halo rows per layer: 1, or 4 = the banding-invariant geometry of conv3x3_wino4
    int T = 0;                // 16-bit terms per activation (0 = fp32)
    int band = 0;             // output rows per band
    bool last_direct = false, all_out = false;
    W2xcKernelKind last_kind = W2XC_K_DIRECT;
    size_t need[2] = {0, 0};  // bytes of the two ping-pong workspaces for one band
    void region(int k, int y0, int y1, int &T_, int &B_) const;
    void ws_need(const w2xc_model *m, int rows, size_t need_[2]) const;
};
int plan_rows(const w2xc_model *m, const w2xc_opts &o_in, int w, int vh, int vy0, int ra, int rb, int plane_h, int n_in, bool all_out, RowPlan *p);

// ---- w2xc_rows.cpp ----
int launch_layer(DevCtx *c, const w2xc_model *m, int l, W2xcKernelKind kind, W2xcConvDesc d, hipStream_t st, const w2xc_opts &o);

// Hooks of the host->host tile farm into the band loop (all optional; enqueue-only, never synchronise the device):
struct BandHooks {
    int out_chunk_rows = 0;                              // > 0: the last layer of a band is launched in row chunks of at most this size,
    int out_chunk_min = 0;                               //      tapering to this size at the end of the band (the exposed D2H tail)
    std::function<int(int, int)> input_needed;           // before layer 1 of band [y0, y1): make the launch stream wait for its input rows
    // layer 1 in row chunks while the band's input is still arriving: in_chunk(y0, y1) > 0 = output rows of layer 1 per chunk
    // (0: the band's rows are already staged / one launch); input_upto(v) = make the launch stream wait for view rows <= v
    std::function<int(int, int)> in_chunk;
    // optional: rows of the chunk that starts at output row c0 of the launch's region (multiples of 8; <= 0 = in_chunk's value) -- the first chunks of a call
    // are short, so that the first launch follows the first kilobytes of the upload and not its first 2 MiB slice
    std::function<int(int)> in_chunk_at;
    std::function<int(int)> input_upto;
    std::function<int(int, int)> prefetch;               // layers 1..n-1 of the current band are enqueued; [y0, y1) = the NEXT band
    std::function<int(int, int)> output_ready;           // output rows [r0, r1) have been enqueued on the launch stream
    // conv3x3_wino4 PROG (the launch of layer n - 1 finishes the last layer itself, rows completing in order): prog_begin hands out where the band's output
    // rows [y0, y1) go -- page-locked host memory the kernel writes over PCIe -- and the job flags (tile_rows x groups words) with the value a finished job
    // stores; out == nullptr on return: not available, the chunked path runs.  prog_launched: the launch is enqueued; job (jr, jg) holds the band's rows
    // [16 jr - first, 16 jr - first + 16) clipped, columns [256 jg, 256 jg + 256).
    struct ProgTail { float *out = nullptr; long long out_stride_f = 0; unsigned *flags = nullptr; unsigned epoch = 0; };
    std::function<int(int, int, int, int, ProgTail *)> prog_begin;
    std::function<int(int, int, int, int, int)> prog_launched;
};

int run_rows(w2xc_model *m, DevCtx *c, const float *d_in, size_t in_stride_f, int vh, int vy0, int w, int ra, int rb,
             float *d_out, size_t out_stride_f, hipStream_t st, const w2xc_opts &o_in, int up = 0, int n_in = 1,
             long long in_cs = 0, long long out_cs = 0, const BandHooks *hk = nullptr, int plane_h = 0);
int check_plane_args(const w2xc_model *m, const void *in, size_t in_stride, int w, int h, const void *out, size_t out_stride);

}  // namespace w2xc_eng
