// w2xc_host_pipeline.cpp -- host plane in -> host plane out (w2xc_convert_plane / _nn2x / _rows): per (model, device) a
// persistent pipe of three HIP streams, device copies of the unit's rows and pinned staging rings; a feeder and a drainer
// thread per unit; units fan out over devices.  The parallel replacement of the sequential block walk of
// src/convertRoutine.cpp:114-165 (host-side gather only, no exchange between units).
#include "w2xc_engine.hpp"
#include <sched.h>
#include <pthread.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <thread>

#include "w2xc_copy_pool.hpp"

namespace w2xc_eng {

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
// on the CPU node the GPU hangs off, or every staged byte crosses the inter-socket link twice).  w2xc_opts.host_numa = 1 disables. ----
int numa_node_of_device(int dev)
{
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
    explicit NodeAffinity(int dev, bool enabled)
    {
        if (!enabled) return;
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
    NodeAffinity node_guard(dev, o.host_numa == 0);
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
    // slot >= 0: rows [r0, r1) arrive in staging slot `slot` behind its D2H event.  slot < 0: a band whose last layer the launch of layer n - 1 finishes itself
    // (conv3x3_wino4 PROG): its gather jobs write the rows into the page-locked band buffer `src` over PCIe and flag them (job (jr, jg) = rows
    // [16 jr - first, 16 jr - first + 16) x columns [256 jg, 256 jg + 256) of the band); the drainer follows the flags and stitches rows while the launch runs
    struct Chunk { int r0, r1, slot; int trows = 0, groups = 0, first = 0; const volatile unsigned *flags = nullptr; unsigned epoch = 0; const char *src = nullptr; };
    std::mutex qmu;
    std::condition_variable qcv;
    std::deque<Chunk> pending;     // D2H queued, not yet stitched (drainer consumes in order)
    long queued = 0, drained = 0;  // chunk counters (slots are used round-robin)
    long prog_bands = 0, prog_bands_drained = 0;   // PROG bands handed to the drainer / stitched (the band buffers alternate)
    bool feeder_done = false;
    std::atomic<int> drain_rc{W2XC_OK};
    std::string drain_err;
    // (started with the first chunk it is handed, not at the call's start: creating a thread costs tens of microseconds the first upload would wait behind)
    std::thread drainer;
    auto start_drainer = [&] {
        if (out_pinned || drainer.joinable()) return;
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
                if (ch.slot < 0) {
                    // follow the job flags: tile rows complete top to bottom (roughly); every run of finished tile rows is stitched at once
                    const int R = ch.r1 - ch.r0;
                    int jr = 0, runs = 0;
                    long spins = 0;
                    bool launch_over = false;
                    double t_first_seen = 0, t_last_seen = 0;
                    while (jr < ch.trows && drain_rc.load() == W2XC_OK) {
                        int ready = jr;
                        while (ready < ch.trows) {
                            bool all = true;
                            for (int g = 0; g < ch.groups && all; g++) all = (int)(ch.flags[(size_t)ready * ch.groups + g] - ch.epoch) >= 0;
                            if (!all && !launch_over) break;
                            ready++;
                        }
                        if (ready == jr) {
                            // nothing new: the launch may be over (every row written, the flags' last stores included -- or it failed)
                            if ((++spins & 1023) == 0) {
                                const hipError_t q = hipStreamQuery(p.s_compute);
                                if (q == hipSuccess) launch_over = true;
                                else if (q != hipErrorNotReady) {
                                    drain_err = std::string("the launch that finishes the last layer failed: ") + hipGetErrorString(q);
                                    drain_rc.store(W2XC_ERR_HIP);
                                }
                            }
#if defined(__x86_64__)
                            __builtin_ia32_pause();
#endif
                            continue;
                        }
                        std::atomic_thread_fence(std::memory_order_acquire);
                        const int a = std::max(0, 16 * jr - ch.first), b = std::min(R, 16 * ready - ch.first);
                        const double t_seen = trace ? ms_since(t0) : 0.0;
                        if (b > a) {
                            try {
                                w2xc_host::CopyPool::get().copy_rows((char *)out + (size_t)(ch.r0 + a) * out_stride, out_stride, ch.src + (size_t)a * out_row, out_row, out_row,
                                                                     b - a, copy_threads);
                            } catch (const std::exception &ex) {
                                drain_err = std::string("host copy of finished rows failed: ") + ex.what();
                                drain_rc.store(W2XC_ERR_NOMEM);
                            } catch (...) {
                                drain_err = "host copy of finished rows failed";
                                drain_rc.store(W2XC_ERR_NOMEM);
                            }
                        }
                        if (trace) {
                            if (runs == 0) t_first_seen = t_seen;
                            runs++;
                            t_last_seen = t_seen;
                        }
                        jr = ready;
                    }
                    if (trace) fprintf(stderr, "[w2xc host] rows %d..%d finished by the launch of layer n-1 itself: %d tile rows stitched in %d runs, first seen %.3f ms, last seen %.3f ms, "
                                               "stitched %.3f ms%s\n", ch.r0, ch.r1, ch.trows, runs, t_first_seen, t_last_seen, ms_since(t0), launch_over ? " (the launch was over before its last flags were seen)" : "");
                    {
                        std::lock_guard<std::mutex> ql(qmu);
                        prog_bands_drained++;
                    }
                    qcv.notify_all();
                    continue;
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
    };
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
    // the call's first chunks are an eighth, a quarter, a half of a slice: the first launch starts ~30 us into the call instead of behind the first 2 MiB
    // (pageable: staged by the copy threads first); later chunks are whole slices, every launch of the persistent kernel has a ramp
    hk.in_chunk_at = [&](int c0) -> int {
        const int full = std::max(8, ((in_chunk_rows << up) + 7) & ~7);
        if (uploaded > 0 && c0 == 0) return 0;   // (a later band whose first rows were prefetched)
        return c0 < full / 8 ? full / 8 : c0 < full / 8 + full / 4 ? full / 4 : c0 < full / 8 + full / 4 + full / 2 ? full / 2 : full;
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
            start_drainer();
            {
                std::lock_guard<std::mutex> ql(qmu);
                pending.push_back({a, b2, slot});
                queued++;
            }
            qcv.notify_all();
        }
        return W2XC_OK;
    };

    // conv3x3_wino4 PROG: the band's rows are written by the launch of layer n - 1 itself, into the caller's plane when it is page-locked, else into one of
    // two page-locked band buffers (bands alternate; band s waits for band s - 2 to have been stitched), and flagged per job for the drainer
    hk.prog_begin = [&](int y0, int y1, int trows, int groups, BandHooks::ProgTail *pt) -> int {
        if (o.fusion == W2XC_FUSION_GATHER_LAUNCH) return W2XC_OK;   // (pt->out stays null: the chunked launches + gather of rounds 4 / 5)
        if (out_pinned) {   // nothing to stitch: the rows are complete when the launch is (the closing synchronisation)
            pt->out = (float *)((char *)out + (size_t)y0 * out_stride);
            pt->out_stride_f = (long long)(out_stride / 4);
            pt->flags = nullptr;
            return W2XC_OK;
        }
        const int par = (int)(prog_bands & 1);
        const size_t need = (size_t)(y1 - y0) * out_row, nflags = (size_t)trows * groups;
        if (p.band_bytes[par] < need) {
            HIP_TRY(hipStreamSynchronize(p.s_compute));   // (an earlier launch may still write the old buffer)
            if (p.pin_band[par]) { HIP_TRY(hipHostFree(p.pin_band[par])); p.pin_band[par] = nullptr; p.band_bytes[par] = 0; }
            // (COHERENT: uncached on the GPU side, every store goes out over PCIe at once -- default host allocations may be cached in the GPU's L2 until the launch
            //  ends: the system-scope flag then arrived long before the rows it announces, measured)
            if (hipHostMalloc((void **)&p.pin_band[par], need, hipHostMallocCoherent) != hipSuccess) return fail(W2XC_ERR_NOMEM, "hipHostMalloc(%zu MiB) of the band buffer failed", need >> 20);
            p.band_bytes[par] = need;
        }
        if (p.flags_n[par] < nflags) {
            HIP_TRY(hipStreamSynchronize(p.s_compute));
            if (p.pin_flags[par]) { HIP_TRY(hipHostFree(p.pin_flags[par])); p.pin_flags[par] = nullptr; p.flags_n[par] = 0; }
            if (hipHostMalloc((void **)&p.pin_flags[par], nflags * sizeof(unsigned), hipHostMallocCoherent) != hipSuccess) return fail(W2XC_ERR_NOMEM, "hipHostMalloc of the job flags failed");
            memset(p.pin_flags[par], 0, nflags * sizeof(unsigned));
            p.flags_n[par] = nflags;
            p.flags_epoch[par] = 0;
        }
        {   // the buffer's previous band has been stitched
            std::unique_lock<std::mutex> ql(qmu);
            qcv.wait(ql, [&] { return prog_bands_drained >= prog_bands - 1 || drain_rc.load() != W2XC_OK; });
        }
        if (drain_rc.load()) return drain_rc.load();
        if (p.flags_epoch[par] == 0x7FFFFFFFu) { memset(p.pin_flags[par], 0, p.flags_n[par] * sizeof(unsigned)); p.flags_epoch[par] = 0; }   // (the epoch comparison is signed: start over, nothing is in flight on this buffer)
        pt->out = (float *)p.pin_band[par];
        pt->out_stride_f = (long long)W;
        pt->flags = p.pin_flags[par];
        pt->epoch = ++p.flags_epoch[par];
        return W2XC_OK;
    };
    hk.prog_launched = [&](int y0, int y1, int trows, int groups, int first) -> int {
        if (trace && t_first_out < 0) t_first_out = ms_since(t0);
        if (out_pinned) return W2XC_OK;
        const int par = (int)(prog_bands & 1);
        Chunk ch{y0, y1, -1};
        ch.trows = trows; ch.groups = groups; ch.first = first;
        ch.flags = p.pin_flags[par]; ch.epoch = p.flags_epoch[par]; ch.src = p.pin_band[par];
        start_drainer();
        {
            std::lock_guard<std::mutex> ql(qmu);
            pending.push_back(ch);
            prog_bands++;
        }
        qcv.notify_all();
        return W2XC_OK;
    };

    rc = run_rows(m, c, p.d_in, w, svh << up, sy0 << up, W, ra, rb, p.d_out, W, p.s_compute, o, up, 1, 0, 0, &hk, H);
    const double t_enq = ms_since(t0);
    double t_comp = 0;
    // (not while a PROG band is being followed: the drainer polls hipStreamQuery on the same stream, and a synchronise in flight here holds it up until the launch ends)
    if (trace && prog_bands == 0) { hipStreamSynchronize(p.s_compute); t_comp = ms_since(t0); }
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

}  // namespace

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

}  // namespace w2xc_eng

using namespace w2xc_eng;

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

}  // extern "C"
