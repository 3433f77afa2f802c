// w2xc_rows.cpp -- the band loop that replaces convertWithModels / convertWithModelsBasic / convertWithModelsBlockSplit
// (src/convertRoutine.cpp:21-169) on MI355X, and the device-pointer entry points of include/w2xc_hip.h.
//
// Data layout in HBM: two ping-pong workspaces per (model, device) sized for one band; each boundary between layers picks
// NHWC or planar fp32 (planar_between, w2xc_select.cpp); layer 1 reads the caller's planar plane with clamp-to-edge
// addressing (= copyMakeBorder, convertRoutine.cpp:35,96) and the last layer writes the planar output rows in place
// (= crop + stitch, :40-46,143-161).  A band is `band_rows` output rows x full width; layer k of n computes a valid conv
// on the haloed band (SURVEY invariants I1/I2).
#include "w2xc_engine.hpp"
#include <cmath>
#include <iostream>

namespace w2xc_eng {

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
             float *d_out, size_t out_stride_f, hipStream_t st, const w2xc_opts &o_in, int up, int n_in,
             long long in_cs, long long out_cs, const BandHooks *hk, int plane_h)
{
    RowPlan P;
    {
        int rc = plan_rows(m, o_in, w, vh, vy0, ra, rb, plane_h, n_in, out_cs != 0, &P);
        if (rc) return rc;
    }
    const w2xc_opts &o = P.o;
    const int n = P.n, HL = P.HL, T = P.T, band = P.band;
    const bool all_out = P.all_out, last_direct = P.last_direct;   // (all_out: multi-plane output)
    const W2xcKernelKind last_kind = P.last_kind;
    auto region = [&](int k, int y0, int y1, int &Tk_, int &Bk_) { P.region(k, y0, y1, Tk_, Bk_); };
    for (int i = 0; i < 2; i++)
        if (P.need[i]) { int rc = ensure_ws(c, i, (P.need[i] + 3) / 4); if (rc) return rc; }

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
        bool gathered_in_producer = false;
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
                    // (where the producing launch finishes the last layer itself -- conv3x3_wino4 PROG --, rows start on 128-byte lines: a tile's 32-pixel row
                    //  segment of a tap plane is then exactly ONE line, written whole, and a gather job never pulls a line into its XCD's L2 that holds
                    //  columns of a tile it does not depend on -- with rows of out_w floats such a line, cached before its last columns were written, was
                    //  served stale to the neighbouring job later: intermittent mismatches on small planes)
                    d.out_rs = (T == 0 && gather_in_producer(m, o)) ? ((d.out_w + 31) & ~31) : d.out_w; d.out_ps = 1;
                    d.out_gs = (long long)d.out_h * d.out_rs;
                    d.out_ts = 9 * d.out_gs;
                    d.halves = fused_halves(T, hl.nout);   // (fp32: conv3x3_wino4 writes planar partial planes G[64-plane block][tap][y][x]: its epilogue sums the four plane tiles of a block on chip)
                }
            }
            // fp32, the launch of layer n - 1 FINISHES the fused one-plane last layer itself, rows completing top to bottom while it runs (conv3x3_wino4 PROG):
            //   * host pipeline (hk->prog_begin): ONE launch of layer n - 1, no gather launch, no chunking -- its gather jobs write the band's output rows
            //     straight into page-locked host memory and flag them; the drainer ships rows while the launch is still running (the 0.15-0.25 ms that
            //     three chunked launches of the persistent kernel cost, the six gather launches and their events are gone: DESIGN 7);
            //   * device entry points: only on request (w2xc_opts.fusion = W2XC_FUSION_PROG) -- with planes resident nothing waits for rows, and the launch
            //     is 0.2 ms slower than layer n - 1 + its gather launch (the row-ordered walk, the write-through tap planes).
            // Bit-identical to the gather launch either way (same sum, same order).
            if (T == 0 && k == n - 1 && kind == W2XC_K_MFMA && d.out_terms == 9 && last_kind == W2XC_K_LAST_GATHER && last_direct && !all_out && d.in_ps == 1 &&
                gather_in_producer(m, o) && w >= 4 && (long long)d.out_h * d.out_rs * 4 < (1ll << 32) && ((hk && hk->prog_begin) || (!hk && o.fusion == W2XC_FUSION_PROG))) {
                int trows = 0, groups = 0;
                w2xc_wino4_prog_jobs(d.out_w, d.out_h, d.wino_py, &trows, &groups);
                BandHooks::ProgTail pt;
                pt.out = d_out + (size_t)(y0 - ra) * out_stride_f;
                pt.out_stride_f = (long long)out_stride_f;
                if (hk) {
                    pt.out = nullptr;
                    int rc = hk->prog_begin(y0, y1, trows, groups, &pt);
                    if (rc) return rc;
                }
                if (pt.out) {
                    const size_t nc = w2xc_wino4_prog_counters(d.out_w, d.out_h, d.wino_py);
                    if (c->prog_cnt_n < nc) {
                        if (c->prog_cnt) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(c->prog_cnt)); c->prog_cnt = nullptr; c->prog_cnt_n = 0; }
                        HIP_TRY(hipMalloc((void **)&c->prog_cnt, nc * sizeof(unsigned)));
                        c->prog_cnt_n = nc;
                    }
                    d.prog_cnt = c->prog_cnt;
                    d.g_out = pt.out;
                    d.g_out_rs = pt.out_stride_f;
                    d.g_h = y1 - y0;
                    d.g_w = w;
                    d.g_off = y0 - 1 - Tk;   // rows of this launch's region above the last layer's first input row
                    d.g_bias = c->layers[n - 1].bias;
                    d.prog_flags = pt.flags;
                    d.prog_epoch = pt.epoch;
                    if (hk && hk->prefetch && y1 < rb) { int rc = hk->prefetch(y1, std::min(rb, y1 + band)); if (rc) return rc; }
                    int rc = launch_layer(c, m, k - 1, kind, d, st, o);
                    if (rc) return rc;
                    // job (tile row jr, group jg) holds the output rows [16 jr - first, 16 jr - first + 16) (clipped to the band) x columns [256 jg, 256 jg + 256)
                    if (hk && hk->prog_launched) { rc = hk->prog_launched(y0, y1, trows, groups, d.wino_py + d.g_off); if (rc) return rc; }
                    gathered_in_producer = true;
                    break;
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
                for (int c0 = 0, step = in_chunk; c0 < d.out_h; c0 += step) {
                    step = in_chunk;
                    if (hk->in_chunk_at) { const int s_ = hk->in_chunk_at(c0); if (s_ > 0) step = std::max(8, s_ & ~7); }
                    W2xcConvDesc dd = d;
                    dd.out_h = std::min(step, d.out_h - c0);
                    dd.out = d.out + (size_t)c0 * d.out_rs;
                    dd.off_y = d.off_y + c0;
                    // last view row this chunk reads.  Layer 1 alone: its last row + 2.  The fused launch: the chunk ends on a multiple of 8 LOCAL rows, which
                    // is a 4x4-block edge only when wino_py = 0; the block that straddles the end reads its whole 6-row patch -- every row of it enters every
                    // output row of the block at rounding level (and as NaN if the row holds one) -- so the wait covers the last TOUCHED block: its last row + 4
                    int last = c0 + dd.out_h - 1 + 2;
                    if (kind == W2XC_K_FIRST2_WINO4) last = (((c0 + dd.out_h + d.wino_py + 3) & ~3) - d.wino_py) - 1 + 4;
                    const int vlast = std::min(std::max(last + d.off_y, 0), d.in_h - 1);
                    int rc = hk->input_upto(vlast);
                    if (rc) return rc;
                    rc = launch_layer(c, m, k - 1, kind, dd, st, o);
                    if (rc) return rc;
                }
                src = d.out; src_rs = d.out_rs; src_ps = d.out_ps; src_cs = d.out_cs; src_ts = d.out_ts; src_gs = d.out_gs; src_halves = d.halves;
                src_h = d.out_h; src_w = d.out_w;
                continue;
            }
            if (k == n && gathered_in_producer) break;   // (conv3x3_wino4 PROG finished this layer inside the previous launch; the host pipeline follows its job flags)
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

}  // namespace w2xc_eng

using namespace w2xc_eng;

extern "C" {

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
