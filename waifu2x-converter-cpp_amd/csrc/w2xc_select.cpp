// w2xc_select.cpp -- which kernel runs which layer, which layers fuse, and the layouts between layers.
#include "w2xc_engine.hpp"

#include <cmath>

namespace w2xc_eng {

// 16-bit terms per activation value between the layers of the split pipeline, w2xc_split.hip (0 = not that pipeline)
int split_terms(const w2xc_opts &o)
{
    if (o.precision == W2XC_PRECISION_BF16) return 1;   // plain bf16 = the same pipeline with ONE term
    return (o.precision == W2XC_PRECISION_BF16X2 || o.precision == W2XC_PRECISION_FP16X2) ? 2 : o.precision == W2XC_PRECISION_BF16X3 ? 3 : 0;
}
int split_fmt(const w2xc_opts &o) { return o.precision == W2XC_PRECISION_FP16X2 ? 1 : 0; }

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

// w2xc_opts.fusion: which of the two cross-layer fusions a call may use (both precisions; no environment switch takes part)
static bool fusion_first_on(const w2xc_opts &o) { return o.fusion != W2XC_FUSION_OFF && o.fusion != W2XC_FUSION_LAST; }
static bool fusion_last_on(const w2xc_opts &o) { return o.fusion != W2XC_FUSION_OFF && o.fusion != W2XC_FUSION_FIRST; }

// fp32, last layer fused (fuse_last_fp32): does the producing launch also FINISH it (conv3x3_wino4 PROG: the partial tap planes are summed by the workgroup
// whose arrival completes a 16-row x 256-column job, rows complete top to bottom while the launch runs), or does a conv3x3_last_gather launch follow
// (rounds 4 / 5; w2xc_opts.fusion = W2XC_FUSION_GATHER_LAUNCH, and the shapes PROG has no instantiation for)?  The two are bit-identical.
bool gather_in_producer(const w2xc_model *m, const w2xc_opts &o)
{
    const int n = (int)m->layers.size();
    if (!fuse_last_fp32(m, o) || o.fusion == W2XC_FUSION_GATHER_LAUNCH) return false;
    if (!w2xc_wino4_prog_supported(m->layers[n - 2].nin, m->layers[n - 2].nout)) return false;
    return planar_between(m, n - 3, o);   // (the PROG instantiations read planar planes)
}

// 16-bit modes: layers 1 (ONE plane -> 32) and 2 (32 -> {32,64,128}) run as one kernel (conv3x3_first2_split) when
// layer 2 is an ordinary split mid layer.  w2xc_opts.fusion = W2XC_FUSION_OFF / _LAST disables.
bool fuse_first(const w2xc_model *m, const w2xc_opts &o)
{
    const int n = (int)m->layers.size();
    if (!fusion_first_on(o) || o.kernel == W2XC_KERNEL_DIRECT || n < 3 || split_terms(o) == 0) return false;
    if (m->layers[0].nin != 1 || m->layers[0].nout != 32) return false;
    if (w2xc_pick_kernel(m->layers[1].nin, m->layers[1].nout) != W2XC_K_MFMA) return false;
    return !(n == 3 && fuse_last(m, o));   // (layer 2 would be the fused-last producer: keep that fusion instead)
}

// 16-bit modes: the last layer (cin in {32,64,128} -> ONE plane) is computed inside the epilogue of the mid layer
// before it (conv3x3_split, out_terms = 9) and finished by conv3x3_last_gather.  w2xc_opts.fusion = W2XC_FUSION_OFF / _FIRST disables.
bool fuse_last(const w2xc_model *m, const w2xc_opts &o)
{
    const int n = (int)m->layers.size();
    if (!fusion_last_on(o) || o.kernel == W2XC_KERNEL_DIRECT || n < 3) return false;
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

// fp32 path, layers with 32 / 64 / 128 planes in and out (W2XC_K_MFMA): which kernel runs them.
//   MID_MFMA    conv3x3_mfma2: direct implicit GEMM, a k-ordered fp32 fma chain (the closest MFMA analogue of modelHandler.cpp:134-145)
//   MID_WINO32  conv3x3_wino:   Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32, one wave per SIMD (round 2)
// Same arithmetic type (fp32 throughout); Winograd does 2.25x fewer multiplies in another summation order and is held to the same
// rtol 1e-4 gate against the CPU oracle by the same tests.  w2xc_opts.kernel picks per call (W2XC_KERNEL_MFMA / _WINOGRAD /
// _WINOGRAD32 / _WINOGRAD4); W2XC_KERNEL_AUTO = W2XC_KERNEL_WINOGRAD4: conv3x3_wino4 (F(4x4,3x3)) where it applies (>= 64 output planes),
// conv3x3_wino for the rest.  No environment variable takes part in the choice.
//   MID_WINO4   conv3x3_wino4:  Winograd F(4x4,3x3) on v_mfma_f32_16x16x4_f32 (round 3, the default: 1.78x fewer multiplies again, ~1.3x the rounding error of
//               F(2x2); needs the four-rows-per-layer band geometry of run_rows to stay banding-invariant)
namespace {
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
}  // namespace

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
    if (!fusion_last_on(o)) return false;   // (W2XC_FUSION_AUTO = on)
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
    if (split_terms(o) != 0 || o.precision != W2XC_PRECISION_FP32 || n < 3 || !fusion_first_on(o)) return false;
    if (o.kernel != W2XC_KERNEL_AUTO && o.kernel != W2XC_KERNEL_WINOGRAD4) return false;
    const HostLayer &a = m->layers[0], &b = m->layers[1];
    if (!w2xc_first2_wino4_supported(a.nin, a.nout, b.nout) || b.nin != a.nout) return false;
    if (n == 4 && fuse_last_fp32(m, o)) return false;   // (layer 3 would carry the fused last layer: that instantiation reads 32 NHWC planes only)
    return is_wino4_layer(m, 2, o);   // (layer 3 = conv3x3_wino4: it reads layer 2's planar planes)
}


// ------------------------------------------------------------------------------------------------
// The band geometry of run_rows (w2xc_rows.cpp) as pure host arithmetic: no device, no allocation.  w2xc_plan_rows exposes it
// (tests/test_plan.py runs it on the CPU box).
//
// conv3x3_wino4 (F(4x4,3x3)): an output of a 4x4 block depends, at rounding level, on all 36 patch values, so a block cut by the edge of a
// band's region (clamped rows instead of the plane's) would make results depend on the banding.  HL = 4: every layer k < n computes the rows
//     [floor4(y0) - 4 (n - k), ceil4(y1) + 4 (n - k))  clipped to the layer's plane extent [-(n - k), H + (n - k))
// of a band [y0, y1) instead of [y0 - (n - k), y1 + (n - k)): every region edge that is not a plane edge is a block edge (blocks sit on rows
// = 0 mod 4 of the plane), and layer k + 1 finds the rows it reads (one more each side) inside.  Costs up to 3 + 3 (n - k) more rows per side.
void RowPlan::region(int k, int y0, int y1, int &T_, int &B_) const   // plane rows [T, B) layer k computes for the band [y0, y1)
{
    if (HL == 1 || k == n) { T_ = y0 - (n - k); B_ = y1 + (n - k); return; }
    T_ = std::max(-(n - k), (y0 & ~3) - 4 * (n - k));
    B_ = std::min(plane_h + (n - k), ((y1 + 3) & ~3) + 4 * (n - k));
}

// BYTES per band of `rows` output rows for the two ping-pong buffers (layer k's output goes to ws[(k - 1) & 1])
void RowPlan::ws_need(const w2xc_model *m, int rows, size_t need[2]) const
{
    need[0] = need[1] = 0;
    for (int k = 1; k <= n; k++) {
        if (k == n && last_direct) break;   // written straight to d_out
        if (k == 1 && layer_kind(m, 0, o) == W2XC_K_FUSED_AWAY) continue;   // layer 1's activations stay on chip
        const size_t hk = (size_t)rows + ((HL == 1 || k == n) ? 2 * (n - k) : 6 + 8 * (n - k)), wk = (size_t)w + 2 * (n - k);
        const int ot = out_terms_of(m, k - 1, o);
        const bool fused = ot == 9;   // partial G planes of the fused last layer
        const size_t bpe = (ot >= 1 && ot <= 3) ? 2 * (size_t)ot : 4;   // bytes per activation element of layer k's output
        const size_t px_bytes = fused ? (size_t)fused_halves(T, m->layers[k - 1].nout) * 9 * 4 : m->layers[k - 1].nout * bpe;
        const size_t wk_mem = (planar_between(m, k - 1, o) || (fused && T == 0 && gather_in_producer(m, o))) ? ((wk + 31) & ~(size_t)31) : wk;   // planar rows (and the tap planes a PROG launch gathers itself) start on 128-byte lines
        need[(k - 1) & 1] = std::max(need[(k - 1) & 1], hk * wk_mem * px_bytes);
    }
}

// Output rows [ra, rb) of an h-row plane of which the view holds rows [vy0, vy0 + vh): halo geometry, band height, workspace bytes, and the
// options the call really runs with (W2XC_FUSION_AUTO gives up a fusion whose kernel cannot address the plane; an explicit request fails instead).
int plan_rows(const w2xc_model *m, const w2xc_opts &o_in, int w, int vh, int vy0, int ra, int rb, int plane_h, int n_in, bool all_out, RowPlan *p)
{
    RowPlan &P = *p;
    const int n = (int)m->layers.size();
    if (n == 0) return fail(W2XC_ERR_ARG, "model has no layers");
    P.o = o_in;
    P.n = n; P.w = w; P.plane_h = plane_h; P.all_out = all_out;
    P.HL = 1;
    if (uses_wino4(m, P.o)) {
        const int hs = 4 * n;
        if (plane_h > 0 && vy0 <= std::max(0, ra - hs) && vy0 + vh >= std::min(plane_h, rb + hs)) P.HL = 4;
        else if (P.o.kernel == W2XC_KERNEL_AUTO)   // a view with n halo rows only: no silent change of kernel (and rounding) -- the caller decides
            return fail(W2XC_ERR_ARG, "row-band view [%d,%d) of rows [%d,%d): the default F(4x4) kernel needs %d halo rows (4 per layer) for banding-invariant results; "
                                      "pass the wide view or choose w2xc_opts.kernel explicitly (W2XC_KERNEL_WINOGRAD32: F(2x2), banding-invariant on the minimum view)",
                        vy0, vy0 + vh, ra, rb, hs);
        // (an explicit W2XC_KERNEL_WINOGRAD4 on a narrow view runs as asked: results then depend on the banding at rounding level)
    }
    if (m->layers[0].nin != n_in)   // convertWithModelsBasic pushes exactly one plane (convertRoutine.cpp:63-64)
        return fail(W2XC_ERR_PLANES, "Error : Model-filter : \nnumber of input planes mismatch.\n%d,%d", n_in, m->layers[0].nin);
    for (int l = 1; l < n; l++)
        if (m->layers[l].nin != m->layers[l - 1].nout)
            return fail(W2XC_ERR_PLANES, "Error : Model-filter : \nnumber of input planes mismatch.\n%d,%d", m->layers[l - 1].nout, m->layers[l].nin);
    P.T = split_terms(P.o);
    if (P.o.precision != W2XC_PRECISION_FP32 && P.T == 0) return fail(W2XC_ERR_ARG, "unknown precision %d", P.o.precision);
    if (P.o.fusion < W2XC_FUSION_AUTO || P.o.fusion > W2XC_FUSION_PROG) return fail(W2XC_ERR_ARG, "unknown w2xc_opts.fusion %d", P.o.fusion);
    if (P.T > 0)
        for (int l = 0; l < n; l++)
            if (layer_kind(m, l, P.o) == W2XC_K_DIRECT)
                return fail(W2XC_ERR_UNSUPPORTED, "16-bit precision modes: layer %d (%d->%d) has no kernel ({1,3}->{32,64,128} first, {32,64,128}->{32,64,128}, ->{1,3} last only)",
                            l + 1, m->layers[l].nin, m->layers[l].nout);
    // conv3x3_first2_wino4 addresses its 32 output planes with 32-bit lane offsets (12 plane strides + a row): planes of at most 64 Mi floats.
    // A plane too wide for that even in the shortest band: W2XC_FUSION_AUTO runs the first two layers unfused, an explicit request is refused.
    size_t first2_max_rows = 0, first2_halo = 0;
    if (fuse_first_fp32(m, P.o)) {
        const size_t wk = ((size_t)w + 2 * (n - 2) + 31) & ~(size_t)31;
        first2_max_rows = ((size_t)64 << 20) / wk;
        first2_halo = P.HL == 1 ? 2 * (size_t)(n - 2) : 6 + 8 * (size_t)(n - 2);
        if (first2_max_rows < first2_halo + 8) {
            if (P.o.fusion != W2XC_FUSION_AUTO)
                return fail(W2XC_ERR_UNSUPPORTED, "plane too wide (%d pixels) for the fused first layers; use w2xc_opts.fusion = W2XC_FUSION_AUTO, _LAST or _OFF", w);
            P.o.fusion = W2XC_FUSION_LAST;
            first2_max_rows = 0;
        }
    }
    // the last layer stores straight into the caller's planar plane(s) when its kernel can address planar
    // output (conv3x3_last / conv3x3_direct); otherwise it goes through the NHWC workspace + a repack
    P.last_kind = layer_kind(m, n - 1, P.o);
    P.last_direct = (m->layers[n - 1].nout == 1 || all_out) &&
                    (P.last_kind == W2XC_K_LAST || P.last_kind == W2XC_K_LAST_GATHER || P.last_kind == W2XC_K_DIRECT ||
                     (m->layers[n - 1].nout == 1 && P.last_kind != W2XC_K_MFMA && P.last_kind != W2XC_K_FIRST));
    int band = P.o.band_rows;
    const int total = rb - ra;
    if (band <= 0) {
        const size_t budget = (size_t)(P.o.workspace_mb > 0 ? P.o.workspace_mb : 16384) << 20;
        size_t need[2];
        P.ws_need(m, total, need);
        if (need[0] + need[1] <= budget) band = total;
        else {
            // bytes grow linearly in rows: solve on two probes
            size_t n1[2], n2[2];
            P.ws_need(m, 1, n1);
            P.ws_need(m, 2, n2);
            const double per_row = (double)((n2[0] + n2[1]) - (n1[0] + n1[1]));
            const double base = (double)(n1[0] + n1[1]) - per_row;
            band = (int)std::floor(((double)budget - base) / per_row);
            if (band < 1) band = 1;
            if (band > total) band = total;
            // each buffer's need is a MAX over layers, so the slope measured at 1..2 rows is that of the wide-halo,
            // few-plane layers and under-estimates large bands: re-evaluate the real need and shrink until it fits
            for (int it = 0; it < 64 && band > 1; it++) {
                P.ws_need(m, band, need);
                if (need[0] + need[1] <= budget) break;
                const int nb2 = (int)((double)band * (double)budget / (double)(need[0] + need[1]));
                band = std::max(1, std::min(band - 1, nb2));
            }
            const int nb = (total + band - 1) / band;
            band = (total + nb - 1) / nb;   // equalise (never larger than the band that was just checked)
        }
    }
    // (the addressing limit of the fused first layers also bounds a band the caller asked for: w2xc_plan_rows reports the height that runs)
    if (first2_max_rows && (size_t)band + first2_halo > first2_max_rows) band = (int)(first2_max_rows - first2_halo);
    if (P.HL > 1 && band < total) band = std::max(4, band & ~3);   // (band edges on block rows: no rounding-out rows)
    P.band = std::min(band, total);
    P.ws_need(m, P.band, P.need);
    return W2XC_OK;
}

}  // namespace w2xc_eng
