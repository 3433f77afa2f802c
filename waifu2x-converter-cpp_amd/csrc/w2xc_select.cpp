// w2xc_select.cpp -- which kernel runs which layer, which layers fuse, and the layouts between layers.
#include "w2xc_engine.hpp"

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

}  // namespace w2xc_eng
