// w2xc_kernels.h -- launch interface between the engine (w2xc_rows.cpp, see w2xc_engine.hpp) and the gfx950
// kernels (w2xc_kernels.hip).  One "layer launch" computes
//     out(y,x,o) = leaky( bias[o] + sum_{i,r,c} W[o][i][r][c] * in(clamp(y+r+off_y), clamp(x+c+off_x), i) )
// which is Model::filterWorker (/root/reference/src/modelHandler.cpp:117-159) on one haloed
// band: off = 0 gives the shrinking "valid" conv used inside convertWithModels,
// off = -1 with out dims == in dims gives the same-size BORDER_REPLICATE conv of Model::filter,
// off = -n_layers on layer 1 folds cv::copyMakeBorder (convertRoutine.cpp:35,96) into the load.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct W2xcConvDesc {
    const float *in;
    float *out;
    const float *wpk;    // weights packed for the chosen kernel
    const float *bias;   // float[cout]  ((float)double, modelHandler.cpp:147 via cv::add scalar rule)
    int cin, cout;
    // element (y,x,c) lives at base[y*rs + x*ps + c*cs] (units: floats)
    long long in_rs, in_ps, in_cs;
    long long out_rs, out_ps, out_cs;
    int in_h, in_w;      // extent of the input view (clamp bounds)
    int out_h, out_w;    // region to compute
    int off_y, off_x;
    // N1 (nearest-neighbour 2x of main.cpp:132-140 folded into the load): in_h/in_w and all offsets are
    // in UPSCALED coordinates, memory is addressed at (y >> in_shift, x >> in_shift).  0 or 1; only the
    // first-layer kernels (conv3x3_first, conv3x3_direct) honour it.
    int in_shift;
    // split kernels (w2xc_split.hip): an activation tensor is `terms` 16-bit (bf16 / fp16) term planes, each channel-group
    // blocked: element (t, c, y, x) at t*ts + (c / G)*gs + y*rs + x*G + c % G  (ELEMENTS; G = 16).
    // out_terms = 0 stores plain fp32 NHWC.
    int terms, out_terms;
    long long in_ts, out_ts, in_gs, out_gs;
    int fmt;             // 0 = bf16 terms, 1 = fp16 terms (W2XC_PRECISION_FP16X2)
    float acc_scale;     // fp16: 1 / (power-of-two weight scale of this layer), applied to the accumulators
    // out_terms = 9: the LAST layer (cout = 1) is computed inside this layer's epilogue; `out` receives its partial sums
    // G[half][tap][y][x] fp32 (out_ts = half stride, out_gs = tap-plane stride, out_rs = row stride, in floats) and
    // W2XC_K_LAST_GATHER adds halves and taps (in = G, in_ts / in_gs / in_rs as written, `halves` halves).
    // W2XC_K_FIRST2_SPLIT (layers 1 + 2 in one kernel): `in` / in_* / off_* / in_shift describe LAYER 1's input plane,
    // wpk / bias / acc_scale / terms / fmt / out_* layer 2; layer 1's own weights (W2XC_K_FIRST image) and bias:
    const float *w1pk;
    const float *bias1;
    const void *w7pk;    // last layer's weights as MFMA A fragments (w2xc_split_pack_last)
    float g_scale;       // fp16: 1 / (power-of-two scale of the last layer's weights)
    int halves;
    // conv3x3_wino: 0 / 1 = the 2x2 output blocks start at row -wino_py of this launch's region, so that they sit on EVEN rows of the
    // layer's whole output whatever row the band starts at -- a pixel's arithmetic then does not depend on the banding.
    int wino_py;
    // conv3x3_wino4 with out_terms = 9 and prog_cnt != NULL (PROG): the launch finishes the fused one-plane last layer itself, in row order -- g_out = that
    // layer's output rows (g_h x g_w, row stride g_out_rs floats), output row y reads the partial tap rows y + g_off .. y + g_off + 2 of `out`, g_bias its
    // bias; prog_cnt = w2xc_wino4_prog_counters() job counters (zeroed by the launcher on the stream).
    unsigned *prog_cnt;
    float *g_out;
    const float *g_bias;
    long long g_out_rs;
    int g_h, g_w, g_off;
    // ... with prog_flags != NULL (page-locked HOST memory, one word per job = (tile row, group of 8 tile columns), w2xc_wino4_prog_jobs()) every finished job
    // stores prog_epoch there at system scope behind its output rows: the host pipeline's drainer ships rows while the launch is still running
    unsigned *prog_flags;
    unsigned prog_epoch;
};

enum W2xcKernelKind {
    W2XC_K_DIRECT = 0,   // any shape, any strides, reference summation order (VALU)
    W2XC_K_MFMA = 1,     // cin, cout in {32,64,128}; NHWC in/out; fp32 MFMA implicit GEMM
    W2XC_K_FIRST = 2,    // cin <= 3 -> cout multiple of 32: planar in, NHWC out, fp32 MFMA (K = 9*cin)
    W2XC_K_LAST = 3,     // cin multiple of 32 -> cout <= 3: NHWC in, planar out, taps-as-N fp32 MFMA
    // W2XC_PRECISION_BF16X2 / BF16X3 (and BF16 through the same pipeline): fp32 values carried as d.terms bf16 terms
    W2XC_K_MID_SPLIT = 7,      // cin, cout in {32,64,128}: term planes in, term planes (or fp32 when out_terms = 0) out
    W2XC_K_FIRST_SPLIT = 8,    // W2XC_K_FIRST storing d.out_terms term planes
    W2XC_K_LAST_GATHER = 9,    // sums the partial G planes of a fused W2XC_K_MID_SPLIT (out_terms = 9) into the output plane
    W2XC_K_FIRST2_SPLIT = 10,  // layers 1 (1 -> 32) and 2 (32 -> {32,64,128}) in one kernel; launched in layer 2's slot
    W2XC_K_FUSED_AWAY = 11,    // layer 1 when W2XC_K_FIRST2_SPLIT / W2XC_K_FIRST2_WINO4 computes it: no launch
    W2XC_K_FIRST2_WINO4 = 12,  // fp32: layers 1 (1 -> 32) and 2 (32 -> 32, Winograd F(4x4,3x3)) in one kernel (w2xc_first2_wino4.hip); launched in layer 2's slot
};

// Which kernel kind the fast path has for a (cin, cout) layer; W2XC_K_DIRECT when none.
W2xcKernelKind w2xc_pick_kernel(int cin, int cout);
const char *w2xc_kernel_name(W2xcKernelKind kind, int cin, int cout);

// Size in floats of the packed weight image for `kind`, and the packer (host side).
// w is [cout][cin][3][3] (index o*cin+i, modelHandler.cpp:102).
size_t w2xc_packed_weight_floats(W2xcKernelKind kind, int cin, int cout);
void w2xc_pack_weights(W2xcKernelKind kind, int cin, int cout, const float *w, float *dst);

// Enqueue one layer on `stream`.  Returns hipSuccess or the launch error.
hipError_t w2xc_launch_conv(W2xcKernelKind kind, const W2xcConvDesc &d, hipStream_t stream);

// Winograd F(2x2, 3x3) on the fp32 MFMA (w2xc_wino.hip): the same layer as W2XC_K_MFMA (NHWC fp32 in / out) with 2.25x fewer MFMAs,
// for the shapes w2xc_wino_supported() names; d.wpk = the w2xc_wino_pack image (16 * cin * cout floats).
bool w2xc_wino_supported(int cin, int cout);
size_t w2xc_wino_packed_floats(int cin, int cout);
void w2xc_wino_pack(int cin, int cout, const float *w, float *dst);
hipError_t w2xc_launch_wino(const W2xcConvDesc &d, hipStream_t stream);
// Winograd F(4x4,3x3) on PLANAR activations (w2xc_wino4.hip): in_ps = 1 / in_cs = plane stride; out planar (out_ps = 1) or NHWC (out_cs = 1, out_ps = cout);
// d.wpk = the w2xc_wino4_pack image (36 * cin * cout floats), d.wino_py = first output row mod 4, off_x a non-negative multiple of 4
bool w2xc_wino4_supported(int cin, int cout);
void w2xc_wino4_pack(int cin, int cout, const float *w, float *dst);
hipError_t w2xc_launch_wino4(const W2xcConvDesc &d, hipStream_t stream);
// layers 1 + 2 of the fp32 path in one launch (w2xc_first2_wino4.hip): `in` / in_* / in_h / in_w / in_shift describe LAYER 1's one-plane input, off_y / off_x =
// layer 1's offsets + layer 2's, w1pk / bias1 = layer 1's W2XC_K_FIRST image and bias; wpk = the w2xc_first2_wino4_pack image (36 * 32 * 32 floats), bias,
// planar out, out_h / out_w / wino_py (first output row mod 4) = layer 2's region
bool w2xc_first2_wino4_supported(int cin1, int cout1, int cout2);
void w2xc_first2_wino4_pack(const float *w, float *dst);
hipError_t w2xc_launch_first2_wino4(const W2xcConvDesc &d, hipStream_t stream);
// d.out_terms = 9: the one-plane LAST layer in conv3x3_wino4's epilogue; d.w7pk = w2xc_wino4_pack_last image, `out` = partial tap planes
// G[64-plane block][tap][y][x] (out_ts / out_gs / out_rs), finished by W2XC_K_LAST_GATHER with halves = cout / 64
bool w2xc_wino4_prog_supported(int cin, int cout);
size_t w2xc_wino4_prog_counters(int out_w, int out_h, int wino_py);
// the job grid of such a launch: tile rows (16 rows each, the first one starting wino_py rows above the region) x groups of 8 tile columns (256 pixels)
void w2xc_wino4_prog_jobs(int out_w, int out_h, int wino_py, int *tile_rows, int *groups);
size_t w2xc_wino4_pack_last_floats(int cin);
void w2xc_wino4_pack_last(int cin, const float *w, float *dst);

// split kernels (w2xc_split.hip).  Packed weights of a mid layer: `terms` 16-bit terms of every weight in
// fragment order; W2XC_K_FIRST_SPLIT uses the W2XC_K_FIRST image.
int w2xc_split_kg(int terms, int cin);
size_t w2xc_split_packed_bytes(int cin, int cout, int terms);
float w2xc_split_pack(int cin, int cout, int terms, int fmt, const float *w, void *dst);   // returns the weight scale (1 for bf16)
hipError_t w2xc_launch_split_mid(const W2xcConvDesc &d, hipStream_t stream);
hipError_t w2xc_launch_split_first(const W2xcConvDesc &d, hipStream_t stream);
hipError_t w2xc_launch_first2_split(const W2xcConvDesc &d, hipStream_t stream);
// last layer fused into a two-term mid layer
int w2xc_split_halves(int terms, int cout);
size_t w2xc_split_pack_last_bytes(int cin, int terms);   // terms = 2 (one- and two-term modes) or 3
float w2xc_split_pack_last(int cin, int terms, int fmt, const float *w, void *dst);
hipError_t w2xc_launch_last_gather(const W2xcConvDesc &d, hipStream_t stream);

// strided element copy (planar <-> NHWC repack at the Model::filter boundary)
hipError_t w2xc_launch_repack(const float *src, long long s_rs, long long s_ps, long long s_cs,
                              float *dst, long long d_rs, long long d_ps, long long d_cs,
                              int h, int w, int c, hipStream_t stream);

// replicate-padded planar copy (Model::filter's BORDER_REPLICATE made explicit for conv3x3_wino4): dst is (h + 2 pad) x (w + 2 pad) per plane
hipError_t w2xc_launch_pad_planar(const float *src, long long s_rs, long long s_ps, long long s_cs, float *dst, long long d_rs, long long d_cs,
                                  int h, int w, int c, int pad, hipStream_t stream);

// N2 (w2xc_color.hip): colour front/back end and U/V bicubic of the CLI scale loop (main.cpp:74-76,144,171-172)
hipError_t w2xc_launch_u8_to_yuv(const unsigned char *src, size_t stride, int w, int h, float *y, float *u, float *v, hipStream_t st);
hipError_t w2xc_launch_yuv_to_u8(const float *y, const float *u, const float *v, int w, int h, unsigned char *dst, size_t stride, hipStream_t st);
hipError_t w2xc_launch_resize2x_cubic(const float *src, int w, int h, float *dst, hipStream_t st);
hipError_t w2xc_launch_resize_linear(const float *src, int sw, int sh, float *dst, int dw, int dh, hipStream_t st);
