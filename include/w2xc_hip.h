/*
 * w2xc_hip.h -- C ABI of libw2xc_hip.so, the MI355X (gfx950) engine for the one hot path of
 * WL-Amigo/waifu2x-converter-cpp v1:
 *
 *     w2xc::convertWithModels -> Model::filter -> Model::filterWorker
 *     (/root/reference/src/convertRoutine.cpp:21-169, src/modelHandler.cpp:26-72,117-159)
 *
 * Plain pointers and sizes only: this is what a cgo/JNI/ctypes/C++ binding of the reference's
 * API for this path binds to.  include/w2xc/modelHandler.hpp and convertRoutine.hpp re-expose the
 * reference's exact C++ signatures (w2xc::Model, w2xc::modelUtility, w2xc::convertWithModels) on
 * top of these entry points; INTEGRATION.md shows the swap.
 *
 * Every function returns W2XC_OK (0) or a negative W2XC_ERR_* code; w2xc_last_error() holds a
 * message for the calling thread.  There is NO CPU fallback: if no HIP device is usable the
 * compute calls fail with W2XC_ERR_HIP.
 */
#ifndef W2XC_HIP_H_
#define W2XC_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define W2XC_OK               0
#define W2XC_ERR_IO          -1   /* model file could not be opened (modelHandler.cpp:175-179)      */
#define W2XC_ERR_JSON        -2   /* JSON parse / schema error       (modelHandler.cpp:181-187)      */
#define W2XC_ERR_ARG         -3   /* bad argument (null pointer, non-positive size, bad stride ...) */
#define W2XC_ERR_PLANES      -4   /* number of input planes mismatch (modelHandler.cpp:29-35)        */
#define W2XC_ERR_HIP         -5   /* HIP runtime error / no device                                   */
#define W2XC_ERR_UNSUPPORTED -6   /* kernel size != 3 (hpp:52-58 only demands square; see Q8)        */
#define W2XC_ERR_NOMEM       -7

/* One loaded model file == the reference's std::vector<std::unique_ptr<w2xc::Model>>
 * (one w2xc::Model per conv layer, modelHandler.hpp:24-90). */
typedef struct w2xc_model w2xc_model;

#define W2XC_PRECISION_FP32 0   /* fp32 throughout on the fp32 MFMA (v_mfma_f32_16x16x4_f32 / 32x32x2): Winograd F(4x4,3x3) for the layers
                                 * with 64 / 128 output planes (conv3x3_wino4), F(2x2,3x3) for 32 output planes (conv3x3_wino); per call
                                 * w2xc_opts.kernel = W2XC_KERNEL_MFMA selects the exact-f32 fma chains of conv3x3_mfma2 instead.
                                 * rtol 1e-4 vs the reference either way                                                       */
#define W2XC_PRECISION_BF16 1   /* w2xc_convert_* only: activations BETWEEN layers are bf16 (RNE), layers
                                 * 2..n-1 use bf16 weights on v_mfma_f32_32x32x16_bf16 with fp32
                                 * accumulate, bias and LeakyReLU; the first layer stays fp32, a one-plane
                                 * last layer is fused into layer n-1's epilogue as one more bf16 layer.
                                 * Not the reference's arithmetic: tolerance in DESIGN.md 4.          */
#define W2XC_PRECISION_BF16X2 2 /* w2xc_convert_* only: split products.  Every fp32 activation / weight of layers
                                 * 2..n-1 is carried as the sum of 2 bf16 terms (hi + lo, ~16 mantissa bits)
                                 * and each product is 3 bf16 MFMA products accumulated in fp32.          */
#define W2XC_PRECISION_BF16X3 3 /* as above with 3 terms (~24 mantissa bits) and 6 products: the error level
                                 * of an fp32 FMA chain at 2.7x the fp32 MFMA rate of CDNA4.  First / last
                                 * layer arithmetic stays fp32 MFMA in both.  Tolerances in DESIGN.md 4.   */
#define W2XC_PRECISION_FP16X2 4 /* split into 2 fp16 terms (~22 mantissa bits), 3 products on the fp16 MFMA: the
                                 * speed of BF16X2 at nearly the accuracy of BF16X3.  fp16 has 5 exponent bits:
                                 * weights are pre-scaled per layer by a power of two (exact), activations of
                                 * layers 1..n-2 saturate at +-65504 and carry 2^-25 ABSOLUTE precision below
                                 * 2^-3 -- meant for image planes in [0, 1].                   */

#define W2XC_KERNEL_AUTO    0   /* the fast kernel of each layer shape; for the fp32 layers with 32 / 64 / 128 planes in and out that is
                                 * W2XC_KERNEL_WINOGRAD4.  On a row-band view WITHOUT the wide halo below (w2xc_convert_rows_device /
                                 * w2xc_convert_plane_rows) it is REFUSED with W2XC_ERR_ARG: the kernel (and with it the rounding)
                                 * never changes silently with the view; such a caller passes the wide view or names a kernel.
                                 * No environment switches: the choice is the caller's, per call.                            */
#define W2XC_KERNEL_DIRECT  1   /* every layer: reference-ordered direct conv on VALU (bit-exact vs the oracle)             */
#define W2XC_KERNEL_MFMA    2   /* mid layers: direct implicit GEMM, a k-ordered fp32 fma chain on v_mfma_f32_32x32x2_f32
                                 * (conv3x3_mfma2) -- the closest MFMA analogue of modelHandler.cpp:134-145                */
#define W2XC_KERNEL_WINOGRAD 3  /* = W2XC_KERNEL_WINOGRAD32 (the value is kept for callers compiled against rounds 3 / 4, whose own
                                 * F(2x2) kernel on 16x16x4 tiles, conv3x3_wino16, was retired in round 5)                  */
#define W2XC_KERNEL_WINOGRAD32 4 /* mid layers: Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32 (conv3x3_wino), fp32 throughout;
                                 * banding-invariant on minimum-halo row views; the last layer is its own launch             */
#define W2XC_KERNEL_WINOGRAD4 5 /* mid layers with >= 64 output planes: Winograd F(4x4,3x3) (conv3x3_wino4) on planar activations, fp32
                                 * throughout: 2.25 multiplies per output instead of 4, interpolation points 0, +-3/4, +-3/2 (error against the
                                 * fp64 truth 1.6-1.9x the CPU oracle's own, 0.2-0.3 of the rtol 1e-4 gate on whole frames:
                                 * tests/test_gpu_configs.py).  What W2XC_KERNEL_AUTO picks; asked for explicitly it also runs on row-band
                                 * views with the minimum halo, where its results depend on the banding at rounding level          */

typedef struct w2xc_opts {
    int      struct_size;     /* sizeof(w2xc_opts): ABI versioning                                   */
    int      precision;       /* W2XC_PRECISION_*  (opts == NULL: the process defaults, w2xc_set_default_opts)  */
    int      kernel;          /* W2XC_KERNEL_*                                                       */
    int      device;          /* device-pointer entry points: HIP device ordinal, -1 = current      */
    unsigned device_mask;     /* host-pointer entry points: bit i = use device i; 0 = all devices   */
    int      band_rows;       /* output rows per band (tile height); 0 = derive from workspace_mb   */
    int      workspace_mb;    /* activation workspace budget per device in MiB; 0 = default (16384) */
    int      profile;         /* 1 = bracket every layer launch with hipEvents (see below)          */
    int      verbose;         /* 1 = print the reference's progress lines (convertRoutine.cpp:67)   */
    int      filter_resident; /* w2xc_layer_filter: 1 = when the input planes are exactly the planes the previous
                               * w2xc_layer_filter call on this model wrote (same pointers, count, size) the caller
                               * promises they are unmodified, and the copy still on the device is used instead of
                               * uploading them again (chained Model::filter, test.cpp:72-85).  opts == NULL: the
                               * process defaults (w2xc_set_default_opts).  Default 0: every call uploads what it is given. */
    int      fusion;          /* W2XC_FUSION_*: cross-layer fusion on the fp32 path -- convertRoutine.cpp:66-76's loop collapsed by two launches:
                               * (a) the one-plane last layer inside the epilogue of the layer before it (the default kernel conv3x3_wino4
                               *     carries it; conv3x3_last_gather finishes), (b) layers 1 (1 -> 32) and 2 (32 -> 32) in one launch,
                               *     conv3x3_first2_wino4: layer 1's activations never reach HBM.  W2XC_FUSION_AUTO = both on where the
                               *     model's shapes allow (a fusion whose kernel cannot address the plane is given up, never an error),
                               *     W2XC_FUSION_OFF runs every layer as its own launch, W2XC_FUSION_FIRST / _LAST allow only (b) / only (a),
                               *     W2XC_FUSION_ON = both, and W2XC_ERR_UNSUPPORTED where AUTO would give one up.  Results stay inside the
                               *     fp32 gate either way (fused vs unfused <= 4e-6 of the output range, tests/test_gpu_winograd.py).
                               * The 16-bit modes have the same two fusions (conv3x3_first2_split, conv3x3_split + gather) under the same switch. */
    int      host_units;      /* host-pointer entry points, test aid: cut the rows into this many units, round-robin over the selected
                               * devices, so a one-GPU box runs the multi-device arithmetic; 0 = one unit per device              */
    int      host_chunk_kb;   /* host-pointer entry points, test aid: maximum size of a staged output chunk in KiB; 0 = 8192     */
    int      host_numa;       /* host-pointer entry points: 0 = a unit's feeder / drainer threads and pinned rings are placed on the CPU
                               * node its device hangs off (multi-socket hosts), 1 = the caller's affinity is left alone            */
} w2xc_opts;                  /* (verbose: bit 0 = the reference's progress lines, bit 1 = the host pipeline's phase timestamps on stderr) */

#define W2XC_FUSION_AUTO  0
#define W2XC_FUSION_OFF   1
#define W2XC_FUSION_ON    2
#define W2XC_FUSION_FIRST 3   /* only layers 1 + 2 in one launch */
#define W2XC_FUSION_LAST  4   /* only the last layer inside the epilogue of the layer before it */
/* Round 6: where the fused last layer is FINISHED.  The host-pointer entry points let the launch of layer n - 1 finish it itself, in row order
 * (conv3x3_wino4 PROG: its gather jobs write the output rows straight into page-locked host memory and flag them, so rows leave for the caller's plane
 * while the launch is still running); the device-pointer entry points follow it with a conv3x3_last_gather launch (0.2 ms faster when nothing waits
 * for rows).  Both are the same sum in the same order: BIT-identical.  For A/B runs and tests: */
#define W2XC_FUSION_GATHER_LAUNCH 5   /* as AUTO, but the gather launch in every entry point (the form of rounds 4 / 5) */
#define W2XC_FUSION_PROG          6   /* as AUTO, but finished inside the producing launch in every entry point        */

/* Fill *o with defaults (fp32, auto kernels, current device, all devices, auto banding).  Writes sizeof(w2xc_opts) bytes of THIS header's
 * struct: a binary compiled against an older, shorter w2xc_opts must call w2xc_opts_init_sized with ITS sizeof (or be rebuilt) --
 * the ABI version string (w2xc_version) changes whenever the struct grows. */
void w2xc_opts_init(w2xc_opts *o);
/* The same for a caller that knows only the first `struct_size` bytes of w2xc_opts (an older header): nothing past them is written, and
 * o->struct_size is set to it, which is what every entry point honours when it reads the options back. */
void w2xc_opts_init_sized(w2xc_opts *o, size_t struct_size);
/* What `opts == NULL` means for this process (the C++ adapter behind the reference's unmodified callers passes no options).  Initially:
 * w2xc_opts_init's defaults with precision from the environment variable W2XC_PRECISION (fp32 | fp16x2 | bf16x3 | bf16x2 | bf16) and
 * filter_resident from W2XC_FILTER_RESIDENT (0 | 1) -- the only two environment variables the library reads, once, here.  Passing a
 * struct replaces the defaults; passing NULL re-reads the environment.  Thread-safe; calls already running keep what they resolved. */
int w2xc_set_default_opts(const w2xc_opts *defaults);

/* ---- model container (modelHandler.hpp:24-90, modelHandler.cpp:74-115,170-197) ------------- */

/* == modelUtility::generateModelFromJSON (modelHandler.cpp:170-197).  The file is a JSON array of
 * {kW,kH,nInputPlane,nOutputPlane,bias[nOut],weight[nOut][nIn][kH][kW]} objects
 * (appendix/waifu2x-nocuda/export_model_nocuda.lua:12-19).  Numbers are parsed with strtod and
 * weights narrowed double->float exactly like modelHandler.cpp:95-97; biases stay double. */
int w2xc_model_load_json(const char *path, w2xc_model **out);

/* Same container from arrays: weight[l] is [nout][nin][3][3] floats (index o*nin+i, :102),
 * bias[l] is nout doubles.  Data is copied. */
int w2xc_model_from_arrays(int n_layers, const int *nin, const int *nout,
                           const float *const *weight, const double *const *bias, w2xc_model **out);

void w2xc_model_free(w2xc_model *m);
/* Release what grew with the largest plane converted so far -- activation workspaces, the host pipeline's device rows and
 * pinned rings, Model::filter's buffers -- on every device the model has run on.  Weights, streams and events stay; the
 * next call re-grows what it needs.  For a long-lived process after an unusually large image (a 16384^2 plane leaves
 * 17 GiB of workspace and 1.3 GiB of plane copies behind).  Not to be called concurrently with a conversion on `m`. */
int  w2xc_model_trim(w2xc_model *m);
int  w2xc_model_layers(const w2xc_model *m);             /* models.size()                          */
int  w2xc_model_nin(const w2xc_model *m, int layer);     /* Model::getNInputPlanes  (:18-20)        */
int  w2xc_model_nout(const w2xc_model *m, int layer);    /* Model::getNOutputPlanes (:22-24)        */
/* copy out one layer's weights ([nout][nin][3][3]) and biases; either pointer may be NULL.
 * Backs Model::printWeightMatrix / printBiases (:229-242). */
int  w2xc_model_get_layer(const w2xc_model *m, int layer, float *weight, double *bias);

/* ---- modelUtility singleton knobs (modelHandler.hpp:92-113, .cpp:199-224) ------------------- */
int  w2xc_set_jobs(int n);                    /* setNumberOfJobs: rejects n < 1 (:199-203)          */
int  w2xc_get_jobs(void);                     /* default 4 (hpp:99)                                 */
int  w2xc_set_block_size(int w, int h);       /* setBlockSize: rejects negatives (:209-213)         */
int  w2xc_set_block_size_exp2(int exp);       /* setBlockSizeExp2Square (:215-220)                  */
void w2xc_get_block_size(int *w, int *h);     /* default 512x512 (hpp:99)                           */

/* ---- the hot path --------------------------------------------------------------------------- */

/* == w2xc::convertWithModels(inputPlane, outputPlane, models, blockSplitting)
 * (convertRoutine.cpp:21-51).  `in`/`out` are HOST pointers to h rows of w floats, strides in
 * bytes (a cv::Mat ROI's step).  out(y,x) = valid-conv CNN(replicate_pad(in, n_layers))(y,x):
 * identical math per output pixel to both the unsplit path (:32-46) and the 512/498 block walk
 * (:84-169), so `block_splitting` and the singleton block size do not change results and the
 * engine bands the plane by its own workspace budget.  No clipping (Q2).  Uses every device in
 * opts->device_mask: row bands are striped over devices, one host thread per device, no
 * inter-device traffic.  opts may be NULL.
 * Host pipeline (per model and device, created once and kept): three HIP streams (H2D, layers, D2H), device copies of
 * this device's rows, and rings of pinned staging slots.  The caller's pageable planes are copied into / out of the
 * slots by modelUtility's nJob host threads (w2xc_set_jobs) while the DMA engines and the kernels run: the upload of
 * band k+1 and the download + stitch of the previous rows overlap the layers of band k; the last layer is launched in
 * ~2 MiB row chunks so its rows leave while the rest is still computed.  Planes that are already page-locked
 * (hipHostMalloc / hipHostRegister) are DMA'd in place.  The call returns when `out` is complete. */
int w2xc_convert_plane(w2xc_model *m, const float *in, size_t in_stride_bytes, int w, int h,
                       float *out, size_t out_stride_bytes, int block_splitting,
                       const w2xc_opts *opts);

/* Same contract with DEVICE pointers on device opts->device, enqueued on `hip_stream`
 * (a hipStream_t; NULL = the null stream) and NOT synchronised on return.  Input and output
 * stay resident in HBM: this is what bench.py times.  The enqueued kernels use the model's per-device
 * activation workspace, so asynchronous calls on the same (model, device) must share one stream (or be
 * serialised by the caller); different models or devices are independent. */
int w2xc_convert_plane_device(w2xc_model *m, const float *d_in, size_t in_stride_bytes, int w, int h,
                              float *d_out, size_t out_stride_bytes, void *hip_stream,
                              const w2xc_opts *opts);

/* Row-band form for sharding ONE plane over several processes / GPUs (the reference's block walk,
 * convertRoutine.cpp:114-165, made parallel): computes output rows [row_begin, row_end) of the
 * plane_h x w conversion.  `d_view` holds plane rows [view_y0, view_y0 + view_h) and must cover
 * [row_begin - n_layers, row_end + n_layers) clipped to the plane; `d_out` points at output row
 * row_begin.  Device pointers, asynchronous on `hip_stream`, no exchange between bands.
 * Bands stitch BIT-identically with the whole-plane call when the view holds the WIDE halo,
 * [row_begin - 4 n_layers, row_end + 4 n_layers) clipped: the default F(4x4) mid-layer kernel works on 4x4 blocks
 * and needs every band region to end on a block row, four rows of halo per layer.  On a view with only the minimum
 * halo W2XC_KERNEL_AUTO is refused (W2XC_ERR_ARG) -- the rounding never changes silently with the view; a named kernel
 * (W2XC_KERNEL_WINOGRAD32 / _WINOGRAD / _MFMA / _DIRECT: banding-invariant there) runs on it.                        */
int w2xc_convert_rows_device(w2xc_model *m, const float *d_view, size_t view_stride_bytes, int view_h,
                             int view_y0, int w, int plane_h, int row_begin, int row_end, float *d_out,
                             size_t out_stride_bytes, void *hip_stream, const w2xc_opts *opts);

/* Multi-plane generalisation (BASELINE.json configs[4]: a 3 -> 128 -> ... -> 3 model).  The reference
 * can only reach such a model by chaining Model::filter by hand (convertWithModels pushes one plane,
 * convertRoutine.cpp:63-64, and returns only outputPlanes[0], :78); this runs the same wrapper --
 * replicate pad by the layer count, all layers, crop -- on n_in_planes planar DEVICE planes and writes
 * ALL planes of the last layer, planar.  With one input plane and a one-plane last layer it equals
 * w2xc_convert_plane_device.  fp32 only. */
int w2xc_convert_planes_device(w2xc_model *m, int n_in_planes, const float *d_in, size_t in_plane_stride_bytes,
                               size_t in_stride_bytes, int w, int h, float *d_out, size_t out_plane_stride_bytes,
                               size_t out_stride_bytes, void *hip_stream, const w2xc_opts *opts);

/* N1: the scale loop of the CLI -- cv::resize(INTER_NEAREST, 2x) of the luma plane
 * (main.cpp:132-140) followed by convertWithModels (:148) -- as ONE call.  `in` is the h x w plane
 * BEFORE the resize, `out` is 2h x 2w.  The nearest-neighbour upscale is folded into layer 1's load
 * (source pixel (y>>1, x>>1)), so the 4x larger plane is never materialised nor copied over PCIe.
 * Result == w2xc_convert_plane on the explicitly upscaled plane. */
int w2xc_convert_plane_nn2x(w2xc_model *m, const float *in, size_t in_stride_bytes, int w, int h,
                            float *out, size_t out_stride_bytes, const w2xc_opts *opts);
int w2xc_convert_plane_nn2x_device(w2xc_model *m, const float *d_in, size_t in_stride_bytes, int w, int h,
                                   float *d_out, size_t out_stride_bytes, void *hip_stream,
                                   const w2xc_opts *opts);

/* One UNIT of the tile farm from host memory (the reference's block walk, convertRoutine.cpp:114-165, made parallel across
 * processes): output rows [row_begin, row_end) of the conversion of a w x h source plane (nn2x = 1: of its nearest-
 * neighbour 2x, main.cpp:132-140, so the output plane is 2w x 2h and row numbers are in OUTPUT coordinates).
 * `in_view` points at source row view_y0 and holds view_h rows, which must cover every source row the range reads
 * (rows [row_begin - n_layers, row_end + n_layers) of the plane, halved for nn2x, clipped -- 4 n_layers instead of n_layers
 * for units that stitch bit-identically with the whole-plane call, see w2xc_convert_rows_device); `out` points at output row
 * row_begin.  Same pinned-staged, overlapped pipeline and device_mask semantics as w2xc_convert_plane; units never
 * exchange data, so N processes each calling this with their own row range ARE the multi-GPU farm (host-side gather only). */
int w2xc_convert_plane_rows(w2xc_model *m, const float *in_view, size_t in_stride_bytes, int view_y0, int view_h, int w, int h,
                            int nn2x, int row_begin, int row_end, float *out, size_t out_stride_bytes, const w2xc_opts *opts);

/* N2: the whole SCALE PHASE of the CLI for one image, device-resident:
 *   convertTo(CV_32F, 1/255) + cvtColor(COLOR_RGB2YUV) on the three channels as given  (main.cpp:75-76)
 *   `iterations` times: Y <- convertWithModels(resize(Y, 2x, INTER_NEAREST)),
 *                       U, V <- resize(2x, INTER_CUBIC)                                  (main.cpp:126-156)
 *   cvtColor(COLOR_YUV2RGB) + convertTo(CV_8U, 255)  (the only clip, Q2)                 (main.cpp:171-172)
 * `in` is h rows of w interleaved 3-channel uint8 pixels, `out` is (h << iterations) rows of (w << iterations).
 * The image stays float YUV between iterations, exactly like the reference.  The *_device form takes device
 * pointers and is asynchronous on `hip_stream`; the host form uses opts->device (default: current). */
int w2xc_scale2x_image_u8_device(w2xc_model *scale_model, const unsigned char *d_in, size_t in_stride_bytes, int w, int h,
                                 unsigned char *d_out, size_t out_stride_bytes, int iterations, void *hip_stream,
                                 const w2xc_opts *opts);
int w2xc_scale2x_image_u8(w2xc_model *scale_model, const unsigned char *in, size_t in_stride_bytes, int w, int h,
                          unsigned char *out, size_t out_stride_bytes, int iterations, const w2xc_opts *opts);
/* All three processing modes of the CLI (main.cpp:46-48, -m noise | scale | noise_scale) on one uint8 image:
 * an optional noise model is applied to Y first (main.cpp:83-98), then `iterations` 2x steps with the scale
 * model.  noise_model or scale_model may be NULL (iterations must be 0 without a scale model). */
int w2xc_process_image_u8_device(w2xc_model *noise_model, w2xc_model *scale_model, const unsigned char *d_in,
                                 size_t in_stride_bytes, int w, int h, unsigned char *d_out, size_t out_stride_bytes,
                                 int iterations, void *hip_stream, const w2xc_opts *opts);
int w2xc_process_image_u8(w2xc_model *noise_model, w2xc_model *scale_model, const unsigned char *in, size_t in_stride_bytes,
                          int w, int h, unsigned char *out, size_t out_stride_bytes, int iterations, const w2xc_opts *opts);
/* ... plus the final INTER_LINEAR shrink the CLI applies for scale ratios that are not powers of two
 * (main.cpp:107-114,158-167): shrink_ratio in (0,1) resizes the float YUV image to
 * int((w << iterations) * shrink_ratio) x int((h << iterations) * shrink_ratio) before the conversion back to
 * uint8; 0 = no shrink. */
int w2xc_process_image_u8_ex_device(w2xc_model *noise_model, w2xc_model *scale_model, const unsigned char *d_in,
                                    size_t in_stride_bytes, int w, int h, unsigned char *d_out, size_t out_stride_bytes,
                                    int iterations, double shrink_ratio, void *hip_stream, const w2xc_opts *opts);
int w2xc_process_image_u8_ex(w2xc_model *noise_model, w2xc_model *scale_model, const unsigned char *in, size_t in_stride_bytes,
                             int w, int h, unsigned char *out, size_t out_stride_bytes, int iterations, double shrink_ratio,
                             const w2xc_opts *opts);
/* the building blocks on contiguous float planes (device pointers): main.cpp:144 on one plane, :75-76, :171-172 */
int w2xc_resize2x_cubic_device(const float *d_src, int w, int h, float *d_dst, void *hip_stream);
int w2xc_u8_to_yuv_device(const unsigned char *d_in, size_t in_stride_bytes, int w, int h, float *d_y, float *d_u,
                          float *d_v, void *hip_stream);
int w2xc_yuv_to_u8_device(const float *d_y, const float *d_u, const float *d_v, int w, int h, unsigned char *d_out,
                          size_t out_stride_bytes, void *hip_stream);

/* == Model::filter(inputPlanes, outputPlanes) for layer `layer` (modelHandler.cpp:26-72):
 * n_in_planes host planes of h x w floats in, nout planes out, SAME size, per-layer
 * BORDER_REPLICATE (:141-142), bias, LeakyReLU(0.1) (:147-152).  Returns W2XC_ERR_PLANES when
 * n_in_planes != nInputPlanes (the reference returns false, :29-35). */
int w2xc_layer_filter(w2xc_model *m, int layer, int n_in_planes, const float *const *in_planes,
                      size_t in_stride_bytes, int w, int h, float *const *out_planes,
                      size_t out_stride_bytes, const w2xc_opts *opts);

/* The same layer on DEVICE data with arbitrary element strides (in floats): element (plane c, row y, pixel x) is at
 * base[c*plane_stride + y*row_stride + x*pixel_stride] -- planar planes (pixel_stride 1) as Model::filter has them, or
 * NHWC (plane_stride 1, pixel_stride = plane count, 16-byte aligned), which the MFMA kernels use directly so a chain
 * of calls that hands NHWC from layer to layer never repacks.  Asynchronous on `hip_stream`. */
int w2xc_layer_filter_device(w2xc_model *m, int layer, int n_in_planes, const float *d_in, long long in_plane_stride,
                             long long in_row_stride, long long in_pixel_stride, int w, int h, float *d_out,
                             long long out_plane_stride, long long out_row_stride, long long out_pixel_stride,
                             void *hip_stream, const w2xc_opts *opts);

/* ---- measurement / introspection ------------------------------------------------------------ */

/* With opts->profile = 1 every layer launch of the device entry point is bracketed by hipEvents
 * on the launch stream.  After the stream has been synchronised, this returns for each layer the
 * SUM of its launch durations (ms) and the number of launches since the last reset. */
int  w2xc_profile_read(w2xc_model *m, int device, float *layer_ms, int *layer_launches, int n_layers);
void w2xc_profile_reset(w2xc_model *m, int device);
/* name of the kernel the engine picks for a layer (for matching rocprofv3 kernel traces) */
const char *w2xc_layer_kernel_name(const w2xc_model *m, int layer, const w2xc_opts *opts);

/* The band geometry a w2xc_convert_rows_device call with these arguments would run with, computed on the host (no device needed, nothing is
 * allocated or launched): halo rows per layer (1, or 4 = the banding-invariant geometry of the default F(4x4) kernel), output rows per band,
 * number of bands, bytes of the two activation workspaces, and whether the two cross-layer fusions are in effect.  Fails exactly where the
 * conversion would refuse its arguments (W2XC_ERR_ARG for W2XC_KERNEL_AUTO on a minimum-halo view, W2XC_ERR_PLANES ...). */
typedef struct w2xc_row_plan {
    int struct_size;
    int n_layers;
    int halo_rows_per_layer;
    int band_rows;
    int n_bands;
    int fused_first, fused_last;
    unsigned long long workspace_bytes[2];
} w2xc_row_plan;
int w2xc_plan_rows(const w2xc_model *m, int w, int view_y0, int view_h, int plane_h, int row_begin, int row_end, const w2xc_opts *opts,
                   w2xc_row_plan *plan);
/* plane rows [*top, *bottom) that layer `layer` (1 .. n_layers) computes for the band of output rows [y0, y1) under `plan`
 * (negative rows / rows >= plane_h + ...: the replicate padding of convertRoutine.cpp:35 seen from that layer) */
int w2xc_plan_region(const w2xc_row_plan *plan, int plane_h, int layer, int y0, int y1, int *top, int *bottom);

int w2xc_device_count(void);          /* hipGetDeviceCount, 0 when no device / no driver           */
const char *w2xc_last_error(void);    /* thread-local message of the last failing call             */
const char *w2xc_version(void);

#ifdef __cplusplus
}
#endif
#endif /* W2XC_HIP_H_ */
