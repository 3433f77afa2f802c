/*
 * w2xc_oracle.c -- CPU restatement of the waifu2x-converter-cpp (v1) hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product path may call, link
 * or import this file; it is the checker for tests/, __graft_entry__.smoke() and
 * the cpu_baseline leg of bench.py.
 *
 * PARITY PINNING: the reference's arithmetic lives in OpenCV 3.0.0-rc1
 * (README.md:33), which is un-vendored and absent here, and the reference ships
 * no golden vectors (src/test.cpp asserts nothing).  The OpenCV primitives
 * (filter2D / add / max / min / scaleAdd / copyMakeBorder) are therefore
 * "parity unpinned" -- they are restated from OpenCV's documented CPU semantics.
 * What IS pinned: oracle/_ref builds the reference's own modelHandler.cpp and
 * convertRoutine.cpp against a small OpenCV shim (oracle/cvshim), and
 * tests/test_oracle.py demands bit-equality between this restatement and
 * that build (thread partition, weight indexing, JSON loading, pad, block walk,
 * crop, stitch are the reference's own code there).
 *
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference).  Compile with -ffp-contract=off: OpenCV's SSE path is
 * mul-then-add, not fma.
 */
#include <math.h>
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "w2xc_oracle.h"

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------------------- */
/* cv::filter2D(src, dst, -1, K3x3, Point(-1,-1), 0.0, BORDER_REPLICATE)     */
/* modelHandler.cpp:141-142.  Correlation (no kernel flip), anchor = centre,  */
/* same-size output, clamp-to-edge border.  fp32 accumulator starts at        */
/* delta = 0 and taps are added in row-major order, mul then add.             */
/* ------------------------------------------------------------------------- */
static void filter2d_3x3_f32(const float *src, int w, int h, const float *k9, float *dst)
{
    for (int y = 0; y < h; y++) {
        const float *r0 = src + (size_t)clampi(y - 1, 0, h - 1) * w;
        const float *r1 = src + (size_t)y * w;
        const float *r2 = src + (size_t)clampi(y + 1, 0, h - 1) * w;
        float *d = dst + (size_t)y * w;
        /* left edge, interior (vectorisable), right edge */
        for (int x = 0; x < w; x++) {
            if (x == 1 && w > 2) {
                for (; x < w - 1; x++) {
                    float t = 0.0f;
                    t = t + k9[0] * r0[x - 1];
                    t = t + k9[1] * r0[x];
                    t = t + k9[2] * r0[x + 1];
                    t = t + k9[3] * r1[x - 1];
                    t = t + k9[4] * r1[x];
                    t = t + k9[5] * r1[x + 1];
                    t = t + k9[6] * r2[x - 1];
                    t = t + k9[7] * r2[x];
                    t = t + k9[8] * r2[x + 1];
                    d[x] = t;
                }
            }
            int xm = clampi(x - 1, 0, w - 1), xp = clampi(x + 1, 0, w - 1);
            float t = 0.0f;
            t = t + k9[0] * r0[xm];
            t = t + k9[1] * r0[x];
            t = t + k9[2] * r0[xp];
            t = t + k9[3] * r1[xm];
            t = t + k9[4] * r1[x];
            t = t + k9[5] * r1[xp];
            t = t + k9[6] * r2[xm];
            t = t + k9[7] * r2[x];
            t = t + k9[8] * r2[xp];
            d[x] = t;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Model::filterWorker  -- modelHandler.cpp:117-159                           */
/*   for o in [begin, begin+n): acc = 0; for i: acc += filter2D(in[i],W[o,i]) */
/*   acc += (float)bias[o]; acc = 0.1f*min(acc,0) + max(acc,0)                */
/* ------------------------------------------------------------------------- */
typedef struct {
    const w2xc_oracle_layer *L;
    const float *in;   /* [nin][h][w] */
    float *out;        /* [nout][h][w] */
    int w, h;
    unsigned begin, n;
} worker_arg;

static void *filter_worker(void *p)
{
    worker_arg *a = (worker_arg *)p;
    const w2xc_oracle_layer *L = a->L;
    const size_t plane = (size_t)a->w * a->h;
    float *tmp = (float *)malloc(plane * sizeof(float));
    for (unsigned o = a->begin; o < a->begin + a->n; o++) {
        float *acc = a->out + (size_t)o * plane;
        memset(acc, 0, plane * sizeof(float));                       /* :131-132 */
        for (int i = 0; i < L->nin; i++) {                           /* :134 */
            const float *k9 = L->weight + ((size_t)o * L->nin + i) * 9; /* index o*nIn+i, :130,137 */
            filter2d_3x3_f32(a->in + (size_t)i * plane, a->w, a->h, k9, tmp); /* :141-142 */
            for (size_t q = 0; q < plane; q++) acc[q] = acc[q] + tmp[q];   /* cv::add :144 */
        }
        const float b = (float)L->bias[o];   /* cv::add(UMat, double): scalar cast to the array depth, :147 */
        for (size_t q = 0; q < plane; q++) {
            float v = acc[q] + b;
            float pos = v > 0.0f ? v : 0.0f;   /* cv::max(v, 0.0)  :150 */
            float neg = v < 0.0f ? v : 0.0f;   /* cv::min(v, 0.0)  :151 */
            acc[q] = neg * 0.1f + pos;         /* cv::scaleAdd(neg, 0.1, pos) :152 (alpha cast to float) */
        }
    }
    free(tmp);
    return NULL;
}

/* Model::filter -- modelHandler.cpp:26-72: plane-count check, nJob std::threads
 * over contiguous output-plane ranges of floor(nOut/nJob); the last thread takes
 * the remainder when nJob does not divide nOut (:46-65). */
int w2xc_oracle_filter(const w2xc_oracle_layer *L, int n_in_planes, const float *in, int w, int h,
                       float *out, int njob)
{
    if (n_in_planes != L->nin) return -1;                 /* :29-35 returns false */
    if (njob < 1) njob = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * njob);
    worker_arg *args = (worker_arg *)malloc(sizeof(worker_arg) * njob);
    int per = L->nout / njob;                             /* :46 */
    for (int idx = 0; idx < njob; idx++) {
        worker_arg *a = &args[idx];
        a->L = L; a->in = in; a->out = out; a->w = w; a->h = h;
        a->begin = (unsigned)(per * idx);
        if (!(idx == njob - 1 && per * njob != L->nout)) a->n = (unsigned)per;       /* :48-55 */
        else a->n = (unsigned)(L->nout - per * idx);                                   /* :56-64 */
        pthread_create(&th[idx], NULL, filter_worker, a);
    }
    for (int idx = 0; idx < njob; idx++) pthread_join(th[idx], NULL);   /* :67-69 */
    free(th); free(args);
    return 0;
}

/* convertWithModelsBasic -- convertRoutine.cpp:53-82: layer loop, ping-pong
 * plane vectors, result = outputPlanes[0].  `plane` is w x h contiguous. */
static int convert_basic(const w2xc_oracle_layer *layers, int nlayers, const float *plane, int w, int h,
                         float *out, int njob)
{
    const size_t px = (size_t)w * h;
    int maxc = 1;
    for (int l = 0; l < nlayers; l++) {
        if (layers[l].nin > maxc) maxc = layers[l].nin;
        if (layers[l].nout > maxc) maxc = layers[l].nout;
    }
    float *a = (float *)malloc(px * maxc * sizeof(float));
    float *b = (float *)malloc(px * maxc * sizeof(float));
    if (!a || !b) { free(a); free(b); return -2; }
    memcpy(a, plane, px * sizeof(float));
    int nplanes = 1;                                               /* :63-64 */
    for (int l = 0; l < nlayers; l++) {                            /* :66 */
        if (w2xc_oracle_filter(&layers[l], nplanes, a, w, h, b, njob) != 0) { /* :68-70 (exit(-1)) */
            free(a); free(b); return -1;
        }
        nplanes = layers[l].nout;
        float *t = a; a = b; b = t;                                /* :71-75 */
    }
    memcpy(out, a, px * sizeof(float));                            /* outputPlanes[0], :78 */
    free(a); free(b);
    return 0;
}

/* cv::copyMakeBorder(src, dst, p,p,p,p, BORDER_REPLICATE) -- convertRoutine.cpp:35,96 */
static void pad_replicate(const float *src, size_t src_stride, int w, int h, int p, float *dst)
{
    const int W = w + 2 * p, H = h + 2 * p;
    for (int y = 0; y < H; y++) {
        const float *s = src + (size_t)clampi(y - p, 0, h - 1) * src_stride;
        float *d = dst + (size_t)y * W;
        for (int x = 0; x < W; x++) d[x] = s[clampi(x - p, 0, w - 1)];
    }
}

/* convertWithModels -- convertRoutine.cpp:21-51 (+ BlockSplit :84-169). */
int w2xc_oracle_convert(const w2xc_oracle_layer *layers, int nlayers, const float *in,
                        size_t in_stride, int w, int h, float *out, size_t out_stride,
                        int block_splitting, int block_w, int block_h, int njob)
{
    const int nModel = nlayers;                                    /* :33, :91 */
    const int W = w + 2 * nModel, H = h + 2 * nModel;
    float *padded = (float *)malloc((size_t)W * H * sizeof(float));
    if (!padded) return -2;
    pad_replicate(in, in_stride, w, h, nModel, padded);            /* :35 / :94-97 */
    int rc = 0;
    const int require_split = (w * h) > block_w * block_h * 3 / 2; /* :25-26 (int math) */
    if (!(block_splitting && require_split)) {
        float *full = (float *)malloc((size_t)W * H * sizeof(float));
        rc = convert_basic(layers, nlayers, padded, W, H, full, njob);         /* :38 */
        if (rc == 0)
            for (int y = 0; y < h; y++)                                        /* crop :40-46 */
                memcpy(out + (size_t)y * out_stride, full + (size_t)(y + nModel) * W + nModel,
                       (size_t)w * sizeof(float));
        free(full);
    } else {
        const int sw = block_w - 2 * nModel, sh = block_h - 2 * nModel;
        const unsigned splitColumns = (unsigned)ceilf((float)w / (float)sw);   /* :100-102 */
        const unsigned splitRows = (unsigned)ceilf((float)h / (float)sh);      /* :103-105 */
        for (int y = 0; y < h; y++) memset(out + (size_t)y * out_stride, 0, (size_t)w * sizeof(float)); /* :113 */
        float *blk = (float *)malloc((size_t)(block_w > W ? block_w : W) * (block_h > H ? block_h : H) * sizeof(float));
        float *blk_out = (float *)malloc((size_t)(block_w > W ? block_w : W) * (block_h > H ? block_h : H) * sizeof(float));
        for (unsigned r = 0; r < splitRows && rc == 0; r++) {                  /* :114 */
            const int y0 = (int)r * sh;
            const int y1 = (r == splitRows - 1) ? H : y0 + block_h;            /* :115-121 */
            for (unsigned c = 0; c < splitColumns && rc == 0; c++) {           /* :122 */
                const int x0 = (int)c * sw;
                const int x1 = (c == splitColumns - 1) ? W : x0 + block_w;     /* :123-131 */
                const int bw = x1 - x0, bh = y1 - y0;
                /* The reference hands filter2D a Mat ROI; without BORDER_ISOLATED OpenCV may
                 * read real neighbours beyond the ROI for layer 1.  That only changes the
                 * 7-px rim that is cropped below, so the oracle copies the block (isolated). */
                for (int y = 0; y < bh; y++)
                    memcpy(blk + (size_t)y * bw, padded + (size_t)(y0 + y) * W + x0, (size_t)bw * sizeof(float));
                rc = convert_basic(layers, nlayers, blk, bw, bh, blk_out, njob); /* :135 */
                if (rc) break;
                /* stitch interior: :143-161.  (The reference uses blockSize.height for the
                 * column offset, :153-155; blocks are square in v1 so this is c*sw.) */
                for (int y = 0; y < bh - 2 * nModel; y++)
                    memcpy(out + (size_t)(y0 + y) * out_stride + x0,
                           blk_out + (size_t)(y + nModel) * bw + nModel,
                           (size_t)(bw - 2 * nModel) * sizeof(float));
            }
        }
        free(blk); free(blk_out);
    }
    free(padded);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* fp64 "truth" variant: same loop nest, double accumulation, for error       */
/* budgeting (not a restatement of the reference's rounding).                 */
/* Computes valid-conv CNN(replicate_pad(plane, nlayers)) (SURVEY invariant   */
/* I1), whole plane, single thread-pool over output planes.                   */
/* ------------------------------------------------------------------------- */
typedef struct {
    const w2xc_oracle_layer *L;
    const double *in; double *out; int w, h; int o0, o1;
} w64_arg;

static void *worker64(void *p)
{
    w64_arg *a = (w64_arg *)p;
    const w2xc_oracle_layer *L = a->L;
    const int w = a->w, h = a->h;
    const size_t plane = (size_t)w * h;
    for (int o = a->o0; o < a->o1; o++) {
        double *acc = a->out + (size_t)o * plane;
        for (size_t q = 0; q < plane; q++) acc[q] = 0.0;
        for (int i = 0; i < L->nin; i++) {
            const float *k9 = L->weight + ((size_t)o * L->nin + i) * 9;
            const double *src = a->in + (size_t)i * plane;
            for (int y = 0; y < h; y++) {
                const double *r[3] = { src + (size_t)clampi(y - 1, 0, h - 1) * w, src + (size_t)y * w,
                                       src + (size_t)clampi(y + 1, 0, h - 1) * w };
                for (int x = 0; x < w; x++) {
                    const int xs[3] = { clampi(x - 1, 0, w - 1), x, clampi(x + 1, 0, w - 1) };
                    double t = 0.0;
                    for (int kr = 0; kr < 3; kr++)
                        for (int kc = 0; kc < 3; kc++) t += (double)k9[kr * 3 + kc] * r[kr][xs[kc]];
                    acc[(size_t)y * w + x] += t;
                }
            }
        }
        for (size_t q = 0; q < plane; q++) {
            double v = acc[q] + L->bias[o];
            acc[q] = v > 0.0 ? v : 0.1 * v;
        }
    }
    return NULL;
}

int w2xc_oracle_convert_f64(const w2xc_oracle_layer *layers, int nlayers, const float *in,
                            size_t in_stride, int w, int h, double *out, int njob)
{
    const int p = nlayers, W = w + 2 * p, H = h + 2 * p;
    const size_t px = (size_t)W * H;
    int maxc = 1;
    for (int l = 0; l < nlayers; l++) {
        if (layers[l].nin > maxc) maxc = layers[l].nin;
        if (layers[l].nout > maxc) maxc = layers[l].nout;
    }
    double *a = (double *)malloc(px * maxc * sizeof(double));
    double *b = (double *)malloc(px * maxc * sizeof(double));
    if (!a || !b) { free(a); free(b); return -2; }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            a[(size_t)y * W + x] = in[(size_t)clampi(y - p, 0, h - 1) * in_stride + clampi(x - p, 0, w - 1)];
    if (njob < 1) njob = 1;
    int nplanes = 1;
    for (int l = 0; l < nlayers; l++) {
        if (nplanes != layers[l].nin) { free(a); free(b); return -1; }
        pthread_t th[64]; w64_arg args[64];
        int nj = njob > 64 ? 64 : njob;
        if (nj > layers[l].nout) nj = layers[l].nout;
        for (int t = 0; t < nj; t++) {
            args[t] = (w64_arg){ &layers[l], a, b, W, H, (int)((long)layers[l].nout * t / nj),
                                 (int)((long)layers[l].nout * (t + 1) / nj) };
            pthread_create(&th[t], NULL, worker64, &args[t]);
        }
        for (int t = 0; t < nj; t++) pthread_join(th[t], NULL);
        nplanes = layers[l].nout;
        double *t = a; a = b; b = t;
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) out[(size_t)y * w + x] = a[(size_t)(y + p) * W + (x + p)];
    free(a); free(b);
    return 0;
}
