/*
 * w2xc_oracle_color.c -- CPU restatement of the colour front/back end and the U/V resize of the reference
 * CLI's scale loop (row N2 of SURVEY.md 8f).  TEST INFRASTRUCTURE ONLY (see w2xc_oracle.c).
 *
 * Reference call sites (/root/reference/src/main.cpp):
 *   :75   image.convertTo(image, CV_32F, 1.0/255.0)
 *   :76   cv::cvtColor(image, image, cv::COLOR_RGB2YUV)     (applied to imread's BGR data as-is, Q3)
 *   :136  cv::resize(image, image2xNearest, 2x, INTER_NEAREST)   -> Y plane for the CNN
 *   :144  cv::resize(image, image2xBicubic, 2x, INTER_CUBIC)     -> U, V planes
 *   :171  cv::cvtColor(image, image, cv::COLOR_YUV2RGB)
 *   :172  image.convertTo(image, CV_8U, 255.0)              (the only clip in the pipeline, Q2)
 *
 * PARITY PINNING STATUS.  These are OpenCV 3.0 core/imgproc functions; OpenCV is un-vendored and absent here and the
 * reference holds no golden vectors, so the formulas below are restated from the OpenCV 3.0 sources AS REMEMBERED (no copy
 * of the tree is reachable offline) and each one is cross-checked against an INDEPENDENT implementation in
 * tests/test_oracle_color.py.  What each restatement follows, and what pins it:
 *
 *   convertTo 8U->32F    modules/core/src/convert.cpp, cvtScale_<uchar, float, float>: dst = saturate_cast<float>(src*scale
 *                        + shift) with scale narrowed to the float work type: (float)u * (float)(1/255.0).
 *                        Pinned by: numpy float32 product (bit-exact), test_convert_to_float.
 *   cvtColor RGB2YUV     modules/imgproc/src/color.cpp, RGB2YCrCb_f<float> built with the YUV coefficient table
 *     (float)             {0.114f, 0.587f, 0.299f, 0.492f, 0.877f} for blueIdx 2... i.e. with channel 0 = "R":
 *                        Y = c0*0.299f + c1*0.587f + c2*0.114f;  U = (c2 - Y)*0.492f + 0.5f;  V = (c0 - Y)*0.877f + 0.5f
 *                        (delta = 0.5f for float images; dst order Y, U, V).  The CLI feeds imread's BGR bytes, so "c0" is
 *                        really blue (quirk Q3) -- irrelevant here, the formula is applied to the channels as given.
 *                        Pinned by: an fp64 numpy matrix form of the documented equations (OpenCV docs, "RGB <-> YUV":
 *                        Y = 0.299R + 0.587G + 0.114B, U = 0.492(B - Y) + 0.5, V = 0.877(R - Y) + 0.5), test_yuv_matrix.
 *   cvtColor YUV2RGB     color.cpp, YCrCb2RGB_f<float> with {2.032f, -0.395f, -0.581f, 1.140f}:
 *     (float)             c2 = Y + (U-0.5f)*2.032f;  c1 = Y + (U-0.5f)*-0.395f + (V-0.5f)*-0.581f;  c0 = Y + (V-0.5f)*1.140f
 *                        Pinned by: the same fp64 matrix form inverted, and the u8 -> YUV -> u8 identity on all sampled bytes.
 *   convertTo 32F->8U    convert.cpp, cvtScale_<float, uchar, float>: saturate_cast<uchar>(cvRound(v*255.f)); cvRound is
 *                        lrint / _mm_cvtss_si32 under the default rounding mode = round half to EVEN; saturate_cast clips
 *                        to [0, 255].  Pinned by: numpy rint on exact .5 products and the clip cases, test_round_saturate.
 *   resize INTER_NEAREST modules/imgproc/src/imgwarp.cpp, resizeNN: sx = min(cvFloor(dx * (1/fx)), sw-1) = dx >> 1 for 2x.
 *                        Pinned by: np.repeat.
 *   resize INTER_CUBIC   imgwarp.cpp, resizeGeneric_ with HResizeCubic / VResizeCubic and interpolateCubic():
 *                        fx = (dx+0.5)*scale - 0.5; sx = cvFloor(fx); t = fx - sx; A = -0.75f;
 *                        c0 = ((A*(t+1) - 5A)*(t+1) + 8A)*(t+1) - 4A;  c1 = ((A+2)*t - (A+3))*t*t + 1;
 *                        c2 = ((A+2)*(1-t) - (A+3))*(1-t)*(1-t) + 1;  c3 = 1 - c0 - c1 - c2;
 *                        taps sx-1..sx+2, indices clipped to [0, sw-1] (border replicate); float work type for CV_32F:
 *                        horizontal pass into float rows, then the vertical pass; each a left-to-right 4-term sum.
 *                        Pinned by: torch F.interpolate(mode="bicubic", align_corners=False) (independent code, A = -0.75,
 *                        half-pixel centres, clamped taps) at 2x on several sizes incl. 1-pixel-wide planes, test_cubic.
 *   resize INTER_LINEAR  imgwarp.cpp, resizeGeneric_ with HResizeLinear / VResizeLinear: fx = (dx+0.5)*scale - 0.5;
 *                        sx = cvFloor(fx); fx -= sx; sx < 0 -> (0, 0); sx >= sw-1 -> (sw-1, 0); weights (1-fx, fx).
 *                        Pinned by: torch F.interpolate(mode="bilinear", align_corners=False, antialias=False) for the
 *                        CLI's shrink ratios (0.75, 0.625, 0.6 ...) and an enlargement, test_linear.
 *
 * STILL UNPINNED (cannot be settled without an OpenCV 3.0.0-rc1 build): (1) whether 3.0.0-rc1 pairs the U / V coefficients
 * and output order exactly as above (later 3.x releases touched the YUV channel order); (2) bit-level summation order of
 * the SSE/AVX paths of the resize passes and of cvtColor versus the scalar order restated here (differences <= 1 float ulp
 * per pass; the independent checks above hold to 1e-6); (3) cvRound under a non-default MXCSR rounding mode.
 * Compile with -ffp-contract=off.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* main.cpp:75-76 + cv::split: interleaved 3-channel u8 (row stride in bytes) -> planar Y,U,V floats */
void w2xc_oracle_u8_to_yuv(const uint8_t *src, size_t stride, int w, int h, float *y, float *u, float *v)
{
    const float s = (float)(1.0 / 255.0);
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++) {
            const uint8_t *p = src + (size_t)r * stride + (size_t)c * 3;
            const float c0 = (float)p[0] * s, c1 = (float)p[1] * s, c2 = (float)p[2] * s;
            float Y = c0 * 0.299f;
            Y = Y + c1 * 0.587f;
            Y = Y + c2 * 0.114f;
            const size_t q = (size_t)r * w + c;
            y[q] = Y;
            u[q] = (c2 - Y) * 0.492f + 0.5f;
            v[q] = (c0 - Y) * 0.877f + 0.5f;
        }
}

/* main.cpp:171-172 + cv::merge: planar Y,U,V floats -> interleaved 3-channel u8 */
void w2xc_oracle_yuv_to_u8(const float *y, const float *u, const float *v, int w, int h, uint8_t *dst, size_t stride)
{
    for (int r = 0; r < h; r++)
        for (int c = 0; c < w; c++) {
            const size_t q = (size_t)r * w + c;
            const float Y = y[q], U = u[q] - 0.5f, V = v[q] - 0.5f;
            float ch[3];
            ch[2] = Y + U * 2.032f;
            ch[1] = (Y + U * -0.395f) + V * -0.581f;
            ch[0] = Y + V * 1.140f;
            uint8_t *p = dst + (size_t)r * stride + (size_t)c * 3;
            for (int k = 0; k < 3; k++) {
                const float t = ch[k] * 255.0f;
                long iv = lrintf(t);                 /* cvRound: round half to even (default FP mode) */
                p[k] = (uint8_t)(iv < 0 ? 0 : (iv > 255 ? 255 : iv));
            }
        }
}

static void cubic_coeffs(float t, float *c)
{
    const float A = -0.75f;
    c[0] = ((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A;
    c[1] = ((A + 2) * t - (A + 3)) * t * t + 1;
    c[2] = ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

/* main.cpp:144 on one plane: cv::resize(src, dst, 2x, INTER_CUBIC) */
void w2xc_oracle_resize2x_cubic(const float *src, int w, int h, float *dst)
{
    const int W = 2 * w, H = 2 * h;
    float *tmp = (float *)malloc((size_t)W * h * sizeof(float));   /* horizontal pass */
    for (int dx = 0; dx < W; dx++) {
        float fx = (float)((dx + 0.5) * 0.5 - 0.5);
        int sx = (int)floorf(fx);
        float c[4];
        cubic_coeffs(fx - sx, c);
        int xs[4];
        for (int k = 0; k < 4; k++) xs[k] = clampi(sx - 1 + k, 0, w - 1);
        for (int r = 0; r < h; r++) {
            const float *S = src + (size_t)r * w;
            float a = S[xs[0]] * c[0];
            a = a + S[xs[1]] * c[1];
            a = a + S[xs[2]] * c[2];
            a = a + S[xs[3]] * c[3];
            tmp[(size_t)r * W + dx] = a;
        }
    }
    for (int dy = 0; dy < H; dy++) {
        float fy = (float)((dy + 0.5) * 0.5 - 0.5);
        int sy = (int)floorf(fy);
        float c[4];
        cubic_coeffs(fy - sy, c);
        const float *R[4];
        for (int k = 0; k < 4; k++) R[k] = tmp + (size_t)clampi(sy - 1 + k, 0, h - 1) * W;
        float *D = dst + (size_t)dy * W;
        for (int x = 0; x < W; x++) {
            float a = R[0][x] * c[0];
            a = a + R[1][x] * c[1];
            a = a + R[2][x] * c[2];
            a = a + R[3][x] * c[3];
            D[x] = a;
        }
    }
    free(tmp);
}

/* main.cpp:136 on one plane: cv::resize(src, dst, 2x, INTER_NEAREST) */
void w2xc_oracle_resize2x_nearest(const float *src, int w, int h, float *dst)
{
    for (int y = 0; y < 2 * h; y++)
        for (int x = 0; x < 2 * w; x++) dst[(size_t)y * 2 * w + x] = src[(size_t)(y >> 1) * w + (x >> 1)];
}

/* main.cpp:158-167 on one plane: cv::resize(src, dst, Size(dw, dh), 0, 0, INTER_LINEAR) for CV_32F.
 * OpenCV semantics: fx = (dx + 0.5) * (sw/dw) - 0.5; sx = floor(fx); fx -= sx; sx < 0 -> (sx, fx) = (0, 0);
 * sx >= sw-1 -> (sw-1, 0); weights (1-fx, fx); horizontal pass to float rows, then vertical pass; each a
 * two-term left-to-right sum.  PARITY UNPINNED (OpenCV absent). */
void w2xc_oracle_resize_linear(const float *src, int sw, int sh, float *dst, int dw, int dh)
{
    const double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
    float *tmp = (float *)malloc((size_t)dw * sh * sizeof(float));
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        const int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
        const float a0 = 1.f - fx, a1 = fx;
        for (int r = 0; r < sh; r++) {
            const float *S = src + (size_t)r * sw;
            float a = S[sx] * a0;
            a = a + S[sx1] * a1;
            tmp[(size_t)r * dw + dx] = a;
        }
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        if (sy < 0) { fy = 0; sy = 0; }
        if (sy >= sh - 1) { fy = 0; sy = sh - 1; }
        const int sy1 = sy + 1 < sh ? sy + 1 : sh - 1;
        const float b0 = 1.f - fy, b1 = fy;
        const float *R0 = tmp + (size_t)sy * dw, *R1 = tmp + (size_t)sy1 * dw;
        for (int x = 0; x < dw; x++) {
            float a = R0[x] * b0;
            a = a + R1[x] * b1;
            dst[(size_t)dy * dw + x] = a;
        }
    }
    free(tmp);
}
