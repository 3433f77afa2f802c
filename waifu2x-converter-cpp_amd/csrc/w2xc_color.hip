// w2xc_color.hip -- row N2 of SURVEY.md 8f: the colour front/back end and the U/V resize of the reference
// CLI's scale loop (/root/reference/src/main.cpp:74-76,136,144,171-172), so that the whole scale phase of
// one image runs device-resident.  All three kernels are HBM-streaming (a few bytes per pixel) and keep
// OpenCV's float evaluation order with unfused mul/add (the file is built with -ffp-contract=off).
#include "w2xc_kernels.h"

static __device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// main.cpp:75-76 (+ cv::split): convertTo(CV_32F, 1/255) and COLOR_RGB2YUV on the channels AS GIVEN
// (the reference feeds imread's BGR order, Q3): Y = .299 c0 + .587 c1 + .114 c2, U = (c2-Y)*.492+.5, V = (c0-Y)*.877+.5
__global__ void __launch_bounds__(256) k_u8_to_yuv(const unsigned char *src, long long stride, int w, int h, float *y, float *u, float *v)
{
    const long long total = (long long)w * h;
    const float s = (float)(1.0 / 255.0);
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
        const int r = (int)(q / w), c = (int)(q - (long long)r * w);
        const unsigned char *p = src + r * stride + (long long)c * 3;
        const float c0 = (float)p[0] * s, c1 = (float)p[1] * s, c2 = (float)p[2] * s;
        float Y = c0 * 0.299f;
        Y = Y + c1 * 0.587f;
        Y = Y + c2 * 0.114f;
        y[q] = Y;
        u[q] = (c2 - Y) * 0.492f + 0.5f;
        v[q] = (c0 - Y) * 0.877f + 0.5f;
    }
}

// main.cpp:171-172 (+ cv::merge): COLOR_YUV2RGB and convertTo(CV_8U, 255) = saturate(cvRound(v*255))
__global__ void __launch_bounds__(256) k_yuv_to_u8(const float *y, const float *u, const float *v, int w, int h, unsigned char *dst, long long stride)
{
    const long long total = (long long)w * h;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
        const int r = (int)(q / w), c = (int)(q - (long long)r * w);
        const float Y = y[q], U = u[q] - 0.5f, V = v[q] - 0.5f;
        float ch[3];
        ch[2] = Y + U * 2.032f;
        ch[1] = (Y + U * -0.395f) + V * -0.581f;
        ch[0] = Y + V * 1.140f;
        unsigned char *p = dst + r * stride + (long long)c * 3;
#pragma unroll
        for (int k = 0; k < 3; k++) p[k] = (unsigned char)clampi(__float2int_rn(ch[k] * 255.0f), 0, 255);   // round half to even
    }
}

static __device__ __forceinline__ void cubic_coeffs(float t, float *c)   // Keys cubic, A = -0.75 (OpenCV interpolateCubic)
{
    const float A = -0.75f;
    c[0] = ((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A;
    c[1] = ((A + 2) * t - (A + 3)) * t * t + 1;
    c[2] = ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

// main.cpp:144 on one plane: cv::resize(2x, INTER_CUBIC): horizontal pass to float, then vertical pass,
// taps sx-1..sx+2 clipped to the image.  One thread per output pixel (4 x 4 source taps from L2/L1).
__global__ void __launch_bounds__(256) k_resize2x_cubic(const float *src, int w, int h, float *dst)
{
    const int W = 2 * w, H = 2 * h;
    const long long total = (long long)W * H;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
        const int dy = (int)(q / W), dx = (int)(q - (long long)dy * W);
        const float fx = (float)((dx + 0.5) * 0.5 - 0.5), fy = (float)((dy + 0.5) * 0.5 - 0.5);
        const int sx = (int)floorf(fx), sy = (int)floorf(fy);
        float cx[4], cy[4];
        cubic_coeffs(fx - sx, cx);
        cubic_coeffs(fy - sy, cy);
        int xs[4];
#pragma unroll
        for (int k = 0; k < 4; k++) xs[k] = clampi(sx - 1 + k, 0, w - 1);
        float rowv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float *S = src + (long long)clampi(sy - 1 + j, 0, h - 1) * w;
            float a = S[xs[0]] * cx[0];
            a = a + S[xs[1]] * cx[1];
            a = a + S[xs[2]] * cx[2];
            a = a + S[xs[3]] * cx[3];
            rowv[j] = a;
        }
        float a = rowv[0] * cy[0];
        a = a + rowv[1] * cy[1];
        a = a + rowv[2] * cy[2];
        a = a + rowv[3] * cy[3];
        dst[q] = a;
    }
}

// main.cpp:158-167 on one plane: cv::resize(Size(dw, dh), INTER_LINEAR): half-pixel centres, the two taps
// clipped to the image, horizontal pass to float then vertical pass (no antialiasing, like OpenCV).
__global__ void __launch_bounds__(256) k_resize_linear(const float *src, int sw, int sh, float *dst, int dw, int dh, double scale_x, double scale_y)
{
    const long long total = (long long)dw * dh;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
        const int dy = (int)(q / dw), dx = (int)(q - (long long)dy * dw);
        float fx = (float)((dx + 0.5) * scale_x - 0.5), fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sx = (int)floorf(fx), sy = (int)floorf(fy);
        fx -= sx;
        fy -= sy;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        if (sy < 0) { fy = 0; sy = 0; }
        if (sy >= sh - 1) { fy = 0; sy = sh - 1; }
        const int sx1 = sx + 1 < sw ? sx + 1 : sw - 1, sy1 = sy + 1 < sh ? sy + 1 : sh - 1;
        const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
        const float *R0 = src + (long long)sy * sw, *R1 = src + (long long)sy1 * sw;
        float h0 = R0[sx] * a0;
        h0 = h0 + R0[sx1] * a1;
        float h1 = R1[sx] * a0;
        h1 = h1 + R1[sx1] * a1;
        float a = h0 * b0;
        a = a + h1 * b1;
        dst[q] = a;
    }
}

static unsigned grid_for(long long total)
{
    long long b = (total + 255) / 256;
    return (unsigned)(b > 65536 ? 65536 : (b < 1 ? 1 : b));
}

hipError_t w2xc_launch_u8_to_yuv(const unsigned char *src, size_t stride, int w, int h, float *y, float *u, float *v, hipStream_t st)
{
    hipLaunchKernelGGL(k_u8_to_yuv, dim3(grid_for((long long)w * h)), dim3(256), 0, st, src, (long long)stride, w, h, y, u, v);
    return hipGetLastError();
}
hipError_t w2xc_launch_yuv_to_u8(const float *y, const float *u, const float *v, int w, int h, unsigned char *dst, size_t stride, hipStream_t st)
{
    hipLaunchKernelGGL(k_yuv_to_u8, dim3(grid_for((long long)w * h)), dim3(256), 0, st, y, u, v, w, h, dst, (long long)stride);
    return hipGetLastError();
}
hipError_t w2xc_launch_resize2x_cubic(const float *src, int w, int h, float *dst, hipStream_t st)
{
    hipLaunchKernelGGL(k_resize2x_cubic, dim3(grid_for(4LL * w * h)), dim3(256), 0, st, src, w, h, dst);
    return hipGetLastError();
}
hipError_t w2xc_launch_resize_linear(const float *src, int sw, int sh, float *dst, int dw, int dh, hipStream_t st)
{
    hipLaunchKernelGGL(k_resize_linear, dim3(grid_for((long long)dw * dh)), dim3(256), 0, st, src, sw, sh, dst, dw, dh,
                       (double)sw / dw, (double)sh / dh);
    return hipGetLastError();
}
