// w2xc_wino4_math.h -- the Winograd F(4x4, 3x3) transforms shared by conv3x3_wino4 (w2xc_wino4.hip) and conv3x3_wino4s (w2xc_wino4s.hip).
// Cook-Toom on the points 0, +-3/4, +-3/2, inf (tools/winograd_points.py: every entry a dyadic rational <= 27/8):
//   B^T = [81/64 0 -45/16 0 1 0; 0 -27/16 -9/4 3/4 1 0; 0 27/16 -9/4 -3/4 1 0; 0 -27/32 -9/16 3/2 1 0; 0 27/32 -9/16 -3/2 1 0; 0 81/64 0 -45/16 0 1]
//   G   = [64/81 0 0; -128/243 -32/81 -8/27; -128/243 32/81 -8/27; 32/243 16/81 8/27; 32/243 -16/81 8/27; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 3/4 -3/4 3/2 -3/2 0; 0 9/16 9/16 9/4 9/4 0; 0 27/64 -27/64 27/8 -27/8 1]
// Y = A^T [ (G g G^T) (.) (B^T d B) ] A per 4x4 output block and 6x6 input patch, the 3x3 correlation of Model::filterWorker
// (/root/reference/src/modelHandler.cpp:134-145) with 2.25 multiplies per output.
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));

// y = B^T x for a 6-vector, in place (14 fma / mul / add)
static __device__ __forceinline__ void bt6(float &x0, float &x1, float &x2, float &x3, float &x4, float &x5)
{
    const float y0 = __builtin_fmaf(-2.8125f, x2, __builtin_fmaf(1.265625f, x0, x4));
    const float p = __builtin_fmaf(-2.25f, x2, x4), q = __builtin_fmaf(-1.6875f, x1, 0.75f * x3);
    const float u = __builtin_fmaf(-0.5625f, x2, x4), v = __builtin_fmaf(-0.84375f, x1, 1.5f * x3);
    const float y5 = __builtin_fmaf(-2.8125f, x3, __builtin_fmaf(1.265625f, x1, x5));
    x0 = y0;
    x1 = p + q;
    x2 = p - q;
    x3 = u + v;
    x4 = u - v;
    x5 = y5;
}

// y = A^T m for a 6-vector (12 fma / mul / add)
static __device__ __forceinline__ void at6(float m0, float m1, float m2, float m3, float m4, float m5, float &y0, float &y1, float &y2, float &y3)
{
    const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
    y0 = m0 + s1 + s2;
    y1 = __builtin_fmaf(1.5f, d2, 0.75f * d1);
    y2 = __builtin_fmaf(2.25f, s2, 0.5625f * s1);
    y3 = __builtin_fmaf(3.375f, d2, __builtin_fmaf(0.421875f, d1, m5));
}

// G as doubles (host side: the weight images U = G g G^T are formed in double and rounded once)
static const double W2XC_WINO4_G[6][3] = {{64.0 / 81, 0, 0},
                                          {-128.0 / 243, -32.0 / 81, -8.0 / 27},
                                          {-128.0 / 243, 32.0 / 81, -8.0 / 27},
                                          {32.0 / 243, 16.0 / 81, 8.0 / 27},
                                          {32.0 / 243, -16.0 / 81, 8.0 / 27},
                                          {0, 0, 1}};
