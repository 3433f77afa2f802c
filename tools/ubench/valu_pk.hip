// Is packed fp32 (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32) twice the scalar rate when NO MFMA shares the SIMD?  (r3_sweeps.log 4 measured it beside MFMAs:
// a packed instruction cost twice a plain one there.  conv3x3_wino4's epilogue is a pure VALU phase -- ~900 scalar fp32 instructions per wave and item, two
// waves per SIMD, 7.2k cycles per item with the matrix pipe idle: DESIGN 9.)
//   hipcc --offload-arch=gfx950 -O3 valu_pk.hip -o valu_pk && ./valu_pk
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>   // 0: v_fma_f32 x 16 per iteration, 1: v_pk_fma_f32 x 16 per iteration (twice the FLOPs), 2: v_pk_add_f32, 3: v_add_f32, 4: v_pk_mul_f32
__global__ void __launch_bounds__(512) k(float *out, int iters, float a, float b)
{
    f32x2 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = f32x2{(float)threadIdx.x + i, (float)i};
    const f32x2 A = {a, a}, B = {b, b};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if constexpr (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i][0]) : "v"(a), "v"(b));
            if constexpr (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(A), "v"(B));
            if constexpr (MODE == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(B));
            if constexpr (MODE == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i][0]) : "v"(b));
            if constexpr (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(A));
            if constexpr (MODE == 5) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i][0]) : "v"(a), "v"(b));                 // VOP2: dst += a * b   (3 register reads)
            if constexpr (MODE == 6) asm volatile("v_fmamk_f32 %0, %0, 0x3f400000, %1" : "+v"(v[i][0]) : "v"(b));            // dst = dst * K + b     (2 register reads + literal)
            if constexpr (MODE == 7) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f400000" : "+v"(v[i][0]) : "v"(a));            // dst = dst * a + K
            if constexpr (MODE == 8) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i][0]) : "v"(a));
            if constexpr (MODE == 9) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[i][0]) : "v"(v[i][1]), "s"(b));       // 2 register reads + scalar
            if constexpr (MODE == 10) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i][0]) : "s"(a), "v"(b));              // fma with one SCALAR operand (2 register reads)
            if constexpr (MODE == 11) asm volatile("v_fma_f32 %0, %0, 0.5, %1" : "+v"(v[i][0]) : "v"(b));                     // fma with an inline constant
            if constexpr (MODE == 12) asm volatile("v_mul_f32 %0, 0x3dcccccd, %0" : "+v"(v[i][0]));                           // mul by a literal
            if constexpr (MODE == 13) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "s"(A), "v"(B));               // packed fma, one scalar pair
            if constexpr (MODE == 14) asm volatile("v_mov_b32 %0, %1" : "+v"(v[i][0]) : "v"(v[i][1]));
            if constexpr (MODE == 16) asm volatile("v_fmac_f32 %0, 0x3fc00000, %1" : "+v"(v[i][0]) : "v"(v[i][1]));   // dst += K * x (literal: 2 register reads)
            if constexpr (MODE == 17) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i][0]) : "v"(v[i][1]));
            if constexpr (MODE == 18) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[i][0]) : "v"(v[i][1]), "v"(b));
            if constexpr (MODE == 15) asm volatile("v_accvgpr_write_b32 a0, %0" : : "v"(v[i][0]) : "a0");
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += v[i][0] + v[i][1];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int MODE>
static void run(const char *name, float *d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    float best = 1e9f;
    for (int r = 0; r < 4; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, d, iters, 1.0001f, 0.5f);   // 8 waves per CU = 2 per SIMD, one workgroup per CU
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r && ms < best) best = ms;
    }
    // per SIMD: 2 waves x 16 instructions x iters
    const double instr = 2.0 * 16 * iters;
    printf("%-16s %.3f ms  = %.2f ns per wave-instruction per SIMD  (%.2f cycles at 2.4 GHz)\n", name, best, best * 1e6 / instr, best * 1e6 / instr * 2.4);
}
int main()
{
    float *d; hipMalloc(&d, 256 * 512 * 4);
    for (int rep = 0; rep < 2; rep++) {
        run<0>("v_fma_f32 vvv", d); run<1>("v_pk_fma_f32", d); run<3>("v_add_f32", d); run<2>("v_pk_add_f32", d); run<4>("v_pk_mul_f32", d);
        run<5>("v_fmac_f32", d); run<6>("v_fmamk_f32", d); run<7>("v_fmaak_f32", d); run<8>("v_mul_f32", d); run<9>("v_med3 vvs", d);
        run<10>("v_fma_f32 vsv", d); run<11>("v_fma inline c", d); run<12>("v_mul literal", d); run<13>("v_pk_fma s-pair", d); run<14>("v_mov_b32", d);
        run<16>("v_fmac literal", d); run<17>("v_max_f32", d); run<18>("v_med3 vvv", d);
    }
    return 0;
}
