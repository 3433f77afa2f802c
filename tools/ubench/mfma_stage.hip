// What can a barrier-coupled stage loop reach on the fp32 matrix pipe?  (tuning aid for conv3x3_wino4: DESIGN.md 3)
// One workgroup of 8 waves per CU (two per SIMD), every wave per "stage": NM independent MFMAs on its own accumulators (36 x v_mfma_f32_16x16x4_f32 = 144
// registers, or 18 x v_mfma_f32_32x32x2_f32 = 9 x 16 registers: the same 1152 cycles of pipe time), optionally its operand quads from LDS (ds_read_b128,
// one group of MFMAs ahead), optionally 42 VALU instructions (three 6-point transforms) behind MFMA slots 3, 4, 5, and one s_barrier in front of the
// last group of MFMAs.  Prints cycles per stage at the measured rate (the pipe's floor is 2304 for the two waves of a SIMD).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off mfma_stage.hip -o mfma_stage && ./mfma_stage
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int B, int E, class F>
static __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}
static __device__ __forceinline__ void bt6(float &x0, float &x1, float &x2, float &x3, float &x4, float &x5)
{
    const float y0 = __builtin_fmaf(-2.8125f, x2, __builtin_fmaf(1.265625f, x0, x4));
    const float p = __builtin_fmaf(-2.25f, x2, x4), q = __builtin_fmaf(-1.6875f, x1, 0.75f * x3);
    const float u = __builtin_fmaf(-0.5625f, x2, x4), v = __builtin_fmaf(-0.84375f, x1, 1.5f * x3);
    const float y5 = __builtin_fmaf(-2.8125f, x3, __builtin_fmaf(1.265625f, x1, x5));
    x0 = y0; x1 = p + q; x2 = p - q; x3 = u + v; x4 = u - v; x5 = y5;
}
// BIG: 1 = 32x32x2 (18 per stage) instead of 16x16x4 (36); 2 = 27 x v_mfma_f32_16x16x16_f16 (three fp16 products per position: 9 positions x 16 channels, the same
// operand bytes as one fp32 stage -- DESIGN.md 9.2); LDSOPS: operand quads from LDS; VALU: the 42 transform instructions; BAR: the barrier
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <int BIG, bool LDSOPS, bool VALU, bool BAR>
__global__ void __launch_bounds__(512, 2) k(const float *in, float *out, int stages)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 18 * 1024; i += 512) lds[i] = in[i & 1023];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char *ua = reinterpret_cast<const char *>(lds) + (wave & 3) * 1024 + lane * 16;           // + g * 4096
    const char *va = reinterpret_cast<const char *>(lds) + 36864 + (wave >> 2) * 1024 + lane * 16;   // + g * 2048
    constexpr int NQ = BIG == 1 ? 9 : 36;
    typedef typename std::conditional<BIG == 1, f32x16, f32x4>::type acc_t;
    acc_t acc[NQ];
    for (int i = 0; i < NQ; i++) for (int e = 0; e < (BIG == 1 ? 16 : 4); e++) acc[i][e] = 0.0f;
    float dd[18];
    for (int i = 0; i < 18; i++) dd[i] = in[(lane + i) & 1023];
    f32x4 a4[3], b4[3];
    for (int g = 0; g < 3; g++) { a4[g] = *reinterpret_cast<const f32x4 *>(ua + g * 4096); b4[g] = *reinterpret_cast<const f32x4 *>(va + g * 2048); }
    for (int s = 0; s < stages; s++) {
        static_for<0, 36>([&](auto XI) {
            constexpr int xi = decltype(XI)::value, g = xi >> 2;
            if constexpr (xi == 32 && BAR) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if constexpr (LDSOPS) {
                    a4[0] = *reinterpret_cast<const f32x4 *>(ua); b4[0] = *reinterpret_cast<const f32x4 *>(va);
                    a4[1] = *reinterpret_cast<const f32x4 *>(ua + 4096); b4[1] = *reinterpret_cast<const f32x4 *>(va + 2048);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (LDSOPS && (xi & 3) == 0 && g + 2 < 9 && (BIG != 1 || (g & 1) == 0)) {   // (32x32x2: half the operand dwords per stage)
                a4[(g + 2) % 3] = *reinterpret_cast<const f32x4 *>(ua + (g + 2) * 4096);
                b4[(g + 2) % 3] = *reinterpret_cast<const f32x4 *>(va + (g + 2) * 2048);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (BIG == 0) {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[xi]) : "v"(a4[g % 3][xi & 3]), "v"(b4[g % 3][xi & 3]));
            } else if constexpr (BIG == 2) {
                if constexpr (xi < 27) {   // position xi / 3, product xi % 3; an operand = four fp16 = half a quad
                    const f16x4 av = __builtin_bit_cast(f16x4, __builtin_shufflevector(a4[g % 3], a4[g % 3], (xi & 1) * 2, (xi & 1) * 2 + 1));
                    const f16x4 bv = __builtin_bit_cast(f16x4, __builtin_shufflevector(b4[g % 3], b4[g % 3], (xi & 1) * 2, (xi & 1) * 2 + 1));
                    asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc[xi / 3]) : "v"(av), "v"(bv));
                }
            } else if constexpr ((xi & 1) == 0) {
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[(xi >> 1) % 9]) : "v"(a4[g % 3][xi & 3]), "v"(b4[g % 3][xi & 3]));
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (VALU && xi >= 3 && xi < 6) {
                constexpr int r = xi - 3;
                bt6(dd[6 * r], dd[6 * r + 1], dd[6 * r + 2], dd[6 * r + 3], dd[6 * r + 4], dd[6 * r + 5]);
                asm volatile("" : "+v"(dd[6 * r]), "+v"(dd[6 * r + 1]), "+v"(dd[6 * r + 2]), "+v"(dd[6 * r + 3]), "+v"(dd[6 * r + 4]), "+v"(dd[6 * r + 5]));
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float sum = 0;
    for (int i = 0; i < NQ; i++) for (int e = 0; e < (BIG == 1 ? 16 : 4); e++) sum += acc[i][e];
    for (int i = 0; i < 18; i++) sum += dd[i];
    out[blockIdx.x * 512 + threadIdx.x] = sum;
}
template <int BIG, bool LDSOPS, bool VALU, bool BAR>
static void run(const float *in, float *out, const char *name)
{
    const int stages = 4000;
    auto kern = k<BIG, LDSOPS, VALU, BAR>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 72 * 1024, 0, in, out, stages);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // report against the pure-MFMA time of the same number of stages: 2304 cycles per stage
    printf("%-44s %.3f ms  = %.0f ns per stage (%s)\n", name, best, best * 1e6 / stages, hipGetErrorString(hipGetLastError()));
}
int main()
{
    float *in, *out;
    std::vector<float> h(1024);
    for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    (void)hipMalloc(&in, 4096); (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; rep++) {
        run<0, false, false, false>(in, out, "16x16x4: MFMA only");
        run<0, false, false, true>(in, out, "16x16x4: + barrier");
        run<0, true, false, true>(in, out, "16x16x4: + barrier + LDS operands");
        run<0, false, true, true>(in, out, "16x16x4: + barrier + 42 VALU");
        run<0, true, true, true>(in, out, "16x16x4: + barrier + LDS operands + 42 VALU");
        run<0, true, true, false>(in, out, "16x16x4: LDS operands + 42 VALU, no barrier");
        run<1, false, false, false>(in, out, "32x32x2: MFMA only");
        run<1, false, false, true>(in, out, "32x32x2: + barrier");
        run<1, true, false, true>(in, out, "32x32x2: + barrier + LDS operands");
        run<1, false, true, true>(in, out, "32x32x2: + barrier + 42 VALU");
        run<1, true, true, true>(in, out, "32x32x2: + barrier + LDS operands + 42 VALU");
        run<2, false, false, false>(in, out, "16x16x16 f16 x 27: MFMA only");
        run<2, true, false, true>(in, out, "16x16x16 f16 x 27: + barrier + LDS operands");
        run<2, true, true, true>(in, out, "16x16x16 f16 x 27: + barrier + LDS operands + 42 VALU");
    }
    return 0;
}
