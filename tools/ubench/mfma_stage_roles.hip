// Does it matter WHICH wave of a SIMD carries the stage's VALU work?  (round 5 study for conv3x3_wino4, r5_sweeps.log 9; the loop of mfma_stage.hip:
// 8 waves, per stage 36 tied MFMAs + 18 operand ds_read_b128 per wave + one barrier) with the 2 x 42 transform instructions of a SIMD's two waves
//   ROLE 0: 42 in each wave (what conv3x3_wino4 does)   1: all 84 in the OLDER wave (waves 0..3)   2: all 84 in the YOUNGER wave (waves 4..7)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off mfma_stage_roles.hip -o mfma_stage_roles && ./mfma_stage_roles
// (original header:) What can a barrier-coupled stage loop reach on the fp32 matrix pipe?
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
template <int ROLE, int SLOT0>
__global__ void __launch_bounds__(512, 2) k(const float *in, float *out, int stages)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 18 * 1024; i += 512) lds[i] = in[i & 1023];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char *ua = reinterpret_cast<const char *>(lds) + (wave & 3) * 1024 + lane * 16;           // + g * 4096
    const char *va = reinterpret_cast<const char *>(lds) + 36864 + (wave >> 2) * 1024 + lane * 16;   // + g * 2048
    f32x4 acc[36];
    for (int i = 0; i < 36; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dd[36];
    for (int i = 0; i < 36; i++) dd[i] = in[(lane + i) & 1023];
    f32x4 a4[3], b4[3];
    for (int g = 0; g < 3; g++) { a4[g] = *reinterpret_cast<const f32x4 *>(ua + g * 4096); b4[g] = *reinterpret_cast<const f32x4 *>(va + g * 2048); }
    const bool heavy = ROLE == 1 ? wave < 4 : wave >= 4;
    auto stage = [&](auto HEAVY) {
        constexpr int nt = ROLE == 0 ? 3 : (decltype(HEAVY)::value ? 6 : 0);   // 6-point transforms in this wave's stage
        static_for<0, 36>([&](auto XI) {
            constexpr int xi = decltype(XI)::value, g = xi >> 2;
            if constexpr (xi == 32) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                a4[0] = *reinterpret_cast<const f32x4 *>(ua); b4[0] = *reinterpret_cast<const f32x4 *>(va);
                a4[1] = *reinterpret_cast<const f32x4 *>(ua + 4096); b4[1] = *reinterpret_cast<const f32x4 *>(va + 2048);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr ((xi & 3) == 0 && g + 2 < 9) {
                a4[(g + 2) % 3] = *reinterpret_cast<const f32x4 *>(ua + (g + 2) * 4096);
                b4[(g + 2) % 3] = *reinterpret_cast<const f32x4 *>(va + (g + 2) * 2048);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[xi]) : "v"(a4[g % 3][xi & 3]), "v"(b4[g % 3][xi & 3]));
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (xi >= SLOT0 && xi < SLOT0 + nt) {
                constexpr int r = xi - SLOT0;
                bt6(dd[6 * r], dd[6 * r + 1], dd[6 * r + 2], dd[6 * r + 3], dd[6 * r + 4], dd[6 * r + 5]);
                asm volatile("" : "+v"(dd[6 * r]), "+v"(dd[6 * r + 1]), "+v"(dd[6 * r + 2]), "+v"(dd[6 * r + 3]), "+v"(dd[6 * r + 4]), "+v"(dd[6 * r + 5]));
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };
    if (ROLE == 0 || heavy) { for (int s = 0; s < stages; s++) stage(std::true_type{}); }
    else { for (int s = 0; s < stages; s++) stage(std::false_type{}); }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float sum = 0;
    for (int i = 0; i < 36; i++) for (int e = 0; e < 4; e++) sum += acc[i][e];
    for (int i = 0; i < 36; i++) sum += dd[i];
    out[blockIdx.x * 512 + threadIdx.x] = sum;
}
template <int ROLE, int SLOT0>
static void run(const float *in, float *out, const char *name)
{
    const int stages = 4000;
    auto kern = k<ROLE, SLOT0>;
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
    printf("%-60s %.3f ms  = %.0f ns per stage (%s)\n", name, best, best * 1e6 / stages, hipGetErrorString(hipGetLastError()));
}
int main()
{
    float *in, *out;
    std::vector<float> h(1024);
    for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    (void)hipMalloc(&in, 4096); (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; rep++) {
        run<0, 3>(in, out, "42 VALU in every wave (behind MFMA slots 3..5)");
        run<1, 3>(in, out, "84 VALU in the OLDER wave of a SIMD (slots 3..8)");
        run<2, 3>(in, out, "84 VALU in the YOUNGER wave of a SIMD (slots 3..8)");
        run<1, 20>(in, out, "84 VALU in the OLDER wave, late (slots 20..25)");
        run<2, 20>(in, out, "84 VALU in the YOUNGER wave, late (slots 20..25)");
        run<0, 20>(in, out, "42 VALU in every wave, late (slots 20..22)");
    }
    return 0;
}
