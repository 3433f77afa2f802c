// Can the A operands (U) of conv3x3_wino4's stage come STRAIGHT from L2 into registers instead of through LDS?  (round 5 study)
// The synthetic stage of mfma_stage.hip -- 8 waves per CU, per stage 36 tied v_mfma_f32_16x16x4_f32 per wave, B operands (V) from LDS, 42 VALU, one barrier --
// with the nine A quads of a stage loaded by global_load_dwordx4 from a 2.36 MB weight image (128 -> 128: [ob][stage][group][pt][lane][4], every CU reads all of
// it: L2-resident), D groups (of four MFMAs) ahead.  Prints ns per stage; the LDS-operand form of mfma_stage.hip needs ~1130.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off mfma_stage_gl.hip -o mfma_stage_gl && ./mfma_stage_gl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
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
// D: look-ahead of the A quads in groups (a ring of D + 1 register quads); GLOBAL_A: 1 = A from global memory, 0 = A from LDS (the reference form); VALU, BAR as in mfma_stage.hip
template <int D, bool GLOBAL_A, bool VALU, bool BAR>
__global__ void __launch_bounds__(512, 2) k(const float *in, const float *wimg, float *out, int stages)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 18 * 1024; i += 512) lds[i] = in[i & 1023];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char *ua = reinterpret_cast<const char *>(lds) + (wave & 3) * 1024 + lane * 16;           // + g * 4096
    const char *va = reinterpret_cast<const char *>(lds) + 36864 + (wave >> 2) * 1024 + lane * 16;   // + g * 2048
    // the weight image: 64 (ob, stage) slabs of 36 KiB; quad g of the wave's plane tile at slab * 36864 + g * 4096 + pt * 1024 + lane * 16
    const char *ga = reinterpret_cast<const char *>(wimg) + (wave & 3) * 1024 + lane * 16;
    f32x4 acc[36];
    for (int i = 0; i < 36; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dd[18];
    for (int i = 0; i < 18; i++) dd[i] = in[(lane + i) & 1023];
    constexpr int R = D + 1;
    f32x4 a4[R], b4[3];
    auto load_a = [&](int slab, int g, int slot) {
        if constexpr (GLOBAL_A) a4[slot] = *reinterpret_cast<const f32x4 *>(ga + (size_t)slab * 36864 + g * 4096);
        else a4[slot] = *reinterpret_cast<const f32x4 *>(ua + g * 4096);
    };
    // prologue: groups 0 .. D-1 of stage 0
    static_for<0, D>([&](auto G) { constexpr int g = decltype(G)::value; load_a((g / 9) & 63, g % 9, g % R); });
    for (int g = 0; g < 2; g++) b4[g] = *reinterpret_cast<const f32x4 *>(va + g * 2048);
    int slab = blockIdx.x & 63;   // (CUs start at different slabs: the L2 serves 32 CUs of an XCD at different addresses)
    // the stage loop, unrolled by R stages so that the ring slots are compile-time (9 groups per stage; slot of group (s, g) = (9 s + g) mod R)
    for (int s0 = 0; s0 < stages; s0 += R) {
        static_for<0, R>([&](auto SS) {
            constexpr int ss = decltype(SS)::value;
            static_for<0, 36>([&](auto XI) {
                constexpr int xi = decltype(XI)::value, g = xi >> 2;
                if constexpr (xi == 32 && BAR) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    b4[0] = *reinterpret_cast<const f32x4 *>(va); b4[1] = *reinterpret_cast<const f32x4 *>(va + 2048);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr ((xi & 3) == 0) {
                    // A quad D groups ahead: group index gi = 9 ss + g + D of this unrolled block (possibly in the next stage(s))
                    constexpr int gi = 9 * ss + g + D;
                    const int sl = (slab + gi / 9) & 63;
                    load_a(sl, gi % 9, gi % R);
                    if constexpr (g + 2 < 9) b4[(g + 2) % 3] = *reinterpret_cast<const f32x4 *>(va + (g + 2) * 2048);
                    __builtin_amdgcn_sched_barrier(0);
                }
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[xi]) : "v"(a4[(9 * ss + g) % R][xi & 3]), "v"(b4[g % 3][xi & 3]));
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (VALU && xi >= 3 && xi < 6) {
                    constexpr int r = xi - 3;
                    bt6(dd[6 * r], dd[6 * r + 1], dd[6 * r + 2], dd[6 * r + 3], dd[6 * r + 4], dd[6 * r + 5]);
                    asm volatile("" : "+v"(dd[6 * r]), "+v"(dd[6 * r + 1]), "+v"(dd[6 * r + 2]), "+v"(dd[6 * r + 3]), "+v"(dd[6 * r + 4]), "+v"(dd[6 * r + 5]));
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        });
        slab = (slab + R) & 63;
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float sum = 0;
    for (int i = 0; i < 36; i++) for (int e = 0; e < 4; e++) sum += acc[i][e];
    for (int i = 0; i < 18; i++) sum += dd[i];
    for (int i = 0; i < R; i++) sum += a4[i][0];
    out[blockIdx.x * 512 + threadIdx.x] = sum;
}
template <int D, bool GLOBAL_A, bool VALU, bool BAR>
static void run(const float *in, const float *wimg, float *out, const char *name)
{
    const int stages = 4000 / (D + 1) * (D + 1);
    auto kern = k<D, GLOBAL_A, VALU, BAR>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 72 * 1024, 0, in, wimg, out, stages);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-64s %.3f ms  = %.0f ns per stage (%s)\n", name, best, best * 1e6 / stages, hipGetErrorString(hipGetLastError()));
}
int main()
{
    float *in, *out, *wimg;
    std::vector<float> h(1024);
    for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    (void)hipMalloc(&in, 4096); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&wimg, 64 * 36864);
    (void)hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    (void)hipMemset(wimg, 0, 64 * 36864);
    for (int rep = 0; rep < 2; rep++) {
        run<2, false, true, true>(in, wimg, out, "A from LDS, 2 groups ahead, + barrier + 42 VALU (reference)");
        run<2, true, true, true>(in, wimg, out, "A from L2, 2 groups ahead");
        run<4, true, true, true>(in, wimg, out, "A from L2, 4 groups ahead");
        run<6, true, true, true>(in, wimg, out, "A from L2, 6 groups ahead");
        run<8, true, true, true>(in, wimg, out, "A from L2, 8 groups ahead (a whole stage)");
        run<8, true, false, true>(in, wimg, out, "A from L2, 8 groups ahead, no VALU");
        run<8, true, true, false>(in, wimg, out, "A from L2, 8 groups ahead, no barrier");
    }
    return 0;
}
