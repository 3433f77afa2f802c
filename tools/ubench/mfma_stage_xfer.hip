// Does a U stream at HALF-stage granularity buy anything?  (tuning aid for conv3x3_wino4)
// The synthetic stage of mfma_stage.hip -- 8 waves, per stage 36 tied v_mfma_f32_16x16x4_f32 + 18 operand ds_read_b128 + 42 VALU per wave -- with the ONE thing
// it lacked: the U transfers of the real kernel (36 pieces of 1 KiB per stage and workgroup, global_load_lds_dwordx4 from a 2.25 MiB L2-resident image, issued
// by the four older waves behind the stage's first MFMA slots, counted vmcnt + barrier in front of the last four MFMAs).  Three forms:
//   N  no transfers (the 0.86-of-the-pipe loop of mfma_stage.hip)
//   W  whole stages: two U slots of 36 KiB, the pieces of stage s + 1 issued during stage s (what conv3x3_wino4 does: 72 KiB, one close per stage)
//   H  half stages: a ring of three 20 KiB slots (groups 0..4 | 5..8 of a stage's nine operand groups), the pieces of half h + 2 issued during half h
//      (a full stage of look-ahead instead of "the rest of this stage"), TWO closes per stage, 60 KiB
// Timing only (operands are whatever lands; the accumulators are summed so nothing is optimised away).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Wno-inline-asm mfma_stage_xfer.hip -o mfma_stage_xfer && ./mfma_stage_xfer
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
template <unsigned LDS_IMM>
static __device__ __forceinline__ void dma16(const void *sbase, unsigned voff, unsigned lds_base)
{
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_base), "n"(LDS_IMM) : "memory", "m0", "scc");
}
#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// LDS: V at 0 (18 KiB, static), U ring behind it.  U operand of group g of the slot at `ub`: ub + (g - g0) * 4096 + pt * 1024 + lane * 16.
constexpr unsigned V_BYTES = 18 * 1024, U_AT = V_BYTES;
template <int MODE>   // 0 N, 1 W, 2 H
__global__ void __launch_bounds__(512, 2) k(const float *in, const char *wimg, float *out, int stages)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int LDS_FLOATS = (V_BYTES + (MODE == 2 ? 3 * 20 : 2 * 36) * 1024) / 4;
    for (int i = threadIdx.x; i < LDS_FLOATS; i += 512) lds[i] = in[i & 1023];
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    char *ldsb = reinterpret_cast<char *>(lds);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pt = wave & 3, bt = wave >> 2;
    const unsigned voff = (unsigned)lane * 16u;
    const unsigned wbase = (unsigned)__builtin_amdgcn_readfirstlane(lds0 + U_AT + (unsigned)pt * 1024u);   // a DMA wave (bt = 0) sends the pieces of ITS plane tile pt
    const char *va = ldsb + bt * 1024 + lane * 16;            // V: + g * 2048
    f32x4 acc[36];
    for (int i = 0; i < 36; i++) for (int e = 0; e < 4; e++) acc[i][e] = 0.0f;
    float dd[18];
    for (int i = 0; i < 18; i++) dd[i] = in[(lane + i) & 1023];
    f32x4 a4[3], b4[3];
    // U bytes of stage s in the image: 64 stages of 36 KiB, walked cyclically (2.25 MiB: L2-resident, as the 128 -> 128 layer's 2.36 MB are)
    auto stage_img = [&](int s) { return wimg + (size_t)(s & 63) * 36864 + (size_t)pt * 1024; };
    auto ua_of = [&](unsigned slot_off, int g_rel) { return ldsb + U_AT + slot_off + g_rel * 4096 + pt * 1024 + lane * 16; };

    if constexpr (MODE == 1) {
        // prologue: U of stage 0 into slot 0
        if (bt == 0) static_for<0, 9>([&](auto G) { dma16<decltype(G)::value * 4096u>(stage_img(0) + decltype(G)::value * 4096, voff, wbase); });
        WAIT_VMCNT(0);
        __syncthreads();
    }
    if constexpr (MODE == 2) {
        // prologue: halves 0 and 1 (stage 0) into ring slots 0 and 1
        if (bt == 0) {
            static_for<0, 5>([&](auto G) { dma16<decltype(G)::value * 4096u>(stage_img(0) + decltype(G)::value * 4096, voff, wbase); });
            static_for<5, 9>([&](auto G) { dma16<20480u + (decltype(G)::value - 5) * 4096u>(stage_img(0) + decltype(G)::value * 4096, voff, wbase); });
        }
        WAIT_VMCNT(0);
        __syncthreads();
    }
    int ring = 0;   // MODE 2: ring slot of the half being read (0..2)
    for (int s = 0; s < stages; s++) {
        const unsigned uslot = MODE == 1 ? (unsigned)(s & 1) * 36864u : 0u;
        const unsigned unext = MODE == 1 ? (unsigned)((s + 1) & 1) * 36864u : 0u;
        const char *img_next = stage_img(s + 1);
        // MODE 2: slots of this stage's two halves and of the two halves being fetched
        const unsigned h0 = (unsigned)ring * 20480u, h1 = (unsigned)((ring + 1) % 3) * 20480u, f0 = (unsigned)((ring + 2) % 3) * 20480u, f1 = h0;
        auto ua = [&](int g) -> const char * {
            if constexpr (MODE == 2) return g < 5 ? ua_of(h0, g) : ua_of(h1, g - 5);
            else return ua_of(uslot, g);
        };
        if (s == 0) {
            for (int g = 0; g < 2; g++) { a4[g] = *reinterpret_cast<const f32x4 *>(ua(g)); b4[g] = *reinterpret_cast<const f32x4 *>(va + g * 2048); }
        }
        static_for<0, 36>([&](auto XI) {
            constexpr int xi = decltype(XI)::value, g = xi >> 2;
            if constexpr (MODE == 2 && xi == 12) {
                // first close: half 1 of this stage (issued during the second half of the LAST stage) has landed; this stage's first fetch (5 pieces per
                // DMA wave, issued behind slots 1..5) may still fly
                if (bt == 0) WAIT_VMCNT(5);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if constexpr (xi == 32) {
                // the stage's close, in front of its last four MFMAs
                if constexpr (MODE == 1) { if (bt == 0) WAIT_VMCNT(0); }
                if constexpr (MODE == 2) { if (bt == 0) WAIT_VMCNT(4); }   // (half 0 of the next stage has landed; the 4 pieces of its half 1 may fly)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                // first operands of the NEXT stage
                const char *n0, *n1;
                if constexpr (MODE == 2) { n0 = ua_of(f0, 0); n1 = ua_of(f0, 1); }
                else { n0 = ua_of(unext, 0); n1 = ua_of(unext, 1); }
                a4[0] = *reinterpret_cast<const f32x4 *>(n0); b4[0] = *reinterpret_cast<const f32x4 *>(va);
                a4[1] = *reinterpret_cast<const f32x4 *>(n1); b4[1] = *reinterpret_cast<const f32x4 *>(va + 2048);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr ((xi & 3) == 0 && g + 2 < 9) {
                a4[(g + 2) % 3] = *reinterpret_cast<const f32x4 *>(ua(g + 2));
                b4[(g + 2) % 3] = *reinterpret_cast<const f32x4 *>(va + (g + 2) * 2048);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[xi]) : "v"(a4[g % 3][xi & 3]), "v"(b4[g % 3][xi & 3]));
            __builtin_amdgcn_sched_barrier(0);
            // transfers behind the MFMA slots (the four older waves, nine pieces each per stage)
            if constexpr (MODE == 1 && xi >= 1 && xi <= 9) {
                constexpr int q = xi - 1;
                if (bt == 0) dma16<0>(img_next + q * 4096, voff, wbase + unext + q * 4096u);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (MODE == 2 && xi >= 1 && xi <= 5) {      // first half of the stage: fetch half 0 of stage s + 1 into the slot half 1 of stage s - 1 has left
                constexpr int q = xi - 1;
                if (bt == 0) dma16<0>(img_next + q * 4096, voff, wbase + f0 + q * 4096u);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (MODE == 2 && xi >= 21 && xi <= 24) {    // second half (behind the first close: half 0's slot is free): fetch half 1 of stage s + 1 into it
                constexpr int q = xi - 21;
                if (bt == 0) dma16<0>(img_next + (5 + q) * 4096, voff, wbase + f1 + q * 4096u);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (xi >= 3 && xi < 6) {
                constexpr int r = xi - 3;
                bt6(dd[6 * r], dd[6 * r + 1], dd[6 * r + 2], dd[6 * r + 3], dd[6 * r + 4], dd[6 * r + 5]);
                asm volatile("" : "+v"(dd[6 * r]), "+v"(dd[6 * r + 1]), "+v"(dd[6 * r + 2]), "+v"(dd[6 * r + 3]), "+v"(dd[6 * r + 4]), "+v"(dd[6 * r + 5]));
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        if constexpr (MODE == 2) ring = (ring + 2) % 3;
    }
    WAIT_VMCNT(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float sum = 0;
    for (int i = 0; i < 36; i++) for (int e = 0; e < 4; e++) sum += acc[i][e];
    for (int i = 0; i < 18; i++) sum += dd[i];
    out[blockIdx.x * 512 + threadIdx.x] = sum;
}
template <int MODE>
static void run(const float *in, const char *wimg, float *out, const char *name)
{
    const int stages = 4000;
    auto kern = k<MODE>;
    const size_t lds = V_BYTES + (MODE == 2 ? 3 * 20 : 2 * 36) * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, in, wimg, out, stages);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-64s %.3f ms  = %.0f ns per stage (%s)\n", name, best, best * 1e6 / stages, hipGetErrorString(hipGetLastError()));
}
int main()
{
    float *in, *out; char *wimg;
    std::vector<float> h(1024);
    for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    (void)hipMalloc(&in, 4096); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&wimg, 65 * 36864 + 65536);
    (void)hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    (void)hipMemset(wimg, 0, 65 * 36864 + 65536);
    for (int rep = 0; rep < 2; rep++) {
        run<0>(in, wimg, out, "N  no transfers (36 MFMA + 18 ds_read_b128 + 42 VALU + barrier)");
        run<1>(in, wimg, out, "W  + U whole stages, 2 x 36 KiB, one close per stage");
        run<2>(in, wimg, out, "H  + U half stages, 3 x 20 KiB ring, two closes per stage");
    }
    return 0;
}
