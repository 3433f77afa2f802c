// What does one non-MFMA instruction cost beside v_mfma_f32_16x16x4_f32?  (tuning aid for conv3x3_wino16: DESIGN.md 3)
// Each wave runs a stream of independent MFMAs (32 accumulator quads) with NF filler instructions of one KIND behind every MFMA;
// 1 or 2 waves per SIMD.  Prints cycles of matrix-pipe time per MFMA (32 = the floor) -> cost per filler = (cyc - base) / NF.
//   hipcc --offload-arch=gfx950 -O3 mfma_fillers.hip -o mfma_fillers && ./mfma_fillers
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
enum { NONE, VADD, VPKADD, VFMA, VMOV, SNOP, DSREAD64, DSREAD128, VPKMUL, VMAX };
template <int KIND, int NF, int WAVES>
__global__ void __launch_bounds__(WAVES * 256, 2) k(const float *in, float *out, int iters)
{
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = in[i & 511];
    __syncthreads();
    f32x4 acc[32];
    for (int i = 0; i < 32; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = in[threadIdx.x & 255], b = in[(threadIdx.x & 255) + 256];
    float v[8];
    f32x2 p[8];
    for (int i = 0; i < 8; i++) { v[i] = in[i + (threadIdx.x & 63)]; p[i] = f32x2{v[i], v[i] + 1.f}; }
    f32x4 q[4] = {};
    const char *lb = reinterpret_cast<const char *>(lds) + (threadIdx.x & 63) * 16;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 32; i++) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < NF; f++) {
                const int j = (i * NF + f) & 7;
                if constexpr (KIND == VADD) { v[j] = v[j] + v[(j + 3) & 7]; asm volatile("" : "+v"(v[j])); }
                if constexpr (KIND == VFMA) { v[j] = __builtin_fmaf(v[j], v[(j + 3) & 7], v[(j + 5) & 7]); asm volatile("" : "+v"(v[j])); }
                if constexpr (KIND == VMAX) { v[j] = __builtin_fmaxf(v[j], v[(j + 3) & 7]); asm volatile("" : "+v"(v[j])); }
                if constexpr (KIND == VMOV) { asm volatile("v_mov_b32 %0, %1" : "=v"(v[j]) : "v"(v[(j + 3) & 7])); }
                if constexpr (KIND == VPKADD) { p[j] = p[j] + p[(j + 3) & 7]; asm volatile("" : "+v"(p[j])); }
                if constexpr (KIND == VPKMUL) { p[j] = p[j] * p[(j + 3) & 7]; asm volatile("" : "+v"(p[j])); }
                if constexpr (KIND == SNOP) asm volatile("s_nop 0");
                if constexpr (KIND == DSREAD64) { f32x2 t = *reinterpret_cast<const f32x2 *>(lb + ((i * NF + f) & 15) * 1024); asm volatile("" : "+v"(t)); p[j] = t; }
                if constexpr (KIND == DSREAD128) { f32x4 t = *reinterpret_cast<const f32x4 *>(lb + ((i * NF + f) & 15) * 1024); asm volatile("" : "+v"(t)); q[j & 3] = t; }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 32; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; i++) s += v[i] + p[i][0] + p[i][1];
    for (int i = 0; i < 4; i++) s += q[i][0] + q[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND, int NF, int WAVES>
static double run(const float *in, float *out, const char *name, double base)
{
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND, NF, WAVES>), dim3(256), dim3(WAVES * 256), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // MFMAs per SIMD = WAVES * iters * 32; time per MFMA in ns; report as "cycles at 2.4 GHz" (the clock moves: compare rows, not absolutes)
    const double ns_per = best * 1e6 / ((double)WAVES * iters * 32);
    const double cyc = ns_per * 2.4;
    printf("%d wave(s)/SIMD  %-10s x%d : %.3f ms  %.2f cyc/MFMA @2.4GHz", WAVES, name, NF, best, cyc);
    if (base > 0 && NF > 0) printf("   cost per filler %.2f cyc", (cyc - base) / NF);
    printf("\n");
    return cyc;
}
int main()
{
    float *in, *out;
    std::vector<float> h(1024);
    for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMalloc(&in, 4096); hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
#define ROW(W) { double b = run<NONE, 0, W>(in, out, "none", 0); \
    run<VADD, 1, W>(in, out, "v_add_f32", b); run<VADD, 2, W>(in, out, "v_add_f32", b); run<VADD, 4, W>(in, out, "v_add_f32", b); \
    run<VPKADD, 1, W>(in, out, "v_pk_add", b); run<VPKADD, 2, W>(in, out, "v_pk_add", b); \
    run<VFMA, 1, W>(in, out, "v_fma_f32", b); run<VMAX, 1, W>(in, out, "v_max_f32", b); run<VPKMUL, 1, W>(in, out, "v_pk_mul", b); \
    run<VMOV, 1, W>(in, out, "v_mov_b32", b); run<VMOV, 2, W>(in, out, "v_mov_b32", b); run<SNOP, 1, W>(in, out, "s_nop", b); run<SNOP, 4, W>(in, out, "s_nop", b); \
    run<DSREAD64, 1, W>(in, out, "ds_read64", b); run<DSREAD128, 1, W>(in, out, "ds_read128", b); }
    ROW(1) ROW(2)
    return 0;
}
