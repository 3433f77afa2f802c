// Pure v_mfma_f32_32x32x2_f32 issue-rate microbenchmark (calibration of the fp32 MFMA ceiling
// on this box, incl. data-dependent DVFS).  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) k(const float *in, float *out, int iters)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char **argv)
{
    int wg_per_cu = argc > 1 ? atoi(argv[1]) : 1;
    int zero = argc > 2 ? atoi(argv[2]) : 0;
    int iters = 20000;
    float *in, *out;
    std::vector<float> h(512);
    for (auto &v : h) v = zero ? 0.f : (float)rand() / RAND_MAX - 0.5f;
    hipMalloc(&in, 2048); hipMalloc(&out, 256 * 8 * 256 * 4);
    hipMemcpy(in, h.data(), 2048, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * wg_per_cu;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)grid * 4 * iters * 8 * 4 * 4096.0;
        printf("wg/cu=%d zero=%d: %.3f ms  %.1f TFLOP/s\n", wg_per_cu, zero, ms, flops / ms / 1e9);
    }
    return 0;
}
