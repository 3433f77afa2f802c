// libw2xc_probe.so -- MEASUREMENT AID for bench.py, not part of the drop-in (include/w2xc_hip.h does not declare it, libw2xc_hip.so does not contain it).
// gpurun boxes differ by a few percent (bench.py's frame: 13.75-14.29 ms over five boxes of one afternoon, profiles/r5_sweeps.log); the roofline fraction
// is priced at the nominal 2.4 GHz, so bench.py records beside it what the matrix pipes of THIS box deliver: every SIMD of the chip runs a stream of
// independent v_mfma_f32_16x16x4_f32 on random operands (8 passes = 32 cycles each, two waves per SIMD so that the loop overhead hides), and
//     MHz = MFMAs per SIMD per second x 32 / 1e6.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512, 1) probe_mfma(const float *in, float *out, int iters)
{
    f32x4 acc[16];
    for (int i = 0; i < 16; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
    for (int i = 1; i < 16; i++) s += acc[i];
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[threadIdx.x] = s[0];   // (keeps the chain alive)
}

// Runs the stream for about `ms` milliseconds on `device` and returns the matrix pipes' MHz-equivalent (< 0: a HIP error, printed to stderr).
extern "C" double w2xc_probe_mfma_mhz(int device, int ms)
{
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "w2xc_probe: %s: %s\n", #x, hipGetErrorString(e_)); return -1.0; } } while (0)
    CK(hipSetDevice(device));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, device));
    const int cus = prop.multiProcessorCount;
    float *buf = nullptr;
    CK(hipMalloc(&buf, 4096));
    float h[1024];                                        // operands in (-1, 1): a datapath that toggles is what the clock is granted for (zeros run faster)
    unsigned r = 12345u;
    for (int i = 0; i < 1024; i++) { r = r * 1664525u + 1013904223u; h[i] = (float)(int)(r >> 8) * (1.0f / 8388608.0f) - 1.0f; }
    CK(hipMemcpy(buf, h, sizeof h, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double mhz = -1.0;
    int iters = 2000;                                     // 2000 x 16 MFMAs x 2 waves x 32 cycles = 2.05 M cycles (~0.85 ms at 2.4 GHz)
    for (int pass = 0; pass < 3; pass++) {                // pass 0 warms up, pass 1 sizes the run to `ms`, pass 2 is the measurement
        CK(hipEventRecord(e0, 0));
        probe_mfma<<<cus, 512, 0, 0>>>(buf, buf + 512, iters);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t = 0.f;
        CK(hipEventElapsedTime(&t, e0, e1));
        mhz = (double)iters * 16 * 2 * 32 / (t * 1e-3) / 1e6;
        if (pass == 1 && t > 0.f) {
            const double want = (double)iters * (ms > 0 ? ms : 20) / t;
            iters = want > 2e6 ? 2000000 : want < 2000 ? 2000 : (int)want;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(buf);
    return mhz;
#undef CK
}
