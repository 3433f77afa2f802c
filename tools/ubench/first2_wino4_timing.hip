// conv3x3_first2_wino4 alone (layers 1 + 2 fused: 1 -> 32 -> 32): correctness against a double-precision direct evaluation of both layers on sampled outputs, time per launch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize [-DW4S_ABL=n] [-DW4S_TIMING] -I../../waifu2x-converter-cpp_amd/csrc first2_wino4_timing.hip -o first2_wino4_timing
//   ./first2_wino4_timing [h w [wino_py [off [in_shift]]]]     off: layer 1's offset (<= 0: replicate padding folded into the loads), h x w = layer 2's output
#include "w2xc_first2_wino4.hip"
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void ref_check(const float *src, long long srs, int sh, int sw, int shift, int off, const float *w1, const float *b1, const float *w2, const float *b2, const float *out,
                          long long ors, long long ocs, int h, int wd, int ystep, int xstep, double *maxerr, double *maxref, unsigned long long *nbad)
{
    const int o = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int ny = (h + ystep - 1) / ystep, nx = (wd + xstep - 1) / xstep;
    if (idx >= ny * nx) return;
    int y = (idx / nx) * ystep, x = (idx % nx) * xstep;
    y = (y + o * 5) % h;
    x = (x + o * 3) % wd;
    double acc = 0.0;
    for (int c = 0; c < 32; c++)
        for (int r = 0; r < 3; r++)
            for (int q = 0; q < 3; q++) {
                double a = (double)b1[c];
                for (int rr = 0; rr < 3; rr++)
                    for (int ss = 0; ss < 3; ss++) {
                        int gy = y + r + rr + off, gx = x + q + ss + off;
                        gy = gy < 0 ? 0 : gy > sh - 1 ? sh - 1 : gy;
                        gx = gx < 0 ? 0 : gx > sw - 1 ? sw - 1 : gx;
                        a += (double)w1[c * 9 + rr * 3 + ss] * (double)src[(long long)(gy >> shift) * srs + (gx >> shift)];
                    }
                a = a > 0 ? a : 0.1 * a;
                acc += (double)w2[((size_t)o * 32 + c) * 9 + r * 3 + q] * a;
            }
    acc += (double)b2[o];
    acc = acc > 0 ? acc : 0.1 * acc;
    const double got = (double)out[o * ocs + (long long)y * ors + x];
    const double err = fabs(got - acc);
    atomicMax(reinterpret_cast<unsigned long long *>(maxerr), (unsigned long long)__double_as_longlong(err));          // (non-negative doubles order like integers)
    atomicMax(reinterpret_cast<unsigned long long *>(maxref), (unsigned long long)__double_as_longlong(fabs(acc)));
    if (!(err <= 1e-5 + 1e-4 * fabs(acc))) atomicAdd(nbad, 1ull);
}

int main(int argc, char **argv)
{
    const int h = argc > 1 ? atoi(argv[1]) : 2160, w = argc > 2 ? atoi(argv[2]) : 3840;
    const int wino_py = argc > 3 ? atoi(argv[3]) : 0, off = argc > 4 ? atoi(argv[4]) : 0, shift = argc > 5 ? atoi(argv[5]) : 0;
    // the source plane in UPSCALED coordinates: exactly what a valid layer 1 + layer 2 need when off = 0, smaller (clamped reads) when off < 0
    const int sh = h + 4 + 2 * off, sw = w + 4 + 2 * off;
    const int mh = (sh + shift) >> shift, mw = (sw + shift) >> shift;   // rows / columns in memory
    const long long srs = mw + 5;
    const long long ors_p = (w + 31) & ~31, ocs_p = ors_p * h;
    std::vector<float> hsrc((size_t)srs * mh), hw1(32 * 9), hb1(32), hw2((size_t)32 * 32 * 9), hb2(32);
    {
        unsigned x = 12345u;
        for (auto &v : hsrc) { x = x * 1664525u + 1013904223u; v = (float)(x >> 8) * (1.0f / 16777216.0f); }
        for (int y = 0; y < mh; y++)
            for (long long xx = mw; xx < srs; xx++) hsrc[y * srs + xx] = (y & 1) ? NAN : 1e30f;   // (pad columns: never read)
    }
    for (auto &v : hw1) v = ((float)rand() / RAND_MAX - 0.5f) * 0.9f;
    for (int c = 0; c < 32; c++) hb1[c] = 0.02f * (float)(c % 5) - 0.03f;
    for (auto &v : hw2) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
    for (int o = 0; o < 32; o++) hb2[o] = 0.01f * (float)(o % 7) - 0.02f;
    std::vector<float> pk2((size_t)36 * 32 * 32), pk1(5 * 64, 0.0f);
    w2xc_first2_wino4_pack(hw2.data(), pk2.data());
    for (int k = 0; k < 9; k++)
        for (int c = 0; c < 32; c++) pk1[(k >> 1) * 64 + (k & 1) * 32 + c] = hw1[c * 9 + k];   // the W2XC_K_FIRST image for 1 -> 32
    float *dsrc, *dout, *dpk2, *dpk1, *dw1, *db1, *dw2, *db2;
    hipMalloc(&dsrc, hsrc.size() * 4); hipMalloc(&dout, (size_t)ocs_p * 32 * 4); hipMalloc(&dpk2, pk2.size() * 4); hipMalloc(&dpk1, pk1.size() * 4);
    hipMalloc(&dw1, hw1.size() * 4); hipMalloc(&db1, 128); hipMalloc(&dw2, hw2.size() * 4); hipMalloc(&db2, 128);
    hipMemcpy(dsrc, hsrc.data(), hsrc.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dpk2, pk2.data(), pk2.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dpk1, pk1.data(), pk1.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dw1, hw1.data(), hw1.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db1, hb1.data(), 128, hipMemcpyHostToDevice);
    hipMemcpy(dw2, hw2.data(), hw2.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db2, hb2.data(), 128, hipMemcpyHostToDevice);
    hipMemset(dout, 0xff, (size_t)ocs_p * 32 * 4);
    W2xcConvDesc d;
    memset(&d, 0, sizeof d);
    d.in = dsrc; d.out = dout; d.wpk = dpk2; d.bias = db2; d.w1pk = dpk1; d.bias1 = db1; d.cin = 32; d.cout = 32;
    d.in_rs = srs; d.in_ps = 1; d.in_cs = 0; d.in_h = sh; d.in_w = sw; d.in_shift = shift; d.off_y = off; d.off_x = off;
    d.out_rs = ors_p; d.out_ps = 1; d.out_cs = ocs_p; d.out_h = h; d.out_w = w; d.wino_py = wino_py;
    {
        int nb = -1;
        hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_first2_wino4), hipFuncAttributeMaxDynamicSharedMemorySize, 77184);
        hipError_t eo = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv3x3_first2_wino4, 256, 77184);
        printf("occupancy: %d workgroups per CU (%s)\n", nb, hipGetErrorString(eo));
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipError_t e = w2xc_launch_first2_wino4(d, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("first2_wino4 1->32->32 %dx%d py %d off %d shift %d: %.3f ms (%s / %s)\n", h, w, wino_py, off, shift, ms, hipGetErrorString(e), hipGetErrorString(hipGetLastError()));
    }
#if W4S_ABL == 0
    {
        double *dm; unsigned long long *dbad;
        hipMalloc(&dm, 16); hipMalloc(&dbad, 8); hipMemset(dm, 0, 16); hipMemset(dbad, 0, 8);
        const int ystep = h > 600 ? 7 : 1, xstep = w > 600 ? 5 : 1;
        const int ny = (h + ystep - 1) / ystep, nx = (w + xstep - 1) / xstep;
        dim3 grid((ny * nx + 255) / 256, 32);
        hipLaunchKernelGGL(ref_check, grid, dim3(256), 0, 0, dsrc, srs, sh, sw, shift, off, dw1, db1, dw2, db2, dout, d.out_rs, d.out_cs, h, w, ystep, xstep, dm, dm + 1, dbad);
        double hm[2]; unsigned long long bad;
        hipMemcpy(hm, dm, 16, hipMemcpyDeviceToHost); hipMemcpy(&bad, dbad, 8, hipMemcpyDeviceToHost);
        printf("check (%d x %d samples x 32 planes): max |err| %.3g, max |ref| %.3g, outside 1e-5 + 1e-4 |ref|: %llu  %s\n", ny, nx, hm[0], hm[1], bad,
               bad == 0 && hm[1] > 0 ? "OK" : "FAILED");
    }
#endif
#ifdef W4S_TIMING
    {
        static unsigned long long st[4096];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(w4s_stamps), sizeof st);
        double ph[8] = {0};
        int tiles = 0;
        for (int it = 2; it < 60 && (it + 1) * 8 < 4096 && st[(it + 1) * 8]; it++) {
            const unsigned long long *q = &st[it * 8];
            for (int k = 0; k < 7; k++) ph[k] += (double)(q[k + 1] - q[k]);
            ph[7] += (double)(q[8] - q[7]);
            tiles++;
        }
        if (tiles) printf("workgroup 0 wave 0 (%d tiles), s_memtime ticks: S %.0f | T %.0f | bar %.0f | G %.0f | X %.0f | bar %.0f | O %.0f | bar %.0f = %.0f per tile\n", tiles, ph[0] / tiles,
                          ph[1] / tiles, ph[2] / tiles, ph[3] / tiles, ph[4] / tiles, ph[5] / tiles, ph[6] / tiles, ph[7] / tiles,
                          (ph[0] + ph[1] + ph[2] + ph[3] + ph[4] + ph[5] + ph[6] + ph[7]) / tiles);
    }
#endif
    return 0;
}
