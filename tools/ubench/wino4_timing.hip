// conv3x3_wino4 alone: correctness against a double-precision direct sum on sampled outputs (every tile edge included), time per launch, and --
// with -DW4_TIMING -- where the cycles go (s_memtime stamps of waves 0 and 4 of workgroup 0 after every stage close and every epilogue).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize [-DW4_TIMING] [-DW4_ABL=n] -I../../waifu2x-converter-cpp_amd/csrc wino4_timing.hip -o wino4_timing
//   ./wino4_timing <cin> <cout> [h w [nhwc_out]]
#include "w2xc_wino4.hip"
#include <cmath>
#include <cstdio>
#include <vector>

// planar in: in[c][y][x] (row stride irs, plane stride ics); out element (c, y, x) at c * ocs + y * ors + x * ops
__global__ void ref_check(const float *in, long long irs, long long ics, long long ips, const float *w, const float *bias, const float *out, long long ors, long long ops,
                          long long ocs, int cin, int cout, int h, int wd, int ystep, int xstep, int off_y, double *maxerr, double *maxref, unsigned long long *nbad)
{
    const int o = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int ny = (h + ystep - 1) / ystep, nx = (wd + xstep - 1) / xstep;
    if (idx >= ny * nx) return;
    int y = (idx / nx) * ystep, x = (idx % nx) * xstep;
    // walk the sample grid with an offset that depends on the plane so that all residues mod 4 / 16 / 32 are hit
    y = (y + o * 5) % h;
    x = (x + o * 3) % wd;
    double acc = 0.0;
    for (int c = 0; c < cin; c++)
        for (int r = 0; r < 3; r++)
            for (int q = 0; q < 3; q++) acc += (double)w[((size_t)o * cin + c) * 9 + r * 3 + q] * (double)in[c * ics + (long long)(y + off_y + r) * irs + (x + q) * ips];
    acc += (double)bias[o];
    acc = acc > 0 ? acc : 0.1 * acc;
    const double got = (double)out[o * ocs + (long long)y * ors + (long long)x * ops];
    const double err = fabs(got - acc);
    atomicMax(reinterpret_cast<unsigned long long *>(maxerr), (unsigned long long)__double_as_longlong(err));          // (non-negative doubles order like integers)
    atomicMax(reinterpret_cast<unsigned long long *>(maxref), (unsigned long long)__double_as_longlong(fabs(acc)));
    if (!(err <= 1e-5 + 1e-4 * fabs(acc))) atomicAdd(nbad, 1ull);
}

// fused last layer: G[ob][tap][y][x] = sum over the 64 planes of block ob of w7[plane][tap] * leaky(bias + conv(in)[plane]) -- against a double-precision sum
__global__ void ref_check_fused(const float *in, long long irs, long long ics, long long ips, const float *w, const float *bias, const float *w7, const float *G,
                                long long gts, long long ggs, long long grs, int cin, int cout, int h, int wd, int nsamp, int off_y, double *maxerr, double *maxref,
                                unsigned long long *nbad)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nsamp) return;
    unsigned r = 2654435761u * (unsigned)(idx + 1);
    const int x = (int)(r % (unsigned)wd); r = r * 1664525u + 1013904223u;
    const int y = (int)((r >> 8) % (unsigned)h); r = r * 1664525u + 1013904223u;
    const int tap = (int)((r >> 8) % 9u); r = r * 1664525u + 1013904223u;
    const int ob = (int)((r >> 8) % (unsigned)(cout / 64));
    double sum = 0.0;
    for (int o = ob * 64; o < ob * 64 + 64; o++) {
        double acc = 0.0;
        for (int c = 0; c < cin; c++)
            for (int rr = 0; rr < 3; rr++)
                for (int q = 0; q < 3; q++) acc += (double)w[((size_t)o * cin + c) * 9 + rr * 3 + q] * (double)in[c * ics + (long long)(y + off_y + rr) * irs + (x + q) * ips];
        acc += (double)bias[o];
        acc = acc > 0 ? acc : 0.1 * acc;
        sum += (double)w7[o * 9 + tap] * acc;
    }
    const double got = (double)G[ob * gts + tap * ggs + (long long)y * grs + x];
    const double err = fabs(got - sum);
    atomicMax(reinterpret_cast<unsigned long long *>(maxerr), (unsigned long long)__double_as_longlong(err));
    atomicMax(reinterpret_cast<unsigned long long *>(maxref), (unsigned long long)__double_as_longlong(fabs(sum)));
    if (!(err <= 2e-5 + 1e-4 * fabs(sum))) atomicAdd(nbad, 1ull);
}

int main(int argc, char **argv)
{
    const int cin = argc > 1 ? atoi(argv[1]) : 128, cout = argc > 2 ? atoi(argv[2]) : 128;
    const int h = argc > 3 ? atoi(argv[3]) : 2160, w = argc > 4 ? atoi(argv[4]) : 3840;
    const int nhwc_out = argc > 5 ? atoi(argv[5]) : 0;
    const int wino_py = argc > 6 ? atoi(argv[6]) : 0, off_y = argc > 7 ? atoi(argv[7]) : 0;   // (block phase of the first row; rows skipped in the input)
    const int ih = h + 2 + off_y + (argc > 8 ? atoi(argv[8]) : 0), iw = w + 2;
    const int nhwc_in = argc > 9 ? atoi(argv[9]) : 0;   // (cin = 32 only)
    const int fuse = argc > 10 ? atoi(argv[10]) : 0;     // the one-plane last layer in the epilogue: the output is the partial tap planes
    const long long irs = nhwc_in ? (long long)iw * cin : (iw + 3) & ~3, ics = nhwc_in ? 1 : irs * ih, ips = nhwc_in ? cin : 1;
    const long long ors_p = (w + 3) & ~3, ocs_p = ors_p * h;
    std::vector<float> hin(nhwc_in ? (size_t)irs * ih : (size_t)ics * cin), hw((size_t)cout * cin * 9), hb(cout);
    {   // (a 32-bit LCG instead of rand(): a gigabyte of input is filled in about a second -- box time is GPU budget)
        unsigned x = 12345u;
        for (auto &v : hin) { x = x * 1664525u + 1013904223u; v = (float)(x >> 8) * (1.0f / 16777216.0f); }
        // the pad columns of every row hold garbage on purpose: results must not depend on them
        for (int c = 0; c < cin && !nhwc_in; c++)
            for (int y = 0; y < ih; y++)
                for (long long xx = iw; xx < irs; xx++) hin[c * ics + y * irs + xx] = (y & 1) ? NAN : 1e30f;
    }
    for (auto &v : hw) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
    for (int o = 0; o < cout; o++) hb[o] = 0.01f * (float)(o % 7) - 0.02f;
    std::vector<float> pk((size_t)36 * cin * cout);
    w2xc_wino4_pack(cin, cout, hw.data(), pk.data());
    float *din, *dout, *dw, *dwraw, *db;
    const size_t out_floats = fuse ? (size_t)(cout / 64) * 9 * h * w : nhwc_out ? (size_t)h * w * cout : (size_t)ocs_p * cout;
    hipMalloc(&din, hin.size() * 4); hipMalloc(&dout, out_floats * 4); hipMalloc(&dw, pk.size() * 4); hipMalloc(&db, cout * 4); hipMalloc(&dwraw, hw.size() * 4);
    hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dw, pk.data(), pk.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dwraw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), cout * 4, hipMemcpyHostToDevice);
    hipMemset(dout, 0xff, out_floats * 4);
    W2xcConvDesc d;
    memset(&d, 0, sizeof d);
    d.in = din; d.out = dout; d.wpk = dw; d.bias = db; d.cin = cin; d.cout = cout;
    d.in_rs = irs; d.in_ps = ips; d.in_cs = ics;
    if (nhwc_out) { d.out_rs = (long long)w * cout; d.out_ps = cout; d.out_cs = 1; }
    else { d.out_rs = ors_p; d.out_ps = 1; d.out_cs = ocs_p; }
    d.in_h = ih; d.in_w = iw; d.out_h = h; d.out_w = w; d.wino_py = wino_py; d.off_y = off_y;
    std::vector<float> hw7((size_t)cout * 9), pk7(w2xc_wino4_pack_last_floats(cout));
    float *dw7 = nullptr, *dw7raw = nullptr;
    if (fuse) {
        for (auto &v : hw7) v = ((float)rand() / RAND_MAX - 0.5f) * 0.2f;
        w2xc_wino4_pack_last(cout, hw7.data(), pk7.data());
        hipMalloc(&dw7, pk7.size() * 4); hipMalloc(&dw7raw, hw7.size() * 4);
        hipMemcpy(dw7, pk7.data(), pk7.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dw7raw, hw7.data(), hw7.size() * 4, hipMemcpyHostToDevice);
        d.out_terms = 9; d.w7pk = dw7; d.out_rs = w; d.out_ps = 1; d.out_gs = (long long)h * w; d.out_ts = 9 * d.out_gs;
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipError_t e = w2xc_launch_wino4(d, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("wino4 %d->%d %dx%d %s py %d off_y %d: %.3f ms (%s / %s)\n", cin, cout, h, w, fuse ? "fused-last" : nhwc_out ? "nhwc-out" : "planar-out", wino_py, off_y, ms, hipGetErrorString(e), hipGetErrorString(hipGetLastError()));
    }
#if W4_ABL == 0
    if (fuse) {
        double *dm; unsigned long long *dbad;
        hipMalloc(&dm, 16); hipMalloc(&dbad, 8); hipMemset(dm, 0, 16); hipMemset(dbad, 0, 8);
        const int nsamp = h * w * 9 > 40000 ? 40000 : h * w * 9;
        hipLaunchKernelGGL(ref_check_fused, dim3((nsamp + 255) / 256), dim3(256), 0, 0, din, irs, ics, ips, dwraw, db, dw7raw, dout, d.out_ts, d.out_gs, d.out_rs, cin, cout, h, w,
                           nsamp, off_y, dm, dm + 1, dbad);
        double hm[2]; unsigned long long bad;
        hipMemcpy(hm, dm, 16, hipMemcpyDeviceToHost); hipMemcpy(&bad, dbad, 8, hipMemcpyDeviceToHost);
        printf("fused check (%d samples of the partial tap planes): max |err| %.3g, max |ref| %.3g, outside 2e-5 + 1e-4 |ref|: %llu  %s\n", nsamp, hm[0], hm[1], bad,
               bad == 0 && hm[1] > 0 ? "OK" : "FAILED");
    } else {
        double *dm; unsigned long long *dbad;
        hipMalloc(&dm, 16); hipMalloc(&dbad, 8); hipMemset(dm, 0, 16); hipMemset(dbad, 0, 8);
        const int ystep = h > 600 ? 7 : 1, xstep = w > 600 ? 5 : 1;
        const int ny = (h + ystep - 1) / ystep, nx = (w + xstep - 1) / xstep;
        dim3 grid((ny * nx + 255) / 256, cout);
        hipLaunchKernelGGL(ref_check, grid, dim3(256), 0, 0, din, irs, ics, ips, dwraw, db, dout, d.out_rs, d.out_ps, d.out_cs, cin, cout, h, w, ystep, xstep, off_y, dm, dm + 1, dbad);
        double hm[2]; unsigned long long bad;
        hipMemcpy(hm, dm, 16, hipMemcpyDeviceToHost); hipMemcpy(&bad, dbad, 8, hipMemcpyDeviceToHost);
        printf("check (%d x %d samples x %d planes): max |err| %.3g, max |ref| %.3g, outside 1e-5 + 1e-4 |ref|: %llu  %s\n", ny, nx, cout, hm[0], hm[1], bad,
               bad == 0 && hm[1] > 0 ? "OK" : "FAILED");
    }
#endif
#ifdef W4_TIMING
    static unsigned long long st[2][8192];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(w4_stamps), sizeof st);
    const int nst = cin / 4, per_item = 3 * nst + 1;
    for (int g = 0; g < 2; g++) {
        // stamps: [0] = start, then per item: per stage (before the closing wait, after it, after the barrier), then the epilogue's end
        std::vector<double> run(nst, 0), wait(nst, 0), bar(nst, 0);
        double epi = 0; int items = 0;
        for (int it = 2; it < 40 && 1 + (it + 1) * per_item < 8192 && st[g][(it + 1) * per_item]; it++) {
            const unsigned long long *s = &st[g][it * per_item];   // s[0] = end of the previous item's epilogue
            for (int k = 0; k < nst; k++) {
                run[k] += (double)(s[3 * k + 1] - s[3 * k]);
                wait[k] += (double)(s[3 * k + 2] - s[3 * k + 1]);
                bar[k] += (double)(s[3 * k + 3] - s[3 * k + 2]);
            }
            epi += (double)(s[3 * nst + 1] - s[3 * nst]);
            items++;
        }
        if (!items) { printf("wave %d: no stamps\n", 4 * g); continue; }
        printf("wave %d (%d items), cycles per stage: issue phase / vmcnt+lgkm wait / barrier wait   (ideal: 2304 per stage for the two waves of a SIMD)\n", 4 * g, items);
        double tot = epi / items;
        for (int k = 0; k < nst; k++) {
            if (k < 8 || k >= nst - 4) printf("   stage %2d: %5.0f %5.0f %5.0f\n", k, run[k] / items, wait[k] / items, bar[k] / items);
            tot += (run[k] + wait[k] + bar[k]) / items;
        }
        printf("   epilogue %5.0f   item %.0f  (%.0f per stage)\n", epi / items, tot, tot / nst);
    }
#endif
    return 0;
}
