// Where do the cycles of conv3x3_wino4 go?  Includes the kernel file with -DW4_TIMING (s_memtime stamps of waves 0 and 4 of workgroup 0 after every
// stage close and after every epilogue) and runs one layer on synthetic data.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DW4_TIMING -I../../waifu2x-converter-cpp_amd/csrc wino4_timing.hip -o wino4_timing
//   ./wino4_timing <cin> <cout> [h w]
#include "w2xc_wino4.hip"
#include <cstdio>
#include <vector>
int main(int argc, char **argv)
{
    const int cin = argc > 1 ? atoi(argv[1]) : 128, cout = argc > 2 ? atoi(argv[2]) : 128;
    const int h = argc > 3 ? atoi(argv[3]) : 2160, w = argc > 4 ? atoi(argv[4]) : 3840;
    const int ih = h + 2, iw = w + 2;
    std::vector<float> hin((size_t)ih * iw * cin), hw((size_t)cout * cin * 9), hb(cout, 0.01f);
    {   // (a 32-bit LCG instead of rand(): a gigabyte of input is filled in about a second -- box time is GPU budget)
        unsigned x = 12345u;
        for (auto &v : hin) { x = x * 1664525u + 1013904223u; v = (float)(x >> 8) * (1.0f / 16777216.0f); }
    }
    for (auto &v : hw) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
    std::vector<float> pk((size_t)36 * cin * cout);
    w2xc_wino4_pack(cin, cout, hw.data(), pk.data());
    float *din, *dout, *dw, *db;
    hipMalloc(&din, hin.size() * 4); hipMalloc(&dout, (size_t)h * w * cout * 4); hipMalloc(&dw, pk.size() * 4); hipMalloc(&db, cout * 4);
    hipMemcpy(din, hin.data(), hin.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dw, pk.data(), pk.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), cout * 4, hipMemcpyHostToDevice);
    W2xcConvDesc d;
    memset(&d, 0, sizeof d);
    d.in = din; d.out = dout; d.wpk = dw; d.bias = db; d.cin = cin; d.cout = cout;
    d.in_rs = (long long)iw * cin; d.in_ps = cin; d.in_cs = 1; d.out_rs = (long long)w * cout; d.out_ps = cout; d.out_cs = 1;
    d.in_h = ih; d.in_w = iw; d.out_h = h; d.out_w = w;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipError_t e = w2xc_launch_wino4(d, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%d->%d %dx%d: %.3f ms (%s)\n", cin, cout, h, w, ms, hipGetErrorString(e));
    }
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
#ifdef W4_SLOT_STAMPS   // (needs the per-slot W4_SLOT stamps compiled into the kernel: the patch is described in profiles/r3_sweeps.log block 21)
    // slot stamps inside a stage: the transforming stage of waves 0 / 4 and a plain stage of the same waves (every third MFMA, then the stage's end)
    static unsigned long long sl[2][2][64][40];
    hipMemcpyFromSymbol(sl, HIP_SYMBOL(w4_slots), sizeof sl);
    for (int g = 0; g < 2; g++)
        for (int tr = 0; tr < 2; tr++) {
            printf("wave %d %s stage, cycles from the first MFMA to MFMA #: ", 4 * g, tr ? "TRANSFORMING" : "plain");
            for (int x = 3; x <= 36; x += 3) {
                double sum = 0; int cnt = 0;
                for (int k = 0; k < 64; k++)
                    if (sl[g][tr][k][0] && sl[g][tr][k][x] > sl[g][tr][k][0]) { sum += (double)(sl[g][tr][k][x] - sl[g][tr][k][0]); cnt++; }
                printf("%d:%.0f ", x, cnt ? sum / cnt : 0.0);
            }
            printf("\n");
        }
#endif
    return 0;
}
