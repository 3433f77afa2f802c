// Where do the cycles of conv3x3_wino16 go?  Includes the kernel file with -DW16_TIMING (s_memtime stamps of one wave of each plane
// group of workgroup 0 after every stage close and after every epilogue) and runs one layer on synthetic data.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DW16_TIMING -I../../waifu2x-converter-cpp_amd/csrc wino16_timing.hip -o wino16_timing
//   ./wino16_timing <cin> <cout> [h w]
#include "w2xc_wino16.hip"
#include <cstdio>
#include <vector>
int main(int argc, char **argv)
{
    const int cin = argc > 1 ? atoi(argv[1]) : 128, cout = argc > 2 ? atoi(argv[2]) : 128;
    const int h = argc > 3 ? atoi(argv[3]) : 2160, w = argc > 4 ? atoi(argv[4]) : 3840;
    const int ih = h + 2, iw = w + 2;
    std::vector<float> hin((size_t)ih * iw * cin), hw((size_t)cout * cin * 9), hb(cout, 0.01f);
    for (auto &v : hin) v = (float)rand() / RAND_MAX;
    for (auto &v : hw) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
    std::vector<float> pk((size_t)16 * cin * cout);
    w2xc_wino16_pack(cin, cout, hw.data(), pk.data());
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
        hipError_t e = w2xc_launch_wino16(d, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%d->%d %dx%d: %.3f ms (%s)\n", cin, cout, h, w, ms, hipGetErrorString(e));
    }
    static unsigned long long st[2][4096];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(w16_stamps), sizeof st);
    const int nsl = cin / 8, per_item = 3 * nsl + 1;
    for (int g = 0; g < 2; g++) {
        // stamps: [0] = start, then per item: per stage (before the closing wait, after it, after the barrier), then the epilogue's end
        std::vector<double> run(nsl, 0), wait(nsl, 0), bar(nsl, 0);
        double epi = 0; int items = 0;
        for (int it = 2; it < 40 && 1 + (it + 1) * per_item < 4096 && st[g][(it + 1) * per_item]; it++) {
            const unsigned long long *s = &st[g][it * per_item];   // s[0] = end of the previous item's epilogue
            for (int k = 0; k < nsl; k++) {
                run[k] += (double)(s[3 * k + 1] - s[3 * k]);
                wait[k] += (double)(s[3 * k + 2] - s[3 * k + 1]);
                bar[k] += (double)(s[3 * k + 3] - s[3 * k + 2]);
            }
            epi += (double)(s[3 * nsl + 1] - s[3 * nsl]);
            items++;
        }
        if (!items) { printf("group %d: no stamps\n", g); continue; }
        printf("group %d (%d items), cycles per stage: issue phase / vmcnt wait / barrier wait   (ideal: 4096 per stage for the two waves of a SIMD)\n", g, items);
        double tot = epi / items;
        for (int k = 0; k < nsl; k++) {
            printf("   stage %2d: %5.0f %5.0f %5.0f\n", k, run[k] / items, wait[k] / items, bar[k] / items);
            tot += (run[k] + wait[k] + bar[k]) / items;
        }
        printf("   epilogue %5.0f   item %.0f\n", epi / items, tot);
    }
    return 0;
}
