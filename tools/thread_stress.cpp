// tools/thread_stress.cpp -- concurrent callers of the C ABI (include/w2xc_hip.h), as a C++ program so that the engine's host code can run
// under ThreadSanitizer (`make -C waifu2x-converter-cpp_amd/csrc tsan` builds libw2xc_hip_tsan.so + this file with -fsanitize=thread;
// tests/test_gpu_threads.py builds and runs the plain form).
//
//   thread_stress [threads = 4] [rounds = 6]
//
// Phases, every result compared bit for bit with the same call made alone:
//   1  all threads call w2xc_convert_plane on ONE model, mixed plane sizes (the context serialises them; pipe buffers grow and shrink)
//   2  half of the threads on model A, half on model B, same device; every third call is w2xc_convert_plane_nn2x
//   3  w2xc_convert_plane with host_units = 3 from two threads (unit fan-out + shared copy pool inside concurrent calls)
//   4  Model::filter chains with filter_resident on one model from all threads
//   5  w2xc_process_image_u8 with (A, B) and (B, A) in opposite roles (lock order)
//   6  modelUtility knobs + w2xc_set_default_opts hammered by one thread while the others convert
// The contract being exercised: SURVEY 8b "engine internally thread-safe per Model"; the reference's own threading is
// /root/reference/src/modelHandler.cpp:42-69 (workers on disjoint planes) and the unguarded singleton of :163-168.
#include <w2xc_hip.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

namespace {

struct Rng {   // xorshift64*: the weights only have to be the same in every thread, not good
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 1) {}
    double uni() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return (double)((s * 0x2545F4914F6CDD1Dull) >> 11) / 9007199254740992.0; }
    double gauss() { double a = 0; for (int i = 0; i < 12; i++) a += uni(); return a - 6.0; }
};

w2xc_model *make_model(const std::vector<int> &planes, uint64_t seed)
{
    const int n = (int)planes.size() - 1;
    std::vector<int> nin(n), nout(n);
    std::vector<std::vector<float>> w(n);
    std::vector<std::vector<double>> b(n);
    std::vector<const float *> wp(n);
    std::vector<const double *> bp(n);
    Rng r(seed);
    for (int l = 0; l < n; l++) {
        nin[l] = planes[l]; nout[l] = planes[l + 1];
        w[l].resize((size_t)nin[l] * nout[l] * 9);
        b[l].resize(nout[l]);
        const double sd = std::sqrt(2.0 / (9.0 * nin[l]));   // He-style, like tools/gen_model.py
        for (auto &v : w[l]) v = (float)(sd * r.gauss());
        for (auto &v : b[l]) v = 0.05 * r.gauss();
        wp[l] = w[l].data(); bp[l] = b[l].data();
    }
    w2xc_model *m = nullptr;
    if (w2xc_model_from_arrays(n, nin.data(), nout.data(), wp.data(), bp.data(), &m) != W2XC_OK) { fprintf(stderr, "model: %s\n", w2xc_last_error()); exit(2); }
    return m;
}

std::vector<float> plane(int h, int w, uint64_t seed)
{
    std::vector<float> p((size_t)h * w);
    Rng r(seed);
    for (auto &v : p) v = (float)r.uni();
    return p;
}

std::atomic<int> g_bad{0};
void check(bool ok, const char *what, int t, int it)
{
    if (!ok) { g_bad++; fprintf(stderr, "MISMATCH %s thread %d call %d: %s\n", what, t, it, w2xc_last_error()); }
}

void run(int n, const std::function<void(int)> &fn)
{
    std::vector<std::thread> th;
    for (int t = 0; t < n; t++) th.emplace_back(fn, t);
    for (auto &x : th) x.join();
}

const int SZ[5][2] = {{96, 160}, {301, 423}, {64, 64}, {257, 130}, {40, 500}};

std::vector<float> convert(w2xc_model *m, const std::vector<float> &x, int h, int w, const w2xc_opts *o, bool nn2x = false)
{
    const int up = nn2x ? 2 : 1;
    std::vector<float> out((size_t)h * up * w * up, -1.0f);
    const int rc = nn2x ? w2xc_convert_plane_nn2x(m, x.data(), (size_t)w * 4, w, h, out.data(), (size_t)w * up * 4, o)
                        : w2xc_convert_plane(m, x.data(), (size_t)w * 4, w, h, out.data(), (size_t)w * 4, 1, o);
    if (rc != W2XC_OK) { fprintf(stderr, "convert failed: %s\n", w2xc_last_error()); g_bad++; }
    return out;
}

}  // namespace

int main(int argc, char **argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 4, R = argc > 2 ? atoi(argv[2]) : 6;
    if (w2xc_device_count() < 1) { fprintf(stderr, "no HIP device (libw2xc_hip has no CPU fallback)\n"); return 3; }
    w2xc_model *A = make_model({1, 32, 32, 64, 64, 128, 128, 1}, 102), *B = make_model({1, 32, 32, 32, 32, 32, 32, 1}, 7);
    std::vector<std::vector<float>> xs;
    for (int i = 0; i < 5; i++) xs.push_back(plane(SZ[i][0], SZ[i][1], 100 + i));
    std::vector<std::vector<float>> wantA, wantB, wantA2;
    for (int i = 0; i < 5; i++) {
        wantA.push_back(convert(A, xs[i], SZ[i][0], SZ[i][1], nullptr));
        wantB.push_back(convert(B, xs[i], SZ[i][0], SZ[i][1], nullptr));
        wantA2.push_back(convert(A, xs[i], SZ[i][0], SZ[i][1], nullptr, true));
    }
    // 1
    run(T, [&](int t) {
        for (int it = 0; it < 3 * R; it++) {
            const int i = (t * 3 + it) % 5;
            check(convert(A, xs[i], SZ[i][0], SZ[i][1], nullptr) == wantA[i], "one model", t, it);
        }
    });
    printf("phase 1 done (%d bad)\n", g_bad.load());
    // 2
    run(T, [&](int t) {
        w2xc_model *m = t < T / 2 ? A : B;
        for (int it = 0; it < 3 * R; it++) {
            const int i = (t + 2 * it) % 5;
            if (it % 3 == 0 && m == A) check(convert(A, xs[i], SZ[i][0], SZ[i][1], nullptr, true) == wantA2[i], "nn2x", t, it);
            else check(convert(m, xs[i], SZ[i][0], SZ[i][1], nullptr) == (m == A ? wantA[i] : wantB[i]), "two models", t, it);
        }
    });
    printf("phase 2 done (%d bad)\n", g_bad.load());
    // 3
    run(2, [&](int t) {
        w2xc_opts o;
        w2xc_opts_init(&o);
        o.host_units = 3;
        for (int it = 0; it < R; it++) check(convert(A, xs[1], SZ[1][0], SZ[1][1], &o) == wantA[1], "units", t, it);
    });
    printf("phase 3 done (%d bad)\n", g_bad.load());
    // 4: layers 1..3 of B through Model::filter by hand, planes handed straight back (filter_resident)
    {
        const int h = 61, w = 83;
        const std::vector<float> x = plane(h, w, 4);
        auto chain = [&](int t, int it, std::vector<float> *res) {
            w2xc_opts o;
            w2xc_opts_init(&o);
            o.filter_resident = 1;
            std::vector<std::vector<float>> cur(1, x), nxt;
            for (int l = 0; l < 3; l++) {
                const int ni = w2xc_model_nin(B, l), no = w2xc_model_nout(B, l);
                nxt.assign(no, std::vector<float>((size_t)h * w));
                std::vector<const float *> ip(ni);
                std::vector<float *> op(no);
                for (int i = 0; i < ni; i++) ip[i] = cur[i].data();
                for (int i = 0; i < no; i++) op[i] = nxt[i].data();
                if (w2xc_layer_filter(B, l, ni, ip.data(), (size_t)w * 4, w, h, op.data(), (size_t)w * 4, &o) != W2XC_OK) { check(false, "filter rc", t, it); return; }
                cur.swap(nxt);
            }
            res->clear();
            for (auto &p : cur) res->insert(res->end(), p.begin(), p.end());
        };
        std::vector<float> want;
        chain(-1, 0, &want);
        run(T, [&](int t) {
            for (int it = 0; it < R; it++) {
                std::vector<float> got;
                chain(t, it, &got);
                check(got == want, "filter chain", t, it);
            }
        });
    }
    printf("phase 4 done (%d bad)\n", g_bad.load());
    // 5
    {
        const int h = 40, w = 56;
        std::vector<unsigned char> img((size_t)h * w * 3);
        Rng r(3);
        for (auto &v : img) v = (unsigned char)(r.uni() * 256.0);
        auto proc = [&](w2xc_model *noise, w2xc_model *scale) {
            std::vector<unsigned char> out((size_t)4 * h * w * 3);
            if (w2xc_process_image_u8(noise, scale, img.data(), (size_t)w * 3, w, h, out.data(), (size_t)2 * w * 3, 1, nullptr) != W2XC_OK) { g_bad++; fprintf(stderr, "process_image: %s\n", w2xc_last_error()); }
            return out;
        };
        const auto wantAB = proc(A, B), wantBA = proc(B, A);
        run(2, [&](int t) {
            for (int it = 0; it < 4 * R; it++) check(proc(t ? B : A, t ? A : B) == (t ? wantBA : wantAB), "opposite roles", t, it);
        });
    }
    printf("phase 5 done (%d bad)\n", g_bad.load());
    // 6
    run(T < 2 ? 2 : T, [&](int t) {
        for (int it = 0; it < 5 * R; it++) {
            if (t == 0) {
                w2xc_set_jobs(1 + it % 6);
                w2xc_set_block_size(256 + it, 256);
                w2xc_set_default_opts(nullptr);
                int bw, bh;
                w2xc_get_block_size(&bw, &bh);
                (void)w2xc_get_jobs();
            } else {
                check(convert(B, xs[2], SZ[2][0], SZ[2][1], nullptr) == wantB[2], "knobs", t, it);
            }
        }
    });
    w2xc_set_jobs(4);
    w2xc_set_block_size(512, 512);
    printf("phase 6 done (%d bad)\n", g_bad.load());
    w2xc_model_free(A);
    w2xc_model_free(B);
    if (g_bad.load()) { printf("thread_stress: %d MISMATCHES\n", g_bad.load()); return 1; }
    printf("thread_stress: ok (%d threads, %d rounds)\n", T, R);
    return 0;
}
