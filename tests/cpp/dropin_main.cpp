// tests/cpp/dropin_main.cpp -- ONE caller, written against the reference's public API only
// (w2xc::modelUtility, w2xc::convertWithModels, w2xc::Model::filter; the call sequence of
// /root/reference/src/main.cpp:79-98 and src/test.cpp:72-85), compiled TWICE by tests/cpp/Makefile:
//   dropin_ref : -I/root/reference/src + the reference's modelHandler.cpp/convertRoutine.cpp (OpenCV shim)
//   dropin_hip : -Iinclude/w2xc      + libw2xc_hip.so                                   (same shim for cv::Mat)
// usage: dropin_xxx convert model.json in.f32 w h out.f32 [block_splitting]
//        dropin_xxx filter  model.json layer in.f32 nplanes w h out.f32
//        dropin_xxx chain   model.json in.f32 w h out.f32          (every layer through Model::filter, test.cpp:72-85)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <vector>

#include "modelHandler.hpp"
#include "convertRoutine.hpp"

static bool read_all(const char *path, std::vector<float> &v)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    const size_t n = fread(v.data(), sizeof(float), v.size(), f);
    fclose(f);
    return n == v.size();
}
static bool write_all(const char *path, const float *p, size_t n)
{
    FILE *f = fopen(path, "wb");
    if (!f) return false;
    const size_t k = fwrite(p, sizeof(float), n, f);
    fclose(f);
    return k == n;
}

#ifdef W2XC_HIP_MODEL_HANDLER_HPP_
// the adapter also answers to the later upstream namespace name (BASELINE.json's north_star: w2xconv::Model)
#include <type_traits>
static_assert(std::is_same<w2xconv::Model, w2xc::Model>::value && std::is_same<w2xconv::modelUtility, w2xc::modelUtility>::value,
              "namespace w2xconv = w2xc");
static bool (*const w2xconv_entry)(cv::Mat &, cv::Mat &, std::vector<std::unique_ptr<w2xconv::Model> > &, bool) = &w2xconv::convertWithModels;
#endif

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    // the reference prints progress lines on stdout; mute them for the lifetime of main
    struct Mute {
        std::ostringstream sink;
        std::streambuf *old;
        Mute() : old(std::cout.rdbuf(sink.rdbuf())) {}
        ~Mute() { std::cout.rdbuf(old); }
    } mute;
    std::vector<std::unique_ptr<w2xc::Model> > models;
    w2xc::modelUtility::getInstance().setNumberOfJobs(4);                       // main.cpp:79
    int rc = 0;
    if (!strcmp(argv[1], "convert") && argc >= 7) {
        if (!w2xc::modelUtility::generateModelFromJSON(argv[2], models)) return 3;   // main.cpp:88
        const int w = atoi(argv[4]), h = atoi(argv[5]);
        std::vector<float> in((size_t)w * h);
        if (!read_all(argv[3], in)) return 4;
        cv::Mat src(h, w, CV_32FC1, in.data());
        cv::Mat dst;
        const bool split = argc > 7 ? atoi(argv[7]) != 0 : true;
        if (!w2xc::convertWithModels(src, dst, models, split)) rc = 1;               // main.cpp:96
        else {
            std::vector<float> out((size_t)w * h);
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++) out[(size_t)y * w + x] = dst.at<float>(y, x);
            if (dst.rows != h || dst.cols != w || !write_all(argv[6], out.data(), out.size())) rc = 5;
        }
    } else if (!strcmp(argv[1], "filter") && argc >= 9) {
        if (!w2xc::modelUtility::generateModelFromJSON(argv[2], models)) return 3;
        const int layer = atoi(argv[3]), np = atoi(argv[5]), w = atoi(argv[6]), h = atoi(argv[7]);
        std::vector<float> in((size_t)np * w * h);
        if (!read_all(argv[4], in)) return 4;
        std::vector<cv::Mat> ip, op;
        for (int i = 0; i < np; i++) ip.push_back(cv::Mat(h, w, CV_32FC1, in.data() + (size_t)i * w * h));
        if (!models.at(layer)->filter(ip, op)) rc = 1;                                // test.cpp:76
        else {
            std::vector<float> out(op.size() * (size_t)w * h);
            for (size_t o = 0; o < op.size(); o++)
                for (int y = 0; y < h; y++)
                    for (int x = 0; x < w; x++) out[(o * h + y) * w + x] = op[o].at<float>(y, x);
            if (!write_all(argv[8], out.data(), out.size())) rc = 5;
        }
    } else if (!strcmp(argv[1], "chain") && argc >= 7) {
        // the reference's test.cpp:72-85 pattern: filter() layer after layer, each call's output vector handed to the next
        if (!w2xc::modelUtility::generateModelFromJSON(argv[2], models)) return 3;
        const int w = atoi(argv[4]), h = atoi(argv[5]);
        std::vector<float> in((size_t)w * h);
        if (!read_all(argv[3], in)) return 4;
        std::vector<cv::Mat> cur, nxt;
        cur.push_back(cv::Mat(h, w, CV_32FC1, in.data()));
        for (size_t l = 0; l < models.size() && rc == 0; l++) {
            if (!models[l]->filter(cur, nxt)) rc = 1;
            else cur = nxt;
        }
        if (rc == 0) {
            std::vector<float> out(cur.size() * (size_t)w * h);
            for (size_t o = 0; o < cur.size(); o++)
                for (int y = 0; y < h; y++)
                    for (int x = 0; x < w; x++) out[(o * h + y) * w + x] = cur[o].at<float>(y, x);
            if (!write_all(argv[6], out.data(), out.size())) rc = 5;
        }
    } else {
        rc = 2;
    }
    return rc;
}
