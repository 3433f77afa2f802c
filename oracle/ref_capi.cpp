/*
 * oracle/ref_capi.cpp -- extern "C" handle around the REFERENCE'S OWN hot-path code
 * (w2xc::modelUtility::generateModelFromJSON, w2xc::convertWithModels, w2xc::Model::filter)
 * as compiled from /root/reference/src/{modelHandler,convertRoutine}.cpp against the OpenCV
 * shim in oracle/cvshim.  TEST INFRASTRUCTURE ONLY -- built into oracle/_ref/libw2xc_ref.so
 * by oracle/Makefile; used to pin oracle/w2xc_oracle.c and as a second checker in tests.
 *
 * No reference source is copied: the two .cpp files are compiled where they lie.
 */
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "modelHandler.hpp"     /* /root/reference/src, via -I */
#include "convertRoutine.hpp"

namespace {
struct RefModels {
    std::vector<std::unique_ptr<w2xc::Model> > models;
};
/* the reference prints progress on std::cout (convertRoutine.cpp:67,133) */
struct MuteCout {
    std::streambuf *old;
    std::ostringstream sink;
    MuteCout() : old(std::cout.rdbuf(sink.rdbuf())) {}
    ~MuteCout() { std::cout.rdbuf(old); }
};
}  // namespace

extern "C" {

void *w2xc_ref_load(const char *json_path)
{
    RefModels *m = new RefModels();
    if (!w2xc::modelUtility::generateModelFromJSON(json_path, m->models)) {
        delete m;
        return nullptr;
    }
    return m;
}

void w2xc_ref_free(void *h) { delete static_cast<RefModels *>(h); }

int w2xc_ref_nlayers(void *h) { return (int)static_cast<RefModels *>(h)->models.size(); }
int w2xc_ref_nin(void *h, int l) { return static_cast<RefModels *>(h)->models[l]->getNInputPlanes(); }
int w2xc_ref_nout(void *h, int l) { return static_cast<RefModels *>(h)->models[l]->getNOutputPlanes(); }

/* convertWithModels (convertRoutine.cpp:21).  in/out: contiguous w*h floats.  block_w/h <= 0
 * keeps the singleton's current block size (default 512x512, modelHandler.hpp:99). */
int w2xc_ref_convert(void *h, const float *in, int w, int hgt, float *out, int block_splitting, int njob,
                     int block_w, int block_h)
{
    RefModels *m = static_cast<RefModels *>(h);
    MuteCout mute;
    w2xc::modelUtility &u = w2xc::modelUtility::getInstance();
    if (njob > 0) u.setNumberOfJobs(njob);
    if (block_w > 0 && block_h > 0) u.setBlockSize(cv::Size(block_w, block_h));
    cv::Mat src(hgt, w, CV_32FC1, const_cast<float *>(in));
    cv::Mat dst;
    bool ok = w2xc::convertWithModels(src, dst, m->models, block_splitting != 0);
    if (!ok || dst.rows != hgt || dst.cols != w) return -1;
    for (int y = 0; y < hgt; y++)
        for (int x = 0; x < w; x++) out[(size_t)y * w + x] = dst.at<float>(y, x);
    return 0;
}

/* Model::filter (modelHandler.cpp:26).  in: [n_in][h][w] planar, out: [nout][h][w]. */
int w2xc_ref_filter(void *h, int layer, int n_in, const float *in, int w, int hgt, float *out, int njob)
{
    RefModels *m = static_cast<RefModels *>(h);
    MuteCout mute;
    std::ostringstream errsink;
    std::streambuf *olderr = std::cerr.rdbuf(errsink.rdbuf());
    if (njob > 0) w2xc::modelUtility::getInstance().setNumberOfJobs(njob);
    std::vector<cv::Mat> ip, op;
    for (int i = 0; i < n_in; i++)
        ip.push_back(cv::Mat(hgt, w, CV_32FC1, const_cast<float *>(in) + (size_t)i * w * hgt));
    bool ok = m->models[layer]->filter(ip, op);
    std::cerr.rdbuf(olderr);
    if (!ok) return -1;
    for (size_t o = 0; o < op.size(); o++)
        for (int y = 0; y < hgt; y++)
            for (int x = 0; x < w; x++) out[(o * hgt + y) * w + x] = op[o].at<float>(y, x);
    return 0;
}

int w2xc_ref_get_jobs(void) { return w2xc::modelUtility::getInstance().getNumberOfJobs(); }
int w2xc_ref_set_jobs(int n) { return w2xc::modelUtility::getInstance().setNumberOfJobs(n) ? 0 : -1; }

}  // extern "C"
