/*
 * include/w2xc/modelHandler.hpp -- drop-in for /root/reference/src/modelHandler.hpp.
 *
 * Same namespace, class names, public member signatures and error behaviour as the reference's
 * w2xc::Model / w2xc::modelUtility (modelHandler.hpp:24-113), implemented on the C ABI of
 * libw2xc_hip.so (include/w2xc_hip.h).  A caller such as the reference's main.cpp compiles
 * unchanged with `-I<repo>/include/w2xc` in place of `-I<reference>/src` and links
 * `-lw2xc_hip` in place of modelHandler.o / convertRoutine.o.
 *
 * Header-only.  cv::Mat comes from the caller's OpenCV (>= 3.0); the test build uses the small
 * stand-in at oracle/cvshim.  picojson is only needed for the Model(picojson::object&)
 * constructor and is included when it is on the include path, exactly like the reference does.
 *
 * Differences from the reference, by design:
 *   - nothing in here calls std::exit: constructor / filter failures surface as `false` from the
 *     calling API (a failed Model constructor leaves an invalid model whose filter() fails);
 *   - setNumberOfJobs() sizes the HOST STAGING threads (the copies between the caller's planes and the pinned rings
 *     that feed the DMA engines) instead of a pool of convolution threads -- the layer runs on the GPU(s); the block
 *     size does not change results and is not used for tiling;
 *   - Model::filter called in a chain (test.cpp:72-85) re-uploads its input planes on every call unless the process
 *     sets W2XC_FILTER_RESIDENT=1, which promises that planes handed back unchanged may be taken from the device copy.
 */
#ifndef W2XC_HIP_MODEL_HANDLER_HPP_
#define W2XC_HIP_MODEL_HANDLER_HPP_
#define MODEL_HANDLER_HPP_   /* shadow the reference's include guard */

#include <opencv2/opencv.hpp>
#if defined(__has_include)
#if __has_include(<opencv2/core/ocl.hpp>)
#include <opencv2/core/ocl.hpp>
#endif
#if __has_include("picojson.h")
#include "picojson.h"
#define W2XC_HIP_HAVE_PICOJSON 1
#endif
#endif
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "../w2xc_hip.h"

namespace w2xc {

namespace detail {
struct ModelHandle {
    w2xc_model *m = nullptr;
    explicit ModelHandle(w2xc_model *p) : m(p) {}
    ~ModelHandle() { if (m) w2xc_model_free(m); }
    ModelHandle(const ModelHandle &) = delete;
    ModelHandle &operator=(const ModelHandle &) = delete;
};
}  // namespace detail

class Model {
private:
    std::shared_ptr<detail::ModelHandle> set;   // the loaded file (all layers), shared by its Models
    int layer = 0;
    int nInputPlanes = 0;
    int nOutputPlanes = 0;

    friend class modelUtility;
    friend bool convertWithModels(cv::Mat &, cv::Mat &, std::vector<std::unique_ptr<Model> > &, bool);

    Model(std::shared_ptr<detail::ModelHandle> s, int l) : set(std::move(s)), layer(l)
    {
        nInputPlanes = w2xc_model_nin(set->m, layer);
        nOutputPlanes = w2xc_model_nout(set->m, layer);
    }

public:
#ifdef W2XC_HIP_HAVE_PICOJSON
    /* One layer from its JSON object (the contract of modelHandler.hpp:48-71 + modelHandler.cpp:74-115: weight[o][i][kh][kw]
     * doubles narrowed to float, kernel index o * nInputPlanes + i, biases kept as doubles).  Flat, bounds-checked walk:
     * a malformed object yields an INVALID model (filter() returns false) instead of the reference's exit(-1) / UB. */
    Model(picojson::object &jsonObj)
    {
        auto number = [&](const char *key, double &v) {
            auto it = jsonObj.find(key);
            if (it == jsonObj.end() || !it->second.is<double>()) return false;
            v = it->second.get<double>();
            return true;
        };
        double nin = 0, nout = 0, kw = 0, kh = 0;
        if (!number("nInputPlane", nin) || !number("nOutputPlane", nout) || !number("kW", kw) || !number("kH", kh) ||
            !(nin >= 1 && nin <= 4096 && nout >= 1 && nout <= 4096)) {
            std::cerr << "Error : Model-Constructor : \nbad layer object." << std::endl;
            return;
        }
        nInputPlanes = static_cast<int>(nin);
        nOutputPlanes = static_cast<int>(nout);
        if (static_cast<int>(kw) != static_cast<int>(kh) || static_cast<int>(kw) != 3) {
            std::cerr << "Error : Model-Constructor : \nkernel in model is not square (or not 3x3).\n" << std::endl;
            return;   /* invalid model: filter() will fail (the reference exit(-1)s here) */
        }
        auto wit = jsonObj.find("weight"), bit = jsonObj.find("bias");
        if (wit == jsonObj.end() || bit == jsonObj.end() || !wit->second.is<picojson::array>() || !bit->second.is<picojson::array>()) return;
        const picojson::array &W = wit->second.get<picojson::array>(), &B = bit->second.get<picojson::array>();
        if ((int)W.size() != nOutputPlanes || (int)B.size() < nOutputPlanes) return;
        std::vector<float> taps((size_t)nInputPlanes * nOutputPlanes * 9);
        std::vector<double> bias(nOutputPlanes);
        for (size_t k = 0; k < taps.size(); k++) {          /* k = ((o * nIn + i) * 3 + r) * 3 + c */
            const size_t c = k % 3, r = (k / 3) % 3, i = (k / 9) % (size_t)nInputPlanes, o = k / (9 * (size_t)nInputPlanes);
            const picojson::value *v = &W[o];
            for (size_t idx : {i, r, c}) {
                if (!v->is<picojson::array>() || v->get<picojson::array>().size() <= idx) return;
                v = &v->get<picojson::array>()[idx];
            }
            if (!v->is<double>()) return;
            taps[k] = static_cast<float>(v->get<double>());
        }
        for (int o = 0; o < nOutputPlanes; o++) {
            if (!B[o].is<double>()) return;
            bias[o] = B[o].get<double>();
        }
        const float *wp = taps.data();
        const double *bp = bias.data();
        w2xc_model *m = nullptr;
        if (w2xc_model_from_arrays(1, &nInputPlanes, &nOutputPlanes, &wp, &bp, &m) == W2XC_OK)
            set = std::make_shared<detail::ModelHandle>(m);
    }
#endif
    ~Model() {}

    /* for debugging -- modelHandler.cpp:229-242 */
    void printWeightMatrix()
    {
        if (!set) return;
        std::vector<float> w((size_t)nInputPlanes * nOutputPlanes * 9);
        w2xc_model_get_layer(set->m, layer, w.data(), nullptr);
        for (size_t k = 0; k < w.size(); k += 9) {
            cv::Mat m(3, 3, CV_32FC1, &w[k]);
            std::cout << m << std::endl;
        }
    }
    void printBiases()
    {
        if (!set) return;
        std::vector<double> b(nOutputPlanes);
        w2xc_model_get_layer(set->m, layer, nullptr, b.data());
        for (double v : b) std::cout << v << std::endl;
    }

    int getNInputPlanes() { return nInputPlanes; }
    int getNOutputPlanes() { return nOutputPlanes; }

    /* bool filter(inputPlanes, outputPlanes) -- modelHandler.cpp:26-72: same-size planes,
     * per-layer BORDER_REPLICATE, bias, LeakyReLU(0.1); false on a plane-count mismatch. */
    bool filter(std::vector<cv::Mat> &inputPlanes, std::vector<cv::Mat> &outputPlanes)
    {
        if ((int)inputPlanes.size() != nInputPlanes) {
            std::cerr << "Error : Model-filter : \n"
                         "number of input planes mismatch." << std::endl;
            std::cerr << inputPlanes.size() << "," << nInputPlanes << std::endl;
            return false;
        }
        if (!set) return false;
        const int rows = inputPlanes[0].rows, cols = inputPlanes[0].cols;
        std::vector<cv::Mat> in(inputPlanes.size());
        std::vector<const float *> ip(inputPlanes.size());
        size_t istep = (size_t)cols * sizeof(float);
        for (size_t i = 0; i < inputPlanes.size(); i++) {
            /* the C ABI takes one stride for all planes: compact any plane that differs */
            if (inputPlanes[i].rows != rows || inputPlanes[i].cols != cols) return false;
            if (inputPlanes[i].isContinuous()) in[i] = inputPlanes[i];
            else inputPlanes[i].copyTo(in[i]);
            ip[i] = reinterpret_cast<const float *>(in[i].data);
        }
        std::vector<cv::Mat> out(nOutputPlanes);
        std::vector<float *> op(nOutputPlanes);
        for (int o = 0; o < nOutputPlanes; o++) {
            out[o] = cv::Mat::zeros(rows, cols, CV_32FC1);      /* :37-40 */
            op[o] = reinterpret_cast<float *>(out[o].data);
        }
        const int rc = w2xc_layer_filter(set->m, layer, (int)ip.size(), ip.data(), istep, cols, rows, op.data(),
                                         (size_t)cols * sizeof(float), nullptr);
        if (rc != W2XC_OK) {
            std::cerr << "Error : Model-filter : " << w2xc_last_error() << std::endl;
            return false;
        }
        outputPlanes.swap(out);
        return true;
    }
};

class modelUtility {
private:
    modelUtility() {}

public:
    /* modelHandler.cpp:170-197 */
    static bool generateModelFromJSON(const std::string &fileName, std::vector<std::unique_ptr<Model> > &models)
    {
        w2xc_model *m = nullptr;
        if (w2xc_model_load_json(fileName.c_str(), &m) != W2XC_OK) return false;   /* message already on stderr */
        std::shared_ptr<detail::ModelHandle> set = std::make_shared<detail::ModelHandle>(m);
        const int n = w2xc_model_layers(m);
        for (int l = 0; l < n; l++) models.push_back(std::unique_ptr<Model>(new Model(set, l)));
        return true;
    }
    static modelUtility &getInstance()
    {
        static modelUtility instance;
        return instance;
    }
    bool setNumberOfJobs(int setNJob) { return w2xc_set_jobs(setNJob) == W2XC_OK; }
    int getNumberOfJobs() { return w2xc_get_jobs(); }
    bool setBlockSize(cv::Size size) { return w2xc_set_block_size(size.width, size.height) == W2XC_OK; }
    bool setBlockSizeExp2Square(int exp) { return w2xc_set_block_size_exp2(exp) == W2XC_OK; }
    cv::Size getBlockSize()
    {
        int w = 0, h = 0;
        w2xc_get_block_size(&w, &h);
        return cv::Size(w, h);
    }
};

}  // namespace w2xc

// Later upstream releases renamed the namespace (w2xconv::Model, BASELINE.json's north_star uses that name); this snapshot's is
// w2xc (/root/reference/src/modelHandler.hpp:22, convertRoutine.hpp:20).  Both spellings name the same classes and functions.
namespace w2xconv = w2xc;

#endif /* W2XC_HIP_MODEL_HANDLER_HPP_ */
