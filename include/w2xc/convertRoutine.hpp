/*
 * include/w2xc/convertRoutine.hpp -- drop-in for /root/reference/src/convertRoutine.hpp:
 *
 *   bool w2xc::convertWithModels(cv::Mat &inputPlane, cv::Mat &outputPlane,
 *                                std::vector<std::unique_ptr<Model> > &models,
 *                                bool blockSplitting = true);          (convertRoutine.hpp:25-28)
 *
 * outputPlane is (re)allocated to the input size, CV_32FC1, exactly like the reference's final
 * copyTo (convertRoutine.cpp:46,78); inputPlane may be a strided ROI; in/out may alias.  The pad-7
 * / block-split / crop logic of convertRoutine.cpp:21-169 is replaced by the GPU band loop behind
 * w2xc_convert_plane (identical math per output pixel -- SURVEY I1/I2), so `blockSplitting`
 * does not change the result.  Returns false (and prints to std::cerr) on any failure instead of
 * calling std::exit (convertRoutine.cpp:68-70).
 */
#ifndef W2XC_HIP_CONVERTROUTINE_HPP_
#define W2XC_HIP_CONVERTROUTINE_HPP_
#define CONVERTROUTINE_HPP_   /* shadow the reference's include guard */

#include "modelHandler.hpp"
#include <memory>
#include <vector>

namespace w2xc {

inline bool convertWithModels(cv::Mat &inputPlane, cv::Mat &outputPlane,
                              std::vector<std::unique_ptr<Model> > &models, bool blockSplitting = true)
{
    if (models.empty() || inputPlane.empty()) {
        std::cerr << "w2xc::convertWithModels : empty model list or input plane" << std::endl;
        return false;
    }
    /* the common case: `models` is exactly one loaded file, in order -> its weights are already
     * resident on the GPUs.  Any other vector of layers gets a temporary container. */
    std::shared_ptr<detail::ModelHandle> set = models[0]->set;
    bool whole = set && (int)models.size() == w2xc_model_layers(set->m);
    for (size_t l = 0; whole && l < models.size(); l++)
        whole = models[l]->set == set && models[l]->layer == (int)l;
    if (!whole) {
        const int n = (int)models.size();
        std::vector<int> nin(n), nout(n);
        std::vector<std::vector<float> > w(n);
        std::vector<std::vector<double> > b(n);
        std::vector<const float *> wp(n);
        std::vector<const double *> bp(n);
        for (int l = 0; l < n; l++) {
            if (!models[l]->set) return false;
            nin[l] = models[l]->getNInputPlanes();
            nout[l] = models[l]->getNOutputPlanes();
            w[l].resize((size_t)nin[l] * nout[l] * 9);
            b[l].resize(nout[l]);
            w2xc_model_get_layer(models[l]->set->m, models[l]->layer, w[l].data(), b[l].data());
            wp[l] = w[l].data();
            bp[l] = b[l].data();
        }
        w2xc_model *m = nullptr;
        if (w2xc_model_from_arrays(n, nin.data(), nout.data(), wp.data(), bp.data(), &m) != W2XC_OK) return false;
        set = std::make_shared<detail::ModelHandle>(m);
    }
    /* the reference (re)allocates the output to the input's size and type (copyTo, :46 / :78).  The engine writes the rows
     * straight into it -- no staging Mat, no extra 33 MB copy per 4K plane; the C ABI is safe for in == out (in-place). */
    const int rows = inputPlane.rows, cols = inputPlane.cols;
    const float *src = reinterpret_cast<const float *>(inputPlane.data);
    const size_t src_step = (size_t)inputPlane.step;
    cv::Mat keep = inputPlane;           /* outputPlane may BE inputPlane: create() below must not free the rows being read */
    outputPlane.create(rows, cols, CV_32FC1);
    const int rc = w2xc_convert_plane(set->m, src, src_step, cols, rows, reinterpret_cast<float *>(outputPlane.data),
                                      (size_t)outputPlane.step, blockSplitting ? 1 : 0, nullptr);
    if (rc != W2XC_OK) {
        std::cerr << "w2xc::convertWithModels : " << w2xc_last_error() << std::endl;
        return false;
    }
    return true;
}

}  // namespace w2xc

#endif /* W2XC_HIP_CONVERTROUTINE_HPP_ */
