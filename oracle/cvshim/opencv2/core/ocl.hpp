/* oracle/cvshim/opencv2/core/ocl.hpp -- cv::ocl::setUseOpenCL lives in the shim's opencv.hpp
 * (TEST INFRASTRUCTURE ONLY; see that file). */
#include "../opencv.hpp"
