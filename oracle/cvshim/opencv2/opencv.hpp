/*
 * oracle/cvshim/opencv2/opencv.hpp -- the handful of OpenCV 3.0 types and functions that
 * the reference's hot path touches (src/modelHandler.{hpp,cpp}, src/convertRoutine.cpp),
 * restated so that those two reference source files compile UNMODIFIED into oracle/_ref
 * where OpenCV is not installed.
 *
 * TEST INFRASTRUCTURE ONLY (also used as the stand-in cv::Mat when the C++ drop-in adapter
 * in include/w2xc/ is exercised without OpenCV).  This is NOT OpenCV: it is a restatement of
 * the documented CPU semantics of
 *     cv::Mat (refcounted header + ROI), cv::UMat (aliased to Mat), cv::Size, cv::Range,
 *     cv::Point, cv::filter2D (3x3, CV_32F, BORDER_REPLICATE, non-isolated ROI),
 *     cv::add (Mat+Mat, Mat+scalar), cv::max / cv::min (Mat, scalar), cv::scaleAdd,
 *     cv::copyMakeBorder (BORDER_REPLICATE), cv::ocl::setUseOpenCL
 * for CV_32FC1 data only.  Arithmetic notes are at each function.
 */
#ifndef W2XC_CVSHIM_OPENCV_HPP_
#define W2XC_CVSHIM_OPENCV_HPP_

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <memory>
#include <ostream>
#include <vector>

#define W2XC_CVSHIM 1
#define CV_32F 5
#define CV_32FC1 5
#define CV_8U 0

namespace cv {

enum { ACCESS_READ = 1 << 24, ACCESS_WRITE = 1 << 25, ACCESS_RW = 3 << 24 };
enum { BORDER_CONSTANT = 0, BORDER_REPLICATE = 1, BORDER_ISOLATED = 16 };

struct Size {
    int width, height;
    Size() : width(0), height(0) {}
    Size(int w, int h) : width(w), height(h) {}
    bool operator==(const Size &o) const { return width == o.width && height == o.height; }
    bool operator!=(const Size &o) const { return !(*this == o); }
};
struct Point {
    int x, y;
    Point() : x(0), y(0) {}
    Point(int x_, int y_) : x(x_), y(y_) {}
};
struct Range {
    int start, end;
    Range(int s, int e) : start(s), end(e) {}
};

class Mat;
typedef Mat UMat;   /* the reference only round-trips Mat<->UMat (modelHandler.cpp:132-138,153) */

class Mat {
public:
    int rows, cols;
    size_t step;      /* bytes between rows */
    float *data;      /* first element of this (ROI) view */
    /* parent bookkeeping so that filter2D can see beyond a ROI like OpenCV does (locateROI) */
    int wholeRows, wholeCols, ofsY, ofsX;
    std::shared_ptr<float> buf;

    Mat() : rows(0), cols(0), step(0), data(nullptr), wholeRows(0), wholeCols(0), ofsY(0), ofsX(0) {}
    Mat(int r, int c, int type) : Mat() { (void)type; create(r, c, type); }
    Mat(Size s, int type) : Mat() { create(s.height, s.width, type); }
    Mat(Size s, int type, double v) : Mat() { create(s.height, s.width, type); setTo((float)v); }
    /* wrap caller memory (no ownership) */
    Mat(int r, int c, int type, void *ext, size_t step_bytes = 0) : Mat()
    {
        (void)type;
        rows = wholeRows = r; cols = wholeCols = c;
        step = step_bytes ? step_bytes : (size_t)c * sizeof(float);
        data = (float *)ext;
    }

    void create(int r, int c, int type = CV_32FC1)
    {
        (void)type;
        if (data && r == rows && c == cols) return;   /* OpenCV: create() is a no-op on a match */
        rows = wholeRows = r; cols = wholeCols = c; ofsY = ofsX = 0;
        step = (size_t)c * sizeof(float);
        buf = std::shared_ptr<float>(new float[(size_t)r * c + 1], std::default_delete<float[]>());
        data = buf.get();
    }
    void create(Size s, int type = CV_32FC1) { create(s.height, s.width, type); }
    void setTo(float v)
    {
        for (int y = 0; y < rows; y++) std::fill(ptr(y), ptr(y) + cols, v);
    }
    static Mat zeros(int r, int c, int type) { Mat m(r, c, type); m.setTo(0.f); return m; }
    static Mat zeros(Size s, int type) { return zeros(s.height, s.width, type); }

    Size size() const { return Size(cols, rows); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    bool isContinuous() const { return step == (size_t)cols * sizeof(float); }
    int type() const { return CV_32FC1; }
    float *ptr(int y) { return (float *)((char *)data + (size_t)y * step); }
    const float *ptr(int y) const { return (const float *)((const char *)data + (size_t)y * step); }
    template <typename T> T &at(int r, int c) { return ((T *)ptr(r))[c]; }
    template <typename T> const T &at(int r, int c) const { return ((const T *)ptr(r))[c]; }

    Mat operator()(const Range &rr, const Range &cr) const
    {
        Mat m(*this);
        m.rows = rr.end - rr.start; m.cols = cr.end - cr.start;
        m.data = (float *)((char *)data + (size_t)rr.start * step) + cr.start;
        m.ofsY = ofsY + rr.start; m.ofsX = ofsX + cr.start;
        return m;
    }
    Mat rowRange(int a, int b) const { return (*this)(Range(a, b), Range(0, cols)); }
    Mat colRange(int a, int b) const { return (*this)(Range(0, rows), Range(a, b)); }

    void copyTo(Mat &dst) const
    {
        dst.create(rows, cols);
        for (int y = 0; y < rows; y++) std::memmove(dst.ptr(y), ptr(y), (size_t)cols * sizeof(float));
    }
    Mat clone() const { Mat m; copyTo(m); return m; }

    Mat getUMat(int) const { return *this; }
    Mat getMat(int) const { return *this; }
};

inline std::ostream &operator<<(std::ostream &os, const Mat &m)
{
    os << "[";
    for (int y = 0; y < m.rows; y++) {
        for (int x = 0; x < m.cols; x++) os << m.at<float>(y, x) << (x + 1 < m.cols ? ", " : "");
        os << (y + 1 < m.rows ? ";\n " : "");
    }
    return os << "]";
}

/* filter2D, ddepth=-1, 3x3 CV_32F kernel, anchor (-1,-1) = centre, delta, BORDER_REPLICATE.
 * OpenCV semantics: CORRELATION (dst(y,x) = sum K(r,c) * src(y+r-1, x+c-1)); for a small
 * kernel the direct FilterEngine is used: accumulator starts at delta, non-zero taps are
 * visited in row-major order, each tap is mul then add in fp32 (SSE, unfused).  Skipping
 * zero taps is numerically identical to adding 0*x for finite x, so all 9 are visited.
 * Without BORDER_ISOLATED, pixels outside a ROI but inside the parent matrix are read
 * as-is; replicate applies at the parent's edge only. */
inline void filter2D(const Mat &src, Mat &dst, int ddepth, const Mat &kernel, Point anchor = Point(-1, -1),
                     double delta = 0.0, int borderType = BORDER_REPLICATE)
{
    (void)ddepth; (void)anchor;
    assert(kernel.rows == 3 && kernel.cols == 3);
    const bool isolated = (borderType & BORDER_ISOLATED) != 0;
    const int H = src.rows, W = src.cols;
    /* coordinates are expressed in the parent frame unless isolated */
    const int oy = isolated ? 0 : src.ofsY, ox = isolated ? 0 : src.ofsX;
    const int PH = isolated ? H : src.wholeRows, PW = isolated ? W : src.wholeCols;
    const float *base = (const float *)((const char *)src.data - (size_t)oy * src.step) - ox;
    Mat out;
    const bool alias = dst.data == src.data;
    Mat &d = alias ? out : dst;
    d.create(H, W);
    float k[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) k[r * 3 + c] = kernel.at<float>(r, c);
    const float d0 = (float)delta;
    for (int y = 0; y < H; y++) {
        const float *rp[3];
        for (int r = 0; r < 3; r++) {
            int py = std::min(std::max(oy + y + r - 1, 0), PH - 1);
            rp[r] = (const float *)((const char *)base + (size_t)py * src.step);
        }
        float *o = d.ptr(y);
        for (int x = 0; x < W; x++) {
            int xs[3];
            for (int c = 0; c < 3; c++) xs[c] = std::min(std::max(ox + x + c - 1, 0), PW - 1);
            float s = d0;
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) {
                    float p = k[r * 3 + c] * rp[r][xs[c]];
                    s = s + p;
                }
            o[x] = s;
        }
    }
    if (alias) out.copyTo(dst);
}

/* cv::add(a, b, dst): element-wise fp32 add. */
inline void add(const Mat &a, const Mat &b, Mat &dst)
{
    assert(a.rows == b.rows && a.cols == b.cols);
    Mat d = dst;
    d.create(a.rows, a.cols);
    for (int y = 0; y < a.rows; y++) {
        const float *pa = a.ptr(y), *pb = b.ptr(y);
        float *pd = d.ptr(y);
        for (int x = 0; x < a.cols; x++) pd[x] = pa[x] + pb[x];
    }
    dst = d;
}
/* cv::add(a, Scalar(double), dst) on CV_32F: arithm_op narrows a non-integer double scalar to
 * the array depth (depth2 = CV_32F when depth1 == CV_32F), so the add is a + (float)s. */
inline void add(const Mat &a, double s, Mat &dst)
{
    const float fs = (float)s;
    Mat d = dst;
    d.create(a.rows, a.cols);
    for (int y = 0; y < a.rows; y++) {
        const float *pa = a.ptr(y);
        float *pd = d.ptr(y);
        for (int x = 0; x < a.cols; x++) pd[x] = pa[x] + fs;
    }
    dst = d;
}
/* cv::max / cv::min (array, double, dst): scalar converted to the array type. */
inline void max(const Mat &a, double s, Mat &dst)
{
    const float fs = (float)s;
    Mat d = dst;
    d.create(a.rows, a.cols);
    for (int y = 0; y < a.rows; y++) {
        const float *pa = a.ptr(y);
        float *pd = d.ptr(y);
        for (int x = 0; x < a.cols; x++) pd[x] = pa[x] > fs ? pa[x] : fs;   /* SSE maxps(a, s) */
    }
    dst = d;
}
inline void min(const Mat &a, double s, Mat &dst)
{
    const float fs = (float)s;
    Mat d = dst;
    d.create(a.rows, a.cols);
    for (int y = 0; y < a.rows; y++) {
        const float *pa = a.ptr(y);
        float *pd = d.ptr(y);
        for (int x = 0; x < a.cols; x++) pd[x] = pa[x] < fs ? pa[x] : fs;   /* SSE minps(a, s) */
    }
    dst = d;
}
/* cv::scaleAdd(src1, alpha, src2, dst): dst = src1*alpha + src2, alpha narrowed to float
 * for CV_32F (scaleAdd_32f), mul then add. */
inline void scaleAdd(const Mat &a, double alpha, const Mat &b, Mat &dst)
{
    const float fa = (float)alpha;
    Mat d = dst;
    d.create(a.rows, a.cols);
    for (int y = 0; y < a.rows; y++) {
        const float *pa = a.ptr(y), *pb = b.ptr(y);
        float *pd = d.ptr(y);
        for (int x = 0; x < a.cols; x++) {
            float t = pa[x] * fa;
            pd[x] = t + pb[x];
        }
    }
    dst = d;
}

/* cv::copyMakeBorder(..., BORDER_REPLICATE).  (OpenCV also peeks beyond a ROI here unless
 * BORDER_ISOLATED; the reference only ever passes whole planes, convertRoutine.cpp:35,96.) */
inline void copyMakeBorder(const Mat &src, Mat &dst, int top, int bottom, int left, int right, int borderType)
{
    assert((borderType & ~BORDER_ISOLATED) == BORDER_REPLICATE);
    Mat d;
    d.create(src.rows + top + bottom, src.cols + left + right);
    for (int y = 0; y < d.rows; y++) {
        const float *s = src.ptr(std::min(std::max(y - top, 0), src.rows - 1));
        float *o = d.ptr(y);
        for (int x = 0; x < d.cols; x++) o[x] = s[std::min(std::max(x - left, 0), src.cols - 1)];
    }
    dst = d;
}

namespace ocl {
inline void setUseOpenCL(bool) {}
}

}  // namespace cv

#endif
