// cvlite.h -- the handful of OpenCV types that the reference's ORBextractor.h / ORBmatcher.h and
// their callers (src/Frame.cc:591-597, src/Tracking.cc:816-822) use on this path, for machines
// without OpenCV (neither the dev container nor the GPU box has it).  Same names, namespace `cv`,
// same member names and the same binary layout for KeyPoint (28 bytes) so that code written
// against OpenCV 2.4 compiles unchanged.  When a real OpenCV is available, define
// ORBHIP_USE_OPENCV and this header forwards to <opencv2/core/core.hpp> instead.
#ifndef ORBHIP_CVLITE_H
#define ORBHIP_CVLITE_H

#ifdef ORBHIP_USE_OPENCV
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#else

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

typedef unsigned char uchar;

#define CV_8U 0
#define CV_32F 5
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_PI 3.1415926535897932384626433832795

inline int cvRound(double v) { return (int)lrint(v); }   // SSE2 cvtsd2si: round half to even
inline int cvFloor(double v) { int i = (int)v; return i - (i > v); }
inline int cvCeil(double v) { int i = (int)v; return i + (i < v); }

namespace cv {

template <typename T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T _x, T _y) : x(_x), y(_y) {}
    Point_ &operator*=(T s) { x = x * s; y = y * s; return *this; }
};
typedef Point_<int> Point2i;
typedef Point_<int> Point;
typedef Point_<float> Point2f;

template <typename T> struct Size_ {
    T width, height;
    Size_() : width(0), height(0) {}
    Size_(T w, T h) : width(w), height(h) {}
};
typedef Size_<int> Size;

struct Rect {
    int x, y, width, height;
    Rect() : x(0), y(0), width(0), height(0) {}
    Rect(int _x, int _y, int w, int h) : x(_x), y(_y), width(w), height(h) {}
};

struct Range {
    int start, end;
    Range(int s, int e) : start(s), end(e) {}
};

// cv::KeyPoint: pt, size, angle, response, octave, class_id (28 bytes, same order as OpenCV 2.4).
class KeyPoint {
public:
    KeyPoint() : pt(0, 0), size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(float x, float y, float _size, float _angle = -1, float _response = 0, int _octave = 0,
             int _class_id = -1)
        : pt(x, y), size(_size), angle(_angle), response(_response), octave(_octave), class_id(_class_id) {}
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
};

// Dense 2-D matrix header over reference-counted storage; ROI views share the storage.
class Mat {
public:
    int flags, rows, cols;
    size_t step;       // bytes between rows
    uchar *data;

    Mat() : flags(CV_8UC1), rows(0), cols(0), step(0), data(nullptr) {}
    Mat(int r, int c, int type) : Mat() { create(r, c, type); }
    Mat(Size sz, int type) : Mat() { create(sz.height, sz.width, type); }
    // user-allocated data, not owned
    Mat(int r, int c, int type, void *d, size_t s = 0)
        : flags(type), rows(r), cols(c), step(s ? s : (size_t)c * esz(type)), data((uchar *)d) {}
    // ROI
    Mat(const Mat &m, const Rect &roi)
        : flags(m.flags), rows(roi.height), cols(roi.width), step(m.step),
          data(m.data + (size_t)roi.y * m.step + (size_t)roi.x * esz(m.flags)), buf_(m.buf_) {}

    static Mat zeros(int r, int c, int type)
    {
        Mat m(r, c, type);
        if (m.data) memset(m.data, 0, (size_t)r * m.step);
        return m;
    }
    void create(int r, int c, int type)
    {
        if (r == rows && c == cols && type == flags && data && isContinuous()) return;
        flags = type;
        rows = r;
        cols = c;
        step = (size_t)c * esz(type);
        const size_t n = (size_t)r * step;
        buf_.reset(n ? new uchar[n] : nullptr, std::default_delete<uchar[]>());
        data = buf_.get();
    }
    void release()
    {
        buf_.reset();
        data = nullptr;
        rows = cols = 0;
        step = 0;
    }
    Mat clone() const
    {
        Mat m(rows, cols, flags);
        for (int y = 0; y < rows; y++) memcpy(m.data + (size_t)y * m.step, data + (size_t)y * step, (size_t)cols * elemSize());
        return m;
    }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return flags; }
    int depth() const { return flags & 7; }
    int channels() const { return (flags >> CV_CN_SHIFT) + 1; }
    size_t elemSize() const { return esz(flags); }
    size_t step1() const { return step / (depth() == CV_32F ? 4 : 1); }
    bool isContinuous() const { return step == (size_t)cols * elemSize(); }
    Size size() const { return Size(cols, rows); }
    Mat row(int y) const { return Mat(*this, Rect(0, y, cols, 1)); }
    Mat rowRange(int a, int b) const { return Mat(*this, Rect(0, a, cols, b - a)); }
    Mat colRange(int a, int b) const { return Mat(*this, Rect(a, 0, b - a, rows)); }
    Mat operator()(const Rect &r) const { return Mat(*this, r); }
    uchar *ptr(int y = 0) { return data + (size_t)y * step; }
    const uchar *ptr(int y = 0) const { return data + (size_t)y * step; }
    template <typename T> T *ptr(int y = 0) { return (T *)(data + (size_t)y * step); }
    template <typename T> const T *ptr(int y = 0) const { return (const T *)(data + (size_t)y * step); }
    template <typename T> T &at(int y, int x) { return ((T *)(data + (size_t)y * step))[x]; }
    template <typename T> const T &at(int y, int x) const { return ((const T *)(data + (size_t)y * step))[x]; }

private:
    static size_t esz(int type) { return (size_t)((type & 7) == CV_32F ? 4 : 1) * (size_t)((type >> CV_CN_SHIFT) + 1); }
    std::shared_ptr<uchar> buf_;
};

// InputArray / OutputArray as used by ORBextractor::operator() (include/ORBextractor.h:77-79):
// only Mat is ever passed on this path.
class _InputArray {
public:
    _InputArray() : m_(nullptr) {}
    _InputArray(const Mat &m) : m_(&m) {}
    Mat getMat() const { return m_ ? *m_ : Mat(); }
    bool empty() const { return !m_ || m_->empty(); }
protected:
    const Mat *m_;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray() : o_(nullptr) {}
    _OutputArray(Mat &m) : _InputArray(m), o_(&m) {}
    void create(int rows, int cols, int type) const { if (o_) o_->create(rows, cols, type); }
    void release() const { if (o_) o_->release(); }
    Mat getMat() const { return o_ ? *o_ : Mat(); }
private:
    Mat *o_;
};
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;

}  // namespace cv

#endif  // ORBHIP_USE_OPENCV
#endif
