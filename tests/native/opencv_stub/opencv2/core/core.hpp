// TEST-ONLY STUB -- NOT OpenCV.  Declarations (no definitions) of the handful of OpenCV 2.4 names that
// vi-orb-slam-icra2018_amd/host/*.cc and include/orbhip/*.h use, written from the OpenCV 2.4 API as the reference's
// call sites rely on it (src/ORBextractor.cc, src/ORBmatcher.cc, src/Frame.cc): Mat::step is a MatStep object (not a
// size_t), InputArray / OutputArray are references to proxy classes, KeyPoint has its 28-byte field order.
// Purpose: `g++ -fsyntax-only -DORBHIP_USE_OPENCV` of the drop-in sources (tests/test_host_logic.py), so that the branch of
// include/orbhip/cvlite.h that a build inside the reference tree takes does not rot.  It pins nothing about OpenCV's
// behaviour and links against nothing.
#ifndef ORBHIP_TEST_OPENCV_STUB_CORE_HPP
#define ORBHIP_TEST_OPENCV_STUB_CORE_HPP

#include <cstddef>
#include <vector>

typedef unsigned char uchar;

#define CV_8U 0
#define CV_32F 5
#define CV_CN_SHIFT 3
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_PI 3.1415926535897932384626433832795

int cvRound(double value);
int cvFloor(double value);
int cvCeil(double value);

namespace cv {

template <typename T> class Point_ {
public:
    Point_();
    Point_(T x, T y);
    T x, y;
};
typedef Point_<int> Point2i;
typedef Point2i Point;
typedef Point_<float> Point2f;

template <typename T> class Size_ {
public:
    Size_();
    Size_(T width, T height);
    T width, height;
};
typedef Size_<int> Size;

template <typename T> class Rect_ {
public:
    Rect_();
    Rect_(T x, T y, T width, T height);
    T x, y, width, height;
};
typedef Rect_<int> Rect;

class Range {
public:
    Range(int start, int end);
    int start, end;
};

struct MatStep {
    MatStep();
    operator size_t() const;
    size_t *p;
};

class Mat {
public:
    enum { AUTO_STEP = 0 };
    Mat();
    Mat(int rows, int cols, int type);
    Mat(Size size, int type);
    Mat(int rows, int cols, int type, void *data, size_t step = AUTO_STEP);
    Mat(const Mat &m, const Rect &roi);
    static Mat zeros(int rows, int cols, int type);
    void create(int rows, int cols, int type);
    void release();
    Mat clone() const;
    bool empty() const;
    int type() const;
    int depth() const;
    int channels() const;
    size_t elemSize() const;
    bool isContinuous() const;
    Size size() const;
    Mat row(int y) const;
    Mat rowRange(int startrow, int endrow) const;
    Mat colRange(int startcol, int endcol) const;
    Mat operator()(const Rect &roi) const;
    uchar *ptr(int i0 = 0);
    const uchar *ptr(int i0 = 0) const;
    template <typename T> T *ptr(int i0 = 0);
    template <typename T> const T *ptr(int i0 = 0) const;
    template <typename T> T &at(int i0, int i1);
    template <typename T> const T &at(int i0, int i1) const;
    int flags, dims, rows, cols;
    uchar *data;
    MatStep step;
};

class _InputArray {
public:
    _InputArray();
    _InputArray(const Mat &m);
    virtual ~_InputArray();
    virtual Mat getMat(int i = -1) const;
    virtual bool empty() const;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray();
    _OutputArray(Mat &m);
    virtual void create(int rows, int cols, int type, int i = -1, bool allowTransposed = false, int fixedDepthMask = 0) const;
    virtual void release() const;
};
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;

}  // namespace cv
#endif
