// TEST-ONLY STUB -- NOT OpenCV (see ../core/core.hpp): cv::KeyPoint as OpenCV 2.4 declares it.
#ifndef ORBHIP_TEST_OPENCV_STUB_FEATURES2D_HPP
#define ORBHIP_TEST_OPENCV_STUB_FEATURES2D_HPP
#include "opencv2/core/core.hpp"
namespace cv {
class KeyPoint {
public:
    KeyPoint();
    KeyPoint(Point2f _pt, float _size, float _angle = -1, float _response = 0, int _octave = 0, int _class_id = -1);
    KeyPoint(float x, float y, float _size, float _angle = -1, float _response = 0, int _octave = 0, int _class_id = -1);
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
};
}  // namespace cv
#endif
