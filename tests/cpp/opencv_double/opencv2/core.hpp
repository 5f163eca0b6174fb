// Declaration-only TEST DOUBLE of the few OpenCV 4 types include/jsorb_compat.hpp touches under JSORB_WITH_OPENCV.  OpenCV is not
// installed in this image; this header lets tests/test_abi_and_host.py TYPE-CHECK (g++ -fsyntax-only) the cv::Mat / cv::KeyPoint
// overloads, i.e. the exact signatures of the reference (include/ORBextractor.h:40-42, Frame.cpp:780-803).  Nothing here is linked or run.
#pragma once
#include <cstddef>
#include <string>
#include <vector>
#define CV_8UC1 0
namespace cv {
struct MatStep { size_t v; operator size_t() const { return v; } };
class Mat {
public:
    Mat();
    Mat(int rows, int cols, int type);
    bool empty() const;
    unsigned char *ptr(int row = 0);
    const unsigned char *ptr(int row = 0) const;
    unsigned char *data;
    int rows, cols;
    MatStep step;
};
struct Point2f { float x, y; };
class KeyPoint {
public:
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
};
} // namespace cv
