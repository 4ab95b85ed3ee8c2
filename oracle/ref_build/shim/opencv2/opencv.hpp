// TEST INFRASTRUCTURE: the few OpenCV names the Cerberus feature manager mentions, as inert stand-ins. The only functions that
// would use them for arithmetic (solvePoseByPnP / initFramePoseByPnP, feature_manager.cpp:215-298) are outside the window
// manager rows pinned here; cv::solvePnP reports failure if it is ever reached.
#pragma once
#include <vector>
namespace cv {
struct Point2f { float x, y; Point2f(float a = 0, float b = 0) : x(a), y(b) {} };
struct Point3f { float x, y, z; Point3f(float a = 0, float b = 0, float c = 0) : x(a), y(b), z(c) {} };
struct Mat { };
template <class T> struct Mat_ : Mat {
  Mat_(int, int) {}
  Mat_ &operator<<(T) { return *this; }
  Mat_ &operator,(T) { return *this; }
};
template <class M> inline void eigen2cv(const M &, Mat &) {}
template <class M> inline void cv2eigen(const Mat &, M &) {}
inline void Rodrigues(const Mat &, Mat &) {}
template <class A, class B> inline bool solvePnP(const A &, const B &, const Mat &, const Mat &, Mat &, Mat &, int) { return false; }
}  // namespace cv
