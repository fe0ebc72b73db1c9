// declarations only (see ../../README.md)
#pragma once
namespace cv {
struct Point2f { float x, y; };
struct KeyPoint { Point2f pt; };
struct Mat { template <class T> T at(int r, int c) const; template <class T> T at(int i) const; };
Mat operator*(const Mat&, const Mat&);
}  // namespace cv
