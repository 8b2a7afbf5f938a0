// Minimal OpenCV API stub: ONLY for syntax/type-checking pl-slam_amd/adaptor/*.h in an image without OpenCV.
// Declarations mirror the OpenCV 3.x signatures the adaptor uses; there are no definitions (nothing is linked).
#pragma once
#include <cstddef>
#include <vector>
typedef unsigned char uchar;
#define CV_8U 0
#define CV_8UC1 0
namespace cv {
struct Size { int width, height; bool operator!=(const Size& o) const; };
class Mat;
class _InputArray { public: _InputArray(); _InputArray(const Mat&); bool empty() const; Mat getMat(int i = -1) const; };
class _OutputArray : public _InputArray { public: _OutputArray(); _OutputArray(Mat&); void create(int rows, int cols, int type) const; void release() const; };
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
class Mat {
 public:
  Mat(); Mat(int rows, int cols, int type);
  int rows, cols; uchar* data; size_t step;
  int type() const; bool empty() const; bool isContinuous() const; Size size() const; Mat clone() const;
  void create(int rows, int cols, int type);
  Mat rowRange(int a, int b) const; void copyTo(OutputArray m) const;
  template <typename T> T* ptr(int r = 0); template <typename T> const T* ptr(int r = 0) const;
  template <typename T> T& at(int i); template <typename T> const T& at(int i) const;
  template <typename T> T& at(int r, int c); template <typename T> const T& at(int r, int c) const;
};
struct Point2f { float x, y; };
class KeyPoint { public: Point2f pt; float size, angle, response; int octave, class_id; };
struct DMatch { DMatch(); DMatch(int q, int t, float d); int queryIdx, trainIdx, imgIdx; float distance; };
}  // namespace cv
