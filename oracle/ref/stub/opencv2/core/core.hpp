// Minimal stand-in for the OpenCV headers, just enough to compile two pieces of the REFERENCE from the sources where
// they lie (oracle/ref/build_ref.sh): its vendored DBoW2 (Thirdparty/DBoW2) and its ORB extractor (src/ORBextractor.cc).
// OpenCV itself is not in the image.  TEST INFRASTRUCTURE ONLY.
//
//  * cv::Mat is a reference-counted byte / float matrix with row step and sub-matrix views (rowRange, colRange,
//    operator()(Rect)) that share storage, the few members those sources touch, and OpenCV's "assign a zeros()
//    expression into an existing matrix of the same size" semantics (computeDescriptors relies on it).
//  * The image-processing primitives cv::resize, cv::GaussianBlur, cv::FAST, cv::fastAtan2 forward to the ORACLE's
//    restatements (oracle/img_ops.cc): what oracle/_ref pins for ORB is therefore the reference's own control logic
//    (scale tables, per-cell FAST + fallback, quad-tree distribution, IC_Angle, steered rBRIEF, ordering) running on top
//    of restated primitives -- not OpenCV's primitives themselves.
//  * cv::FileStorage / cv::FileNode only let the YAML members of DBoW2's TemplatedVocabulary compile (never called).
#ifndef PLO_REF_STUB_OPENCV_CORE_HPP
#define PLO_REF_STUB_OPENCV_CORE_HPP
// (the real headers pull these in transitively; the reference sources rely on it)
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "../../../../plo.h"   // the oracle's restated OpenCV primitives

typedef unsigned char uchar;
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_Assert(x) assert(x)
#define CV_PI 3.1415926535897932384626433832795

namespace cv {

inline int cvRound(double v) { return (int)lrint(v); }
inline int cvRound(float v) { return (int)lrintf(v); }
inline int cvRound(int v) { return v; }
inline int cvFloor(double v) { const int i = (int)v; return i - (i > v); }
inline int cvCeil(double v) { const int i = (int)v; return i + (i < v); }
inline float fastAtan2(float y, float x) { return plo_fast_atan2(y, x); }

template <typename T> struct Point_ {
  T x, y;
  Point_() : x(0), y(0) {}
  Point_(T x_, T y_) : x(x_), y(y_) {}
  Point_& operator*=(float s) { x = (T)(x * s); y = (T)(y * s); return *this; }
};
typedef Point_<int> Point2i;
typedef Point2i Point;
typedef Point_<float> Point2f;
struct Size {
  int width, height;
  Size() : width(0), height(0) {}
  Size(int w, int h) : width(w), height(h) {}
};
struct Rect {
  int x, y, width, height;
  Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {}
};
struct KeyPoint {
  Point2f pt;
  float size, angle, response;
  int octave, class_id;
  KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
  KeyPoint(float x, float y, float s, float a = -1, float r = 0, int o = 0, int c = -1)
      : pt(x, y), size(s), angle(a), response(r), octave(o), class_id(c) {}
};

enum { BORDER_REFLECT_101 = 4, BORDER_ISOLATED = 16, INTER_LINEAR = 1 };

struct MatStep {
  size_t p;
  operator size_t() const { return p; }
};
struct ZerosExpr {   // what Mat::zeros returns: assigned INTO an existing matrix of that size, like a cv::MatExpr
  int rows, cols, type;
};

class Mat {
 public:
  int rows = 0, cols = 0;
  unsigned char* data = nullptr;
  MatStep step = {0};
  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(Size s, int type) { create(s.height, s.width, type); }
  Mat(int r, int c, int type, void* ext, size_t st) : rows(r), cols(c), data((unsigned char*)ext), type_(type) { step.p = st; }
  Mat(const ZerosExpr& z) { *this = z; }
  static ZerosExpr zeros(int r, int c, int type) { return ZerosExpr{r, c, type}; }
  Mat& operator=(const ZerosExpr& z) {
    create(z.rows, z.cols, z.type);
    for (int r = 0; r < rows; r++) std::memset(data + (size_t)r * step.p, 0, (size_t)cols * elem());
    return *this;
  }
  void create(int r, int c, int type) {
    if (r == rows && c == cols && type == type_ && data) return;
    rows = r; cols = c; type_ = type;
    step.p = (size_t)c * elem();
    buf_ = std::make_shared<std::vector<unsigned char> >((size_t)r * step.p);
    data = buf_->empty() ? nullptr : buf_->data();
  }
  Mat clone() const {
    Mat m;
    if (data) {
      m.create(rows, cols, type_);
      for (int r = 0; r < rows; r++) std::memcpy(m.data + (size_t)r * m.step.p, data + (size_t)r * step.p, (size_t)cols * elem());
    }
    return m;
  }
  void release() { buf_.reset(); data = nullptr; rows = cols = 0; step.p = 0; }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  int type() const { return type_; }
  Size size() const { return Size(cols, rows); }
  Mat rowRange(int a, int b) const { Mat m(*this); m.data = data + (size_t)a * step.p; m.rows = b - a; return m; }
  Mat colRange(int a, int b) const { Mat m(*this); m.data = data + (size_t)a * elem(); m.cols = b - a; return m; }
  Mat operator()(const Rect& r) const { return rowRange(r.y, r.y + r.height).colRange(r.x, r.x + r.width); }
  size_t step1() const { return step.p / elem(); }
  unsigned char* ptr(int r = 0) { return data + (size_t)r * step.p; }
  const unsigned char* ptr(int r = 0) const { return data + (size_t)r * step.p; }
  template <typename T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * step.p); }
  template <typename T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * step.p); }
  template <typename T> T& at(int r, int c) { return ptr<T>(r)[c]; }
  template <typename T> const T& at(int r, int c) const { return ptr<T>(r)[c]; }

 private:
  size_t elem() const { return type_ == CV_32F ? 4 : 1; }
  int type_ = CV_8U;
  std::shared_ptr<std::vector<unsigned char> > buf_;
};

// InputArray / OutputArray: thin handles on a Mat
class _InputArray {
 public:
  _InputArray(const Mat& m) : m_(const_cast<Mat*>(&m)) {}
  bool empty() const { return m_->empty(); }
  Mat getMat() const { return *m_; }
 protected:
  Mat* m_;
};
class _OutputArray : public _InputArray {
 public:
  _OutputArray(Mat& m) : _InputArray(m) {}
  void create(int r, int c, int type) const { m_->create(r, c, type); }
  void release() const { m_->release(); }
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;

// ---- image-processing primitives: forwarded to the oracle's restatements ----
inline void resize(const Mat& src, Mat& dst, Size sz, double, double, int) {
  dst.create(sz.height, sz.width, src.type());   // no-op for the pre-sized pyramid views
  plo_resize_linear_u8(src.data, src.cols, src.rows, src.step, dst.data, dst.cols, dst.rows, dst.step);
}
inline int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
  return p;
}
// BORDER_REFLECT_101 (with or without BORDER_ISOLATED: the sources are never sub-matrices whose surroundings matter).
// Safe when src is the interior view of dst itself (ComputePyramid does that).
inline void copyMakeBorder(const Mat& src, Mat& dst, int top, int bottom, int left, int right, int) {
  dst.create(src.rows + top + bottom, src.cols + left + right, src.type());
  for (int y = 0; y < src.rows; y++) {
    unsigned char* d = dst.data + (size_t)(y + top) * dst.step + left;
    const unsigned char* s = src.data + (size_t)y * src.step;
    if (d != s) std::memmove(d, s, src.cols);
    for (int x = 0; x < left; x++) d[x - left] = d[reflect101(x - left, src.cols)];
    for (int x = 0; x < right; x++) d[src.cols + x] = d[reflect101(src.cols + x, src.cols)];
  }
  for (int y = 0; y < top; y++)
    std::memcpy(dst.data + (size_t)y * dst.step, dst.data + (size_t)(top + reflect101(y - top, src.rows)) * dst.step, dst.cols);
  for (int y = 0; y < bottom; y++)
    std::memcpy(dst.data + (size_t)(top + src.rows + y) * dst.step,
                dst.data + (size_t)(top + reflect101(src.rows + y, src.rows)) * dst.step, dst.cols);
}
inline void GaussianBlur(const Mat& src, Mat& dst, Size k, double sx, double, int) {
  assert(k.width == k.height);
  Mat tmp(src.rows, src.cols, src.type());
  plo_gaussian_blur_u8(src.data, src.cols, src.rows, src.step, tmp.data, tmp.step, k.width, sx);
  dst.create(src.rows, src.cols, src.type());
  for (int r = 0; r < src.rows; r++) std::memcpy(dst.data + (size_t)r * dst.step, tmp.data + (size_t)r * tmp.step, src.cols);
}
inline void FAST(const Mat& img, std::vector<KeyPoint>& kps, int threshold, bool nonmax) {
  std::vector<plo_keypoint> out((size_t)img.rows * img.cols + 1);
  const int n = plo_fast9_16(img.data, img.cols, img.rows, img.step, threshold, nonmax ? 1 : 0, out.data(), (int)out.size());
  kps.clear();
  for (int i = 0; i < n; i++) kps.push_back(KeyPoint(out[i].x, out[i].y, out[i].size, out[i].angle, out[i].response, out[i].octave, out[i].class_id));
}
struct KeyPointsFilter {   // only ComputeKeyPointsOld (dead code in the reference) uses it
  static void retainBest(std::vector<KeyPoint>& k, int n) {
    if (n >= 0 && (int)k.size() > n) {
      std::stable_sort(k.begin(), k.end(), [](const KeyPoint& a, const KeyPoint& b) { return a.response > b.response; });
      k.resize(n);
    }
  }
};

// ---- YAML storage: compile-only ----
class FileNode {
 public:
  FileNode operator[](const char*) const { return FileNode(); }
  FileNode operator[](const std::string&) const { return FileNode(); }
  FileNode operator[](int) const { return FileNode(); }
  size_t size() const { return 0; }
  bool empty() const { return true; }
  operator int() const { return 0; }
  operator double() const { return 0; }
  operator float() const { return 0; }
  operator std::string() const { return std::string(); }
};
class FileStorage {
 public:
  enum { READ = 0, WRITE = 1 };
  FileStorage() {}
  FileStorage(const char*, int) {}
  FileStorage(const std::string&, int) {}
  bool isOpened() const { return false; }
  void release() {}
  FileNode operator[](const char*) const { return FileNode(); }
  FileNode operator[](const std::string&) const { return FileNode(); }
};
template <typename T> inline FileStorage& operator<<(FileStorage& fs, const T&) { return fs; }

}  // namespace cv
#endif
