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
#include <cfloat>
#include <climits>
#include <map>
#include <stdexcept>
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
#define CV_8S 1
#define CV_8SC1 1
#define CV_16S 3
#define CV_32S 4
#define CV_32SC1 4
#define CV_16SC1 3
#define CV_32FC1 5
#define CV_64F 6
#define CV_64FC1 6
#define CV_8UC3 16
#define CV_EXPORTS
#define CV_EXPORTS_W
#define CV_EXPORTS_W_SIMPLE
#define CV_WRAP
#define CV_OUT
#define CV_IN_OUT
#define CV_PROP_RW
#define CV_PROP
#define CV_Error(code, msg) do { std::cerr << "CV_Error: " << (msg) << std::endl; std::abort(); } while (0)
#define CV_DbgAssert(x) assert(x)

namespace cv {

// cvstd.hpp does this inside namespace cv: unqualified calls in the reference resolve to the std overloads
using std::min; using std::max; using std::abs; using std::swap; using std::sqrt; using std::exp; using std::pow; using std::log;

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
typedef Point_<double> Point2d;
inline Point2i to_point(const Point2f& p) { return Point2i((int)lrintf(p.x), (int)lrintf(p.y)); }   // saturate_cast<int>(float)
struct Size {
  int width, height;
  Size() : width(0), height(0) {}
  Size(int w, int h) : width(w), height(h) {}
  bool operator==(const Size& o) const { return width == o.width && height == o.height; }
  bool operator!=(const Size& o) const { return !(*this == o); }
};
template <typename T, int N> struct Vec {
  T val[N];
  Vec() { for (int i = 0; i < N; i++) val[i] = T(); }
  Vec(T a, T b, T c, T d) { static_assert(N == 4, "4 values"); val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
  T& operator[](int i) { return val[i]; }
  const T& operator[](int i) const { return val[i]; }
};
typedef Vec<float, 4> Vec4f;
typedef Vec<int, 4> Vec4i;
typedef Vec<unsigned char, 3> Vec3b;
struct Scalar {
  double val[4];
  Scalar(double a = 0, double b = 0, double c = 0, double d = 0) { val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
  static Scalar all(double v) { return Scalar(v, v, v, v); }
};
typedef std::string String;
template <typename T> class Ptr : public std::shared_ptr<T> {
 public:
  Ptr() {}
  Ptr(T* p) : std::shared_ptr<T>(p) {}
  template <typename U> Ptr(const Ptr<U>& o) : std::shared_ptr<T>(o) {}
  Ptr(const std::shared_ptr<T>& o) : std::shared_ptr<T>(o) {}
  bool empty() const { return !this->get(); }
  operator T*() const { return this->get(); }
};
template <typename T, typename... A> Ptr<T> makePtr(A&&... a) { return Ptr<T>(new T(std::forward<A>(a)...)); }
class FileNode;
class FileStorage;
class Algorithm {
 public:
  virtual ~Algorithm() {}
  virtual void read(const FileNode&) {}
  virtual void write(FileStorage&) const {}
};
struct DMatch {
  int queryIdx, trainIdx, imgIdx;
  float distance;
  DMatch() : queryIdx(-1), trainIdx(-1), imgIdx(-1), distance(0) {}
  DMatch(int q, int t, float d) : queryIdx(q), trainIdx(t), imgIdx(-1), distance(d) {}
  DMatch(int q, int t, int i, float d) : queryIdx(q), trainIdx(t), imgIdx(i), distance(d) {}
  bool operator<(const DMatch& m) const { return distance < m.distance; }
};
namespace Error { enum { StsBadArg = -5, BadDataPtr = -12, StsBadSize = -201 }; }
enum { NORM_HAMMING = 6, NORM_L2 = 4, NORM_L1 = 2, COLOR_BGR2GRAY = 6, COLOR_GRAY2BGR = 8, THRESH_BINARY = 0, DECOMP_LU = 0 };
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
  int channels() const { return type_ == CV_8UC3 ? 3 : 1; }
  int depth() const { return type_ == CV_8UC3 ? CV_8U : type_; }
  bool isContinuous() const { return step.p == (size_t)cols * elem(); }
  size_t total() const { return (size_t)rows * cols; }
  void copyTo(Mat& m) const { m = clone(); }
  void copyTo(const class _OutputArray& o) const;
  Mat& setTo(const Scalar& v) {
    for (int r = 0; r < rows; r++)
      for (int c = 0; c < cols; c++) {
        if (type_ == CV_8U) at<unsigned char>(r, c) = (unsigned char)v.val[0];
        else if (type_ == CV_16S) at<short>(r, c) = (short)v.val[0];
        else if (type_ == CV_32F) at<float>(r, c) = (float)v.val[0];
        else if (type_ == CV_64F) at<double>(r, c) = v.val[0];
      }
    return *this;
  }
  Mat t() const;
  // compile-only members (stereo / triangulation code of Frame.cc that the harnesses never reach)
  // The stand-in has no channels: an N x 2 CV_32F matrix stands for N two-channel points, so the reshape(2) / reshape(1) pair
  // around cv::undistortPoints (Frame.cc:930-934, 957-959) is the identity here.  Anything else stays compile-only.
  Mat reshape(int cn, int = 0) const {
    if (type_ == CV_32F && cols == 2 && (cn == 1 || cn == 2)) return *this;
    std::cerr << "oracle/ref stub: Mat::reshape is compile-only" << std::endl; std::abort();
  }
  void convertTo(Mat&, int) const { std::cerr << "oracle/ref stub: Mat::convertTo is compile-only" << std::endl; std::abort(); }
  static Mat eye(int r, int c, int type) {
    Mat m = Mat::zeros(r, c, type);
    assert(type == CV_32F);
    for (int i = 0; i < (r < c ? r : c); i++) m.at<float>(i, i) = 1.f;
    return m;
  }
  static Mat ones(int, int, int) { std::cerr << "oracle/ref stub: Mat::ones is compile-only" << std::endl; std::abort(); }
  // 3 x 3 CV_32F only (LSDmatcher::ComputeF12, src/LSDmatcher.cpp:857): the closed form with double products as cv::invert takes for
  // n == 3 -- a stand-in that lets the harness produce a fundamental matrix; nothing derived from it is claimed to be pinned (the
  // product takes F as an input)
  Mat inv(int = 0) const {
    if (!(type_ == CV_32F && rows == 3 && cols == 3)) { std::cerr << "oracle/ref stub: Mat::inv is 3 x 3 CV_32F only" << std::endl; std::abort(); }
    const Mat& S = *this;
    auto m = [&](int i, int j) { return (double)S.at<float>(i, j); };
    double d = m(0, 0) * (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) - m(0, 1) * (m(1, 0) * m(2, 2) - m(1, 2) * m(2, 0)) +
               m(0, 2) * (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0));
    Mat r = Mat::zeros(3, 3, CV_32F);
    if (d == 0.) return r;
    d = 1. / d;
    r.at<float>(0, 0) = (float)((m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) * d);
    r.at<float>(0, 1) = (float)((m(0, 2) * m(2, 1) - m(0, 1) * m(2, 2)) * d);
    r.at<float>(0, 2) = (float)((m(0, 1) * m(1, 2) - m(0, 2) * m(1, 1)) * d);
    r.at<float>(1, 0) = (float)((m(1, 2) * m(2, 0) - m(1, 0) * m(2, 2)) * d);
    r.at<float>(1, 1) = (float)((m(0, 0) * m(2, 2) - m(0, 2) * m(2, 0)) * d);
    r.at<float>(1, 2) = (float)((m(0, 2) * m(1, 0) - m(0, 0) * m(1, 2)) * d);
    r.at<float>(2, 0) = (float)((m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0)) * d);
    r.at<float>(2, 1) = (float)((m(0, 1) * m(2, 0) - m(0, 0) * m(2, 1)) * d);
    r.at<float>(2, 2) = (float)((m(0, 0) * m(1, 1) - m(0, 1) * m(1, 0)) * d);
    return r;
  }
  Mat cross(const Mat& o) const {
    Mat r(rows, cols, CV_32F);
    const float a0 = at<float>(0), a1 = at<float>(1), a2 = at<float>(2), b0 = o.at<float>(0), b1 = o.at<float>(1), b2 = o.at<float>(2);
    r.at<float>(0) = a1 * b2 - a2 * b1; r.at<float>(1) = a2 * b0 - a0 * b2; r.at<float>(2) = a0 * b1 - a1 * b0;
    return r;
  }
  Mat& operator/=(double s) {
    for (int i = 0; i < rows; i++)
      for (int j = 0; j < cols; j++) at<float>(i, j) = (float)(at<float>(i, j) / s);
    return *this;
  }
  double dot(const Mat& o) const {
    double s = 0;
    for (int i = 0; i < rows; i++)
      for (int j = 0; j < cols; j++) s += (double)at<float>(i, j) * o.at<float>(i, j);
    return s;
  }
  Mat row(int r) const { return rowRange(r, r + 1); }
  Mat col(int c) const { return colRange(c, c + 1); }
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
  template <typename T> T& at(int i) { return rows == 1 ? ptr<T>(0)[i] : ptr<T>(i)[0]; }
  template <typename T> const T& at(int i) const { return rows == 1 ? ptr<T>(0)[i] : ptr<T>(i)[0]; }

 private:
  size_t elem() const { return (type_ == CV_32F || type_ == CV_32S) ? 4 : type_ == CV_16S ? 2 : type_ == CV_64F ? 8 : type_ == CV_8UC3 ? 3 : 1; }
  int type_ = CV_8U;
  std::shared_ptr<std::vector<unsigned char> > buf_;
};

template <typename T> struct MatType;
template <> struct MatType<unsigned char> { enum { value = CV_8U }; };
template <> struct MatType<short> { enum { value = CV_16S }; };
template <> struct MatType<int> { enum { value = CV_32S }; };
template <> struct MatType<float> { enum { value = CV_32F }; };
template <> struct MatType<double> { enum { value = CV_64F }; };
template <typename T> class Mat_ : public Mat {
 public:
  Mat_() {}
  Mat_(int r, int c) : Mat(r, c, MatType<T>::value) {}
  Mat_(const Mat& m) : Mat(m) {}
  template <typename U> Mat_& operator=(const Mat_<U>& o) {   // converting assignment (Mat_<float> = Mat_<int>(r, c))
    create(o.rows, o.cols, MatType<T>::value);
    for (int r = 0; r < rows; r++)
      for (int c = 0; c < cols; c++) (*this)[r][c] = (T)o[r][c];
    return *this;
  }
  Mat_& operator=(const Mat& m) { Mat::operator=(m); return *this; }
  T* operator[](int r) { return ptr<T>(r); }
  const T* operator[](int r) const { return ptr<T>(r); }
  T& operator()(int r, int c) { return ptr<T>(r)[c]; }
  Mat_ t() const { stub_unreachable_("Mat_::t"); return Mat_(); }
  static Mat_ eye(int r, int c) {   // Mat_<double>::eye(3, 3): the rectification argument of initUndistortRectifyMap (Frame.cc:221)
    Mat_ m(r, c);
    for (int i = 0; i < r; i++)
      for (int j = 0; j < c; j++) m(i, j) = (T)(i == j);
    return m;
  }
 private:
  static void stub_unreachable_(const char* w) { std::cerr << "oracle/ref stub: " << w << " is compile-only" << std::endl; std::abort(); }
};
template <typename T> struct MatCommaInitializer_ {
  Mat_<T> m; int k;
  template <typename U> MatCommaInitializer_& operator,(U x) { m.template ptr<T>(k / m.cols)[k % m.cols] = (T)x; k++; return *this; }
  operator Mat() const { return m; }
  operator Mat_<T>() const { return m; }
};
template <typename T, typename U> inline MatCommaInitializer_<T> operator<<(const Mat_<T>& m, U x) {
  MatCommaInitializer_<T> c{m, 0};
  c, x;
  return c;
}

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
  _OutputArray(const Mat& m) : _InputArray(m) {}   // temporaries (views) as destinations, as OpenCV allows
  void create(int r, int c, int type) const { m_->create(r, c, type); }
  void release() const { m_->release(); }
  void assign(const Mat& m) const { *m_ = m; }
  // OpenCV's copyTo: create() is a no-op for a destination of the right size and type, so the data lands in the existing
  // storage (this is what makes `a.copyTo(b.rowRange(..))` fill a block of b)
  void copy_from(const Mat& src) const;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
inline void _OutputArray::copy_from(const Mat& src) const {
  Mat& d = *m_;
  if (d.data && src.data && d.data != src.data && d.rows == src.rows && d.cols == src.cols && d.type() == src.type()) {
    const size_t row_bytes = (size_t)src.cols * (src.type() == CV_8U || src.type() == CV_8S ? 1 : src.type() == CV_16S ? 2 : src.type() == CV_64F ? 8 : src.type() == CV_8UC3 ? 3 : 4);
    for (int r = 0; r < src.rows; r++) std::memcpy(d.data + (size_t)r * d.step.p, src.data + (size_t)r * src.step.p, row_bytes);
  } else if (d.data != src.data || !d.data) {
    d = src.clone();
  }
}
inline void Mat::copyTo(const _OutputArray& o) const { o.copy_from(*this); }
[[noreturn]] inline void stub_unreachable(const char* what) { std::cerr << "oracle/ref stub: " << what << " is compile-only" << std::endl; std::abort(); }
// Small dense float algebra (the pose arithmetic of ORBmatcher.cc).  Plain float accumulation in index order; OpenCV's
// gemm may round differently, so nothing computed through these is claimed to be pinned.
inline Mat Mat::t() const {
  assert(type() == CV_32F);
  Mat r(cols, rows, CV_32F);
  for (int i = 0; i < rows; i++)
    for (int j = 0; j < cols; j++) r.at<float>(j, i) = at<float>(i, j);
  return r;
}
inline Mat operator*(const Mat& a, const Mat& b) {
  assert(a.type() == CV_32F && b.type() == CV_32F && a.cols == b.rows);
  Mat r(a.rows, b.cols, CV_32F);
  for (int i = 0; i < a.rows; i++)
    for (int j = 0; j < b.cols; j++) {
      float acc = 0;
      for (int k = 0; k < a.cols; k++) acc += a.at<float>(i, k) * b.at<float>(k, j);
      r.at<float>(i, j) = acc;
    }
  return r;
}
inline Mat mat_scale(const Mat& a, double s) {
  assert(a.type() == CV_32F);
  Mat r(a.rows, a.cols, CV_32F);
  for (int i = 0; i < a.rows; i++)
    for (int j = 0; j < a.cols; j++) r.at<float>(i, j) = (float)(a.at<float>(i, j) * s);
  return r;
}
inline Mat operator*(double s, const Mat& a) { return mat_scale(a, s); }
inline Mat operator*(const Mat& a, double s) { return mat_scale(a, s); }
inline Mat operator-(const Mat& a) { return mat_scale(a, -1.0); }
inline Mat mat_addsub(const Mat& a, const Mat& b, float sgn) {
  assert(a.type() == CV_32F && b.type() == CV_32F && a.rows == b.rows && a.cols == b.cols);
  Mat r(a.rows, a.cols, CV_32F);
  for (int i = 0; i < a.rows; i++)
    for (int j = 0; j < a.cols; j++) r.at<float>(i, j) = a.at<float>(i, j) + sgn * b.at<float>(i, j);
  return r;
}
inline Mat operator-(const Mat& a, const Mat& b) { return mat_addsub(a, b, -1.f); }
inline Mat operator+(const Mat& a, const Mat& b) { return mat_addsub(a, b, 1.f); }
inline double norm(const Mat& a) {
  assert(a.type() == CV_32F);
  double s = 0;
  for (int i = 0; i < a.rows; i++)
    for (int j = 0; j < a.cols; j++) s += (double)a.at<float>(i, j) * a.at<float>(i, j);
  return std::sqrt(s);
}
inline double norm(const Mat& a, const Mat& b, int type) {   // NORM_HAMMING over two byte rows (MapLine.cpp:297)
  assert(type == NORM_HAMMING && a.type() == CV_8U && b.type() == CV_8U && a.rows == 1 && b.rows == 1 && a.cols == b.cols);
  int d = 0;
  for (int i = 0; i < a.cols; i++) d += __builtin_popcount((unsigned)(a.ptr<uchar>(0)[i] ^ b.ptr<uchar>(0)[i]));
  return d;
}
struct SVD {
  enum { MODIFY_A = 1, NO_UV = 2, FULL_UV = 4 };
  static void compute(const Mat&, Mat&, Mat&, Mat&, int = 0) { stub_unreachable("cv::SVD::compute"); }
};
// The three OpenCV calls of the monocular Frame constructor (Frame.cc:220-222, 933, 959), forwarded to the oracle's
// restatements (oracle/img_ops.cc, oracle/frame_search.cc) so that the constructor can be EXECUTED against this stand-in:
// K = 3x3 CV_32F, D = 4x1 or 5x1 CV_32F, R = identity, newK = P = K (the only way the reference calls them).
inline void stub_intrinsics(const Mat& K, const Mat& D, float k[4], float d[5]) {
  k[0] = K.at<float>(0, 0); k[1] = K.at<float>(1, 1); k[2] = K.at<float>(0, 2); k[3] = K.at<float>(1, 2);
  for (int i = 0; i < 5; i++) d[i] = i < D.rows * D.cols ? D.at<float>(i) : 0.f;
}
inline void initUndistortRectifyMap(const Mat& K, const Mat& D, const Mat&, const Mat&, Size sz, int, Mat& m1, Mat& m2) {
  float k[4], d[5];
  stub_intrinsics(K, D, k, d);
  m1.create(sz.height, sz.width, CV_32F);
  m2.create(sz.height, sz.width, CV_32F);
  plo_undistort_maps(k, d, sz.width, sz.height, m1.ptr<float>(0), m2.ptr<float>(0));
}
inline void remap(const Mat& src, Mat& dst, const Mat& mx, const Mat& my, int) {
  assert(src.type() == CV_8U && mx.isContinuous() && my.isContinuous());
  Mat out(src.rows, src.cols, CV_8U);
  plo_remap_linear_u8(src.data, src.cols, src.rows, src.step, mx.ptr<float>(0), my.ptr<float>(0), out.data, out.step);
  dst = out;
}
inline void undistortPoints(const Mat& src, Mat& dst, const Mat& K, const Mat& D, const Mat& = Mat(), const Mat& = Mat()) {
  assert(src.type() == CV_32F && src.cols == 2);   // N points as an N x 2 float matrix (see Mat::reshape)
  float k[4], d[5];
  stub_intrinsics(K, D, k, d);
  std::vector<plo_keypoint> in(src.rows), out(src.rows);
  for (int i = 0; i < src.rows; i++) { in[i] = plo_keypoint(); in[i].x = src.at<float>(i, 0); in[i].y = src.at<float>(i, 1); }
  // plo_undistort_keypoints copies for k1 == 0 (that shortcut is Frame::UndistortKeyPoints' own, :917-921, and the reference
  // never reaches cv::undistortPoints then)
  plo_undistort_keypoints(in.data(), src.rows, k, d, out.data());
  Mat r(src.rows, 2, CV_32F);
  for (int i = 0; i < src.rows; i++) { r.at<float>(i, 0) = out[i].x; r.at<float>(i, 1) = out[i].y; }
  dst = r;
}
inline bool solve(const Mat&, const Mat&, Mat&, int = 0) { stub_unreachable("cv::solve"); }
inline void cvtColor(const Mat&, Mat&, int) { stub_unreachable("cv::cvtColor"); }
inline double threshold(const Mat&, Mat&, double, double, int) { stub_unreachable("cv::threshold"); }
inline Mat abs(const Mat&) { stub_unreachable("cv::abs(Mat)"); }
inline void add(const Mat&, const Mat&, Mat&) { stub_unreachable("cv::add"); }
inline void compare(const Mat&, const Mat&, Mat&, int) { stub_unreachable("cv::compare"); }
inline Mat operator/(const Mat& a, double s) { return mat_scale(a, 1.0 / s); }   // MatExpr: scale by alpha = 1/s
enum { THRESH_TOZERO = 3, CMP_LT = 3, CMP_GT = 1 };

// ---- image-processing primitives: forwarded to the oracle's restatements ----
inline void resize(const Mat& src, Mat& dst, Size sz, double = 0, double = 0, int = INTER_LINEAR) {
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
inline void GaussianBlur(const Mat& src, Mat& dst, Size k, double sx, double = 0, int = BORDER_REFLECT_101) {
  assert(k.width == k.height);
  Mat tmp(src.rows, src.cols, src.type());
  plo_gaussian_blur_u8(src.data, src.cols, src.rows, src.step, tmp.data, tmp.step, k.width, sx);
  dst.create(src.rows, src.cols, src.type());
  for (int r = 0; r < src.rows; r++) std::memcpy(dst.data + (size_t)r * dst.step, tmp.data + (size_t)r * tmp.step, src.cols);
}
// cv::pyrDown(src8u, dst, dsize): the oracle's restatement; OpenCV's size assertion (|2 dsize - ssize| <= 2) becomes the exception
// it throws.  dst may be src (LSDDetector_custom.cpp:69): the result gets a buffer of its own, as Mat::create does for a new size.
inline void pyrDown(const Mat& src, Mat& dst, Size dsize = Size()) {
  if (dsize.width == 0 && dsize.height == 0) dsize = Size((src.cols + 1) / 2, (src.rows + 1) / 2);
  Mat tmp(std::max(dsize.height, 1), std::max(dsize.width, 1), src.type());
  if (plo_pyr_down_u8(src.data, src.cols, src.rows, src.step, tmp.data, dsize.width, dsize.height, tmp.step) != 0)
    throw std::runtime_error("cv::pyrDown: (-215) std::abs(dsize.width*2 - ssize.width) <= 2 && std::abs(dsize.height*2 - ssize.height) <= 2");
  dst = tmp;
}
inline void FAST(const Mat& img, std::vector<KeyPoint>& kps, int threshold, bool nonmax) {
  std::vector<plo_keypoint> out((size_t)img.rows * img.cols + 1);
  const int n = plo_fast9_16(img.data, img.cols, img.rows, img.step, threshold, nonmax ? 1 : 0, out.data(), (int)out.size());
  kps.clear();
  for (int i = 0; i < n; i++) kps.push_back(KeyPoint(out[i].x, out[i].y, out[i].size, out[i].angle, out[i].response, out[i].octave, out[i].class_id));
}
// cv::Sobel(src8u, dst, CV_16S, dx, dy, 3): the oracle computes both derivatives at once
inline void Sobel(const Mat& src, Mat& dst, int ddepth, int dx, int dy, int ksize) {
  assert(ddepth == CV_16S && ksize == 3 && src.type() == CV_8U && dx + dy == 1);
  std::vector<int16_t> gx((size_t)src.rows * src.cols), gy((size_t)src.rows * src.cols);
  plo_sobel3_s16(src.data, src.cols, src.rows, src.step, gx.data(), gy.data());
  dst.create(src.rows, src.cols, CV_16S);
  const std::vector<int16_t>& g = dx ? gx : gy;
  for (int r = 0; r < src.rows; r++) std::memcpy(dst.data + (size_t)r * dst.step, &g[(size_t)r * src.cols], (size_t)src.cols * 2);
}
// cv::LineSegmentDetector (imgproc lsd.cpp) = the oracle's restatement
class LineSegmentDetector {
 public:
  void detect(const Mat& img, std::vector<Vec4f>& lines) {
    const int cap = img.rows * img.cols / 4 + 64;
    std::vector<float> seg((size_t)cap * 4);
    const int n = plo_lsd_detect(img.data, img.cols, img.rows, img.step, seg.data(), cap);
    lines.clear();
    for (int i = 0; i < n && i < cap; i++) lines.push_back(Vec4f(seg[4 * i], seg[4 * i + 1], seg[4 * i + 2], seg[4 * i + 3]));
  }
};
inline Ptr<LineSegmentDetector> createLineSegmentDetector(int = 0, double = 0.8, double = 0.6, double = 2.0, double = 22.5,
                                                          double = 0, double = 0.7, int = 1024) {
  return Ptr<LineSegmentDetector>(new LineSegmentDetector());
}
// cv::LineIterator: only .count is used, for end points inside the image (8-connected: max(|dx|, |dy|) + 1)
class LineIterator {
 public:
  int count;
  LineIterator(const Mat&, Point2f a, Point2f b, int = 8, bool = false) {
    const Point2i p = to_point(a), q = to_point(b);
    count = std::max(std::abs(q.x - p.x), std::abs(q.y - p.y)) + 1;
  }
};
// cv::BFMatcher(NORM_HAMMING, false)::knnMatch(q, t, matches, 2) = the oracle's restatement (plo_knn2)
class BFMatcher {
 public:
  BFMatcher(int = NORM_HAMMING, bool = false) {}
  void knnMatch(const Mat& q, const Mat& t, std::vector<std::vector<DMatch> >& matches, int k) const {
    assert(k == 2 && q.cols == 32 && t.cols == 32 && q.isContinuous() && t.isContinuous());
    std::vector<int32_t> idx((size_t)q.rows * 2 + 2), dist((size_t)q.rows * 2 + 2);
    plo_knn2(q.data, q.rows, t.data, t.rows, idx.data(), dist.data());
    matches.assign(q.rows, std::vector<DMatch>());
    for (int i = 0; i < q.rows; i++)
      for (int j = 0; j < 2 && j < t.rows; j++) matches[i].push_back(DMatch(i, idx[2 * i + j], (float)dist[2 * i + j]));
  }
};
// debugging output of the matchers: no-ops
enum { CV_AA = 16, LINE_AA = 16 };
inline void line(Mat&, Point, Point, const Scalar&, int = 1, int = 8, int = 0) {}
inline bool imwrite(const std::string&, const Mat&) { return true; }
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
