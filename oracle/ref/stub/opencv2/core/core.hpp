// Minimal stand-in for <opencv2/core/core.hpp>, just enough to compile the reference's vendored DBoW2
// (Thirdparty/DBoW2) from the sources where they lie -- see oracle/ref/build_ref.sh.  OpenCV itself is not in the image.
// TEST INFRASTRUCTURE ONLY.  cv::Mat here is a reference-counted byte matrix with the handful of members DBoW2 touches
// (ctor(rows, cols, type), zeros, create, clone, release, ptr<T>(), data, rows, cols, empty); cv::FileStorage /
// cv::FileNode only have to let the YAML save / load members of TemplatedVocabulary compile (they are never called).
#ifndef PLO_REF_STUB_OPENCV_CORE_HPP
#define PLO_REF_STUB_OPENCV_CORE_HPP
// (the real header pulls these in transitively; the DBoW2 sources rely on it)
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#define CV_8U 0
#define CV_32F 5

namespace cv {

class Mat {
 public:
  int rows = 0, cols = 0;
  unsigned char* data = nullptr;
  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  static Mat zeros(int r, int c, int type) {
    Mat m(r, c, type);
    if (m.data) std::memset(m.data, 0, m.bytes());
    return m;
  }
  void create(int r, int c, int type) {
    if (r == rows && c == cols && type == type_ && data) return;
    rows = r; cols = c; type_ = type;
    buf_ = std::make_shared<std::vector<unsigned char> >(bytes());
    data = buf_->empty() ? nullptr : buf_->data();
  }
  Mat clone() const {
    Mat m;
    if (data) { m.create(rows, cols, type_); std::memcpy(m.data, data, bytes()); }
    return m;
  }
  void release() { buf_.reset(); data = nullptr; rows = cols = 0; }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  int type() const { return type_; }
  template <typename T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * cols * elem()); }
  template <typename T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * cols * elem()); }
  template <typename T> T& at(int r, int c) { return ptr<T>(r)[c]; }
  template <typename T> const T& at(int r, int c) const { return ptr<T>(r)[c]; }

 private:
  size_t elem() const { return type_ == CV_32F ? 4 : 1; }
  size_t bytes() const { return (size_t)rows * cols * elem(); }
  int type_ = CV_8U;
  std::shared_ptr<std::vector<unsigned char> > buf_;
};

// ---- YAML storage: compile-only ----
class FileNodeIterator;
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
