// oracle/_ref: the reference's line path compiled from the sources where they lie (oracle/ref/build_ref.sh):
//   src/LineExtractor.cpp (LINEextractor::operator(), :26-93) + include/LineExtractor.h + include/auxiliar.h
//   Thirdparty/line_descriptor/src/LSDDetector_custom.cpp   (detectImpl: Vec4f -> KeyLine, mask filter)
//   Thirdparty/line_descriptor/src/binary_descriptor_custom.cpp (compute -> computeSobel / computeLBD / binaryConversion)
// LineExtractor.cpp is written against opencv_contrib's line_descriptor module, of which the tree only vendors this
// customised twin: oracle/ref/stub/opencv2/line_descriptor/descriptor.hpp maps the names.  OpenCV / Eigen types come
// from oracle/ref/stub; cv::LineSegmentDetector, cv::GaussianBlur, cv::Sobel and cv::LineIterator::count are the ORACLE's
// restatements (oracle/lsd.cc, img_ops.cc), so what this pins is the in-tree control logic and float arithmetic on top of
// them: KeyLine construction, response sort / keep, LBD band accumulation and binarisation, line equations.
// TEST INFRASTRUCTURE ONLY.
#include <cstdint>
#include <cstring>
#include <vector>

#include "LineExtractor.h"

static_assert(sizeof(cv::line_descriptor::KeyLine) == sizeof(plo_keyline), "KeyLine layout");

extern "C" {

void* ref_line_create(int num_octaves, float scale, unsigned n_features, double min_line_length) {
  return new ORB_SLAM2::LINEextractor(num_octaves, scale, n_features, min_line_length);
}
void ref_line_destroy(void* h) { delete static_cast<ORB_SLAM2::LINEextractor*>(h); }

// returns the number of keylines (-1 if cap is too small, -3 if the reference threw -- cv::pyrDown's size assertion with more than
// one octave and (int)scale != 2)
int ref_line_extract(void* h, const uint8_t* img, int rows, int cols, size_t step, const uint8_t* mask, size_t mstep,
                     plo_keyline* kls, uint8_t* desc, double* linefn, int cap) {
  ORB_SLAM2::LINEextractor* ex = static_cast<ORB_SLAM2::LINEextractor*>(h);
  cv::Mat image(rows, cols, CV_8UC1, const_cast<uint8_t*>(img), step), m, d;
  if (mask) m = cv::Mat(rows, cols, CV_8UC1, const_cast<uint8_t*>(mask), mstep);
  std::vector<cv::line_descriptor::KeyLine> k;
  std::vector<Eigen::Vector3d> fn;
  try {
    (*ex)(image, m, k, d, fn);
  } catch (const std::exception&) {
    return -3;
  }
  const int n = (int)k.size();
  if (n > cap) return -1;
  for (int i = 0; i < n; i++) {
    std::memcpy(&kls[i], &k[i], sizeof(plo_keyline));
    std::memcpy(desc + (size_t)i * 32, d.ptr<uint8_t>(i), 32);
    linefn[3 * i] = fn[i](0); linefn[3 * i + 1] = fn[i](1); linefn[3 * i + 2] = fn[i](2);
  }
  return n;
}

}  // extern "C"
