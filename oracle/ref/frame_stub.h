// oracle/_ref: what the reference's include/Frame.h + src/Frame.cc need beyond mapobj_stub.h to compile as they are:
// Converter (g2o behind it) and LocalMapping.h are skipped through their include guards.  TEST INFRASTRUCTURE ONLY.
#ifndef PLO_REF_FRAME_STUB_H
#define PLO_REF_FRAME_STUB_H
#include "mapobj_stub.h"
namespace ORB_SLAM2 {
class Converter {   // include/Converter.h:36 -- the one member Frame.cc uses (ComputeBoW)
 public:
  static std::vector<cv::Mat> toDescriptorVector(const cv::Mat& d) {
    std::vector<cv::Mat> v;
    for (int i = 0; i < d.rows; i++) v.push_back(d.row(i));
    return v;
  }
};
}  // namespace ORB_SLAM2
#endif
