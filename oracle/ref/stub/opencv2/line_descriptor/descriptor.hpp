// src/LineExtractor.cpp includes opencv_contrib's <opencv2/line_descriptor/descriptor.hpp>, which is not in the tree;
// the tree vendors a customised twin of that module (Thirdparty/line_descriptor), whose detector class is named
// LSDDetectorC.  This glue header maps the contrib names LineExtractor.cpp uses onto the vendored twin.
// TEST INFRASTRUCTURE ONLY (oracle/ref/build_ref.sh).
#ifndef PLO_REF_STUB_LINE_DESCRIPTOR_HPP
#define PLO_REF_STUB_LINE_DESCRIPTOR_HPP
#include "line_descriptor/descriptor_custom.hpp"
namespace cv {
namespace line_descriptor {
class LSDDetector : public LSDDetectorC {
 public:
  static Ptr<LSDDetector> createLSDDetector() { return Ptr<LSDDetector>(new LSDDetector()); }
};
}  // namespace line_descriptor
}  // namespace cv
#endif
