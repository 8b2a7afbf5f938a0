// oracle/_ref: the reference's own MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:249-314) and the MapLine twin
// (src/MapLine.cpp:256-326), from include/MapPoint.h + src/MapPoint.cc and include/MapLine.h + src/MapLine.cpp compiled as
// they are against the KeyFrame / Frame / Map stand-ins of mapobj_stub.h.  ORBmatcher::DescriptorDistance, which MapPoint.cc
// calls, forwards to the oracle's popcount (src/ORBmatcher.cc itself is pinned by ref_matcher.cc); the MapLine twin uses
// cv::norm(NORM_HAMMING).  The observations live in a std::map keyed by KeyFrame*: the harness allocates its keyframes in
// one array, so the map iterates them in index order.  TEST INFRASTRUCTURE ONLY.
#include <cstdint>
#include <cstring>
#include <vector>

#include "MapPoint.h"
#include "MapLine.h"
#include "ORBmatcher.h"

namespace ORB_SLAM2 {
int ORBmatcher::DescriptorDistance(const cv::Mat& a, const cv::Mat& b) { return plo_descriptor_distance(a.ptr<uchar>(0), b.ptr<uchar>(0)); }
}  // namespace ORB_SLAM2

using namespace ORB_SLAM2;

namespace {
cv::Mat row_mat(const uint8_t* d) {
  cv::Mat m(1, 32, CV_8U);
  std::memcpy(m.data, d, 32);
  return m;
}
}  // namespace

extern "C" {

// rows[n][32]: the observing keyframes' descriptor rows; kf_bad[n]: pKF->isBad().  Returns 1 and the chosen descriptor, or 0
// when the function leaves the descriptor untouched (no usable observation).
int ref_mappoint_distinctive(const uint8_t* rows, int n, const uint8_t* kf_bad, uint8_t* out) {
  Map map;
  std::vector<KeyFrame> kfs(n > 0 ? n : 1);
  for (int i = 0; i < n; i++) {
    kfs[i].mDescriptors = row_mat(rows + (size_t)i * 32);
    kfs[i].mvuRight.assign(1, -1.f);
    kfs[i].bad = kf_bad[i] != 0;
  }
  cv::Mat pos = cv::Mat::zeros(3, 1, CV_32F);
  MapPoint mp(pos, &kfs[0], &map);
  for (int i = 0; i < n; i++) mp.AddObservation(&kfs[i], 0);
  mp.ComputeDistinctiveDescriptors();
  cv::Mat d = mp.GetDescriptor();
  if (d.empty()) return 0;
  std::memcpy(out, d.ptr<uchar>(0), 32);
  return 1;
}

int ref_mapline_distinctive(const uint8_t* rows, int n, const uint8_t* kf_bad, uint8_t* out) {
  Map map;
  std::vector<KeyFrame> kfs(n > 0 ? n : 1);
  for (int i = 0; i < n; i++) {
    kfs[i].mLineDescriptors = row_mat(rows + (size_t)i * 32);
    kfs[i].bad = kf_bad[i] != 0;
  }
  Vector6d pos;
  pos << 0, 0, 1, 1, 0, 1;
  MapLine ml(pos, &kfs[0], &map);
  for (int i = 0; i < n; i++) ml.AddObservation(&kfs[i], 0);
  ml.ComputeDistinctiveDescriptors();
  cv::Mat d = ml.GetDescriptor();
  if (d.empty()) return 0;
  std::memcpy(out, d.ptr<uchar>(0), 32);
  return 1;
}

}  // extern "C"
