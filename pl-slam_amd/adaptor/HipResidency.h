// Device residency of the frames the matcher adaptors search (round 6).
//
// Tracking searches every frame two to four times -- TrackWithMotionModel or TrackReferenceKeyFrame (Tracking.cc:1321-1357,
// 1151-1159), SearchLocalPoints / SearchLocalLines (:1792-1855), and once more as the LAST frame's successor searches it -- and a
// KeyFrame is searched by every frame tracked against it.  What those searches read from the frame (mvKeysUn, mDescriptors, the
// grid; mvKeylinesUn, mLdesc, mvKeyLineFunctions, the line grid) never changes after the Frame constructor, so the adaptor keeps it
// on the device: a small LRU of plh_frame_points / plh_frame_lines handles keyed by Frame::mnId (a KeyFrame answers to the id of the
// Frame it was made from, KeyFrame::mnFrameId: same features).  A hit uploads nothing and rebuilds no grid; the queries of the
// call are all that crosses PCIe.
//
// Frame ids restart when the tracker is reset (Tracking::Reset sets Frame::nNextId = 0), so a hit is only taken when the entry's
// fingerprint -- feature count, first and last keypoint / keyline, first and last descriptor row, image bounds of the grid -- equals the
// caller's arrays.
// Thread-safe: the map is guarded by a mutex, handles are immutable and reference-counted (an eviction cannot free a handle another
// thread is searching).
#ifndef PLSLAM_HIP_ADAPTOR_RESIDENCY_H
#define PLSLAM_HIP_ADAPTOR_RESIDENCY_H

#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include <opencv2/line_descriptor/descriptor.hpp>
#include <Eigen/Core>

#include <cstring>
#include <list>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "plslam_hip.h"

namespace ORB_SLAM2 {
namespace hip {

struct ResidentPoints {
  plh_frame_points* h = nullptr;
  bool hasNodes = false;
  int n = 0;
  unsigned char print[2 * sizeof(plh_keypoint) + 64 + sizeof(plh_grid_params)];
  ~ResidentPoints() { plh_frame_points_destroy(h); }
};
struct ResidentLines {
  plh_frame_lines* h = nullptr;
  int nl = 0;
  unsigned char print[2 * sizeof(plh_keyline) + 64 + sizeof(plh_grid_params)];
  ~ResidentLines() { plh_frame_lines_destroy(h); }
};

class FrameResidency {
 public:
  static FrameResidency& Instance() {
    static FrameResidency* r = new FrameResidency();   // never destroyed: its handles must not be freed behind the HIP runtime at exit
    return *r;
  }
  void SetCapacity(size_t frames) {
    std::lock_guard<std::mutex> lock(mMutex);
    mCapacity = frames ? frames : 1;
    Trim(mPoints, mPointsIndex);
    Trim(mLines, mLinesIndex);
  }
  void Clear() {
    std::lock_guard<std::mutex> lock(mMutex);
    mPoints.clear(); mPointsIndex.clear(); mLines.clear(); mLinesIndex.clear();
  }
  // counters for tests / tuning: uploads made, searches served from a resident handle
  unsigned long Uploads() const { return mUploads; }
  unsigned long Hits() const { return mHits; }

  // the points of frame `id`; node != NULL: the FeatureVector node of every feature is set on the handle (SearchByBoW)
  std::shared_ptr<ResidentPoints> Points(unsigned long id, const std::vector<cv::KeyPoint>& keysUn, const cv::Mat& desc,
                                         const plh_grid_params& gp, const std::vector<int32_t>* node = nullptr, int device = 0) {
    const int n = (int)keysUn.size();
    unsigned char fp[sizeof(((ResidentPoints*)0)->print)];
    std::memset(fp, 0, sizeof(fp));
    cv::Mat d = desc.isContinuous() ? desc : desc.clone();
    if (n > 0) {
      std::memcpy(fp, &keysUn[0], sizeof(plh_keypoint));
      std::memcpy(fp + sizeof(plh_keypoint), &keysUn[n - 1], sizeof(plh_keypoint));
      std::memcpy(fp + 2 * sizeof(plh_keypoint), d.ptr<uchar>(0), 32);
      std::memcpy(fp + 2 * sizeof(plh_keypoint) + 32, d.ptr<uchar>(n - 1), 32);
    }
    std::memcpy(fp + 2 * sizeof(plh_keypoint) + 64, &gp, sizeof(plh_grid_params));   // (the grid was built for these image bounds)
    std::lock_guard<std::mutex> lock(mMutex);
    std::shared_ptr<ResidentPoints> e;
    auto it = mPointsIndex.find(id);
    if (it != mPointsIndex.end() && it->second->second->n == n && std::memcmp(it->second->second->print, fp, sizeof(fp)) == 0) {
      mPoints.splice(mPoints.begin(), mPoints, it->second);   // most recently used first
      e = it->second->second;
      mHits++;
    } else {
      if (it != mPointsIndex.end()) { mPoints.erase(it->second); mPointsIndex.erase(it); }   // same id, other content: a reset tracker
      e = std::make_shared<ResidentPoints>();
      e->n = n;
      std::memcpy(e->print, fp, sizeof(fp));
      if (plh_frame_points_create(reinterpret_cast<const plh_keypoint*>(keysUn.data()), d.ptr<uchar>(), n, &gp, device, &e->h) != PLH_OK)
        throw std::runtime_error(std::string("plslam_hip: ") + plh_last_error());
      mPoints.emplace_front(id, e);
      mPointsIndex[id] = mPoints.begin();
      mUploads++;
      Trim(mPoints, mPointsIndex);
    }
    if (node && !e->hasNodes) {
      if (plh_frame_points_set_nodes(e->h, node->data()) != PLH_OK) throw std::runtime_error(std::string("plslam_hip: ") + plh_last_error());
      e->hasNodes = true;
    }
    return e;
  }

  std::shared_ptr<ResidentLines> Lines(unsigned long id, const std::vector<cv::line_descriptor::KeyLine>& keylinesUn, const cv::Mat& ldesc,
                                       const std::vector<Eigen::Vector3d>& lineFunctions, const plh_grid_params& gp, int device = 0) {
    static_assert(sizeof(Eigen::Vector3d) == 3 * sizeof(double), "Eigen::Vector3d layout");
    const int nl = (int)keylinesUn.size();
    unsigned char fp[sizeof(((ResidentLines*)0)->print)];
    std::memset(fp, 0, sizeof(fp));
    cv::Mat d = ldesc.isContinuous() ? ldesc : ldesc.clone();
    if (nl > 0) {
      std::memcpy(fp, &keylinesUn[0], sizeof(plh_keyline));
      std::memcpy(fp + sizeof(plh_keyline), &keylinesUn[nl - 1], sizeof(plh_keyline));
      std::memcpy(fp + 2 * sizeof(plh_keyline), d.ptr<uchar>(0), 32);
      std::memcpy(fp + 2 * sizeof(plh_keyline) + 32, d.ptr<uchar>(nl - 1), 32);
    }
    std::memcpy(fp + 2 * sizeof(plh_keyline) + 64, &gp, sizeof(plh_grid_params));
    std::lock_guard<std::mutex> lock(mMutex);
    auto it = mLinesIndex.find(id);
    if (it != mLinesIndex.end() && it->second->second->nl == nl && std::memcmp(it->second->second->print, fp, sizeof(fp)) == 0) {
      mLines.splice(mLines.begin(), mLines, it->second);
      mHits++;
      return it->second->second;
    }
    if (it != mLinesIndex.end()) { mLines.erase(it->second); mLinesIndex.erase(it); }
    std::shared_ptr<ResidentLines> e = std::make_shared<ResidentLines>();
    e->nl = nl;
    std::memcpy(e->print, fp, sizeof(fp));
    if (plh_frame_lines_create(reinterpret_cast<const plh_keyline*>(keylinesUn.data()), d.ptr<uchar>(),
                               reinterpret_cast<const double*>(lineFunctions.data()), nl, &gp, device, &e->h) != PLH_OK)
      throw std::runtime_error(std::string("plslam_hip: ") + plh_last_error());
    mLines.emplace_front(id, e);
    mLinesIndex[id] = mLines.begin();
    mUploads++;
    Trim(mLines, mLinesIndex);
    return e;
  }

 private:
  FrameResidency() : mCapacity(32), mUploads(0), mHits(0) {}
  template <class L, class M>
  void Trim(L& lru, M& index) {
    while (lru.size() > mCapacity) {
      index.erase(lru.back().first);
      lru.pop_back();   // (a handle in use elsewhere lives on in that caller's shared_ptr)
    }
  }
  typedef std::list<std::pair<unsigned long, std::shared_ptr<ResidentPoints> > > PointsList;
  typedef std::list<std::pair<unsigned long, std::shared_ptr<ResidentLines> > > LinesList;
  std::mutex mMutex;
  size_t mCapacity;
  unsigned long mUploads, mHits;
  PointsList mPoints;
  LinesList mLines;
  std::unordered_map<unsigned long, PointsList::iterator> mPointsIndex;
  std::unordered_map<unsigned long, LinesList::iterator> mLinesIndex;
};

}  // namespace hip
}  // namespace ORB_SLAM2

#endif
