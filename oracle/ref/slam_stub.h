// Stand-ins for the reference's SLAM object model (include/MapPoint.h, KeyFrame.h, Frame.h), just the members that
// src/ORBmatcher.cc touches, so that the matcher's own code compiles from the source where it lies
// (oracle/ref/build_ref.sh: -DMAPPOINT_H -DKEYFRAME_H -DFRAME_H -include slam_stub.h).  The harness (ref_matcher.cc) fills
// them from flat arrays.  TEST INFRASTRUCTURE ONLY.
//
// What is real reference code in the resulting library: every function of ORBmatcher.cc.  What is NOT: the methods of these
// stand-in classes.  The grid lookups Frame::GetFeaturesInArea / KeyFrame::GetFeaturesInArea are forwarded to the
// oracle's restatement (oracle/frame_search.cc), MapPoint::PredictScale returns the level the harness stored.
#ifndef PLO_REF_SLAM_STUB_H
#define PLO_REF_SLAM_STUB_H
#include <map>
#include <set>
#include <vector>

#include <opencv2/core/core.hpp>

#include "Thirdparty/DBoW2/DBoW2/BowVector.h"
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"

using namespace std;   // the reference's own headers do this, and ORBmatcher.h relies on it (unqualified vector / pair)

namespace ORB_SLAM2 {

class KeyFrame;
class Frame;

class MapPoint {
 public:
  // tracking state written by Frame::isInFrustum in the reference
  float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = -1;
  bool mbTrackInView = false;
  int mnTrackScaleLevel = 0;
  float mTrackViewCos = 1;
  long unsigned int mnLastFrameSeen = 0, mnFuseCandidateForKF = 0, mnId = 0;
  // harness data
  bool bad = false;
  cv::Mat desc, pos, normal;
  float minDist = 0, maxDist = 1e30f;
  int nobs = 1, predicted = 0;
  std::map<KeyFrame*, size_t> obs;
  MapPoint* replaced = nullptr;

  bool isBad() const { return bad; }
  cv::Mat GetDescriptor() const { return desc.clone(); }
  cv::Mat GetWorldPos() const { return pos.clone(); }
  cv::Mat GetNormal() const { return normal.clone(); }
  float GetMinDistanceInvariance() const { return minDist; }
  float GetMaxDistanceInvariance() const { return maxDist; }
  int PredictScale(const float&, KeyFrame*) const { return predicted; }
  int PredictScale(const float&, Frame*) const { return predicted; }
  int Observations() const { return nobs; }
  void AddObservation(KeyFrame* k, size_t i) { obs[k] = i; }
  bool IsInKeyFrame(KeyFrame* k) const { return obs.count(k) != 0; }
  int GetIndexInKeyFrame(KeyFrame* k) const { auto it = obs.find(k); return it == obs.end() ? -1 : (int)it->second; }
  void Replace(MapPoint* p) { replaced = p; }
};

struct GridLookup {   // the oracle's Frame::GetFeaturesInArea on a CSR grid (oracle/frame_search.cc)
  std::vector<plo_keypoint> kps;
  std::vector<int32_t> cellStart, cellItems;
  float gp[6] = {0, 0, 0, 0, 0, 0};
  void build();
  std::vector<size_t> query(float x, float y, float r, int minLevel, int maxLevel) const;
};

class Frame {
 public:
  std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
  std::vector<MapPoint*> mvpMapPoints;
  std::vector<bool> mvbOutlier;
  std::vector<float> mvuRight, mvDepth, mvScaleFactors, mvInvLevelSigma2, mvLevelSigma2;
  cv::Mat mDescriptors, mTcw;
  DBoW2::BowVector mBowVec;
  DBoW2::FeatureVector mFeatVec;
  int N = 0;
  float fx = 1, fy = 1, cx = 0, cy = 0, mb = 0, mbf = 0;
  static float mnMinX, mnMaxX, mnMinY, mnMaxY;
  GridLookup grid;
  std::vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r, const int minLevel = -1,
                                        const int maxLevel = -1) const {
    return grid.query(x, y, r, minLevel, maxLevel);
  }
};

class KeyFrame {
 public:
  std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
  std::vector<MapPoint*> mvpMapPoints;
  std::vector<float> mvuRight, mvDepth, mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
  cv::Mat mDescriptors, Rcw, tcw, Ow;
  DBoW2::BowVector mBowVec;
  DBoW2::FeatureVector mFeatVec;
  int N = 0;
  float fx = 1, fy = 1, cx = 0, cy = 0, mbf = 0, mb = 0;
  long unsigned int mnId = 0;
  int mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;
  GridLookup grid;
  std::vector<MapPoint*> GetMapPointMatches() const { return mvpMapPoints; }
  std::set<MapPoint*> GetMapPoints() const {
    std::set<MapPoint*> s;
    for (MapPoint* p : mvpMapPoints) if (p && !p->isBad()) s.insert(p);
    return s;
  }
  MapPoint* GetMapPoint(const size_t& i) const { return mvpMapPoints[i]; }
  void AddMapPoint(MapPoint* p, const size_t& i) { mvpMapPoints[i] = p; }
  cv::Mat GetRotation() const { return Rcw.clone(); }
  cv::Mat GetTranslation() const { return tcw.clone(); }
  cv::Mat GetCameraCenter() const { return Ow.clone(); }
  bool IsInImage(const float& x, const float& y) const { return x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY; }
  std::vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r) const { return grid.query(x, y, r, -1, -1); }
};

}  // namespace ORB_SLAM2
#endif
