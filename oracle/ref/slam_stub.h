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
#include <opencv2/line_descriptor/descriptor.hpp>
#include <eigen3/Eigen/Core>

#include "Thirdparty/DBoW2/DBoW2/BowVector.h"
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"

using namespace std;   // the reference's own headers do this, and ORBmatcher.h relies on it (unqualified vector / pair)
using namespace cv;
using namespace cv::line_descriptor;
using namespace Eigen;
#define ORB_SLAM2_MAPLINE_H
#include "auxiliar.h"   // the reference's own header: sort functors, SkewSymmetricMatrix, Vector6d

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
  // harness hook: the map point whose descriptor was fetched last = the query a Fuse loop is working on
  static const MapPoint*& lastQuery() { static thread_local const MapPoint* p = nullptr; return p; }
  cv::Mat GetDescriptor() const { lastQuery() = this; return desc.clone(); }
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

class MapLine {   // include/MapLine.h, the members src/LSDmatcher.cpp touches
 public:
  float mTrackProjX1 = 0, mTrackProjY1 = 0, mTrackProjX2 = 0, mTrackProjY2 = 0;
  bool mbTrackInView = false;
  int mnTrackScaleLevel = 0;
  float mTrackViewCos = 1;
  long unsigned int mnLastFrameSeen = 0, mnFuseCandidateForKF = 0, mnId = 0;
  cv::Mat mLDescriptor;
  bool bad = false;
  Vector6d pos;
  Eigen::Vector3d normal;
  float minDist = 0, maxDist = 1e30f;
  int nobs = 1, predicted = 0;
  std::map<KeyFrame*, size_t> obs;
  MapLine* replaced = nullptr;
  bool isBad() const { return bad; }
  cv::Mat GetDescriptor() const { return mLDescriptor.clone(); }
  Vector6d GetWorldPos() const { return pos; }
  Eigen::Vector3d GetNormal() const { return normal; }
  float GetMinDistanceInvariance() const { return minDist; }
  float GetMaxDistanceInvariance() const { return maxDist; }
  int PredictScale(const float&, const float&) const { return predicted; }
  int PredictScale(const float&, KeyFrame*) const { return predicted; }
  int Observations() const { return nobs; }
  void AddObservation(KeyFrame* k, size_t i) { obs[k] = i; }
  static const MapLine*& lastQuery() { static thread_local const MapLine* p = nullptr; return p; }   // harness hook
  bool IsInKeyFrame(KeyFrame* k) const { lastQuery() = this; return obs.count(k) != 0; }
  int GetIndexInKeyFrame(KeyFrame* k) const { auto it = obs.find(k); return it == obs.end() ? -1 : (int)it->second; }
  void Replace(MapLine* p) { replaced = p; }
};

struct LineGridLookup {   // the oracle's Frame::GetFeaturesInAreaForLine on a CSR grid (oracle/frame_search.cc)
  std::vector<plo_keyline> kl;
  std::vector<double> fn;
  std::vector<int32_t> cellStart, cellItems;
  float gp[6] = {0, 0, 0, 0, 0, 0};
  void build();
  std::vector<size_t> query(float x1, float y1, float x2, float y2, float r, float TH) const;
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
  // lines
  int NL = 0;
  std::vector<KeyLine> mvKeylinesUn;
  std::vector<Eigen::Vector3d> mvKeyLineFunctions;
  std::vector<MapLine*> mvpMapLines;
  std::vector<bool> mvbLineOutlier;
  std::vector<float> mvScaleFactorsLine;
  cv::Mat mLdesc, ImageGray, mK;
  LineGridLookup lineGrid;
  std::vector<size_t> GetFeaturesInAreaForLine(const float& x1, const float& y1, const float& x2, const float& y2, const float& r,
                                               const int minLevel = -1, const int maxLevel = -1, const float TH = 0.998) const {
    return lineGrid.query(x1, y1, x2, y2, r, TH);
  }
  std::vector<size_t> GetLinesInArea(const float& x1, const float& y1, const float& x2, const float& y2, const float& r,
                                     const int minLevel = -1, const int maxLevel = -1, const float TH = 0.998) const {
    return lineGrid.query(x1, y1, x2, y2, r, TH);
  }
  bool isInFrustum(MapLine* p, float) const { return p->mbTrackInView; }   // the harness stores the verdict in the MapLine
  void lineDescriptorMAD(std::vector<std::vector<cv::DMatch> >, double&, double&) const {
    std::cerr << "oracle/ref stub: Frame::lineDescriptorMAD is compile-only" << std::endl;
    std::abort();
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
  // lines
  int NL = 0;
  std::vector<KeyLine> mvKeyLines, mvKeylinesUn;
  std::vector<Eigen::Vector3d> mvKeyLineFunctions;
  std::vector<MapLine*> mvpMapLines;
  std::vector<float> mvScaleFactorsLine;
  float mfLogScaleFactorLine = 0;
  cv::Mat mLineDescriptors, ImageGray, mK;
  mutable std::vector<std::pair<const MapLine*, int> > lineLog;   // harness hook: (query, line) of every GetMapLine call
  MapLine* GetMapLine(const size_t& i) const { lineLog.emplace_back(MapLine::lastQuery(), (int)i); return mvpMapLines[i]; }
  std::vector<MapLine*> GetMapLineMatches() const { return mvpMapLines; }
  void AddMapLine(MapLine* p, const size_t& i) { mvpMapLines[i] = p; }
  // include/KeyFrame.h:108; src/KeyFrame.cc:647-683 cannot be compiled on its own, so the body (ref_lsdmatcher.cc) is a
  // restatement and what the Fuse harness pins is the loop around it
  std::vector<size_t> GetLinesInArea(const float& x1, const float& y1, const float& x2, const float& y2, const float& r,
                                     const float TH = 0.998) const;
  std::vector<MapPoint*> GetMapPointMatches() const { return mvpMapPoints; }
  std::set<MapPoint*> GetMapPoints() const {
    std::set<MapPoint*> s;
    for (MapPoint* p : mvpMapPoints) if (p && !p->isBad()) s.insert(p);
    return s;
  }
  mutable std::vector<std::pair<const MapPoint*, int> > getLog;   // harness hook: (query, keypoint) of every GetMapPoint call
  MapPoint* GetMapPoint(const size_t& i) const { getLog.emplace_back(MapPoint::lastQuery(), (int)i); return mvpMapPoints[i]; }
  void AddMapPoint(MapPoint* p, const size_t& i) { mvpMapPoints[i] = p; }
  cv::Mat GetRotation() const { return Rcw.clone(); }
  cv::Mat GetTranslation() const { return tcw.clone(); }
  cv::Mat GetCameraCenter() const { return Ow.clone(); }
  bool IsInImage(const float& x, const float& y) const { return x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY; }
  std::vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r) const { return grid.query(x, y, r, -1, -1); }
};

}  // namespace ORB_SLAM2
#endif
