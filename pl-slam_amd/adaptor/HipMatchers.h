// Adaptor for the Hamming matching of the reference's ORBmatcher / LSDmatcher over the C ABI of libplslam_hip.so.
//
// The reference's matcher classes take Frame / KeyFrame / MapPoint objects (include/ORBmatcher.h:37-102,
// include/LSDmatcher.h:22-76); the descriptor work inside them only needs flat arrays.  These helpers take exactly
// the members the reference methods read, so the bodies of
//     ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)     src/ORBmatcher.cc:187-327
//     LSDmatcher::SearchDouble(Frame&, Frame&, vector<int>&)             src/LSDmatcher.cpp:427-460
//     LSDmatcher::SearchDouble(KeyFrame*, Frame&)                        src/LSDmatcher.cpp:375-425
//     ORBmatcher::SearchForInitialization(F1, F2, prev, matches, win)     src/ORBmatcher.cc:455-572
//     ORBmatcher::SearchByProjection(F, MapPoints, th) / (Cur, Last, th)  src/ORBmatcher.cc:56-144, 1441-1585
//     LSDmatcher::SearchByProjection(Cur, Last, th) / (F, MapLines, th)   src/LSDmatcher.cpp:72-176, 221-338
// become a few lines of glue (shown in INTEGRATION.md) and keep their signatures.
#ifndef PLSLAM_HIP_ADAPTOR_MATCHERS_H
#define PLSLAM_HIP_ADAPTOR_MATCHERS_H

#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include <opencv2/line_descriptor/descriptor.hpp>
#include <Eigen/Core>

#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "plslam_hip.h"
#include "HipResidency.h"

namespace ORB_SLAM2 {
namespace hip {

inline void check(plh_status st) {
  if (st != PLH_OK) throw std::runtime_error(std::string("plslam_hip: ") + plh_last_error());
}

// ORBmatcher::DescriptorDistance / LSDmatcher::DescriptorDistance (ORBmatcher.cc:1764-1780, LSDmatcher.cpp:654-670)
inline int DescriptorDistance(const cv::Mat& a, const cv::Mat& b) { return plh_descriptor_distance(a.ptr<uchar>(), b.ptr<uchar>()); }

// DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned> >) -> node id per feature, -1 where the feature has no word.
template <class FeatureVector>
inline std::vector<int32_t> NodeOfFeature(const FeatureVector& fv, int nFeatures) {
  std::vector<int32_t> node(nFeatures, -1);
  for (typename FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it)
    for (size_t k = 0; k < it->second.size(); k++) node[it->second[k]] = (int32_t)it->first;
  return node;
}

// The descriptor half of ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches).
//   descKF / keysKF / nodeKF : pKF->mDescriptors, pKF->mvKeysUn, NodeOfFeature(pKF->mFeatVec, N)
//   validKF[i]               : vpMapPointsKF[i] != NULL && !vpMapPointsKF[i]->isBad()
//   descF / keysF / nodeF    : F.mDescriptors, F.mvKeys, NodeOfFeature(F.mFeatVec, F.N)
// Returns nmatches; matchKF[j] = KeyFrame feature whose MapPoint is assigned to Frame feature j (or -1).
inline int SearchByBoW(const cv::Mat& descKF, const std::vector<cv::KeyPoint>& keysKF, const std::vector<int32_t>& nodeKF,
                       const std::vector<uchar>& validKF, const cv::Mat& descF, const std::vector<cv::KeyPoint>& keysF,
                       const std::vector<int32_t>& nodeF, float nnratio, bool checkOri, std::vector<int>& matchKF,
                       int TH_LOW = 50, int device = 0) {
  const int n1 = descKF.rows, n2 = descF.rows;
  matchKF.assign(n2, -1);
  if (n1 == 0 || n2 == 0) return 0;
  std::vector<float> a1(n1), a2(n2);
  for (int i = 0; i < n1; i++) a1[i] = keysKF[i].angle;
  for (int j = 0; j < n2; j++) a2[j] = keysF[j].angle;
  cv::Mat d1 = descKF.isContinuous() ? descKF : descKF.clone(), d2 = descF.isContinuous() ? descF : descF.clone();
  int nmatches = 0;
  check(plh_orb_search_by_bow(d1.ptr<uchar>(), a1.data(), nodeKF.data(), validKF.data(), n1, d2.ptr<uchar>(), a2.data(), nodeF.data(),
                              n2, TH_LOW, nnratio, checkOri ? 1 : 0, matchKF.data(), &nmatches, device));
  return nmatches;
}

// LSDmatcher::SearchDouble on the two LBD descriptor matrices (InitialFrame.mLdesc, CurrentFrame.mLdesc):
// FrameBFMatch both ways (knnMatch k=2 + MAD filter + ratio) and the mutual-consistency check.
inline int SearchDouble(const cv::Mat& ldesc1, const cv::Mat& ldesc2, std::vector<int>& LineMatches, float nnratio,
                        float TH_LOW = 50.f, int device = 0) {
  LineMatches.assign(ldesc1.rows, -1);
  if (ldesc1.rows == 0 || ldesc2.rows == 0) return 0;
  cv::Mat d1 = ldesc1.isContinuous() ? ldesc1 : ldesc1.clone(), d2 = ldesc2.isContinuous() ? ldesc2 : ldesc2.clone();
  int nmatches = 0;
  check(plh_line_search_double(d1.ptr<uchar>(), d1.rows, d2.ptr<uchar>(), d2.rows, TH_LOW, nnratio, LineMatches.data(), &nmatches, device));
  return nmatches;
}

// LSDmatcher::FrameBFMatch(ldesc1, ldesc2, LineMatches, TH) (LSDmatcher.cpp:462-486): one direction, no mutual check
inline void FrameBFMatch(const cv::Mat& ldesc1, const cv::Mat& ldesc2, std::vector<int>& LineMatches, float nnratio, float TH,
                         int device = 0) {
  LineMatches.assign(ldesc1.rows, -1);
  if (ldesc1.rows == 0 || ldesc2.rows == 0) return;
  cv::Mat d1 = ldesc1.isContinuous() ? ldesc1 : ldesc1.clone(), d2 = ldesc2.isContinuous() ? ldesc2 : ldesc2.clone();
  check(plh_line_frame_bfmatch(d1.ptr<uchar>(), d1.rows, d2.ptr<uchar>(), d2.rows, TH, nnratio, LineMatches.data(), device));
}

// LSDmatcher::SearchForTriangulationNew's matching (LSDmatcher.cpp:780-832): FrameBFMatchNew both ways over the two fundamental
// matrices, the mutual check, the MapLine gate -- one call.  F21 / F12: 3 x 3 CV_32F, the reference's ComputeF12 (pKF2, pKF1) / (pKF1, pKF2).
template <class KL, class FN>
inline int SearchForTriangulationNew(const cv::Mat& ldesc1, const cv::Mat& ldesc2, const std::vector<KL>& kls1, const std::vector<KL>& kls2,
                                     const std::vector<FN>& func1, const std::vector<FN>& func2, const cv::Mat& F21, const cv::Mat& F12,
                                     const std::vector<unsigned char>& hasML1, const std::vector<unsigned char>& hasML2, float nnratio,
                                     float TH, bool isDouble, std::vector<int>& matches12, int device = 0) {
  const int n1 = ldesc1.rows, n2 = ldesc2.rows;
  matches12.assign(n1, -1);
  if (n1 == 0 || n2 == 0) return 0;
  cv::Mat d1 = ldesc1.isContinuous() ? ldesc1 : ldesc1.clone(), d2 = ldesc2.isContinuous() ? ldesc2 : ldesc2.clone();
  std::vector<float> s1(4 * (size_t)n1), s2(4 * (size_t)n2), f21(9), f12(9);
  std::vector<double> fn1(3 * (size_t)n1), fn2(3 * (size_t)n2);
  for (int i = 0; i < n1; i++) {
    s1[4 * i] = kls1[i].startPointX; s1[4 * i + 1] = kls1[i].startPointY; s1[4 * i + 2] = kls1[i].endPointX; s1[4 * i + 3] = kls1[i].endPointY;
    for (int k = 0; k < 3; k++) fn1[3 * i + k] = func1[i](k);
  }
  for (int i = 0; i < n2; i++) {
    s2[4 * i] = kls2[i].startPointX; s2[4 * i + 1] = kls2[i].startPointY; s2[4 * i + 2] = kls2[i].endPointX; s2[4 * i + 3] = kls2[i].endPointY;
    for (int k = 0; k < 3; k++) fn2[3 * i + k] = func2[i](k);
  }
  for (int i = 0; i < 9; i++) { f21[i] = F21.at<float>(i / 3, i % 3); f12[i] = F12.at<float>(i / 3, i % 3); }
  int nmatches = 0;
  check(plh_line_search_for_triangulation_new(d1.ptr<uchar>(), n1, d2.ptr<uchar>(), n2, s1.data(), s2.data(), fn1.data(), fn2.data(),
                                              f21.data(), f12.data(), hasML1.data(), hasML2.data(), TH, nnratio, isDouble ? 1 : 0,
                                              matches12.data(), &nmatches, device));
  return nmatches;
}

// cv::BFMatcher(NORM_HAMMING, false).knnMatch(q, t, matches, 2) (LSDmatcher.cpp:468-469, 494-495)
inline void knnMatch2(const cv::Mat& q, const cv::Mat& t, std::vector<std::vector<cv::DMatch> >& matches, int device = 0) {
  matches.clear();
  if (q.rows == 0) return;
  std::vector<int32_t> idx((size_t)q.rows * 2), dist((size_t)q.rows * 2);
  cv::Mat qc = q.isContinuous() ? q : q.clone(), tc = t.isContinuous() ? t : t.clone();
  check(plh_hamming_knn2(qc.ptr<uchar>(), qc.rows, tc.ptr<uchar>(), tc.rows, idx.data(), dist.data(), device));
  matches.resize(q.rows);
  for (int i = 0; i < q.rows; i++)
    for (int k = 0; k < 2; k++)
      if (idx[i * 2 + k] >= 0) matches[i].push_back(cv::DMatch(i, idx[i * 2 + k], (float)dist[i * 2 + k]));
}


// ---------------------------------------------------------------------------------------------------------------
// Grid ("windowed") searches.  cv::KeyPoint and KeyLine are layout-identical to plh_keypoint / plh_keyline.
// ---------------------------------------------------------------------------------------------------------------
static_assert(sizeof(cv::KeyPoint) == sizeof(plh_keypoint), "cv::KeyPoint layout");
static_assert(sizeof(cv::line_descriptor::KeyLine) == sizeof(plh_keyline), "KeyLine layout");
static_assert(sizeof(cv::Point2f) == 2 * sizeof(float), "cv::Point2f layout");

// Frame::mnMinX, mnMinY, mnMaxX, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv (static members, Frame.cc:113-117)
inline plh_grid_params GridParams(float mnMinX, float mnMinY, float mnMaxX, float mnMaxY, float widthInv, float heightInv) {
  plh_grid_params g = {mnMinX, mnMinY, mnMaxX, mnMaxY, widthInv, heightInv};
  return g;
}

// ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize): pass F1.mvKeysUn / mDescriptors,
// F2.mvKeysUn / mDescriptors.  vbPrevMatched is updated in place exactly as the reference does (:567-569).
inline int SearchForInitialization(const std::vector<cv::KeyPoint>& keysUn1, const cv::Mat& desc1,
                                   const std::vector<cv::KeyPoint>& keysUn2, const cv::Mat& desc2, const plh_grid_params& gp2,
                                   std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize,
                                   float nnratio, bool checkOri, int device = 0) {
  vnMatches12.assign(keysUn1.size(), -1);
  cv::Mat d1 = desc1.isContinuous() ? desc1 : desc1.clone(), d2 = desc2.isContinuous() ? desc2 : desc2.clone();
  int nmatches = 0;
  check(plh_orb_search_for_initialization(reinterpret_cast<const plh_keypoint*>(keysUn1.data()), d1.ptr<uchar>(), (int)keysUn1.size(),
                                          reinterpret_cast<const plh_keypoint*>(keysUn2.data()), d2.ptr<uchar>(), (int)keysUn2.size(),
                                          &gp2, reinterpret_cast<float*>(vbPrevMatched.data()), windowSize, nnratio, checkOri ? 1 : 0,
                                          vnMatches12.data(), &nmatches, device));
  return nmatches;
}

// What the projection searches read from the map elements, one row per query (MapPoint / MapLine / last-frame feature).
struct ProjQueries {
  std::vector<uchar> valid;      // MapPoints: mbTrackInView && !isBad();  last frame: pMP && !mvbOutlier[i] && invzc >= 0
  std::vector<uchar> hasObs;     // Observations() > 0
  std::vector<float> pos;        // points: (x, y) per query; lines: (x1, y1, x2, y2) per query
  std::vector<int32_t> level;    // mnTrackScaleLevel / LastFrame.mvKeys[i].octave (points only)
  std::vector<float> aux;        // mTrackViewCos (map elements) | mvKeysUn[i].angle (last-frame points) | lineLength (last-frame lines)
  cv::Mat desc;                  // n x 32 CV_8U: GetDescriptor() of every query
};

// The back end's pose-driven searches (relocalisation / loop-closing SearchByProjection, both Fuse overloads, SearchBySim3) walk the map
// points of the call and gate every one between the pose transform and the window lookup (ORBmatcher.cc:1591-1640, 337-395, 945-975,
// 1096-1128, 1206-1290, 1313-1365).  What only the host can answer -- isBad(), "already found", an empty descriptor -- stays in the
// adaptor's loop, which also copies what the gates read from the MapPoint; the arithmetic (transform, projection, image bounds, distance
// range, viewing angle) runs for all points in one call of plh_map_point_gates.
struct MapPointGateArrays {
  std::vector<uchar> pre;                  // the map-side gates
  std::vector<float> pos, normal;          // GetWorldPos(), GetNormal() (3 floats each)
  std::vector<float> minInv, maxInv;       // GetMinDistanceInvariance(), GetMaxDistanceInvariance()
  void assign(size_t n) { pre.assign(n, 0); pos.assign(3 * n, 0.f); normal.assign(3 * n, 0.f); minInv.assign(n, 0.f); maxInv.assign(n, 0.f); }
  void set(size_t i, const cv::Mat& worldPos, const cv::Mat& normalVec, float minDistInv, float maxDistInv) {
    pre[i] = 1;
    for (int k = 0; k < 3; k++) pos[3 * i + k] = worldPos.at<float>(k);
    if (!normalVec.empty())
      for (int k = 0; k < 3; k++) normal[3 * i + k] = normalVec.at<float>(k);
    minInv[i] = minDistInv; maxInv[i] = maxDistInv;
  }
};
inline void PutMat(float* dst, const cv::Mat& m, int rows, int cols) {
  for (int r = 0; r < rows; r++)
    for (int c = 0; c < cols; c++) dst[r * cols + c] = m.at<float>(r, c);
}
// the camera searched in: pose (Ow may be empty when the distance is the camera point's), intrinsics, image bounds
inline plh_point_gates PointGates(int flags, const cv::Mat& Rcw, const cv::Mat& tcw, const cv::Mat& Ow, float fx, float fy, float cx, float cy,
                                  float minX, float minY, float maxX, float maxY) {
  plh_point_gates g;
  std::memset(&g, 0, sizeof(g));
  PutMat(g.view.Rcw, Rcw, 3, 3);
  PutMat(g.view.tcw, tcw, 3, 1);
  if (!Ow.empty()) PutMat(g.view.Ow, Ow, 3, 1);
  g.view.fx = fx; g.view.fy = fy; g.view.cx = cx; g.view.cy = cy;
  g.view.min_x = minX; g.view.min_y = minY; g.view.max_x = maxX; g.view.max_y = maxY;
  g.view.log_scale_factor = 1.f; g.view.n_scale_levels = 1;   // (PredictScale stays with the MapPoint: mfMaxDistance is protected)
  g.flags = flags;
  return g;
}
// q.valid = pre && gates, q.pos = (u, v); dist[i] = the distance PredictScale wants, for the points that passed
inline void MapPointGates(const plh_point_gates& g, const MapPointGateArrays& a, ProjQueries& q, std::vector<float>& dist, int device = 0) {
  const int n = (int)a.pre.size();
  q.valid = a.pre;
  q.pos.assign(2 * (size_t)n, 0.f);
  q.level.assign(n, 0);
  dist.assign(n, 0.f);
  if (n == 0) return;
  check(plh_map_point_gates(&g, n, a.pos.data(), a.normal.data(), a.minInv.data(), a.maxInv.data(), NULL, q.valid.data(), q.pos.data(),
                            dist.data(), q.level.data(), device));
}

// ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>&, th).  occupied[idx] = F.mvpMapPoints[idx] != NULL &&
// Observations() > 0 (in/out).  assigned[idx] = query whose MapPoint the reference stores in F.mvpMapPoints[idx], or -1.
inline int SearchByProjection(const std::vector<cv::KeyPoint>& keysUn, const cv::Mat& desc, const plh_grid_params& gp,
                              const std::vector<float>& scaleFactors, std::vector<uchar>& occupied, const ProjQueries& q, float th,
                              float nnratio, std::vector<int>& assigned, int device = 0) {
  assigned.assign(keysUn.size(), -1);
  cv::Mat d = desc.isContinuous() ? desc : desc.clone(), qd = q.desc.isContinuous() ? q.desc : q.desc.clone();
  int nmatches = 0;
  check(plh_orb_search_by_projection_mp(reinterpret_cast<const plh_keypoint*>(keysUn.data()), d.ptr<uchar>(), (int)keysUn.size(), &gp,
                                        scaleFactors.data(), (int)scaleFactors.size(), occupied.data(), (int)q.valid.size(),
                                        q.valid.data(), q.pos.data(), q.level.data(), q.aux.data(), qd.ptr<uchar>(), q.hasObs.data(),
                                        th, nnratio, assigned.data(), &nmatches, device));
  return nmatches;
}

// ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono); mode 0 mono, 1 forward, 2 backward.
inline int SearchByProjectionLastFrame(const std::vector<cv::KeyPoint>& keysUn, const cv::Mat& desc, const plh_grid_params& gp,
                                       const std::vector<float>& scaleFactors, std::vector<uchar>& occupied, const ProjQueries& q,
                                       float th, int mode, bool checkOri, std::vector<int>& assigned, int device = 0) {
  assigned.assign(keysUn.size(), -1);
  cv::Mat d = desc.isContinuous() ? desc : desc.clone(), qd = q.desc.isContinuous() ? q.desc : q.desc.clone();
  int nmatches = 0;
  check(plh_orb_search_by_projection_frame(reinterpret_cast<const plh_keypoint*>(keysUn.data()), d.ptr<uchar>(), (int)keysUn.size(),
                                           &gp, scaleFactors.data(), (int)scaleFactors.size(), occupied.data(), (int)q.valid.size(),
                                           q.valid.data(), q.pos.data(), q.level.data(), q.aux.data(), qd.ptr<uchar>(),
                                           q.hasObs.data(), th, mode, checkOri ? 1 : 0, assigned.data(), &nmatches, device));
  return nmatches;
}

// LSDmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th)   (lastFrame = true,  aux = lineLength)
// LSDmatcher::SearchByProjection(Frame& F, const vector<MapLine*>&, th)             (lastFrame = false, aux = mTrackViewCos)
inline int LineSearchByProjection(const std::vector<cv::line_descriptor::KeyLine>& keylinesUn, const cv::Mat& ldesc,
                                  const std::vector<Eigen::Vector3d>& lineFunctions, const plh_grid_params& gp,
                                  std::vector<uchar>& occupied, const ProjQueries& q, float th, float nnratio, bool lastFrame,
                                  std::vector<int>& assigned, int device = 0) {
  assigned.assign(keylinesUn.size(), -1);
  cv::Mat d = ldesc.isContinuous() ? ldesc : ldesc.clone(), qd = q.desc.isContinuous() ? q.desc : q.desc.clone();
  static_assert(sizeof(Eigen::Vector3d) == 3 * sizeof(double), "Eigen::Vector3d layout");
  const double* fn = reinterpret_cast<const double*>(lineFunctions.data());
  const plh_keyline* kl = reinterpret_cast<const plh_keyline*>(keylinesUn.data());
  int nmatches = 0;
  if (lastFrame)
    check(plh_line_search_by_projection_frame(kl, d.ptr<uchar>(), fn, (int)keylinesUn.size(), &gp, occupied.data(), (int)q.valid.size(),
                                              q.valid.data(), q.pos.data(), q.aux.data(), qd.ptr<uchar>(), q.hasObs.data(), th,
                                              assigned.data(), &nmatches, device));
  else
    check(plh_line_search_by_projection_ml(kl, d.ptr<uchar>(), fn, (int)keylinesUn.size(), &gp, occupied.data(), (int)q.valid.size(),
                                           q.valid.data(), q.pos.data(), q.aux.data(), qd.ptr<uchar>(), q.hasObs.data(), th, nnratio,
                                           assigned.data(), &nmatches, device));
  return nmatches;
}

// ---------------------------------------------------------------------------------------------------------------
// The same tracking-path searches on RESIDENT frames (round 6, HipResidency.h): keypoints, descriptors and grid already lie on the
// device; the call uploads its queries only.  Arguments as above without the frame-side arrays.
// ---------------------------------------------------------------------------------------------------------------
inline int SearchByProjectionResident(const plh_frame_points* f, const std::vector<float>& scaleFactors, std::vector<uchar>& occupied,
                                      const ProjQueries& q, float th, float nnratio, std::vector<int>& assigned) {
  assigned.assign(plh_frame_points_count(f), -1);
  cv::Mat qd = q.desc.isContinuous() ? q.desc : q.desc.clone();
  int nmatches = 0;
  check(plh_orb_search_by_projection_mp_resident(f, scaleFactors.data(), (int)scaleFactors.size(), occupied.data(), (int)q.valid.size(),
                                                 q.valid.data(), q.pos.data(), q.level.data(), q.aux.data(), qd.ptr<uchar>(),
                                                 q.hasObs.data(), th, nnratio, assigned.data(), &nmatches));
  return nmatches;
}
inline int SearchByProjectionLastFrameResident(const plh_frame_points* f, const std::vector<float>& scaleFactors, std::vector<uchar>& occupied,
                                               const ProjQueries& q, float th, int mode, bool checkOri, std::vector<int>& assigned) {
  assigned.assign(plh_frame_points_count(f), -1);
  cv::Mat qd = q.desc.isContinuous() ? q.desc : q.desc.clone();
  int nmatches = 0;
  check(plh_orb_search_by_projection_frame_resident(f, scaleFactors.data(), (int)scaleFactors.size(), occupied.data(), (int)q.valid.size(),
                                                    q.valid.data(), q.pos.data(), q.level.data(), q.aux.data(), qd.ptr<uchar>(),
                                                    q.hasObs.data(), th, mode, checkOri ? 1 : 0, assigned.data(), &nmatches));
  return nmatches;
}
// ... the same search with the projection on the device: q.pos holds WORLD positions (3 floats per query), `view` the current pose
inline int SearchByProjectionLastFrameResidentWorld(const plh_frame_points* f, const std::vector<float>& scaleFactors, std::vector<uchar>& occupied,
                                                    const plh_frame_view& view, const ProjQueries& q, float th, int mode, bool checkOri,
                                                    std::vector<int>& assigned) {
  const int n = plh_frame_points_count(f), nq = (int)q.valid.size();
  assigned.assign(n, -1);
  int nm = 0;
  check(plh_orb_search_by_projection_frame_resident_world(f, scaleFactors.data(), (int)scaleFactors.size(), occupied.data(), nq, &view,
                                                          q.valid.data(), q.pos.data(), q.level.data(), q.aux.data(), q.desc.ptr<uchar>(),
                                                          q.hasObs.data(), th, mode, checkOri ? 1 : 0, assigned.data(), &nm));
  return nm;
}
inline int SearchForInitializationResident(const plh_frame_points* f1, const plh_frame_points* f2, std::vector<cv::Point2f>& vbPrevMatched,
                                           std::vector<int>& vnMatches12, int windowSize, float nnratio, bool checkOri) {
  vnMatches12.assign(plh_frame_points_count(f1), -1);
  int nmatches = 0;
  check(plh_orb_search_for_initialization_resident(f1, f2, reinterpret_cast<float*>(vbPrevMatched.data()), windowSize, nnratio,
                                                   checkOri ? 1 : 0, vnMatches12.data(), &nmatches));
  return nmatches;
}
// kf / f carry their FeatureVector nodes (FrameResidency::Points(..., &node))
inline int SearchByBoWResident(const plh_frame_points* kf, const std::vector<uchar>& validKF, const plh_frame_points* f, float nnratio,
                               bool checkOri, std::vector<int>& matchKF, int TH_LOW = 50) {
  matchKF.assign(plh_frame_points_count(f), -1);
  int nmatches = 0;
  check(plh_orb_search_by_bow_resident(kf, validKF.data(), f, TH_LOW, nnratio, checkOri ? 1 : 0, matchKF.data(), &nmatches));
  return nmatches;
}
inline int LineSearchByProjectionResident(const plh_frame_lines* f, std::vector<uchar>& occupied, const ProjQueries& q, float th,
                                          float nnratio, bool lastFrame, std::vector<int>& assigned) {
  assigned.assign(plh_frame_lines_count(f), -1);
  cv::Mat qd = q.desc.isContinuous() ? q.desc : q.desc.clone();
  int nmatches = 0;
  if (lastFrame)
    check(plh_line_search_by_projection_frame_resident(f, occupied.data(), (int)q.valid.size(), q.valid.data(), q.pos.data(), q.aux.data(),
                                                       qd.ptr<uchar>(), q.hasObs.data(), th, assigned.data(), &nmatches));
  else
    check(plh_line_search_by_projection_ml_resident(f, occupied.data(), (int)q.valid.size(), q.valid.data(), q.pos.data(), q.aux.data(),
                                                    qd.ptr<uchar>(), q.hasObs.data(), th, nnratio, assigned.data(), &nmatches));
  return nmatches;
}
inline int SearchDoubleResident(const plh_frame_lines* l1, const plh_frame_lines* l2, std::vector<int>& LineMatches, float nnratio,
                                float TH_LOW = 50.f) {
  LineMatches.assign(plh_frame_lines_count(l1), -1);
  int nmatches = 0;
  check(plh_line_search_double_resident(l1, l2, TH_LOW, nnratio, LineMatches.data(), &nmatches));
  return nmatches;
}

// ---------------------------------------------------------------------------------------------------------------
// Back-end searches (LoopClosing / LocalMapping).  The pose algebra and the pre-checks of every map point stay with the
// caller, in the reference's own expressions; a query carries what is left: validity, projection, predicted level, descriptor.
// ---------------------------------------------------------------------------------------------------------------
// ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vpMatches12) (ORBmatcher.cc:574-709): match12[i1] = feature of pKF2 or -1
inline int SearchByBoWKFKF(const std::vector<cv::KeyPoint>& keysUn1, const cv::Mat& desc1, const std::vector<int32_t>& node1,
                           const std::vector<uchar>& valid1, const std::vector<cv::KeyPoint>& keysUn2, const cv::Mat& desc2,
                           const std::vector<int32_t>& node2, const std::vector<uchar>& valid2, float nnratio, bool checkOri,
                           std::vector<int>& match12, int TH_LOW = 50, int device = 0) {
  match12.assign(keysUn1.size(), -1);
  if (keysUn1.empty() || keysUn2.empty()) return 0;
  cv::Mat d1 = desc1.isContinuous() ? desc1 : desc1.clone(), d2 = desc2.isContinuous() ? desc2 : desc2.clone();
  int nmatches = 0;
  check(plh_orb_search_by_bow_kfkf(reinterpret_cast<const plh_keypoint*>(keysUn1.data()), d1.ptr<uchar>(), node1.data(), valid1.data(),
                                   (int)keysUn1.size(), reinterpret_cast<const plh_keypoint*>(keysUn2.data()), d2.ptr<uchar>(), node2.data(),
                                   valid2.data(), (int)keysUn2.size(), TH_LOW, nnratio, checkOri ? 1 : 0, match12.data(), &nmatches, device));
  return nmatches;
}

// ORBmatcher::SearchBySim3 (ORBmatcher.cc:1199-1439): q12 = KeyFrame 1's points projected into KeyFrame 2 (one query per keypoint slot of
// KeyFrame 1), q21 the reverse; match12[i1] = keypoint of KeyFrame 2 both directions agree on, or -1
inline int SearchBySim3(const std::vector<cv::KeyPoint>& keysUn1, const cv::Mat& desc1, const std::vector<cv::KeyPoint>& keysUn2,
                        const cv::Mat& desc2, const plh_grid_params& gp, const std::vector<float>& scaleFactors, const ProjQueries& q12,
                        const ProjQueries& q21, float th, std::vector<int>& match12, int TH_HIGH = 100, int device = 0) {
  match12.assign(keysUn1.size(), -1);
  if (keysUn1.empty() || keysUn2.empty()) return 0;
  cv::Mat d1 = desc1.isContinuous() ? desc1 : desc1.clone(), d2 = desc2.isContinuous() ? desc2 : desc2.clone();
  cv::Mat e12 = q12.desc.isContinuous() ? q12.desc : q12.desc.clone(), e21 = q21.desc.isContinuous() ? q21.desc : q21.desc.clone();
  int nfound = 0;
  check(plh_orb_search_by_sim3(reinterpret_cast<const plh_keypoint*>(keysUn1.data()), d1.ptr<uchar>(), (int)keysUn1.size(),
                               reinterpret_cast<const plh_keypoint*>(keysUn2.data()), d2.ptr<uchar>(), (int)keysUn2.size(), &gp,
                               scaleFactors.data(), (int)scaleFactors.size(), q12.valid.data(), q12.pos.data(), q12.level.data(),
                               e12.ptr<uchar>(), q21.valid.data(), q21.pos.data(), q21.level.data(), e21.ptr<uchar>(), th, TH_HIGH,
                               match12.data(), &nfound, device));
  return nfound;
}

// ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, false) (ORBmatcher.cc:720-912): F = F12 row-major;
// match12[i1] = feature of pKF2 or -1
inline int SearchForTriangulation(const std::vector<cv::KeyPoint>& keysUn1, const cv::Mat& desc1, const std::vector<int32_t>& node1,
                                  const std::vector<uchar>& hasMP1, const std::vector<cv::KeyPoint>& keysUn2, const cv::Mat& desc2,
                                  const std::vector<int32_t>& node2, const std::vector<uchar>& hasMP2, const float F[9], float ex, float ey,
                                  const std::vector<float>& scaleFactors2, const std::vector<float>& levelSigma2_2, bool checkOri,
                                  std::vector<int>& match12, int TH_LOW = 50, int device = 0) {
  match12.assign(keysUn1.size(), -1);
  if (keysUn1.empty() || keysUn2.empty()) return 0;
  cv::Mat d1 = desc1.isContinuous() ? desc1 : desc1.clone(), d2 = desc2.isContinuous() ? desc2 : desc2.clone();
  int nmatches = 0;
  check(plh_orb_search_for_triangulation(reinterpret_cast<const plh_keypoint*>(keysUn1.data()), d1.ptr<uchar>(), node1.data(), hasMP1.data(),
                                         (int)keysUn1.size(), reinterpret_cast<const plh_keypoint*>(keysUn2.data()), d2.ptr<uchar>(),
                                         node2.data(), hasMP2.data(), (int)keysUn2.size(), F, ex, ey, scaleFactors2.data(),
                                         levelSigma2_2.data(), (int)scaleFactors2.size(), TH_LOW, checkOri ? 1 : 0, match12.data(), &nmatches,
                                         device));
  return nmatches;
}

// ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:329-453): occupied[idx] =
// vpMatched[idx] != NULL (in/out); assigned[idx] = query whose MapPoint goes to vpMatched[idx], or -1
inline int SearchByProjectionSim3(const std::vector<cv::KeyPoint>& keysUn, const cv::Mat& desc, const plh_grid_params& gp,
                                  const std::vector<float>& scaleFactors, std::vector<uchar>& occupied, const ProjQueries& q, float th,
                                  std::vector<int>& assigned, int TH_LOW = 50, int device = 0) {
  assigned.assign(keysUn.size(), -1);
  if (keysUn.empty() || q.valid.empty()) return 0;
  cv::Mat d = desc.isContinuous() ? desc : desc.clone(), qd = q.desc.isContinuous() ? q.desc : q.desc.clone();
  int nmatches = 0;
  check(plh_orb_search_by_projection_sim3(reinterpret_cast<const plh_keypoint*>(keysUn.data()), d.ptr<uchar>(), (int)keysUn.size(), &gp,
                                          scaleFactors.data(), (int)scaleFactors.size(), occupied.data(), (int)q.valid.size(),
                                          q.valid.data(), q.pos.data(), q.level.data(), qd.ptr<uchar>(), q.hasObs.data(), th, TH_LOW,
                                          assigned.data(), &nmatches, device));
  return nmatches;
}

// ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, sAlreadyFound, th, ORBdist) (ORBmatcher.cc:1587-1716): q.aux =
// pKF->mvKeysUn[i].angle, occupied[i2] = CurrentFrame.mvpMapPoints[i2] != NULL (in/out)
inline int SearchByProjectionKeyFrame(const std::vector<cv::KeyPoint>& keysUn, const cv::Mat& desc, const plh_grid_params& gp,
                                      const std::vector<float>& scaleFactors, std::vector<uchar>& occupied, const ProjQueries& q, float th,
                                      int ORBdist, bool checkOri, std::vector<int>& assigned, int device = 0) {
  assigned.assign(keysUn.size(), -1);
  if (keysUn.empty() || q.valid.empty()) return 0;
  cv::Mat d = desc.isContinuous() ? desc : desc.clone(), qd = q.desc.isContinuous() ? q.desc : q.desc.clone();
  int nmatches = 0;
  check(plh_orb_search_by_projection_kf(reinterpret_cast<const plh_keypoint*>(keysUn.data()), d.ptr<uchar>(), (int)keysUn.size(), &gp,
                                        scaleFactors.data(), (int)scaleFactors.size(), occupied.data(), (int)q.valid.size(), q.valid.data(),
                                        q.pos.data(), q.level.data(), q.aux.data(), qd.ptr<uchar>(), th, ORBdist, checkOri ? 1 : 0,
                                        assigned.data(), &nmatches, device));
  return nmatches;
}

// The search inside ORBmatcher::Fuse(pKF, vpMapPoints, th) (ORBmatcher.cc:914-1061; invLevelSigma2 = pKF->mvInvLevelSigma2) and
// Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (:1063-1197; no chi-square gate: pass an empty invLevelSigma2): bestIdx[query]
inline int FuseSearch(const std::vector<cv::KeyPoint>& keysUn, const cv::Mat& desc, const plh_grid_params& gp,
                      const std::vector<float>& scaleFactors, const std::vector<float>& invLevelSigma2, const ProjQueries& q, float th,
                      std::vector<int>& bestIdx, int TH_LOW = 50, int device = 0) {
  bestIdx.assign(q.valid.size(), -1);
  if (keysUn.empty() || q.valid.empty()) return 0;
  cv::Mat d = desc.isContinuous() ? desc : desc.clone(), qd = q.desc.isContinuous() ? q.desc : q.desc.clone();
  int nfound = 0;
  check(plh_orb_fuse_search(reinterpret_cast<const plh_keypoint*>(keysUn.data()), d.ptr<uchar>(), (int)keysUn.size(), &gp,
                            scaleFactors.data(), invLevelSigma2.empty() ? NULL : invLevelSigma2.data(), (int)scaleFactors.size(),
                            (int)q.valid.size(), q.valid.data(), q.pos.data(), q.level.data(), qd.ptr<uchar>(), th, TH_LOW, bestIdx.data(),
                            &nfound, device));
  return nfound;
}

// The search inside LSDmatcher::Fuse(pKF, vpMapLines, th) (LSDmatcher.cpp:860-1002) with KeyFrame::GetLinesInArea (KeyFrame.cc:647-683):
// q.pos = 4 floats per query (u1, v1, u2, v2); candDesc = the matrix the reference reads the candidates' rows from
inline int LineFuseSearch(const std::vector<cv::line_descriptor::KeyLine>& keylines, const cv::Mat& candDesc,
                          const std::vector<float>& scaleFactorsLine, const ProjQueries& q, float th, std::vector<int>& bestIdx,
                          float cosTH = 0.998f, int TH_LOW = 50, int device = 0) {
  bestIdx.assign(q.valid.size(), -1);
  if (keylines.empty() || q.valid.empty()) return 0;
  cv::Mat d = candDesc.isContinuous() ? candDesc : candDesc.clone(), qd = q.desc.isContinuous() ? q.desc : q.desc.clone();
  int nfound = 0;
  check(plh_line_fuse_search(reinterpret_cast<const plh_keyline*>(keylines.data()), d.ptr<uchar>(), (int)keylines.size(),
                             scaleFactorsLine.data(), (int)scaleFactorsLine.size(), (int)q.valid.size(), q.valid.data(), q.pos.data(),
                             q.level.data(), qd.ptr<uchar>(), th, cosTH, TH_LOW, bestIdx.data(), &nfound, device));
  return nfound;
}

}  // namespace hip
}  // namespace ORB_SLAM2

#endif
