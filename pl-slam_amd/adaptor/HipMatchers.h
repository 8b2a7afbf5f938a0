// Adaptor for the Hamming matching of the reference's ORBmatcher / LSDmatcher over the C ABI of libplslam_hip.so.
//
// The reference's matcher classes take Frame / KeyFrame / MapPoint objects (include/ORBmatcher.h:37-102,
// include/LSDmatcher.h:22-76); the descriptor work inside them only needs flat arrays.  These helpers take exactly
// the members the reference methods read, so the bodies of
//     ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)     src/ORBmatcher.cc:187-327
//     LSDmatcher::SearchDouble(Frame&, Frame&, vector<int>&)             src/LSDmatcher.cpp:427-460
//     LSDmatcher::SearchDouble(KeyFrame*, Frame&)                        src/LSDmatcher.cpp:375-425
// become a few lines of glue (shown in INTEGRATION.md) and keep their signatures.
#ifndef PLSLAM_HIP_ADAPTOR_MATCHERS_H
#define PLSLAM_HIP_ADAPTOR_MATCHERS_H

#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>

#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "plslam_hip.h"

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

}  // namespace hip
}  // namespace ORB_SLAM2

#endif
