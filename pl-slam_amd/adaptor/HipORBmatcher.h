// Drop-in replacement for the reference's include/ORBmatcher.h: the class ORB_SLAM2::ORBmatcher with the reference's
// constructor and method signatures (include/ORBmatcher.h:37-102), whose tracking-path searches run on the GPU through the
// C ABI of libplslam_hip.so, so that Tracking.cc consumes it unchanged:
//     ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th)          src/ORBmatcher.cc:56-144     (Tracking.cc:1800)
//     ORBmatcher::SearchByProjection(Frame& Cur, const Frame& Last, th, bMono)      :1441-1585                   (Tracking.cc:1321-1357)
//     ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)                :187-327                     (Tracking.cc:1151-1159)
//     ORBmatcher::SearchForInitialization(F1, F2, prevMatched, matches12, window)   :455-572                     (Tracking.cc:706-711)
//     ORBmatcher::DescriptorDistance                                                :1764-1780
// How it coexists with the rest of the reference (plslam_hip_dropin.h, force-included into every translation unit): the
// reference's own header is read here, once, under another class name (ORBmatcher -> ORBmatcherCPU), which also sets its
// include guard, so every later `#include "ORBmatcher.h"` in the tree is a no-op; the class below derives from it, re-declares
// the overloads above -- and the back end's: SearchByBoW(KeyFrame*, KeyFrame*) :574-709, SearchForTriangulation :720-912, both
// Fuse :914-1197, SearchBySim3 :1199-1439, the loop-closing :329-453 and relocalisation :1587-1716 SearchByProjection -- and
// inherits the helpers from the reference's src/ORBmatcher.cc, which the maintainer keeps compiling with
// -DORBmatcher=ORBmatcherCPU -DLSDmatcher=LSDmatcherCPU (one line in CMakeLists.txt).
#ifndef PLSLAM_HIP_ADAPTOR_ORBMATCHER_H
#define PLSLAM_HIP_ADAPTOR_ORBMATCHER_H

#ifndef ORBmatcher   // (defined as a macro only in the translation units of src/ORBmatcher.cc / src/LSDmatcher.cpp themselves)

#include <MapPoint.h>   // read the classes the matcher header pulls in BEFORE the rename is active
#include <KeyFrame.h>
#include <Frame.h>
#define ORBmatcher ORBmatcherCPU
#include <ORBmatcher.h>   // the reference's include/ORBmatcher.h, found on the include path
#undef ORBmatcher

#include <cmath>
#include <set>
#include <stdexcept>

#include "HipMatchers.h"

namespace ORB_SLAM2 {

class ORBmatcher : public ORBmatcherCPU {
 public:
  ORBmatcher(float nnratio = 0.6, bool checkOri = true) : ORBmatcherCPU(nnratio, checkOri) {}

  // every overload that is not re-declared below stays visible
  using ORBmatcherCPU::SearchByProjection;
  using ORBmatcherCPU::SearchByBoW;
  using ORBmatcherCPU::Fuse;

  static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b) { return hip::DescriptorDistance(a, b); }

  // Tracking::SearchLocalPoints: exactly the members the reference loop reads (ORBmatcher.cc:64-88), one call, write back (:137-138)
  int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3) {
    RequireMonocular(F);
    const size_t n = vpMapPoints.size();
    hip::ProjQueries q;
    q.valid.resize(n); q.hasObs.resize(n); q.pos.resize(2 * n); q.level.resize(n); q.aux.resize(n);
    q.desc = cv::Mat::zeros((int)(n ? n : 1), 32, CV_8U);
    for (size_t i = 0; i < n; i++) {
      MapPoint* p = vpMapPoints[i];
      // (a point without a descriptor never matches in the reference: every candidate is skipped, ORBmatcher.cc:113-114)
      q.valid[i] = p->mbTrackInView && !p->isBad() && !p->GetDescriptor().empty();
      q.hasObs[i] = p->Observations() > 0;
      q.pos[2 * i] = p->mTrackProjX; q.pos[2 * i + 1] = p->mTrackProjY;
      q.level[i] = p->mnTrackScaleLevel;
      q.aux[i] = p->mTrackViewCos;
      const cv::Mat d = p->GetDescriptor();
      if (d.data) std::memcpy(q.desc.ptr<uchar>((int)i), d.ptr<uchar>(0), 32);
    }
    std::vector<uchar> occupied(F.N);                     // :98-100
    for (int i = 0; i < F.N; i++) occupied[i] = F.mvpMapPoints[i] && F.mvpMapPoints[i]->Observations() > 0;
    std::vector<int> assigned;
    if (F.N == 0 || n == 0) return 0;
    // (round 6: the frame's keypoints, descriptors and grid are resident on the device -- hip::FrameResidency -- the call uploads the
    // queries only; SearchLocalPoints searches the frame TrackWithMotionModel / TrackReferenceKeyFrame has just searched)
    const std::shared_ptr<hip::ResidentPoints> rf = hip::FrameResidency::Instance().Points(F.mnId, F.mvKeysUn, F.mDescriptors, FrameGrid());
    const int nmatches = hip::SearchByProjectionResident(rf->h, F.mvScaleFactors, occupied, q, th, mfNNratio, assigned);
    for (int i = 0; i < F.N; i++)
      if (assigned[i] >= 0) F.mvpMapPoints[i] = vpMapPoints[assigned[i]];
    return nmatches;
  }

  // TrackWithMotionModel: the pose algebra of :1452-1484 stays here in the reference's own expressions; the window search,
  // the level band, the best-distance scan and the rotation histogram run on the GPU
  int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono) {
    if (!bMono) ThrowStereo();
    RequireMonocular(CurrentFrame);
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat twc = -Rcw.t() * tcw;
    const cv::Mat Rlw = LastFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tlw = LastFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat tlc = Rlw * twc + tlw;
    const bool bForward = tlc.at<float>(2) > CurrentFrame.mb && !bMono;
    const bool bBackward = -tlc.at<float>(2) > CurrentFrame.mb && !bMono;
    const int n = LastFrame.N;
    hip::ProjQueries q;   // (q.pos: the WORLD position of every query, the projection of :1474-1484 runs in front of the search on the device)
    q.valid.assign(n, 0); q.hasObs.assign(n, 0); q.pos.assign(3 * (size_t)n, 0.f); q.level.assign(n, 0); q.aux.assign(n, 0.f);
    q.desc = cv::Mat::zeros(n ? n : 1, 32, CV_8U);
    for (int i = 0; i < n; i++) {
      MapPoint* pMP = LastFrame.mvpMapPoints[i];
      if (!pMP || LastFrame.mvbOutlier[i]) continue;
      const cv::Mat d = pMP->GetDescriptor();
      if (d.empty()) continue;                    // (no descriptor: never matches, ORBmatcher.cc:1530-1531)
      const cv::Mat x3Dw = pMP->GetWorldPos();
      q.valid[i] = 1;
      for (int k = 0; k < 3; k++) q.pos[3 * i + k] = x3Dw.at<float>(k);
      q.level[i] = LastFrame.mvKeys[i].octave;
      q.aux[i] = LastFrame.mvKeysUn[i].angle;
      q.hasObs[i] = pMP->Observations() > 0;
      std::memcpy(q.desc.ptr<uchar>(i), d.ptr<uchar>(0), 32);
    }
    std::vector<uchar> occupied(CurrentFrame.N);          // :1518-1520
    for (int i = 0; i < CurrentFrame.N; i++)
      occupied[i] = CurrentFrame.mvpMapPoints[i] && CurrentFrame.mvpMapPoints[i]->Observations() > 0;
    std::vector<int> assigned;
    if (CurrentFrame.N == 0 || n == 0) return 0;
    const std::shared_ptr<hip::ResidentPoints> rf =
        hip::FrameResidency::Instance().Points(CurrentFrame.mnId, CurrentFrame.mvKeysUn, CurrentFrame.mDescriptors, FrameGrid());
    const plh_point_gates cam = hip::PointGates(0, Rcw, tcw, cv::Mat(), CurrentFrame.fx, CurrentFrame.fy, CurrentFrame.cx, CurrentFrame.cy,
                                                CurrentFrame.mnMinX, CurrentFrame.mnMinY, CurrentFrame.mnMaxX, CurrentFrame.mnMaxY);
    const int nmatches = hip::SearchByProjectionLastFrameResidentWorld(rf->h, CurrentFrame.mvScaleFactors, occupied, cam.view, q, th,
                                                                       bForward ? 1 : bBackward ? 2 : 0, mbCheckOrientation, assigned);
    // :1548 assigns, the rotation-consistency pass (:1567-1580) resets the rejected ones to NULL: `assigned` is the net effect,
    // `occupied` tells which of the previously empty slots stayed empty
    for (int i = 0; i < CurrentFrame.N; i++)
      if (assigned[i] >= 0) CurrentFrame.mvpMapPoints[i] = LastFrame.mvpMapPoints[assigned[i]];
    return nmatches;
  }

  // TrackReferenceKeyFrame / Relocalization
  int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches) {
    const std::vector<MapPoint*> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = std::vector<MapPoint*>(F.N, static_cast<MapPoint*>(NULL));
    std::vector<uchar> valid(vpMapPointsKF.size());
    for (size_t i = 0; i < valid.size(); i++) valid[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad();
    std::vector<int> matchKF;
    if (pKF->N == 0 || F.N == 0) return 0;
    // (a KeyFrame answers to the id of the Frame it was made from: same features; every frame tracked against it finds it resident.
    // The angles SearchByBoW compares are those of mvKeysUn / mvKeys alike: undistortion moves pt only.)
    const std::vector<int32_t> nodeKF = hip::NodeOfFeature(pKF->mFeatVec, pKF->N), nodeF = hip::NodeOfFeature(F.mFeatVec, F.N);
    const std::shared_ptr<hip::ResidentPoints> rk =
        hip::FrameResidency::Instance().Points(pKF->mnFrameId, pKF->mvKeysUn, pKF->mDescriptors, FrameGrid(), &nodeKF);
    const std::shared_ptr<hip::ResidentPoints> rf = hip::FrameResidency::Instance().Points(F.mnId, F.mvKeysUn, F.mDescriptors, FrameGrid(), &nodeF);
    const int n = hip::SearchByBoWResident(rk->h, valid, rf->h, mfNNratio, mbCheckOrientation, matchKF, TH_LOW);
    for (int j = 0; j < F.N; j++)
      if (matchKF[j] >= 0) vpMapPointMatches[j] = vpMapPointsKF[matchKF[j]];
    return n;
  }

  // Tracking::Relocalization (Tracking.cc: matcher2.SearchByProjection(mCurrentFrame, vpCandidateKFs[i], sFound, 10, 100) and the
  // narrower second pass): the KeyFrame's map points projected with the current pose, in the reference's own expressions
  // (:1591-1640); window lookup with the level band, best distance against ORBdist and the rotation histogram on the GPU.
  int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th, const int ORBdist) {
    RequireMonocular(CurrentFrame);
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat Ow = -Rcw.t() * tcw;
    const std::vector<MapPoint*> vpMPs = pKF->GetMapPointMatches();
    const int nq = (int)vpMPs.size();
    hip::ProjQueries q;
    q.hasObs.assign(nq, 1); q.aux.assign(nq, 0.f);
    q.desc = cv::Mat::zeros(nq ? nq : 1, 32, CV_8U);
    hip::MapPointGateArrays in;
    in.assign(nq);
    for (int i = 0; i < nq; i++) {
      MapPoint* pMP = vpMPs[i];
      if (!pMP) continue;
      if (pMP->isBad() || sAlreadyFound.count(pMP)) continue;
      const cv::Mat dMP = pMP->GetDescriptor();
      if (dMP.empty()) continue;
      in.set(i, pMP->GetWorldPos(), cv::Mat(), pMP->GetMinDistanceInvariance(), pMP->GetMaxDistanceInvariance());
      q.aux[i] = pKF->mvKeysUn[i].angle;
      std::memcpy(q.desc.ptr<uchar>(i), dMP.ptr<uchar>(0), 32);
    }
    // :1614-1636: no depth gate, invzc in double, u = fx*xc*invzc + cx, the frame's bounds, |x3Dw - Ow| in the invariance range
    std::vector<float> dist;
    hip::MapPointGates(hip::PointGates(PLH_GATE_INVZ_DOUBLE, Rcw, tcw, Ow, CurrentFrame.fx, CurrentFrame.fy, CurrentFrame.cx, CurrentFrame.cy,
                                       CurrentFrame.mnMinX, CurrentFrame.mnMinY, CurrentFrame.mnMaxX, CurrentFrame.mnMaxY), in, q, dist);
    for (int i = 0; i < nq; i++)
      if (q.valid[i]) q.level[i] = vpMPs[i]->PredictScale(dist[i], &CurrentFrame);
    std::vector<uchar> occupied(CurrentFrame.N);
    for (int i = 0; i < CurrentFrame.N; i++) occupied[i] = CurrentFrame.mvpMapPoints[i] != NULL;
    std::vector<int> assigned;
    if (CurrentFrame.N == 0 || nq == 0) return 0;
    const int nmatches = hip::SearchByProjectionKeyFrame(CurrentFrame.mvKeysUn, CurrentFrame.mDescriptors, FrameGrid(),
                                                         CurrentFrame.mvScaleFactors, occupied, q, th, ORBdist, mbCheckOrientation, assigned);
    // `assigned` is the net effect of the assignments (:1671) and of the rotation-consistency pass that resets some of them (:1704)
    for (int i = 0; i < CurrentFrame.N; i++)
      if (assigned[i] >= 0) CurrentFrame.mvpMapPoints[i] = vpMPs[assigned[i]];
    return nmatches;
  }

  // LoopClosing::ComputeSim3 (LoopClosing.cc:239-375): ORBmatcher(0.75, true).SearchByBoW(mpCurrentKF, pKF, vvpMapPointMatches[i])
  int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12) {
    const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches();
    const std::vector<MapPoint*> vpMapPoints2 = pKF2->GetMapPointMatches();
    vpMatches12 = std::vector<MapPoint*>(vpMapPoints1.size(), static_cast<MapPoint*>(NULL));
    std::vector<uchar> v1(vpMapPoints1.size()), v2(vpMapPoints2.size());
    for (size_t i = 0; i < v1.size(); i++) v1[i] = vpMapPoints1[i] && !vpMapPoints1[i]->isBad();   // :612-616
    for (size_t i = 0; i < v2.size(); i++) v2[i] = vpMapPoints2[i] && !vpMapPoints2[i]->isBad();   // :630-634
    std::vector<int> m12;
    const int n = hip::SearchByBoWKFKF(pKF1->mvKeysUn, pKF1->mDescriptors, hip::NodeOfFeature(pKF1->mFeatVec, pKF1->N), v1, pKF2->mvKeysUn,
                                       pKF2->mDescriptors, hip::NodeOfFeature(pKF2->mFeatVec, pKF2->N), v2, mfNNratio, mbCheckOrientation,
                                       m12, TH_LOW);
    for (size_t i = 0; i < m12.size(); i++)
      if (m12[i] >= 0) vpMatches12[i] = vpMapPoints2[m12[i]];
    return n;
  }

  // LoopClosing::ComputeSim3 (LoopClosing.cc:327): ORBmatcher(0.75, true).SearchBySim3(mpCurrentKF, pKF, vpMapPointMatches, s, R, t, 7.5).
  // Both transforms, the projections and every pre-check (:1206-1290, :1313-1365) are the reference's own expressions; the two window
  // searches (level band l-1..l, TH_HIGH) and the agreement pass (:1421-1436) run on the GPU.
  int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12,
                   const cv::Mat& t12, const float th) {
    RequireMonocularKF(pKF1);
    RequireMonocularKF(pKF2);
    const float &fx = pKF1->fx, &fy = pKF1->fy, &cx = pKF1->cx, &cy = pKF1->cy;
    cv::Mat R1w = pKF1->GetRotation();
    cv::Mat t1w = pKF1->GetTranslation();
    cv::Mat R2w = pKF2->GetRotation();
    cv::Mat t2w = pKF2->GetTranslation();
    cv::Mat sR12 = s12 * R12;
    cv::Mat sR21 = (1.0 / s12) * R12.t();
    cv::Mat t21 = -sR21 * t12;
    const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches();
    const int N1 = vpMapPoints1.size();
    const std::vector<MapPoint*> vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N2 = vpMapPoints2.size();
    std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
    for (int i = 0; i < N1; i++) {
      MapPoint* pMP = vpMatches12[i];
      if (pMP) {
        vbAlreadyMatched1[i] = true;
        int idx2 = pMP->GetIndexInKeyFrame(pKF2);
        if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
      }
    }
    hip::ProjQueries q[2];
    for (int dir = 0; dir < 2; dir++) {
      const std::vector<MapPoint*>& pts = dir == 0 ? vpMapPoints1 : vpMapPoints2;
      const std::vector<bool>& done = dir == 0 ? vbAlreadyMatched1 : vbAlreadyMatched2;
      KeyFrame* pTo = dir == 0 ? pKF2 : pKF1;
      const int n = (int)pts.size();
      hip::ProjQueries& Q = q[dir];
      Q.hasObs.assign(n, 1); Q.aux.assign(n, 0.f);
      Q.desc = cv::Mat::zeros(n ? n : 1, 32, CV_8U);
      hip::MapPointGateArrays in;
      in.assign(n);
      for (int i = 0; i < n; i++) {
        MapPoint* pMP = pts[i];
        if (!pMP || done[i]) continue;
        if (pMP->isBad()) continue;
        const cv::Mat dMP = pMP->GetDescriptor();
        if (dMP.empty()) continue;   // no candidate can lower bestDist from INT_MAX (:1283-1284)
        in.set(i, pMP->GetWorldPos(), cv::Mat(), pMP->GetMinDistanceInvariance(), pMP->GetMaxDistanceInvariance());
        std::memcpy(Q.desc.ptr<uchar>(i), dMP.ptr<uchar>(0), 32);
      }
      // :1232-1268 / :1312-1348: into the point's own camera, then into the other one by the similarity; depth gate, invz in double,
      // x = X*invz, pTo->IsInImage, |p3Dc| of the target camera in the invariance range
      plh_point_gates g = hip::PointGates(PLH_GATE_Z | PLH_GATE_INVZ_DOUBLE | PLH_GATE_UV_NORMALISED | PLH_GATE_KEYFRAME_BOUNDS |
                                              PLH_GATE_DIST_OF_TARGET | PLH_GATE_SECOND,
                                          dir == 0 ? R1w : R2w, dir == 0 ? t1w : t2w, cv::Mat(), fx, fy, cx, cy, pTo->mnMinX, pTo->mnMinY,
                                          pTo->mnMaxX, pTo->mnMaxY);
      hip::PutMat(g.R2, dir == 0 ? sR21 : sR12, 3, 3);
      hip::PutMat(g.t2, dir == 0 ? t21 : t12, 3, 1);
      std::vector<float> dist;
      hip::MapPointGates(g, in, Q, dist);
      for (int i = 0; i < n; i++)
        if (Q.valid[i]) Q.level[i] = pts[i]->PredictScale(dist[i], pTo);
    }
    std::vector<int> m12;
    const int nFound = hip::SearchBySim3(pKF1->mvKeysUn, pKF1->mDescriptors, pKF2->mvKeysUn, pKF2->mDescriptors, FrameGrid(),
                                         pKF1->mvScaleFactors, q[0], q[1], th, m12, TH_HIGH);
    for (int i1 = 0; i1 < N1 && i1 < (int)m12.size(); i1++)
      if (m12[i1] >= 0) vpMatches12[i1] = vpMapPoints2[m12[i1]];
    return nFound;
  }

  // LocalMapping::CreateNewMapPoints (LocalMapping.cc:339-398): ORBmatcher(0.6, false).SearchForTriangulation(mpCurrentKeyFrame, pKF2,
  // F12, vMatchedIndices, false).  The epipole is the reference's own expression (:727-735); the per-node scan with the epipole and
  // epipolar-line gates and the first-come claims on pKF2's features (in FeatureVector order) run on the GPU.
  int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> >& vMatchedPairs,
                             const bool bOnlyStereo) {
    RequireMonocularKF(pKF1);
    RequireMonocularKF(pKF2);
    vMatchedPairs.clear();
    if (bOnlyStereo) return 0;   // monocular KeyFrames hold no stereo feature: the reference skips every idx1 (:778-780)
    cv::Mat Cw = pKF1->GetCameraCenter();
    cv::Mat R2w = pKF2->GetRotation();
    cv::Mat t2w = pKF2->GetTranslation();
    cv::Mat C2 = R2w * Cw + t2w;
    const float invz = 1.0f / C2.at<float>(2);
    const float ex = pKF2->fx * C2.at<float>(0) * invz + pKF2->cx;
    const float ey = pKF2->fy * C2.at<float>(1) * invz + pKF2->cy;
    std::vector<uchar> has1(pKF1->N), has2(pKF2->N);
    for (int i = 0; i < pKF1->N; i++) has1[i] = pKF1->GetMapPoint(i) != NULL;
    for (int i = 0; i < pKF2->N; i++) has2[i] = pKF2->GetMapPoint(i) != NULL;
    std::vector<int> m12;
    float F[9];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) F[3 * r + c] = F12.at<float>(r, c);
    const int n = hip::SearchForTriangulation(pKF1->mvKeysUn, pKF1->mDescriptors, hip::NodeOfFeature(pKF1->mFeatVec, pKF1->N), has1,
                                              pKF2->mvKeysUn, pKF2->mDescriptors, hip::NodeOfFeature(pKF2->mFeatVec, pKF2->N), has2, F, ex,
                                              ey, pKF2->mvScaleFactors, pKF2->mvLevelSigma2, mbCheckOrientation, m12, TH_LOW);
    vMatchedPairs.reserve(n);
    for (size_t i = 0; i < m12.size(); i++)
      if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair(i, (size_t)m12[i]));
    return n;
  }

  // LoopClosing::ComputeSim3 (LoopClosing.cc:360): ORBmatcher(0.75, true).SearchByProjection(mpCurrentKF, mScw, mvpLoopMapPoints,
  // mvpCurrentMatchedPoints, 10).  The Sim3 decomposition, the projection and every pre-check of the loop (:337-395) are the
  // reference's own expressions; the window search with its level band, best distance and TH_LOW runs on the GPU.
  int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched, int th) {
    RequireMonocularKF(pKF);
    const float &fx = pKF->fx, &fy = pKF->fy, &cx = pKF->cx, &cy = pKF->cy;
    cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
    const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
    cv::Mat Rcw = sRcw / scw;
    cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
    cv::Mat Ow = -Rcw.t() * tcw;
    std::set<MapPoint*> spAlreadyFound(vpMatched.begin(), vpMatched.end());
    spAlreadyFound.erase(static_cast<MapPoint*>(NULL));
    const int nq = (int)vpPoints.size();
    hip::ProjQueries q;
    q.hasObs.assign(nq, 1); q.aux.assign(nq, 0.f);
    q.desc = cv::Mat::zeros(nq ? nq : 1, 32, CV_8U);
    hip::MapPointGateArrays in;
    in.assign(nq);
    for (int iMP = 0; iMP < nq; iMP++) {
      MapPoint* pMP = vpPoints[iMP];
      if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
      const cv::Mat dMP = pMP->GetDescriptor();
      if (dMP.empty()) continue;                                   // (:431: no candidate is ever compared)
      in.set(iMP, pMP->GetWorldPos(), pMP->GetNormal(), pMP->GetMinDistanceInvariance(), pMP->GetMaxDistanceInvariance());
      std::memcpy(q.desc.ptr<uchar>(iMP), dMP.ptr<uchar>(0), 32);
    }
    // :362-395: depth gate, invz in float, x = X*invz, pKF->IsInImage, |p3Dw - Ow| in the invariance range, viewing angle under 60 degrees
    std::vector<float> dist;
    hip::MapPointGates(hip::PointGates(PLH_GATE_Z | PLH_GATE_UV_NORMALISED | PLH_GATE_KEYFRAME_BOUNDS | PLH_GATE_NORMAL, Rcw, tcw, Ow, fx, fy, cx,
                                       cy, pKF->mnMinX, pKF->mnMinY, pKF->mnMaxX, pKF->mnMaxY), in, q, dist);
    for (int iMP = 0; iMP < nq; iMP++)
      if (q.valid[iMP]) q.level[iMP] = vpPoints[iMP]->PredictScale(dist[iMP], pKF);
    std::vector<uchar> occupied(vpMatched.size());
    for (size_t i = 0; i < vpMatched.size(); i++) occupied[i] = vpMatched[i] != NULL;
    std::vector<int> assigned;
    if (pKF->N == 0 || nq == 0) return 0;
    const int nmatches = hip::SearchByProjectionSim3(pKF->mvKeysUn, pKF->mDescriptors, FrameGrid(), pKF->mvScaleFactors, occupied, q, (float)th,
                                                     assigned, TH_LOW);
    for (size_t i = 0; i < assigned.size() && i < vpMatched.size(); i++)
      if (assigned[i] >= 0) vpMatched[i] = vpPoints[assigned[i]];
    return nmatches;
  }

  // LocalMapping::SearchInNeighbors (LocalMapping.cc:1535-1573): ORBmatcher().Fuse(pKFi, vpMapPointMatches).  The search for the
  // best keypoint of every map point does not depend on what the loop does to the map, so it runs first, for all points, on the
  // GPU; the loop then runs in the reference's order with the reference's own tests (a point can turn bad, or enter the KeyFrame,
  // through an earlier iteration's Replace / AddObservation) and its replace / add logic (:1029-1058).
  int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th = 3.0) {
    RequireMonocularKF(pKF);
    cv::Mat Rcw = pKF->GetRotation();
    cv::Mat tcw = pKF->GetTranslation();
    const float &fx = pKF->fx, &fy = pKF->fy, &cx = pKF->cx, &cy = pKF->cy;
    cv::Mat Ow = pKF->GetCameraCenter();
    const int nMPs = (int)vpMapPoints.size();
    hip::ProjQueries q;
    q.hasObs.assign(nMPs, 1); q.aux.assign(nMPs, 0.f);
    q.desc = cv::Mat::zeros(nMPs ? nMPs : 1, 32, CV_8U);
    hip::MapPointGateArrays in;
    in.assign(nMPs);
    for (int i = 0; i < nMPs; i++) {
      MapPoint* pMP = vpMapPoints[i];
      if (!pMP) continue;
      const cv::Mat dMP = pMP->GetDescriptor();
      if (dMP.empty()) continue;
      in.set(i, pMP->GetWorldPos(), pMP->GetNormal(), pMP->GetMinDistanceInvariance(), pMP->GetMaxDistanceInvariance());
      std::memcpy(q.desc.ptr<uchar>(i), dMP.ptr<uchar>(0), 32);
    }
    // :945-975: depth gate, invz in float, x = X*invz, pKF->IsInImage, |p3Dw - Ow| in the invariance range, viewing angle under 60 degrees
    std::vector<float> dist;
    hip::MapPointGates(hip::PointGates(PLH_GATE_Z | PLH_GATE_UV_NORMALISED | PLH_GATE_KEYFRAME_BOUNDS | PLH_GATE_NORMAL, Rcw, tcw, Ow, fx, fy, cx,
                                       cy, pKF->mnMinX, pKF->mnMinY, pKF->mnMaxX, pKF->mnMaxY), in, q, dist);
    for (int i = 0; i < nMPs; i++)
      if (q.valid[i]) q.level[i] = vpMapPoints[i]->PredictScale(dist[i], pKF);
    std::vector<int> bestIdx(nMPs, -1);
    if (pKF->N > 0 && nMPs > 0)
      hip::FuseSearch(pKF->mvKeysUn, pKF->mDescriptors, FrameGrid(), pKF->mvScaleFactors, pKF->mvInvLevelSigma2, q, th, bestIdx, TH_LOW);
    int nFused = 0;
    for (int i = 0; i < nMPs; i++) {
      MapPoint* pMP = vpMapPoints[i];
      if (!pMP) continue;
      if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;          // (as it stands when the loop gets here)
      if (!q.valid[i] || bestIdx[i] < 0) continue;
      MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx[i]);
      if (pMPinKF) {
        if (!pMPinKF->isBad()) {
          if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
          else pMPinKF->Replace(pMP);
        }
      } else {
        pMP->AddObservation(pKF, bestIdx[i]);
        pKF->AddMapPoint(pMP, bestIdx[i]);
      }
      nFused++;
    }
    return nFused;
  }

  // LoopClosing::SearchAndFuse (LoopClosing.cc:589-599): ORBmatcher(0.8).Fuse(pKF, cvScw, mvpLoopMapPoints, 4, vpReplacePoints).  As
  // above: the searches first (no chi-square gate in this overload, :1150-1170), then the loop in the reference's order.
  int Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint) {
    RequireMonocularKF(pKF);
    const float &fx = pKF->fx, &fy = pKF->fy, &cx = pKF->cx, &cy = pKF->cy;
    cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
    const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
    cv::Mat Rcw = sRcw / scw;
    cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
    cv::Mat Ow = -Rcw.t() * tcw;
    const std::set<MapPoint*> spAlreadyFound = pKF->GetMapPoints();
    const int nPoints = (int)vpPoints.size();
    hip::ProjQueries q;
    q.hasObs.assign(nPoints, 1); q.aux.assign(nPoints, 0.f);
    q.desc = cv::Mat::zeros(nPoints ? nPoints : 1, 32, CV_8U);
    hip::MapPointGateArrays in;
    in.assign(nPoints);
    for (int iMP = 0; iMP < nPoints; iMP++) {
      MapPoint* pMP = vpPoints[iMP];
      if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
      const cv::Mat dMP = pMP->GetDescriptor();
      if (dMP.empty()) continue;
      in.set(iMP, pMP->GetWorldPos(), pMP->GetNormal(), pMP->GetMinDistanceInvariance(), pMP->GetMaxDistanceInvariance());
      std::memcpy(q.desc.ptr<uchar>(iMP), dMP.ptr<uchar>(0), 32);
    }
    // :1096-1128: as the other overload with invz in double
    std::vector<float> dist;
    hip::MapPointGates(hip::PointGates(PLH_GATE_Z | PLH_GATE_INVZ_DOUBLE | PLH_GATE_UV_NORMALISED | PLH_GATE_KEYFRAME_BOUNDS | PLH_GATE_NORMAL, Rcw,
                                       tcw, Ow, fx, fy, cx, cy, pKF->mnMinX, pKF->mnMinY, pKF->mnMaxX, pKF->mnMaxY), in, q, dist);
    for (int iMP = 0; iMP < nPoints; iMP++)
      if (q.valid[iMP]) q.level[iMP] = vpPoints[iMP]->PredictScale(dist[iMP], pKF);
    std::vector<int> bestIdx(nPoints, -1);
    if (pKF->N > 0 && nPoints > 0)
      hip::FuseSearch(pKF->mvKeysUn, pKF->mDescriptors, FrameGrid(), pKF->mvScaleFactors, std::vector<float>(), q, th, bestIdx, TH_LOW);
    int nFused = 0;
    for (int iMP = 0; iMP < nPoints; iMP++) {
      if (!q.valid[iMP] || bestIdx[iMP] < 0) continue;
      MapPoint* pMP = vpPoints[iMP];
      MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx[iMP]);
      if (pMPinKF) {
        if (!pMPinKF->isBad()) vpReplacePoint[iMP] = pMPinKF;
      } else {
        pMP->AddObservation(pKF, bestIdx[iMP]);
        pKF->AddMapPoint(pMP, bestIdx[iMP]);
      }
      nFused++;
    }
    return nFused;
  }

  // MonocularInitialization
  int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12,
                              int windowSize = 10) {
    // (F1 is the initial frame of every attempt until the initialiser succeeds: it stays resident)
    const std::shared_ptr<hip::ResidentPoints> r1 = hip::FrameResidency::Instance().Points(F1.mnId, F1.mvKeysUn, F1.mDescriptors, FrameGrid());
    const std::shared_ptr<hip::ResidentPoints> r2 = hip::FrameResidency::Instance().Points(F2.mnId, F2.mvKeysUn, F2.mDescriptors, FrameGrid());
    return hip::SearchForInitializationResident(r1->h, r2->h, vbPrevMatched, vnMatches12, windowSize, mfNNratio, mbCheckOrientation);
  }

 protected:
  // The GPU searches implement the monocular form (the project is monocular only, README.md:4): the right-image gates of the
  // stereo / RGB-D form (`er = |mTrackProjXR - mvuRight|`, ORBmatcher.cc:104-109; `ur = u - mbf * invzc`, :1520-1526) are not
  // evaluated.  A frame that carries right coordinates is refused loudly rather than matched without them.
  static void ThrowStereo() {
    throw std::runtime_error("plslam_hip drop-in: ORBmatcher::SearchByProjection supports monocular frames only "
                             "(stereo / RGB-D right-coordinate gates are not implemented on the GPU path)");
  }
  static void RequireMonocular(const Frame& F) {
    for (size_t i = 0; i < F.mvuRight.size(); i++)
      if (F.mvuRight[i] > 0) ThrowStereo();
  }
  static void RequireMonocularKF(const KeyFrame* pKF) {
    for (size_t i = 0; i < pKF->mvuRight.size(); i++)
      if (pKF->mvuRight[i] >= 0) ThrowStereo();
  }
  static plh_grid_params FrameGrid() {
    return hip::GridParams(Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY, Frame::mfGridElementWidthInv,
                           Frame::mfGridElementHeightInv);
  }
};

}  // namespace ORB_SLAM2

#endif  // ORBmatcher
#endif
