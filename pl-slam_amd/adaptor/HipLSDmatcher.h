// Drop-in replacement for the reference's include/LSDmatcher.h: the class ORB_SLAM2::LSDmatcher with the reference's
// constructor and method signatures (include/LSDmatcher.h:22-76), whose tracking-path searches run on the GPU through the
// C ABI of libplslam_hip.so:
//     LSDmatcher::SearchByProjection(Frame& Cur, const Frame& Last, th)     src/LSDmatcher.cpp:72-176    (Tracking.cc:1340-1357)
//     LSDmatcher::SearchByProjection(Frame& Cur, const Frame& Last)         :19-70                       (no caller)
//     LSDmatcher::SearchByProjection(Frame&, const vector<MapLine*>&, th)   :221-338                     (Tracking.cc:1849)
//     LSDmatcher::SearchDouble(Frame&, Frame&, vector<int>&)                :427-460                     (Tracking.cc:711)
//     LSDmatcher::SearchDouble(KeyFrame*, Frame&)                           :375-425                     (Tracking.cc:1159)
//     LSDmatcher::DescriptorDistance                                        :654-670
//     LSDmatcher::SerachForInitialize(Frame&, Frame&, vector<int>&)         :340-373                     (Tracking.cc:710, commented out)
// and the back end's (LocalMapping):
//     LSDmatcher::SearchForTriangulation(pKF1, pKF2, vector<pair>&)         :672-725                     (LocalMapping.cc:679)
//     LSDmatcher::SearchForTriangulation(pKF1, pKF2, vector<int>&, isDouble) :727-778                    (LocalMapping.cc:961)
//     LSDmatcher::SearchForTriangulationNew(pKF1, pKF2, vector<int>&, isDouble) :780-832 (+ FrameBFMatchNew :488-625)  (LocalMapping.cc:960, commented out)
//     LSDmatcher::Fuse(pKF, vpMapLines, th)                                 :860-1002                    (LocalMapping.cc:1600,1627)
// Same construction as adaptor/HipORBmatcher.h: the reference's own class is read as LSDmatcherCPU, the class below derives
// from it and inherits everything it does not re-declare (ComputeF12, RadiusByViewingCos, ...); the maintainer compiles
// src/LSDmatcher.cpp with -DORBmatcher=ORBmatcherCPU -DLSDmatcher=LSDmatcherCPU.  The reference's debugging pictures
// (matchResultTrack.jpg, :67 / :171 / :422; matchResultLocalMapping.jpg, :723 / :776) are not written.
#ifndef PLSLAM_HIP_ADAPTOR_LSDMATCHER_H
#define PLSLAM_HIP_ADAPTOR_LSDMATCHER_H

#ifndef LSDmatcher

#include <MapLine.h>
#include <KeyFrame.h>
#include <Frame.h>
#define LSDmatcher LSDmatcherCPU
#include <LSDmatcher.h>   // the reference's include/LSDmatcher.h
#undef LSDmatcher

#include <limits>

#include "HipMatchers.h"

namespace ORB_SLAM2 {

class LSDmatcher : public LSDmatcherCPU {
 public:
  LSDmatcher(float nnratio = 0.7, bool checkOri = true) : LSDmatcherCPU(nnratio, checkOri) {}

  // The two-argument overload (src/LSDmatcher.cpp:19-70, no caller in the reference): mutual nearest LBD neighbours at TH_LOW between the
  // last frame's lines and the current frame's, the last frame's MapLines carried over
  int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame) {
    if (LastFrame.mLdesc.rows == 0 || CurrentFrame.mLdesc.rows == 0) return 0;
    std::vector<int> m12;
    hip::SearchDouble(LastFrame.mLdesc, CurrentFrame.mLdesc, m12, mfNNratio, (float)TH_LOW);
    int nmatches = 0;
    for (size_t i = 0; i < m12.size() && i < LastFrame.mvpMapLines.size(); i++) {
      if (m12[i] < 0) continue;
      MapLine* mapLine = LastFrame.mvpMapLines[i];
      if (!mapLine) continue;
      CurrentFrame.mvpMapLines[m12[i]] = mapLine;
      nmatches++;
    }
    return nmatches;
  }

  static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b) { return hip::DescriptorDistance(a, b); }

  // TrackWithMotionModel: Frame::isInFrustum(pML, 0.5) (which fills mTrackProj*) stays the reference's, per line, in the
  // reference's order; window lookup, length-ratio gate and best-distance scan run on the GPU
  int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th) {
    const int n = LastFrame.NL;
    hip::ProjQueries q;
    q.valid.assign(n, 0); q.hasObs.assign(n, 0); q.pos.assign(4 * (size_t)n, 0.f); q.aux.assign(n, 0.f);
    q.desc = cv::Mat::zeros(n ? n : 1, 32, CV_8U);
    for (int i = 0; i < n; i++) {
      MapLine* pML = LastFrame.mvpMapLines[i];
      if (!pML || LastFrame.mvbLineOutlier[i]) continue;
      if (!CurrentFrame.isInFrustum(pML, 0.5)) continue;
      q.valid[i] = 1;
      q.pos[4 * i] = pML->mTrackProjX1; q.pos[4 * i + 1] = pML->mTrackProjY1;
      q.pos[4 * i + 2] = pML->mTrackProjX2; q.pos[4 * i + 3] = pML->mTrackProjY2;
      q.aux[i] = LastFrame.mvKeylinesUn[i].lineLength;
      q.hasObs[i] = pML->Observations() > 0;
      const cv::Mat d = pML->GetDescriptor();
      if (d.data) std::memcpy(q.desc.ptr<uchar>(i), d.ptr<uchar>(0), 32);
    }
    std::vector<uchar> occupied(CurrentFrame.NL);
    for (int i = 0; i < CurrentFrame.NL; i++)
      occupied[i] = CurrentFrame.mvpMapLines[i] && CurrentFrame.mvpMapLines[i]->Observations() > 0;
    if (CurrentFrame.NL == 0 || n == 0) return 0;
    std::vector<int> assigned;
    // (round 6: the frame's lines, LBD rows, line equations and line grid are resident on the device, hip::FrameResidency)
    const std::shared_ptr<hip::ResidentLines> rl = hip::FrameResidency::Instance().Lines(
        CurrentFrame.mnId, CurrentFrame.mvKeylinesUn, CurrentFrame.mLdesc, CurrentFrame.mvKeyLineFunctions, FrameGrid());
    const int nmatches = hip::LineSearchByProjectionResident(rl->h, occupied, q, th, mfNNratio, true, assigned);
    for (int i = 0; i < CurrentFrame.NL; i++)
      if (assigned[i] >= 0) CurrentFrame.mvpMapLines[i] = LastFrame.mvpMapLines[assigned[i]];
    return nmatches;
  }

  // Tracking::SearchLocalLines
  int SearchByProjection(Frame& F, const std::vector<MapLine*>& vpMapLines, const float th = 3) {
    const size_t n = vpMapLines.size();
    hip::ProjQueries q;
    q.valid.resize(n); q.hasObs.resize(n); q.pos.resize(4 * n); q.aux.resize(n);
    q.desc.create((int)(n ? n : 1), 32, CV_8U);
    for (size_t i = 0; i < n; i++) {
      MapLine* p = vpMapLines[i];
      q.valid[i] = p->mbTrackInView && !p->isBad();
      q.hasObs[i] = p->Observations() > 0;
      q.pos[4 * i] = p->mTrackProjX1; q.pos[4 * i + 1] = p->mTrackProjY1;
      q.pos[4 * i + 2] = p->mTrackProjX2; q.pos[4 * i + 3] = p->mTrackProjY2;
      q.aux[i] = p->mTrackViewCos;
      const cv::Mat d = p->GetDescriptor();
      if (d.data) std::memcpy(q.desc.ptr<uchar>((int)i), d.ptr<uchar>(0), 32);
    }
    std::vector<uchar> occupied(F.NL);
    for (int i = 0; i < F.NL; i++) occupied[i] = F.mvpMapLines[i] && F.mvpMapLines[i]->Observations() > 0;
    if (F.NL == 0 || n == 0) return 0;
    std::vector<int> assigned;
    const std::shared_ptr<hip::ResidentLines> rl =
        hip::FrameResidency::Instance().Lines(F.mnId, F.mvKeylinesUn, F.mLdesc, F.mvKeyLineFunctions, FrameGrid());
    const int nmatches = hip::LineSearchByProjectionResident(rl->h, occupied, q, th, mfNNratio, false, assigned);
    for (int i = 0; i < F.NL; i++)
      if (assigned[i] >= 0) F.mvpMapLines[i] = vpMapLines[assigned[i]];
    return nmatches;
  }

  // MonocularInitialization
  int SearchDouble(Frame& InitialFrame, Frame& CurrentFrame, std::vector<int>& LineMatches) {
    LineMatches = std::vector<int>(InitialFrame.NL, -1);
    if (InitialFrame.mLdesc.rows == 0 || CurrentFrame.mLdesc.rows == 0) return 0;
    const std::shared_ptr<hip::ResidentLines> r1 = hip::FrameResidency::Instance().Lines(
        InitialFrame.mnId, InitialFrame.mvKeylinesUn, InitialFrame.mLdesc, InitialFrame.mvKeyLineFunctions, FrameGrid());
    const std::shared_ptr<hip::ResidentLines> r2 = hip::FrameResidency::Instance().Lines(
        CurrentFrame.mnId, CurrentFrame.mvKeylinesUn, CurrentFrame.mLdesc, CurrentFrame.mvKeyLineFunctions, FrameGrid());
    return hip::SearchDoubleResident(r1->h, r2->h, LineMatches, mfNNratio, (float)TH_LOW);
  }

  // TrackReferenceKeyFrame: mutual matches between the KeyFrame's and the Frame's lines; a match hands the KeyFrame's
  // MapLine to the Frame (:395-403)
  int SearchDouble(KeyFrame* KF, Frame& CurrentFrame) {
    if (KF->mLineDescriptors.rows == 0 || CurrentFrame.mLdesc.rows == 0) return 0;
    std::vector<int> m12;   // KeyFrame line j -> Frame line, mutual
    hip::SearchDouble(KF->mLineDescriptors, CurrentFrame.mLdesc, m12, mfNNratio, (float)TH_LOW);
    // the reference walks the Frame's lines in index order; the mutual relation is symmetric, so invert it
    std::vector<int> m21(CurrentFrame.NL, -1);
    for (size_t j = 0; j < m12.size(); j++)
      if (m12[j] >= 0) m21[m12[j]] = (int)j;
    int nmatches = 0;
    for (int i = 0; i < CurrentFrame.NL; i++) {
      if (m21[i] < 0) continue;
      MapLine* pML = KF->GetMapLine(m21[i]);
      if (!pML) continue;
      CurrentFrame.mvpMapLines[i] = pML;
      nmatches++;
    }
    return nmatches;
  }

  // Tracking.cc:710 (commented out in the reference; :340-373): the nearest LBD neighbour of every line of the initial frame, kept where
  // the gap to the second nearest exceeds half the MAD of the gaps (Frame::lineDescriptorMAD, Frame.cc:519-544, is LSDmatcher's own,
  // :627-652) -- FrameBFMatch without its distance and ratio tests, i.e. with both thresholds at infinity.
  int SerachForInitialize(Frame& InitialFrame, Frame& CurrentFrame, std::vector<int>& LineMatches) {
    LineMatches.clear();
    LineMatches = std::vector<int>(InitialFrame.NL, -1);
    if (InitialFrame.mLdesc.rows == 0 || CurrentFrame.mLdesc.rows == 0) return 0;
    std::vector<int> m12;
    const float inf = std::numeric_limits<float>::infinity();
    hip::FrameBFMatch(InitialFrame.mLdesc, CurrentFrame.mLdesc, m12, inf, inf);
    int nmatches = 0;
    for (size_t i = 0; i < m12.size() && i < LineMatches.size(); i++) {
      if (m12[i] < 0) continue;
      LineMatches[i] = m12[i];
      nmatches++;
    }
    return nmatches;
  }

  // LocalMapping::CreateNewMapLines (LocalMapping.cc:679): mutual nearest LBD descriptors at TH_LOW between two KeyFrames, kept
  // where neither line carries a MapLine (:692-708)
  int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<std::pair<size_t, size_t> >& vMatchedPairs) {
    vMatchedPairs.clear();
    if (pKF1->mLineDescriptors.rows == 0 || pKF2->mLineDescriptors.rows == 0) return 0;
    std::vector<int> m12;
    hip::SearchDouble(pKF1->mLineDescriptors, pKF2->mLineDescriptors, m12, mfNNratio, (float)TH_LOW);
    int nmatches = 0;
    for (size_t i = 0; i < m12.size(); i++) {
      if (m12[i] < 0) continue;
      if (pKF1->GetMapLine(i) || pKF2->GetMapLine(m12[i])) continue;
      vMatchedPairs.push_back(std::make_pair(i, (size_t)m12[i]));
      nmatches++;
    }
    return nmatches;
  }

  // LocalMapping::CreateNewMapLines2 (LocalMapping.cc:961): as above at TH_HIGH into a per-line vector; isDouble = false keeps the
  // one-directional nearest neighbours (:744-760)
  int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<int>& vMatchedPairs, bool isDouble) {
    vMatchedPairs.clear();
    vMatchedPairs.resize(pKF1->NL, -1);
    if (pKF1->mLineDescriptors.rows == 0 || pKF2->mLineDescriptors.rows == 0) return 0;
    std::vector<int> m12;
    if (isDouble) hip::SearchDouble(pKF1->mLineDescriptors, pKF2->mLineDescriptors, m12, mfNNratio, (float)TH_HIGH);
    else hip::FrameBFMatch(pKF1->mLineDescriptors, pKF2->mLineDescriptors, m12, mfNNratio, (float)TH_HIGH);
    int nmatches = 0;
    for (size_t i = 0; i < m12.size() && i < vMatchedPairs.size(); i++) {
      if (m12[i] < 0) continue;
      if (pKF1->GetMapLine(i) || pKF2->GetMapLine(m12[i])) continue;
      vMatchedPairs[i] = m12[i];
      nmatches++;
    }
    return nmatches;
  }

  // LocalMapping.cc:960 (commented out in the reference): FrameBFMatchNew both ways -- nearest LBD neighbour, the segment carried over
  // the fundamental matrix has to overlap the neighbour's by more than 0.8 (:488-625) -- at TH_LOW, the mutual check if isDouble, only
  // lines without a MapLine (:780-832).  The fundamental matrices are the reference's own ComputeF12 (:834-858).
  int SearchForTriangulationNew(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<int>& vMatchedPairs, bool isDouble = false) {
    vMatchedPairs.clear();
    vMatchedPairs.resize(pKF1->NL, -1);
    if (pKF1->mLineDescriptors.rows == 0 || pKF2->mLineDescriptors.rows == 0) return 0;
    const cv::Mat F21 = ComputeF12(pKF2, pKF1), F12 = ComputeF12(pKF1, pKF2);
    const int n1 = pKF1->mLineDescriptors.rows, n2 = pKF2->mLineDescriptors.rows;
    std::vector<unsigned char> ml1(n1), ml2(n2);
    for (int i = 0; i < n1; i++) ml1[i] = pKF1->GetMapLine(i) ? 1 : 0;
    for (int j = 0; j < n2; j++) ml2[j] = pKF2->GetMapLine(j) ? 1 : 0;
    std::vector<int> m12;
    const int nmatches = hip::SearchForTriangulationNew(pKF1->mLineDescriptors, pKF2->mLineDescriptors, pKF1->mvKeyLines, pKF2->mvKeyLines,
                                                        pKF1->mvKeyLineFunctions, pKF2->mvKeyLineFunctions, F21, F12, ml1, ml2, mfNNratio,
                                                        (float)TH_LOW, isDouble, m12);
    for (size_t i = 0; i < m12.size() && i < vMatchedPairs.size(); i++) vMatchedPairs[i] = m12[i];
    return nmatches;
  }

  // LocalMapping::SearchLineInNeighbors (LocalMapping.cc:1600, :1627).  As ORBmatcher::Fuse in HipORBmatcher.h: the search for the best
  // line of every MapLine does not depend on what the loop does to the map, so it runs first, on the GPU, for every line up to the
  // first one with an endpoint behind the camera (the reference leaves the function there with `return false`, :893-894); the loop
  // then runs in the reference's order with the reference's own tests and its replace / add logic (:975-997).  Candidates' descriptor
  // rows are read from pKF->mDescriptors as the reference does (:963); rows it does not have (the reference reads past the matrix)
  // are zeros here.
  int Fuse(KeyFrame* pKF, const std::vector<MapLine*>& vpMapLines, float th = 3.0) {
    cv::Mat Rcw = pKF->GetRotation();
    cv::Mat tcw = pKF->GetTranslation();
    const float &fx = pKF->fx, &fy = pKF->fy, &cx = pKF->cx, &cy = pKF->cy;
    cv::Mat Ow = pKF->GetCameraCenter();
    const int nMLs = (int)vpMapLines.size();
    const int nLevels = (int)pKF->mvScaleFactorsLine.size();
    hip::ProjQueries q;
    q.valid.assign(nMLs, 0); q.hasObs.assign(nMLs, 1); q.pos.assign(4 * (size_t)nMLs, 0.f); q.level.assign(nMLs, 0); q.aux.assign(nMLs, 0.f);
    q.desc = cv::Mat::zeros(nMLs ? nMLs : 1, 32, CV_8U);
    int stopAt = nMLs;   // the query at which the reference returns
    for (int i = 0; i < nMLs; i++) {
      MapLine* pML = vpMapLines[i];
      if (!pML) continue;
      Vector6d P = pML->GetWorldPos();
      cv::Mat SP = (cv::Mat_<float>(3, 1) << P(0), P(1), P(2));
      cv::Mat EP = (cv::Mat_<float>(3, 1) << P(3), P(4), P(5));
      const cv::Mat SPc = Rcw * SP + tcw;
      const float &SPcX = SPc.at<float>(0), &SPcY = SPc.at<float>(1), &SPcZ = SPc.at<float>(2);
      const cv::Mat EPc = Rcw * EP + tcw;
      const float &EPcX = EPc.at<float>(0), &EPcY = EPc.at<float>(1), &EPcZ = EPc.at<float>(2);
      if (SPcZ < 0.0f || EPcZ < 0.0f) { q.aux[i] = 1.f; continue; }   // (the return happens only if the loop gets here unskipped)
      const float invz1 = 1.0f / SPcZ;
      const float u1 = fx * SPcX * invz1 + cx;
      const float v1 = fy * SPcY * invz1 + cy;
      if (!pKF->IsInImage(u1, v1)) continue;
      const float invz2 = 1.0f / EPcZ;
      const float u2 = fx * EPcX * invz2 + cx;
      const float v2 = fy * EPcY * invz2 + cy;
      if (!pKF->IsInImage(u2, v2)) continue;
      const float maxDistance = pML->GetMaxDistanceInvariance();
      const float minDistance = pML->GetMinDistanceInvariance();
      const cv::Mat OM = 0.5 * (SP + EP) - Ow;
      const float dist = cv::norm(OM);
      if (dist < minDistance || dist > maxDistance) continue;
      Eigen::Vector3d Pn = pML->GetNormal();
      cv::Mat pn = (cv::Mat_<float>(3, 1) << Pn(0), Pn(1), Pn(2));
      if (OM.dot(pn) < 0.5 * dist) continue;
      int nPredictedLevel = pML->PredictScale(dist, pKF->mfLogScaleFactorLine);
      if (nPredictedLevel < 0 || nPredictedLevel >= nLevels) continue;   // (the reference indexes mvScaleFactorsLine out of range here)
      cv::Mat CurrentLineDesc = pML->mLDescriptor;
      if (CurrentLineDesc.empty()) continue;
      q.valid[i] = 1;
      q.pos[4 * i] = u1; q.pos[4 * i + 1] = v1; q.pos[4 * i + 2] = u2; q.pos[4 * i + 3] = v2;
      q.level[i] = nPredictedLevel;
      std::memcpy(q.desc.ptr<uchar>(i), CurrentLineDesc.ptr<uchar>(0), 32);
    }
    (void)stopAt;
    std::vector<int> bestIdx(nMLs, -1);
    if (pKF->NL > 0 && nMLs > 0) {
      cv::Mat cand = cv::Mat::zeros(pKF->NL, 32, CV_8U);
      const int rows = std::min(pKF->NL, pKF->mDescriptors.rows);
      if (rows > 0 && pKF->mDescriptors.cols == 32) pKF->mDescriptors.rowRange(0, rows).copyTo(cand.rowRange(0, rows));
      hip::LineFuseSearch(pKF->mvKeyLines, cand, pKF->mvScaleFactorsLine, q, th, bestIdx, 0.998f, TH_LOW);
    }
    int nFused = 0;
    for (int i = 0; i < nMLs; i++) {
      MapLine* pML = vpMapLines[i];
      if (!pML) continue;
      if (pML->isBad() || pML->IsInKeyFrame(pKF)) continue;           // (as it stands when the loop gets here)
      if (q.aux[i] != 0.f) return false;                               // :893-894
      if (!q.valid[i] || bestIdx[i] < 0) continue;
      MapLine* pMLinKF = pKF->GetMapLine(bestIdx[i]);
      if (pMLinKF) {
        if (!pMLinKF->isBad()) {
          if (pMLinKF->Observations() > pML->Observations()) pML->Replace(pMLinKF);
          else pMLinKF->Replace(pML);
        }
      } else {
        pML->AddObservation(pKF, bestIdx[i]);
        pKF->AddMapLine(pML, bestIdx[i]);
      }
      nFused++;
    }
    return nFused;
  }

 protected:
  static plh_grid_params FrameGrid() {
    return hip::GridParams(Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY, Frame::mfGridElementWidthInv,
                           Frame::mfGridElementHeightInv);
  }
};

}  // namespace ORB_SLAM2

#endif  // LSDmatcher
#endif
