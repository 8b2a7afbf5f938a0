// Drop-in replacement for the reference's include/LSDmatcher.h: the class ORB_SLAM2::LSDmatcher with the reference's
// constructor and method signatures (include/LSDmatcher.h:22-76), whose tracking-path searches run on the GPU through the
// C ABI of libplslam_hip.so:
//     LSDmatcher::SearchByProjection(Frame& Cur, const Frame& Last, th)     src/LSDmatcher.cpp:72-176    (Tracking.cc:1340-1357)
//     LSDmatcher::SearchByProjection(Frame&, const vector<MapLine*>&, th)   :221-338                     (Tracking.cc:1849)
//     LSDmatcher::SearchDouble(Frame&, Frame&, vector<int>&)                :427-460                     (Tracking.cc:711)
//     LSDmatcher::SearchDouble(KeyFrame*, Frame&)                           :375-425                     (Tracking.cc:1159)
//     LSDmatcher::DescriptorDistance                                        :654-670
// Same construction as adaptor/HipORBmatcher.h: the reference's own class is read as LSDmatcherCPU, the class below derives
// from it and inherits everything it does not re-declare (SearchForTriangulation*, Fuse, ...); the maintainer compiles
// src/LSDmatcher.cpp with -DORBmatcher=ORBmatcherCPU -DLSDmatcher=LSDmatcherCPU.  The reference's debugging pictures
// (matchResultTrack.jpg, :67 / :171 / :422) are not written.
#ifndef PLSLAM_HIP_ADAPTOR_LSDMATCHER_H
#define PLSLAM_HIP_ADAPTOR_LSDMATCHER_H

#ifndef LSDmatcher

#include <MapLine.h>
#include <KeyFrame.h>
#include <Frame.h>
#define LSDmatcher LSDmatcherCPU
#include <LSDmatcher.h>   // the reference's include/LSDmatcher.h
#undef LSDmatcher

#include "HipMatchers.h"

namespace ORB_SLAM2 {

class LSDmatcher : public LSDmatcherCPU {
 public:
  LSDmatcher(float nnratio = 0.7, bool checkOri = true) : LSDmatcherCPU(nnratio, checkOri) {}

  using LSDmatcherCPU::SearchByProjection;   // the two-argument overload (:178-219) stays the reference's

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
    const int nmatches = hip::LineSearchByProjection(CurrentFrame.mvKeylinesUn, CurrentFrame.mLdesc, CurrentFrame.mvKeyLineFunctions,
                                                     FrameGrid(), occupied, q, th, mfNNratio, true, assigned);
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
    const int nmatches = hip::LineSearchByProjection(F.mvKeylinesUn, F.mLdesc, F.mvKeyLineFunctions, FrameGrid(), occupied, q, th,
                                                     mfNNratio, false, assigned);
    for (int i = 0; i < F.NL; i++)
      if (assigned[i] >= 0) F.mvpMapLines[i] = vpMapLines[assigned[i]];
    return nmatches;
  }

  // MonocularInitialization
  int SearchDouble(Frame& InitialFrame, Frame& CurrentFrame, std::vector<int>& LineMatches) {
    LineMatches = std::vector<int>(InitialFrame.NL, -1);
    if (InitialFrame.mLdesc.rows == 0 || CurrentFrame.mLdesc.rows == 0) return 0;
    return hip::SearchDouble(InitialFrame.mLdesc, CurrentFrame.mLdesc, LineMatches, mfNNratio, (float)TH_LOW);
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

 protected:
  static plh_grid_params FrameGrid() {
    return hip::GridParams(Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY, Frame::mfGridElementWidthInv,
                           Frame::mfGridElementHeightInv);
  }
};

}  // namespace ORB_SLAM2

#endif  // LSDmatcher
#endif
