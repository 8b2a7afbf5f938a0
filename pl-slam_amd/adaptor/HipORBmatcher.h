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
// the overloads above and inherits every other method (KeyFrame-KeyFrame BoW, SearchForTriangulation, Fuse, SearchBySim3, the
// relocalisation / loop-closing projections) from the reference's src/ORBmatcher.cc, which the maintainer keeps compiling with
// -DORBmatcher=ORBmatcherCPU -DLSDmatcher=LSDmatcherCPU (one line in CMakeLists.txt).  The back-end
// methods have GPU entry points too (plh_orb_*_batch_dev); moving one over is a matter of adding its overload here.
#ifndef PLSLAM_HIP_ADAPTOR_ORBMATCHER_H
#define PLSLAM_HIP_ADAPTOR_ORBMATCHER_H

#ifndef ORBmatcher   // (defined as a macro only in the translation units of src/ORBmatcher.cc / src/LSDmatcher.cpp themselves)

#include <MapPoint.h>   // read the classes the matcher header pulls in BEFORE the rename is active
#include <KeyFrame.h>
#include <Frame.h>
#define ORBmatcher ORBmatcherCPU
#include <ORBmatcher.h>   // the reference's include/ORBmatcher.h, found on the include path
#undef ORBmatcher

#include <stdexcept>

#include "HipMatchers.h"

namespace ORB_SLAM2 {

class ORBmatcher : public ORBmatcherCPU {
 public:
  ORBmatcher(float nnratio = 0.6, bool checkOri = true) : ORBmatcherCPU(nnratio, checkOri) {}

  // every overload that is not re-declared below stays visible
  using ORBmatcherCPU::SearchByProjection;
  using ORBmatcherCPU::SearchByBoW;

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
    const int nmatches = hip::SearchByProjection(F.mvKeysUn, F.mDescriptors, FrameGrid(), F.mvScaleFactors, occupied, q, th, mfNNratio,
                                                 assigned);
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
    hip::ProjQueries q;
    q.valid.assign(n, 0); q.hasObs.assign(n, 0); q.pos.assign(2 * (size_t)n, 0.f); q.level.assign(n, 0); q.aux.assign(n, 0.f);
    q.desc = cv::Mat::zeros(n ? n : 1, 32, CV_8U);
    for (int i = 0; i < n; i++) {
      MapPoint* pMP = LastFrame.mvpMapPoints[i];
      if (!pMP || LastFrame.mvbOutlier[i]) continue;
      cv::Mat x3Dw = pMP->GetWorldPos();
      cv::Mat x3Dc = Rcw * x3Dw + tcw;
      const float xc = x3Dc.at<float>(0);
      const float yc = x3Dc.at<float>(1);
      const float invzc = 1.0 / x3Dc.at<float>(2);
      if (invzc < 0) continue;
      q.valid[i] = !pMP->GetDescriptor().empty();   // (no descriptor: never matches, ORBmatcher.cc:1530-1531)
      q.pos[2 * i] = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
      q.pos[2 * i + 1] = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
      q.level[i] = LastFrame.mvKeys[i].octave;
      q.aux[i] = LastFrame.mvKeysUn[i].angle;
      q.hasObs[i] = pMP->Observations() > 0;
      const cv::Mat d = pMP->GetDescriptor();
      if (d.data) std::memcpy(q.desc.ptr<uchar>(i), d.ptr<uchar>(0), 32);
    }
    std::vector<uchar> occupied(CurrentFrame.N);          // :1518-1520
    for (int i = 0; i < CurrentFrame.N; i++)
      occupied[i] = CurrentFrame.mvpMapPoints[i] && CurrentFrame.mvpMapPoints[i]->Observations() > 0;
    std::vector<int> assigned;
    if (CurrentFrame.N == 0 || n == 0) return 0;
    const int nmatches = hip::SearchByProjectionLastFrame(CurrentFrame.mvKeysUn, CurrentFrame.mDescriptors, FrameGrid(),
                                                          CurrentFrame.mvScaleFactors, occupied, q, th, bForward ? 1 : bBackward ? 2 : 0,
                                                          mbCheckOrientation, assigned);
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
    const int n = hip::SearchByBoW(pKF->mDescriptors, pKF->mvKeysUn, hip::NodeOfFeature(pKF->mFeatVec, pKF->N), valid, F.mDescriptors,
                                   F.mvKeys, hip::NodeOfFeature(F.mFeatVec, F.N), mfNNratio, mbCheckOrientation, matchKF, TH_LOW);
    for (int j = 0; j < F.N; j++)
      if (matchKF[j] >= 0) vpMapPointMatches[j] = vpMapPointsKF[matchKF[j]];
    return n;
  }

  // MonocularInitialization
  int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12,
                              int windowSize = 10) {
    return hip::SearchForInitialization(F1.mvKeysUn, F1.mDescriptors, F2.mvKeysUn, F2.mDescriptors, FrameGrid(), vbPrevMatched,
                                        vnMatches12, windowSize, mfNNratio, mbCheckOrientation);
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
  static plh_grid_params FrameGrid() {
    return hip::GridParams(Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY, Frame::mfGridElementWidthInv,
                           Frame::mfGridElementHeightInv);
  }
};

}  // namespace ORB_SLAM2

#endif  // ORBmatcher
#endif
