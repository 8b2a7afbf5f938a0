// oracle/_ref/libadaptor_{hip,emu}.so: the product's C++ boundary EXECUTED.  The reference's own include/Frame.h + src/Frame.cc
// (and KeyFrame / MapPoint / MapLine / DBoW2 / lineIterator, as in libframe_ref.so) are compiled with
//     -include pl-slam_amd/adaptor/plslam_hip_dropin.h      (+ -I pl-slam_amd/adaptor -I include)
// exactly as INTEGRATION.md section 1 tells a maintainer to, so that ORBextractor, LINEextractor, ORBmatcher and LSDmatcher are the
// adaptor classes in every translation unit; src/ORBextractor.cc and src/LineExtractor.cpp are NOT compiled, src/ORBmatcher.cc and
// src/LSDmatcher.cpp are compiled as the CPU base classes (-DORBmatcher=ORBmatcherCPU -DLSDmatcher=LSDmatcherCPU).  The library
// links libplslam_hip.so (GPU box) or the emulator build of the same sources (CPU tests).
//
// This file drives what ref_frame.cc does not: the monocular Frame constructor (src/Frame.cc:193-276) -- image in,
// Frame::ExtractORB and Frame::ExtractLSD on two threads (:224-227) through the adaptor extractors, UndistortKeyPoints,
// ComputeImageBounds, both grid assignments -- and the two initialisation matchers on constructed frames.  The tracking searches
// (SearchByProjection x3, SearchByBoW) are driven by ref_frame.cc's own harness functions, which in this build instantiate the
// adaptor ORBmatcher / LSDmatcher.  TEST INFRASTRUCTURE ONLY.
#include <cmath>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <memory>
#include <set>
#include <vector>

#define private public
#define protected public
#include "plslam_hip_dropin.h"   // what the reference's own translation units get by -include
#include "Frame.h"
#undef private
#undef protected
#include "ORBmatcher.h"          // no-ops by now: the guards of the reference's headers are set
#include "LSDmatcher.h"

using namespace ORB_SLAM2;

namespace {
struct Tracker {   // what Tracking owns (Tracking.cc:135-148)
  ORBextractor* orb;
  LINEextractor* line;
};
}  // namespace

extern "C" {

// 1 when the matcher classes in this library are the adaptor ones (they derive from the renamed reference classes)
int adx_uses_adaptor_classes() {
  return std::is_base_of<ORBmatcherCPU, ORBmatcher>::value && std::is_base_of<LSDmatcherCPU, LSDmatcher>::value ? 1 : 0;
}

void* adx_tracker_create(int nfeatures, float scale, int nlevels, int ini_th, int min_th, int nlines, double min_line_length) {
  Tracker* t = new Tracker();
  t->orb = new ORBextractor(nfeatures, scale, nlevels, ini_th, min_th);
  t->line = new LINEextractor(1, 1.2f, (unsigned)nlines, min_line_length);
  return t;
}
// LINEextractor::SetRefine: cv::LineSegmentDetector's level (the harness is built with -DPLH_LSD_REFINE_DEFAULT=1, LSD_REFINE_ADV;
// the goldens of the reference's un-linked twin are LSD_REFINE_STD)
void adx_tracker_set_refine(void* h, int level) { ((Tracker*)h)->line->SetRefine(level); }
void adx_tracker_destroy(void* h) {
  Tracker* t = (Tracker*)h;
  delete t->orb;
  delete t->line;
  delete t;
}

// Frame(imGray, timeStamp, extractorORB, extractorLine, voc, K, distCoef, bf, thDepth, mask), src/Frame.cc:193-276.
// Returns NULL (and the message in err) if an adaptor threw.
void* adx_frame_create(void* tracker, const uint8_t* img, int rows, int cols, const float K4[4], const float D5[5], const uint8_t* mask,
                       char* err, int errcap) {
  Tracker* t = (Tracker*)tracker;
  cv::Mat im(rows, cols, CV_8U);
  for (int r = 0; r < rows; r++) std::memcpy(im.ptr<uchar>(r), img + (size_t)r * cols, cols);
  cv::Mat K = cv::Mat::eye(3, 3, CV_32F);
  K.at<float>(0, 0) = K4[0]; K.at<float>(1, 1) = K4[1]; K.at<float>(0, 2) = K4[2]; K.at<float>(1, 2) = K4[3];
  cv::Mat D(5, 1, CV_32F);
  for (int i = 0; i < 5; i++) D.at<float>(i) = D5[i];
  cv::Mat m;
  if (mask) {
    m = cv::Mat(rows, cols, CV_8U);
    for (int r = 0; r < rows; r++) std::memcpy(m.ptr<uchar>(r), mask + (size_t)r * cols, cols);
  }
  Frame::mbInitialComputations = true;   // every call stands for "the first frame after a calibration change"
  try {
    return new Frame(im, 0.0, t->orb, t->line, static_cast<ORBVocabulary*>(NULL), K, D, 0.f, 0.f, m);
  } catch (const std::exception& e) {
    if (err && errcap > 0) { std::strncpy(err, e.what(), errcap - 1); err[errcap - 1] = 0; }
    return NULL;
  }
}
void adx_frame_destroy(void* h) { delete (Frame*)h; }

void adx_frame_counts(void* h, int* n, int* nl) {
  Frame* f = (Frame*)h;
  *n = f->N; *nl = f->NL;
}
// keys / keys_un: 28-byte cv::KeyPoint records; kl: 68-byte KeyLine records; bounds: mnMinX, mnMinY, mnMaxX, mnMaxY, grid inverses
void adx_frame_read(void* h, void* keys, void* keys_un, uint8_t* desc, void* kl, uint8_t* ldesc, double* fn, float bounds[6]) {
  Frame* f = (Frame*)h;
  static_assert(sizeof(cv::KeyPoint) == 28 && sizeof(KeyLine) == 68, "record layouts");
  if (f->N > 0) {
    std::memcpy(keys, f->mvKeys.data(), (size_t)f->N * 28);
    if ((int)f->mvKeysUn.size() == f->N) std::memcpy(keys_un, f->mvKeysUn.data(), (size_t)f->N * 28);
    for (int i = 0; i < f->N; i++) std::memcpy(desc + (size_t)i * 32, f->mDescriptors.ptr<uchar>(i), 32);
  }
  for (int i = 0; i < f->NL; i++) {
    std::memcpy((char*)kl + (size_t)i * 68, &f->mvKeylinesUn[i], 68);
    std::memcpy(ldesc + (size_t)i * 32, f->mLdesc.ptr<uchar>(i), 32);
    for (int k = 0; k < 3; k++) fn[3 * i + k] = f->mvKeyLineFunctions[i](k);
  }
  bounds[0] = Frame::mnMinX; bounds[1] = Frame::mnMinY; bounds[2] = Frame::mnMaxX; bounds[3] = Frame::mnMaxY;
  bounds[4] = Frame::mfGridElementWidthInv; bounds[5] = Frame::mfGridElementHeightInv;
}

// ORBmatcher(nnratio, checkOri).SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize)  (Tracking.cc:706-708)
int adx_search_for_initialization(void* h1, void* h2, float* prev_matched, int window, float nnratio, int check_ori,
                                  int32_t* matches12) {
  Frame &f1 = *(Frame*)h1, &f2 = *(Frame*)h2;
  std::vector<cv::Point2f> prev(f1.N);
  for (int i = 0; i < f1.N; i++) prev[i] = cv::Point2f(prev_matched[2 * i], prev_matched[2 * i + 1]);
  std::vector<int> m;
  ORBmatcher matcher(nnratio, check_ori != 0);
  const int n = matcher.SearchForInitialization(f1, f2, prev, m, window);
  for (int i = 0; i < f1.N; i++) {
    matches12[i] = m[i];
    prev_matched[2 * i] = prev[i].x; prev_matched[2 * i + 1] = prev[i].y;
  }
  return n;
}

// LSDmatcher(nnratio).SearchDouble(InitialFrame, CurrentFrame, LineMatches)  (Tracking.cc:711)
int adx_search_double(void* h1, void* h2, float nnratio, int32_t* matches12) {
  Frame &f1 = *(Frame*)h1, &f2 = *(Frame*)h2;
  std::vector<int> m;
  LSDmatcher matcher(nnratio);
  const int n = matcher.SearchDouble(f1, f2, m);
  for (int i = 0; i < f1.NL && i < (int)m.size(); i++) matches12[i] = m[i];
  return n;
}

// LSDmatcher(nnratio).SearchByProjection(CurrentFrame, LastFrame) -- the two-argument overload, src/LSDmatcher.cpp:19-70, no caller in the
// reference: the reference's method and the drop-in's on the same Frames.  has_ml[i]: line i of the last frame carries a MapLine (the
// method only copies the pointer: a tag address stands for the object).  out_*[j] = the last frame's line whose MapLine line j of the
// current frame received, -1 none.
int adx_line_search_by_projection_two_arg(void* h_cur, void* h_last, const uint8_t* has_ml, float nnratio, int32_t* out_ref, int32_t* out_hip,
                                          int* n_ref) {
  Frame &cur = *(Frame*)h_cur, &last = *(Frame*)h_last;
  char* const tag = reinterpret_cast<char*>(&last);
  const std::vector<MapLine*> keepLast = last.mvpMapLines, keepCur = cur.mvpMapLines;
  last.mvpMapLines.assign(last.NL, nullptr);
  for (int i = 0; i < last.NL; i++) if (has_ml[i]) last.mvpMapLines[i] = reinterpret_cast<MapLine*>(tag + 1 + i);
  int res[2] = {0, 0};
  for (int side = 0; side < 2; side++) {
    cur.mvpMapLines.assign(cur.NL, nullptr);
    if (side == 0) { LSDmatcherCPU m(nnratio); res[0] = m.SearchByProjection(cur, last); }
    else { LSDmatcher m(nnratio); res[1] = m.SearchByProjection(cur, last); }
    int32_t* out = side == 0 ? out_ref : out_hip;
    for (int j = 0; j < cur.NL; j++) out[j] = cur.mvpMapLines[j] ? (int32_t)(reinterpret_cast<char*>(cur.mvpMapLines[j]) - tag - 1) : -1;
  }
  last.mvpMapLines = keepLast; cur.mvpMapLines = keepCur;
  *n_ref = res[0];
  return res[1];
}

// LSDmatcher(nnratio).SerachForInitialize(InitialFrame, CurrentFrame, LineMatches)  (Tracking.cc:710, commented out; src/LSDmatcher.cpp:340-373):
// the reference's own method and the drop-in's on the same two Frames
int adx_serach_for_initialize(void* h1, void* h2, float nnratio, int32_t* out_ref, int32_t* out_hip, int* n_ref) {
  Frame &f1 = *(Frame*)h1, &f2 = *(Frame*)h2;
  std::vector<int> a, b;
  LSDmatcherCPU cpu(nnratio);
  LSDmatcher hipm(nnratio);
  *n_ref = cpu.SerachForInitialize(f1, f2, a);
  const int n = hipm.SerachForInitialize(f1, f2, b);
  if ((int)a.size() != f1.NL || (int)b.size() != f1.NL) return -4;
  for (int i = 0; i < f1.NL; i++) { out_ref[i] = a[i]; out_hip[i] = b[i]; }
  return n;
}

// ---------------------------------------------------------------------------------------------------------------------
// Back-end call sites on real KeyFrame / MapPoint objects: every function builds the scene TWICE and runs the reference's own
// method (ORBmatcherCPU = src/ORBmatcher.cc) on one copy and the adaptor overload (ORBmatcher, GPU) on the other; the caller
// compares what the two left behind.  Poses may rotate: both sides use the same cv::Mat algebra.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct BackScene {
  Map map; KeyFrameDatabase db;
  Frame f;
  KeyFrame* kf = nullptr;
  std::vector<std::unique_ptr<MapPoint> > pts;
  ~BackScene() { delete kf; }
};
// view[24] as in oracle/plo.h: R (9), t (3), unused (3), fx fy cx cy, unused (4), log scale factor [23]
void build_kf(BackScene& s, const void* kps28, const uint8_t* desc, int n, const float gp[6], const float T16[16], const float K4[4], int nlevels,
              float scale, ORBVocabulary* voc) {
  Frame& f = s.f;
  Frame::mnMinX = gp[0]; Frame::mnMinY = gp[1]; Frame::mnMaxX = gp[2]; Frame::mnMaxY = gp[3];
  Frame::mfGridElementWidthInv = gp[4]; Frame::mfGridElementHeightInv = gp[5];
  Frame::fx = K4[0]; Frame::fy = K4[1]; Frame::cx = K4[2]; Frame::cy = K4[3];
  f.N = n;
  f.mvKeysUn.resize(n);
  if (n > 0) std::memcpy(f.mvKeysUn.data(), kps28, (size_t)n * 28);
  f.mvKeys = f.mvKeysUn;
  f.mDescriptors = cv::Mat(n > 0 ? n : 1, 32, CV_8U);
  if (n > 0) std::memcpy(f.mDescriptors.data, desc, (size_t)n * 32);
  f.mvuRight.assign(n, -1.f);
  f.mvDepth.assign(n, -1.f);
  f.mvpMapPoints.assign(n, nullptr);
  f.mvbOutlier.assign(n, false);
  f.mnScaleLevels = nlevels;
  f.mfScaleFactor = scale;
  f.mfLogScaleFactor = std::log(scale);
  f.mvScaleFactors.resize(nlevels); f.mvLevelSigma2.resize(nlevels); f.mvInvLevelSigma2.resize(nlevels);
  f.mvScaleFactors[0] = 1.f; f.mvLevelSigma2[0] = 1.f;
  for (int i = 1; i < nlevels; i++) { f.mvScaleFactors[i] = f.mvScaleFactors[i - 1] * scale; f.mvLevelSigma2[i] = f.mvScaleFactors[i] * f.mvScaleFactors[i]; }
  for (int i = 0; i < nlevels; i++) f.mvInvLevelSigma2[i] = 1.0f / f.mvLevelSigma2[i];
  f.mbf = 0.f;
  f.mK = cv::Mat::eye(3, 3, CV_32F);
  f.mK.at<float>(0, 0) = K4[0]; f.mK.at<float>(1, 1) = K4[1]; f.mK.at<float>(0, 2) = K4[2]; f.mK.at<float>(1, 2) = K4[3];
  f.mpORBvocabulary = voc;
  f.AssignFeaturesToGrid();
  cv::Mat T(4, 4, CV_32F);
  for (int i = 0; i < 16; i++) T.at<float>(i / 4, i % 4) = T16[i];
  f.SetPose(T);
  s.kf = new KeyFrame(f, &s.map, &s.db);
}
MapPoint* add_point(BackScene& s, const float* pos, const float* normal, float dmin, float dmax, const uint8_t* desc, int nobs, long id) {
  cv::Mat P(3, 1, CV_32F);
  for (int k = 0; k < 3; k++) P.at<float>(k) = pos[k];
  s.pts.emplace_back(new MapPoint(P, s.kf, &s.map));
  MapPoint* p = s.pts.back().get();
  p->mNormalVector = cv::Mat(3, 1, CV_32F);
  for (int k = 0; k < 3; k++) p->mNormalVector.at<float>(k) = normal[k];
  p->mfMinDistance = dmin; p->mfMaxDistance = dmax;
  if (desc) { p->mDescriptor = cv::Mat(1, 32, CV_8U); std::memcpy(p->mDescriptor.data, desc, 32); }
  p->nObs = nobs;
  p->mnId = (unsigned long)id;
  return p;
}
}  // namespace

// ORBmatcher(0.75, true).SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (LoopClosing.cc:360).  The KeyFrame holds
// n keypoints; vpMatched[idx] starts with a (dummy) point where matched0[idx] is set.  out_*[idx] = id of the candidate point now
// in vpMatched[idx] (-1: none, -2: the dummy that was there).  Returns nmatches of the adaptor; *n_ref = the reference's.
int adx_loop_search_by_projection(const void* kps28, const uint8_t* desc, int n, const float gp[6], const float T16[16], const float K4[4],
                                  int nlevels, float scale, const float S16[16], const uint8_t* matched0, int npts, const float* pos,
                                  const float* normal, const float* dmin, const float* dmax, const uint8_t* mp_desc, int th,
                                  int32_t* out_ref, int32_t* out_hip, int* n_ref) {
  int res[2] = {0, 0};
  for (int side = 0; side < 2; side++) {
    BackScene s;
    build_kf(s, kps28, desc, n, gp, T16, K4, nlevels, scale, nullptr);
    const float z3[3] = {0, 0, 1};
    std::vector<MapPoint*> vpMatched(n, nullptr), cand(npts);
    for (int i = 0; i < n; i++)
      if (matched0[i]) vpMatched[i] = add_point(s, z3, z3, 0.f, 1e9f, nullptr, 1, -2);
    for (int i = 0; i < npts; i++) cand[i] = add_point(s, pos + 3 * i, normal + 3 * i, dmin[i], dmax[i], mp_desc + (size_t)i * 32, 1, i);
    cv::Mat Scw(4, 4, CV_32F);
    for (int i = 0; i < 16; i++) Scw.at<float>(i / 4, i % 4) = S16[i];
    int32_t* out = side == 0 ? out_ref : out_hip;
    if (side == 0) { ORBmatcherCPU m(0.75f, true); res[0] = m.SearchByProjection(s.kf, Scw, cand, vpMatched, th); }
    else { ORBmatcher m(0.75f, true); res[1] = m.SearchByProjection(s.kf, Scw, cand, vpMatched, th); }
    for (int i = 0; i < n; i++) out[i] = vpMatched[i] ? (int32_t)(long)vpMatched[i]->mnId : -1;
  }
  *n_ref = res[0];
  return res[1];
}

// ORBmatcher().Fuse(pKF, vpMapPoints, th) (LocalMapping.cc:1545).  The KeyFrame already holds a point with kf_obs[idx] observations
// at keypoint idx where kf_obs[idx] > 0; candidate i has cand_obs[i] observations.  What the call leaves behind, per side:
// kf_point[idx] = id of the point now at keypoint idx (candidates: their index; originals: -2 - idx; -1: none), cand_state[i] =
// bit 0 isBad, bit 1 IsInKeyFrame(pKF), bits 8.. = Observations().  Returns nFused of the adaptor; *n_ref the reference's.
int adx_local_mapping_fuse(const void* kps28, const uint8_t* desc, int n, const float gp[6], const float T16[16], const float K4[4], int nlevels,
                           float scale, const int32_t* kf_obs, int npts, const float* pos, const float* normal, const float* dmin,
                           const float* dmax, const uint8_t* mp_desc, const int32_t* cand_obs, const int32_t* order, int norder, float th,
                           int32_t* kf_point_ref, int32_t* cand_state_ref, int32_t* kf_point_hip, int32_t* cand_state_hip, int* n_ref) {
  int res[2] = {0, 0};
  for (int side = 0; side < 2; side++) {
    BackScene s;
    build_kf(s, kps28, desc, n, gp, T16, K4, nlevels, scale, nullptr);
    const float z3[3] = {0, 0, 1};
    for (int i = 0; i < n; i++)
      if (kf_obs[i] > 0) {
        MapPoint* p = add_point(s, z3, z3, 0.f, 1e9f, nullptr, 0, -2 - i);
        // real observations: the point is seen by this KeyFrame at i (Replace() walks them), plus anonymous ones for the count
        p->AddObservation(s.kf, i);
        s.kf->AddMapPoint(p, i);
        p->nObs = kf_obs[i];
      }
    std::vector<MapPoint*> cand(npts);
    for (int i = 0; i < npts; i++) cand[i] = add_point(s, pos + 3 * i, normal + 3 * i, dmin[i], dmax[i], mp_desc + (size_t)i * 32, cand_obs[i], i);
    std::vector<MapPoint*> list(norder);   // the list handed to Fuse: candidates by index, -1 = NULL, repeats allowed
    for (int k = 0; k < norder; k++) list[k] = order[k] >= 0 ? cand[order[k]] : nullptr;
    if (side == 0) { ORBmatcherCPU m; res[0] = m.Fuse(s.kf, list, th); }
    else { ORBmatcher m; res[1] = m.Fuse(s.kf, list, th); }
    int32_t* kp = side == 0 ? kf_point_ref : kf_point_hip;
    int32_t* cs = side == 0 ? cand_state_ref : cand_state_hip;
    for (int i = 0; i < n; i++) { MapPoint* p = s.kf->GetMapPoint(i); kp[i] = p ? (int32_t)(long)p->mnId : -1; }
    for (int i = 0; i < npts; i++)
      cs[i] = (cand[i]->isBad() ? 1 : 0) | (cand[i]->IsInKeyFrame(s.kf) ? 2 : 0) | (cand[i]->Observations() << 8);
  }
  *n_ref = res[0];
  return res[1];
}

// Tracking::Relocalization's ORBmatcher(0.9, true).SearchByProjection(CurrentFrame, pKF, sFound, th, ORBdist).  The current frame =
// kps / desc with pose Tc; the KeyFrame (identity pose) carries candidate point i at its keypoint i (n_kf of them: pos, ranges,
// descriptor, kf_angle = its mvKeysUn[i].angle); found[i]: the point is in sAlreadyFound; cur_has[i2]: the frame already holds a point.
// out_*[i2] = KeyFrame point now at frame keypoint i2 (-1 none, -2 the one that was there).
int adx_relocalization_search(const void* kps28, const uint8_t* desc, int n, const float gp[6], const float Tc16[16], const float K4[4],
                              int nlevels, float scale, const uint8_t* cur_has, int n_kf, const float* kf_angle, const float* pos,
                              const float* dmin, const float* dmax, const uint8_t* mp_desc, const uint8_t* found, float th, int orb_dist,
                              int32_t* out_ref, int32_t* out_hip, int* n_ref) {
  int res[2] = {0, 0};
  const float I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  for (int side = 0; side < 2; side++) {
    BackScene kfs, cur;
    std::vector<cv::KeyPoint> kk(n_kf);
    for (int i = 0; i < n_kf; i++) kk[i] = cv::KeyPoint(10.f + i % 600, 10.f + i % 400, 31.f, kf_angle[i], 50.f, 0, -1);
    std::vector<uint8_t> kd((size_t)std::max(n_kf, 1) * 32, 0);
    build_kf(kfs, kk.data(), kd.data(), n_kf, gp, I16, K4, nlevels, scale, nullptr);
    build_kf(cur, kps28, desc, n, gp, Tc16, K4, nlevels, scale, nullptr);   // (its KeyFrame is not used: the Frame is)
    Frame& F = cur.f;
    const float z3[3] = {0, 0, 1};
    for (int i = 0; i < n; i++)
      if (cur_has[i]) F.mvpMapPoints[i] = add_point(cur, z3, z3, 0.f, 1e9f, nullptr, 1, -2);
    std::set<MapPoint*> sFound;
    for (int i = 0; i < n_kf; i++) {
      MapPoint* p = add_point(kfs, pos + 3 * i, z3, dmin[i], dmax[i], mp_desc + (size_t)i * 32, 1, i);
      kfs.kf->AddMapPoint(p, i);
      if (found[i]) sFound.insert(p);
    }
    if (side == 0) { ORBmatcherCPU m(0.9f, true); res[0] = m.SearchByProjection(F, kfs.kf, sFound, th, orb_dist); }
    else { ORBmatcher m(0.9f, true); res[1] = m.SearchByProjection(F, kfs.kf, sFound, th, orb_dist); }
    int32_t* out = side == 0 ? out_ref : out_hip;
    for (int i = 0; i < n; i++) out[i] = F.mvpMapPoints[i] ? (int32_t)(long)F.mvpMapPoints[i]->mnId : -1;
    F.mvpMapPoints.assign(n, nullptr);
  }
  *n_ref = res[0];
  return res[1];
}

// ORBmatcher(0.8).Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (LoopClosing.cc:595).  The KeyFrame holds a point at keypoint idx where
// kf_has[idx].  replace_*[i] = id of the KeyFrame point candidate i is to replace (-2 - idx) or -1; kf_point_*[idx] as in the other
// Fuse.  Returns nFused of the adaptor; *n_ref the reference's.
int adx_loop_fuse(const void* kps28, const uint8_t* desc, int n, const float gp[6], const float T16[16], const float K4[4], int nlevels,
                  float scale, const float S16[16], const uint8_t* kf_has, int npts, const float* pos, const float* normal, const float* dmin,
                  const float* dmax, const uint8_t* mp_desc, float th, int32_t* replace_ref, int32_t* kf_point_ref, int32_t* replace_hip,
                  int32_t* kf_point_hip, int* n_ref) {
  int res[2] = {0, 0};
  for (int side = 0; side < 2; side++) {
    BackScene s;
    build_kf(s, kps28, desc, n, gp, T16, K4, nlevels, scale, nullptr);
    const float z3[3] = {0, 0, 1};
    for (int i = 0; i < n; i++)
      if (kf_has[i]) { MapPoint* p = add_point(s, z3, z3, 0.f, 1e9f, nullptr, 1, -2 - i); s.kf->AddMapPoint(p, i); }
    std::vector<MapPoint*> cand(npts), repl(npts, nullptr);
    for (int i = 0; i < npts; i++) cand[i] = add_point(s, pos + 3 * i, normal + 3 * i, dmin[i], dmax[i], mp_desc + (size_t)i * 32, 1, i);
    cv::Mat Scw(4, 4, CV_32F);
    for (int i = 0; i < 16; i++) Scw.at<float>(i / 4, i % 4) = S16[i];
    if (side == 0) { ORBmatcherCPU m(0.8f); res[0] = m.Fuse(s.kf, Scw, cand, th, repl); }
    else { ORBmatcher m(0.8f); res[1] = m.Fuse(s.kf, Scw, cand, th, repl); }
    int32_t* rp = side == 0 ? replace_ref : replace_hip;
    int32_t* kp = side == 0 ? kf_point_ref : kf_point_hip;
    for (int i = 0; i < npts; i++) rp[i] = repl[i] ? (int32_t)(long)repl[i]->mnId : -1;
    for (int i = 0; i < n; i++) { MapPoint* p = s.kf->GetMapPoint(i); kp[i] = p ? (int32_t)(long)p->mnId : -1; }
  }
  *n_ref = res[0];
  return res[1];
}

// ORBmatcher(0.75, true).SearchByBoW(pKF1, pKF2, vpMatches12) (LoopClosing.cc:282) with a real vocabulary (DBoW2 text file) behind
// KeyFrame::ComputeBoW.  has1 / has2: the feature carries a MapPoint.  out_*[i1] = feature of KeyFrame 2 whose point was paired, -1.
int adx_loop_search_by_bow(const char* voc_path, const void* kps1, const uint8_t* desc1, const uint8_t* has1, int n1, const void* kps2,
                           const uint8_t* desc2, const uint8_t* has2, int n2, const float gp[6], int32_t* out_ref, int32_t* out_hip,
                           int* n_ref) {
  ORBVocabulary voc;
  if (!voc.loadFromTextFile(voc_path)) return -1;
  const float I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, K4[4] = {500, 500, 320, 240};
  int res[2] = {0, 0};
  for (int side = 0; side < 2; side++) {
    BackScene a, b;
    build_kf(a, kps1, desc1, n1, gp, I16, K4, 8, 1.2f, &voc);
    build_kf(b, kps2, desc2, n2, gp, I16, K4, 8, 1.2f, &voc);
    a.kf->ComputeBoW(); b.kf->ComputeBoW();
    const float z3[3] = {0, 0, 1};
    for (int i = 0; i < n1; i++) if (has1[i]) { MapPoint* p = add_point(a, z3, z3, 0, 1e9f, nullptr, 1, i); a.kf->AddMapPoint(p, i); }
    for (int i = 0; i < n2; i++) if (has2[i]) { MapPoint* p = add_point(b, z3, z3, 0, 1e9f, nullptr, 1, i); b.kf->AddMapPoint(p, i); }
    std::vector<MapPoint*> m12;
    if (side == 0) { ORBmatcherCPU m(0.75f, true); res[0] = m.SearchByBoW(a.kf, b.kf, m12); }
    else { ORBmatcher m(0.75f, true); res[1] = m.SearchByBoW(a.kf, b.kf, m12); }
    int32_t* out = side == 0 ? out_ref : out_hip;
    for (int i = 0; i < n1; i++) out[i] = (i < (int)m12.size() && m12[i]) ? (int32_t)(long)m12[i]->mnId : -1;
  }
  *n_ref = res[0];
  return res[1];
}

// LoopClosing::ComputeSim3's ORBmatcher(0.75, true).SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (LoopClosing.cc:327).
// KeyFrame k holds n_k keypoints at pose T_k; slot i carries a map point where has_k[i] (world position, invariance range,
// descriptor).  pre12[i1] >= 0: vpMatches12[i1] enters as KeyFrame 2's point of that slot.  out_*[i1] = slot of the KeyFrame-2 point
// in vpMatches12[i1] afterwards, -1 none.
int adx_loop_search_by_sim3(const void* kps1, const uint8_t* desc1, int n1, const float T1[16], const uint8_t* has1, const float* pos1,
                            const float* dmin1, const float* dmax1, const uint8_t* mdesc1, const void* kps2, const uint8_t* desc2, int n2,
                            const float T2[16], const uint8_t* has2, const float* pos2, const float* dmin2, const float* dmax2,
                            const uint8_t* mdesc2, const float gp[6], const float K4[4], float s12, const float R12[9], const float t12[3],
                            const int32_t* pre12, float th, int32_t* out_ref, int32_t* out_hip, int* n_ref) {
  int res[2] = {0, 0};
  for (int side = 0; side < 2; side++) {
    BackScene a, b;
    build_kf(a, kps1, desc1, n1, gp, T1, K4, 8, 1.2f, nullptr);
    build_kf(b, kps2, desc2, n2, gp, T2, K4, 8, 1.2f, nullptr);
    const float z3[3] = {0, 0, 1};
    std::vector<MapPoint*> p2(n2, nullptr);
    for (int i = 0; i < n1; i++)
      if (has1[i]) { MapPoint* p = add_point(a, pos1 + 3 * i, z3, dmin1[i], dmax1[i], mdesc1 + (size_t)i * 32, 1, i); a.kf->AddMapPoint(p, i); }
    for (int i = 0; i < n2; i++)
      if (has2[i]) {
        MapPoint* p = add_point(b, pos2 + 3 * i, z3, dmin2[i], dmax2[i], mdesc2 + (size_t)i * 32, 0, i);
        b.kf->AddMapPoint(p, i);
        p->AddObservation(b.kf, i);   // GetIndexInKeyFrame(pKF2) answers from the observations (:1233)
        p2[i] = p;
      }
    std::vector<MapPoint*> m12(n1, nullptr);
    for (int i = 0; i < n1; i++)
      if (pre12[i] >= 0 && pre12[i] < n2) m12[i] = p2[pre12[i]];
    cv::Mat R(3, 3, CV_32F), t(3, 1, CV_32F);
    std::memcpy(R.data, R12, 36);
    std::memcpy(t.data, t12, 12);
    if (side == 0) { ORBmatcherCPU m(0.75f, true); res[0] = m.SearchBySim3(a.kf, b.kf, m12, s12, R, t, th); }
    else { ORBmatcher m(0.75f, true); res[1] = m.SearchBySim3(a.kf, b.kf, m12, s12, R, t, th); }
    int32_t* out = side == 0 ? out_ref : out_hip;
    for (int i = 0; i < n1; i++) out[i] = m12[i] ? (int32_t)(long)m12[i]->mnId : -1;
  }
  *n_ref = res[0];
  return res[1];
}

// LocalMapping::CreateNewMapPoints' ORBmatcher(0.6, false).SearchForTriangulation(pKF1, pKF2, F12, vMatchedIndices, false)
// (LocalMapping.cc:385) with a real vocabulary behind KeyFrame::ComputeBoW and the two poses T1 / T2 (the epipole comes from them).
// out_*[i1] = feature of KeyFrame 2 paired with feature i1, -1 none.
int adx_local_mapping_triangulation(const char* voc_path, const void* kps1, const uint8_t* desc1, const uint8_t* has1, int n1,
                                    const void* kps2, const uint8_t* desc2, const uint8_t* has2, int n2, const float gp[6],
                                    const float T1[16], const float T2[16], const float K4[4], const float F12[9], int check_ori,
                                    int32_t* out_ref, int32_t* out_hip, int* n_ref) {
  ORBVocabulary voc;
  if (!voc.loadFromTextFile(voc_path)) return -1;
  int res[2] = {0, 0};
  for (int side = 0; side < 2; side++) {
    BackScene a, b;
    build_kf(a, kps1, desc1, n1, gp, T1, K4, 8, 1.2f, &voc);
    build_kf(b, kps2, desc2, n2, gp, T2, K4, 8, 1.2f, &voc);
    a.kf->ComputeBoW(); b.kf->ComputeBoW();
    const float z3[3] = {0, 0, 1};
    for (int i = 0; i < n1; i++) if (has1[i]) { MapPoint* p = add_point(a, z3, z3, 0, 1e9f, nullptr, 1, i); a.kf->AddMapPoint(p, i); }
    for (int i = 0; i < n2; i++) if (has2[i]) { MapPoint* p = add_point(b, z3, z3, 0, 1e9f, nullptr, 1, i); b.kf->AddMapPoint(p, i); }
    cv::Mat F(3, 3, CV_32F);
    std::memcpy(F.data, F12, 36);
    std::vector<std::pair<size_t, size_t> > pairs;
    if (side == 0) { ORBmatcherCPU m(0.6f, check_ori != 0); res[0] = m.SearchForTriangulation(a.kf, b.kf, F, pairs, false); }
    else { ORBmatcher m(0.6f, check_ori != 0); res[1] = m.SearchForTriangulation(a.kf, b.kf, F, pairs, false); }
    int32_t* out = side == 0 ? out_ref : out_hip;
    for (int i = 0; i < n1; i++) out[i] = -1;
    size_t last = 0;
    for (size_t k = 0; k < pairs.size(); k++) {
      if (k && pairs[k].first <= last) return -2;   // vMatchedPairs is in index order
      last = pairs[k].first;
      out[pairs[k].first] = (int32_t)pairs[k].second;
    }
    if ((int)pairs.size() != res[side]) return -3;
  }
  *n_ref = res[0];
  return res[1];
}

// ---- the line side of LocalMapping, on real KeyFrame / MapLine objects ----
namespace {
// a KeyFrame that also holds nl keylines (68-byte KeyLine records) with their LBD descriptors; 8 line levels with factor `lscale`
void build_kf_lines(BackScene& s, const void* kps28, const uint8_t* desc, int n, const void* kl68, const uint8_t* ldesc, int nl,
                    const float gp[6], const float T16[16], const float K4[4], float lscale, const double* linefn = nullptr) {
  Frame& f = s.f;
  f.NL = nl;
  f.mvKeylinesUn.resize(nl);
  if (nl > 0) std::memcpy((void*)f.mvKeylinesUn.data(), kl68, (size_t)nl * 68);
  f.mLdesc = cv::Mat(nl > 0 ? nl : 0, 32, CV_8U);
  if (nl > 0) std::memcpy(f.mLdesc.data, ldesc, (size_t)nl * 32);
  f.mvKeyLineFunctions.assign(nl, Eigen::Vector3d(0, 0, 1));
  if (linefn)
    for (int i = 0; i < nl; i++) f.mvKeyLineFunctions[i] = Eigen::Vector3d(linefn[3 * i], linefn[3 * i + 1], linefn[3 * i + 2]);
  f.mvpMapLines.assign(nl, nullptr);
  f.mvbLineOutlier.assign(nl, false);
  f.mnScaleLevelsLine = 8;
  f.mfScaleFactorLine = lscale;
  f.mfLogScaleFactorLine = std::log(lscale);
  f.mvScaleFactorsLine.assign(8, 1.f);
  for (int i = 1; i < 8; i++) f.mvScaleFactorsLine[i] = f.mvScaleFactorsLine[i - 1] * lscale;
  build_kf(s, kps28, desc, n, gp, T16, K4, 8, 1.2f, nullptr);
}
struct LineScene : BackScene { std::vector<std::unique_ptr<MapLine> > lines; };
MapLine* add_line(LineScene& s, const float* p6, const float* normal3, float dmin, float dmax, const uint8_t* ldesc, int nobs, long id) {
  Vector6d P;
  for (int k = 0; k < 6; k++) P(k) = p6[k];
  s.lines.emplace_back(new MapLine(P, s.kf, &s.map));
  MapLine* l = s.lines.back().get();
  l->mNormalVector = Eigen::Vector3d(normal3[0], normal3[1], normal3[2]);
  l->mfMinDistance = dmin; l->mfMaxDistance = dmax;
  if (ldesc) { l->mLDescriptor = cv::Mat(1, 32, CV_8U); std::memcpy(l->mLDescriptor.data, ldesc, 32); }
  l->nObs = nobs;
  l->mnId = (unsigned long)id;
  return l;
}
}  // namespace

// LSDmatcher().SearchForTriangulation between two KeyFrames' lines (LocalMapping.cc:679 / :961).  mode 0: the vector<pair> overload;
// 1: the vector<int> overload with isDouble = true; 2: the same with isDouble = false.  has1 / has2: the line carries a MapLine.
// out_*[i] = line of KeyFrame 2 paired with line i, -1 none.
int adx_local_mapping_line_triangulation(const void* kl1, const uint8_t* ld1, const uint8_t* has1, int n1, const void* kl2,
                                         const uint8_t* ld2, const uint8_t* has2, int n2, int mode, int32_t* out_ref, int32_t* out_hip,
                                         int* n_ref) {
  const float I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, K4[4] = {500, 500, 320, 240};
  const float gp[6] = {0, 0, 640, 480, 0.1f, 0.1f}, z6[6] = {0, 0, 1, 0, 0, 2}, z3[3] = {0, 0, 1};
  int res[2] = {0, 0};
  for (int side = 0; side < 2; side++) {
    LineScene a, b;
    build_kf_lines(a, nullptr, nullptr, 0, kl1, ld1, n1, gp, I16, K4, 1.2f);
    build_kf_lines(b, nullptr, nullptr, 0, kl2, ld2, n2, gp, I16, K4, 1.2f);
    for (int i = 0; i < n1; i++) if (has1[i]) a.kf->AddMapLine(add_line(a, z6, z3, 0, 1e9f, nullptr, 1, i), i);
    for (int i = 0; i < n2; i++) if (has2[i]) b.kf->AddMapLine(add_line(b, z6, z3, 0, 1e9f, nullptr, 1, i), i);
    int32_t* out = side == 0 ? out_ref : out_hip;
    for (int i = 0; i < n1; i++) out[i] = -1;
    std::vector<std::pair<size_t, size_t> > pairs;
    std::vector<int> vec;
    int r;
    if (side == 0) {
      LSDmatcherCPU m;
      r = mode == 0 ? m.SearchForTriangulation(a.kf, b.kf, pairs) : m.SearchForTriangulation(a.kf, b.kf, vec, mode == 1);
    } else {
      LSDmatcher m;
      r = mode == 0 ? m.SearchForTriangulation(a.kf, b.kf, pairs) : m.SearchForTriangulation(a.kf, b.kf, vec, mode == 1);
    }
    res[side] = r;
    if (mode == 0) {
      for (size_t k = 0; k < pairs.size(); k++) {
        if (k && pairs[k].first <= pairs[k - 1].first) return -2;
        out[pairs[k].first] = (int32_t)pairs[k].second;
      }
      if ((int)pairs.size() != r) return -3;
    } else {
      if ((int)vec.size() != n1) return -4;
      for (int i = 0; i < n1; i++) out[i] = vec[i];
    }
  }
  *n_ref = res[0];
  return res[1];
}

// LSDmatcher().SearchForTriangulationNew(pKF1, pKF2, vMatchedPairs, isDouble) (src/LSDmatcher.cpp:780-832; LocalMapping.cc:960) on two real
// KeyFrames with poses T1 / T2 (4 x 4, row-major), their KeyLines (68-byte records), LBD rows and line equations (3 doubles per line);
// has1 / has2: the line carries a MapLine.  The reference's side computes its fundamental matrices with its own ComputeF12, the drop-in
// class with the same inherited method.  out_*[i] = line of KeyFrame 2 paired with line i, -1 none.
static double g_tri_new_us[2] = {0, 0};   // wall microseconds of the last call's two method calls: [0] the reference's, [1] the drop-in's
void adx_local_mapping_line_triangulation_new_us(double out[2]) { out[0] = g_tri_new_us[0]; out[1] = g_tri_new_us[1]; }
int adx_local_mapping_line_triangulation_new(const void* kl1, const uint8_t* ld1, const double* fn1, const uint8_t* has1, int n1,
                                             const void* kl2, const uint8_t* ld2, const double* fn2, const uint8_t* has2, int n2,
                                             const float T1[16], const float T2[16], const float K4[4], int is_double, int32_t* out_ref,
                                             int32_t* out_hip, int* n_ref) {
  const float gp[6] = {0, 0, 640, 480, 0.1f, 0.1f}, z6[6] = {0, 0, 1, 0, 0, 2}, z3[3] = {0, 0, 1};
  int res[2] = {0, 0};
  for (int side = 0; side < 2; side++) {
    LineScene a, b;
    build_kf_lines(a, nullptr, nullptr, 0, kl1, ld1, n1, gp, T1, K4, 1.2f, fn1);
    build_kf_lines(b, nullptr, nullptr, 0, kl2, ld2, n2, gp, T2, K4, 1.2f, fn2);
    for (int i = 0; i < n1; i++) if (has1[i]) a.kf->AddMapLine(add_line(a, z6, z3, 0, 1e9f, nullptr, 1, i), i);
    for (int i = 0; i < n2; i++) if (has2[i]) b.kf->AddMapLine(add_line(b, z6, z3, 0, 1e9f, nullptr, 1, i), i);
    int32_t* out = side == 0 ? out_ref : out_hip;
    for (int i = 0; i < n1; i++) out[i] = -1;
    std::vector<int> vec;
    const auto t0 = std::chrono::steady_clock::now();
    if (side == 0) { LSDmatcherCPU m; res[0] = m.SearchForTriangulationNew(a.kf, b.kf, vec, is_double != 0); }
    else { LSDmatcher m; res[1] = m.SearchForTriangulationNew(a.kf, b.kf, vec, is_double != 0); }
    g_tri_new_us[side] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if ((int)vec.size() != n1) return -4;
    for (int i = 0; i < n1; i++) out[i] = vec[i];
  }
  *n_ref = res[0];
  return res[1];
}

// LSDmatcher().Fuse(pKF, vpMapLines, th) (LocalMapping.cc:1600).  The KeyFrame holds nl keylines (and n ORB rows: the reference reads
// the candidates' descriptor rows from pKF->mDescriptors, LSDmatcher.cpp:963); it already holds a MapLine with kf_obs[idx]
// observations at line idx where kf_obs[idx] > 0.  Candidate i: endpoints pos6, normal, invariance range, LBD descriptor, cand_obs[i]
// observations; `order` is the list handed to Fuse (-1 = NULL, repeats allowed).  What the call leaves behind, per side:
// kf_line[idx] = id of the MapLine now at line idx (candidates: index; originals: -2 - idx; -1 none), cand_state[i] = bit 0 isBad,
// bit 1 IsInKeyFrame(pKF), bits 8.. Observations().
int adx_local_mapping_line_fuse(const void* kps28, const uint8_t* desc, int n, const void* kl68, const uint8_t* ldesc, int nl,
                                const float gp[6], const float T16[16], const float K4[4], float lscale, const int32_t* kf_obs, int nc,
                                const float* pos6, const float* normal, const float* dmin, const float* dmax, const uint8_t* cdesc,
                                const int32_t* cand_obs, const int32_t* order, int norder, float th, int32_t* kf_line_ref,
                                int32_t* cand_state_ref, int32_t* kf_line_hip, int32_t* cand_state_hip, int* n_ref) {
  int res[2] = {0, 0};
  const float z6[6] = {0, 0, 1, 0, 0, 2}, z3[3] = {0, 0, 1};
  for (int side = 0; side < 2; side++) {
    LineScene s;
    build_kf_lines(s, kps28, desc, n, kl68, ldesc, nl, gp, T16, K4, lscale);
    for (int i = 0; i < nl; i++)
      if (kf_obs[i] > 0) {
        MapLine* l = add_line(s, z6, z3, 0.f, 1e9f, ldesc + (size_t)i * 32, 0, -2 - i);
        l->AddObservation(s.kf, i);
        s.kf->AddMapLine(l, i);
        l->nObs = kf_obs[i];
      }
    std::vector<MapLine*> cand(nc);
    for (int i = 0; i < nc; i++)
      cand[i] = add_line(s, pos6 + 6 * i, normal + 3 * i, dmin[i], dmax[i], cdesc + (size_t)i * 32, cand_obs[i], i);
    std::vector<MapLine*> list(norder);
    for (int k = 0; k < norder; k++) list[k] = order[k] >= 0 ? cand[order[k]] : nullptr;
    if (side == 0) { LSDmatcherCPU m; res[0] = m.Fuse(s.kf, list, th); }
    else { LSDmatcher m; res[1] = m.Fuse(s.kf, list, th); }
    int32_t* kp = side == 0 ? kf_line_ref : kf_line_hip;
    int32_t* cs = side == 0 ? cand_state_ref : cand_state_hip;
    for (int i = 0; i < nl; i++) { MapLine* l = s.kf->GetMapLine(i); kp[i] = l ? (int32_t)(long)l->mnId : -1; }
    for (int i = 0; i < nc; i++)
      cs[i] = (cand[i]->isBad() ? 1 : 0) | (cand[i]->IsInKeyFrame(s.kf) ? 2 : 0) | (cand[i]->Observations() << 8);
  }
  *n_ref = res[0];
  return res[1];
}

// the static helpers
int adx_descriptor_distance(const uint8_t* a, const uint8_t* b) {
  cv::Mat ma(1, 32, CV_8U), mb(1, 32, CV_8U);
  std::memcpy(ma.data, a, 32);
  std::memcpy(mb.data, b, 32);
  return ORBmatcher::DescriptorDistance(ma, mb) * 1000 + LSDmatcher::DescriptorDistance(ma, mb);
}

}  // extern "C"
