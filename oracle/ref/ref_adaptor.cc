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
#include <cstdint>
#include <cstring>
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

// the static helpers
int adx_descriptor_distance(const uint8_t* a, const uint8_t* b) {
  cv::Mat ma(1, 32, CV_8U), mb(1, 32, CV_8U);
  std::memcpy(ma.data, a, 32);
  std::memcpy(mb.data, b, 32);
  return ORBmatcher::DescriptorDistance(ma, mb) * 1000 + LSDmatcher::DescriptorDistance(ma, mb);
}

}  // extern "C"
