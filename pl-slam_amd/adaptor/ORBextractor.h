// Drop-in replacement for the reference's include/ORBextractor.h (ORB_SLAM2::ORBextractor, :45-111):
// same class name, constructor, operator() and getters, implemented over the C ABI of libplslam_hip.so.
// Frame::ExtractORB (reference src/Frame.cc:322-328) calls it unchanged:
//     (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors);
// Compiled only where OpenCV headers exist (they do not in the build image; tests/test_adaptor.py
// syntax-checks this file against tests/cv_stub/).
#ifndef PLSLAM_HIP_ADAPTOR_ORBEXTRACTOR_H
#define PLSLAM_HIP_ADAPTOR_ORBEXTRACTOR_H
#define ORBEXTRACTOR_H   // the include guard of the reference's own header: a later `#include "ORBextractor.h"` from include/Frame.h,
// KeyFrame.h or Tracking.h (sibling lookup, which no -I order can override) then finds nothing left to declare

#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>

#include <cassert>
#include <stdexcept>
#include <string>
#include <vector>

#include "plslam_hip.h"

namespace ORB_SLAM2 {

class ORBextractor {
 public:
  enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device = 0)
      : mDevice(device), mHandle(nullptr), mRows(0), mCols(0) {
    static_assert(sizeof(cv::KeyPoint) == sizeof(plh_keypoint), "cv::KeyPoint must be the 28-byte POD plh_keypoint mirrors");
    mParams.nfeatures = nfeatures;
    mParams.scale_factor = scaleFactor;
    mParams.nlevels = nlevels;
    mParams.ini_th_fast = iniThFAST;
    mParams.min_th_fast = minThFAST;
    // scale tables exactly as the reference constructor computes them (ORBextractor.cc:415-431); scaleFactor is
    // held as double there (ORBextractor.h:95)
    const double sf = scaleFactor;
    mvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels);
    mvInvScaleFactor.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
    mvScaleFactor[0] = 1.0f; mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
      mvScaleFactor[i] = (float)(mvScaleFactor[i - 1] * sf);
      mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
    }
    for (int i = 0; i < nlevels; i++) {
      mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
      mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
    }
  }

  ~ORBextractor() { plh_orb_destroy(mHandle); }
  ORBextractor(const ORBextractor&) = delete;
  ORBextractor& operator=(const ORBextractor&) = delete;

  // Compute the ORB features and descriptors on an image.  Mask is ignored, as in the reference.
  void operator()(cv::InputArray _image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint>& _keypoints,
                  cv::OutputArray _descriptors) {
    if (_image.empty()) return;   // ORBextractor.cc:1046-1047
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);
    ensurePlan(image.rows, image.cols);
    const int cap = plh_orb_capacity(mHandle);
    _keypoints.resize(cap);
    mDescBuf.create(cap, 32, CV_8U);
    int n = 0;
    check(plh_orb_extract(mHandle, image.data, image.rows, image.cols, image.step, reinterpret_cast<plh_keypoint*>(_keypoints.data()),
                          mDescBuf.data, cap, &n));
    _keypoints.resize(n);
    if (n == 0) {
      _descriptors.release();   // ORBextractor.cc:1064-1065
    } else {
      mDescBuf.rowRange(0, n).copyTo(_descriptors);   // n x 32 CV_8U (ORBextractor.cc:1068-1069)
    }
  }

  int inline GetLevels() { return mParams.nlevels; }
  float inline GetScaleFactor() { return mParams.scale_factor; }
  std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
  std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
  std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
  std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

  // The reference exposes the pyramid of the last call (only Frame::ComputeStereoMatches reads it, which the
  // monocular project never runs).  It stays on the GPU; call this to materialise border-less host copies.
  std::vector<cv::Mat> mvImagePyramid;
  void DownloadPyramid() {
    mvImagePyramid.resize(mParams.nlevels);
    for (int l = 0; l < mParams.nlevels; l++) {
      const uint8_t* d = nullptr;
      int rows = 0, cols = 0;
      size_t pitch = 0;
      check(plh_orb_pyramid_dev(mHandle, 0, l, &d, &rows, &cols, &pitch));
      mvImagePyramid[l].create(rows, cols, CV_8UC1);
      check(plh_orb_read_level(mHandle, 0, l, mvImagePyramid[l].data, (size_t)rows * cols));
    }
  }

 protected:
  void ensurePlan(int rows, int cols) {
    if (mHandle && rows == mRows && cols == mCols) return;
    plh_orb_destroy(mHandle);
    mHandle = nullptr;
    check(plh_orb_create(&mParams, mDevice, rows, cols, 1, &mHandle));
    mRows = rows;
    mCols = cols;
  }
  static void check(plh_status st) {
    if (st != PLH_OK) throw std::runtime_error(std::string("plslam_hip: ") + plh_last_error());
  }

  plh_orb_params mParams;
  int mDevice;
  plh_orb* mHandle;
  int mRows, mCols;
  cv::Mat mDescBuf;
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

}  // namespace ORB_SLAM2

#endif
