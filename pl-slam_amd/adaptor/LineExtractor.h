// Drop-in replacement for the reference's include/LineExtractor.h (ORB_SLAM2::LINEextractor, :20-62) over the
// C ABI of libplslam_hip.so.  Frame::ExtractLSD (reference src/Frame.cc:331-334) calls it unchanged:
//     (*mpLSDextractorLeft)(im, mask, mvKeylinesUn, mLdesc, mvKeyLineFunctions);
// The per-frame undistortion the reference does in the Frame constructor (Frame.cc:220-222,
// initUndistortRectifyMap + remap) can be folded into the call with SetUndistortion(K, D): the maps are then
// built once instead of once per frame, and Frame passes the raw grey image.
#ifndef PLSLAM_HIP_ADAPTOR_LINEEXTRACTOR_H
#define PLSLAM_HIP_ADAPTOR_LINEEXTRACTOR_H
#define LINEEXTRACTOR_H   // the include guard of the reference's own header: a later `#include "LineExtractor.h"` from include/Frame.h,
// KeyFrame.h or Tracking.h (sibling lookup, which no -I order can override) then finds nothing left to declare

#include <opencv2/core/core.hpp>
#include <opencv2/line_descriptor/descriptor.hpp>

#include <Eigen/Core>
#include <cassert>
#include <stdexcept>
#include <string>
#include <vector>

// Which refine level of cv::LineSegmentDetector the LSDDetector of YOUR opencv_contrib runs is a property of that build
// (src/LineExtractor.cpp:39-40 links the system module; upstream 3.x passes LSD_REFINE_ADV, the twin in the reference's tree
// LSD_REFINE_STD).  A drop-in must not pick silently: the build of the SLAM system says which, e.g.
//     add_definitions(-DPLH_LSD_REFINE_DEFAULT=1)   # 0 = LSD_REFINE_STD, 1 = LSD_REFINE_ADV
#ifndef PLH_LSD_REFINE_DEFAULT
#error "define PLH_LSD_REFINE_DEFAULT (0: LSD_REFINE_STD, 1: LSD_REFINE_ADV) for the LSDDetector your OpenCV build uses -- INTEGRATION.md section 2"
#endif
#include "plslam_hip.h"

namespace ORB_SLAM2 {

class LINEextractor {
 public:
  typedef cv::line_descriptor::KeyLine KeyLine;

  LINEextractor(int _numOctaves, float _scale, unsigned int _nLSDFeature, double _min_line_length, int device = 0)
      : mDevice(device), mHandle(nullptr), mRows(0), mCols(0), mHasUndist(false), mRefine(PLH_LSD_REFINE_DEFAULT), mGrowWaves(-1) {
    static_assert(sizeof(KeyLine) == sizeof(plh_keyline), "KeyLine must be the 68-byte POD plh_keyline mirrors");
    mParams.num_octaves = _numOctaves;
    mParams.scale = _scale;
    mParams.n_lsd_feature = _nLSDFeature;
    mParams.min_line_length = _min_line_length;
    mvScaleFactor.resize(_numOctaves); mvLevelSigma2.resize(_numOctaves);   // LineExtractor.cpp:7-23
    mvInvScaleFactor.resize(_numOctaves); mvInvLevelSigma2.resize(_numOctaves);
    mvScaleFactor[0] = 1.0f; mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < _numOctaves; i++) {
      mvScaleFactor[i] = mvScaleFactor[i - 1] * _scale;
      mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
    }
    for (int i = 0; i < _numOctaves; i++) {
      mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
      mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
    }
  }
  ~LINEextractor() { plh_line_destroy(mHandle); }
  LINEextractor(const LINEextractor&) = delete;
  LINEextractor& operator=(const LINEextractor&) = delete;

  // K = (fx, fy, cx, cy), D = (k1, k2, p1, p2, k3) as Tracking.cc:54-76 reads them from the YAML.
  void SetUndistortion(const float K[4], const float D[5]) {
    for (int i = 0; i < 4; i++) mK[i] = K[i];
    for (int i = 0; i < 5; i++) mD[i] = D[i];
    mHasUndist = true;
    if (mHandle) check(plh_line_set_undistort(mHandle, mK, mD));
  }

  // The refine level of the cv::LineSegmentDetector behind LSDDetector::detect: PLH_LSD_REFINE_STD (what the twin in the
  // reference's tree creates, LSDDetector_custom.cpp:149) or PLH_LSD_REFINE_ADV (what upstream opencv_contrib 3.x passes).  The
  // extractor starts with the level the build chose (PLH_LSD_REFINE_DEFAULT, above); this overrides it at run time.
  void SetRefine(int level) {
    mRefine = level;
    if (mHandle) check(plh_line_set_refine(mHandle, mRefine));
  }
  // Wavefronts per frame of LSD's region growing: -1 automatic, 0 one, 2..16 that many (same segments either way)
  void SetGrowWaves(int waves) {
    mGrowWaves = waves;
    if (mHandle) check(plh_line_set_grow_waves(mHandle, mGrowWaves));
  }

  void operator()(cv::InputArray _image, cv::InputArray _mask, std::vector<KeyLine>& _keylines, cv::OutputArray _descriptors,
                  std::vector<Eigen::Vector3d>& _lineVec2d) {
    if (_image.empty()) return;   // LineExtractor.cpp:29-30
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);
    cv::Mat mask = _mask.getMat();
    if (mask.data != NULL && (mask.size() != image.size() || mask.type() != CV_8UC1))
      throw std::runtime_error("Mask error while detecting lines: please check its dimensions and that data type is CV_8UC1");
    ensurePlan(image.rows, image.cols);
    const int cap = plh_line_capacity(mHandle);
    _keylines.resize(cap);
    cv::Mat desc(cap, 32, CV_8UC1);
    std::vector<double> fn((size_t)cap * 3);
    cv::Mat maskC = mask.empty() ? mask : (mask.isContinuous() ? mask : mask.clone());
    int n = 0;
    check(plh_line_extract(mHandle, image.data, image.rows, image.cols, image.step, maskC.empty() ? NULL : maskC.data,
                           reinterpret_cast<plh_keyline*>(_keylines.data()), desc.data, fn.data(), cap, &n));
    _keylines.resize(n);
    if (n == 0) {
      _descriptors.release();   // LineExtractor.cpp:70-72
      return;
    }
    _lineVec2d.clear();
    for (int i = 0; i < n; i++) _lineVec2d.push_back(Eigen::Vector3d(fn[i * 3], fn[i * 3 + 1], fn[i * 3 + 2]));
    desc.rowRange(0, n).copyTo(_descriptors);
  }

  int inline GetLevels() { return mParams.num_octaves; }
  float inline GetScaleFactor() { return mParams.scale; }
  std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
  std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
  std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
  std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

 protected:
  void ensurePlan(int rows, int cols) {
    if (mHandle && rows == mRows && cols == mCols) return;
    plh_line_destroy(mHandle);
    mHandle = nullptr;
    check(plh_line_create(&mParams, mDevice, rows, cols, 1, &mHandle));
    if (mHasUndist) check(plh_line_set_undistort(mHandle, mK, mD));
    check(plh_line_set_refine(mHandle, mRefine));
    check(plh_line_set_grow_waves(mHandle, mGrowWaves));
    mRows = rows;
    mCols = cols;
  }
  static void check(plh_status st) {
    if (st != PLH_OK) throw std::runtime_error(std::string("plslam_hip: ") + plh_last_error());
  }

  plh_line_params mParams;
  int mDevice;
  plh_line* mHandle;
  int mRows, mCols;
  bool mHasUndist;
  int mRefine, mGrowWaves;
  float mK[4], mD[5];
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

}  // namespace ORB_SLAM2

#endif
