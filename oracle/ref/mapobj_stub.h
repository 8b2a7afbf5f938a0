// oracle/_ref: stand-ins for KeyFrame / Frame / Map so that the reference's own include/MapPoint.h + src/MapPoint.cc and
// include/MapLine.h + src/MapLine.cpp compile as they are (their three headers are skipped by pre-defining the include
// guards KEYFRAME_H, FRAME_H, MAP_H; force-included by oracle/ref/build_ref.sh).  Only the members those two sources touch.
// TEST INFRASTRUCTURE ONLY.
#ifndef PLO_REF_MAPOBJ_STUB_H
#define PLO_REF_MAPOBJ_STUB_H
#include <map>
#include <mutex>
#include <set>
#include <vector>

#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include <opencv2/line_descriptor/descriptor.hpp>
#include <eigen3/Eigen/Core>

using namespace std;   // the reference's headers rely on it
using namespace cv;
using namespace cv::line_descriptor;
using namespace Eigen;
typedef Eigen::Matrix<double, 6, 1> Vector6d;   // include/auxiliar.h (which drags MapLine.h in again) is not needed here

namespace ORB_SLAM2 {

class MapPoint;
class MapLine;

class Map {
 public:
  std::mutex mMutexPointCreation, mMutexLineCreation;
  void EraseMapPoint(MapPoint*) {}
  void EraseMapLine(MapLine*) {}
  void EraseKeyFrame(class KeyFrame*) {}
};

#ifdef PLO_REAL_FRAME   // the Frame.cc build uses the reference's own include/Frame.h
class Frame;
#else
class Frame {
 public:
  long unsigned int mnId = 0;
  int mnScaleLevels = 8, mnScaleLevelsLine = 1;
  float mfLogScaleFactor = 0.1823f, mfLogScaleFactorLine = 0.3466f;
  std::vector<float> mvScaleFactors, mvScaleFactorsLine;
  std::vector<cv::KeyPoint> mvKeysUn;
  std::vector<KeyLine> mvKeylinesUn;
  cv::Mat mDescriptors, mLdesc, mOw;
  cv::Mat GetCameraCenter() const { return mOw.clone(); }
};
#endif

#ifdef PLO_REAL_KEYFRAME   // the KeyFrame.cc build uses the reference's own include/KeyFrame.h
class KeyFrame;
class KeyFrameDatabase {   // include/KeyFrameDatabase.h (skipped through its include guard)
 public:
  void erase(KeyFrame*) {}
};
#else
class KeyFrame {
 public:
  long unsigned int mnId = 0, mnFrameId = 0;
  bool bad = false;
  int mnScaleLevels = 8, mnScaleLevelsLine = 1;
  float mfLogScaleFactor = 0.1823f, mfLogScaleFactorLine = 0.3466f;
  std::vector<float> mvScaleFactors, mvScaleFactorsLine, mvuRight;
  std::vector<cv::KeyPoint> mvKeysUn;
  std::vector<KeyLine> mvKeyLines, mvKeylinesUn;
  cv::Mat mDescriptors, mLineDescriptors, mOw;
  bool isBad() const { return bad; }
  cv::Mat GetCameraCenter() const { return mOw.clone(); }
  void EraseMapPointMatch(const size_t&) {}
  void EraseMapPointMatch(MapPoint*) {}
  void ReplaceMapPointMatch(const size_t&, MapPoint*) {}
  void EraseMapLineMatch(const size_t&) {}
  void EraseMapLineMatch(MapLine*) {}
  void ReplaceMapLineMatch(const size_t&, MapLine*) {}
};
#endif

}  // namespace ORB_SLAM2
#endif
