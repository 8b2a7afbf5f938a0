// Device residency of what the searches read from ONE Frame / KeyFrame (round 6): mvKeysUn + mDescriptors + the 64 x 48 grid of
// Frame::AssignFeaturesToGrid (+ the FeatureVector node of every feature once ComputeBoW ran), and mvKeylinesUn + mLdesc +
// mvKeyLineFunctions + the line grid of AssignFeaturesToGridForLine.  A tracked frame is searched two to four times
// (TrackWithMotionModel / TrackReferenceKeyFrame, SearchLocalPoints, SearchLocalLines, then again as the last frame of the next
// one); the host-buffer calls re-uploaded it and rebuilt its grid every time.  Handles are immutable after creation: any number of
// threads may search them concurrently.
#pragma once
#include "plh_common.h"

struct plh_frame_points {
  int device = 0, n = 0;
  plh_grid_params gp{};
  uint8_t* block = nullptr;          // one allocation behind the pointers below
  plh_keypoint* kps = nullptr;       // [n]
  uint8_t* desc = nullptr;           // [n][32]
  int32_t* dn = nullptr;             // [1] = n
  int32_t* cellStart = nullptr;      // [64 * 48 + 1]
  int32_t* cellItems = nullptr;      // [n]
  int32_t* node = nullptr;           // [n] FeatureVector node of every feature (-1: none); valid once hasNodes
  bool hasNodes = false;
};
struct plh_frame_lines {
  int device = 0, nl = 0, itemCap = 0;
  plh_grid_params gp{};
  uint8_t* block = nullptr;
  plh_keyline* kl = nullptr;         // [nl]
  uint8_t* ldesc = nullptr;          // [nl][32]
  double* fn = nullptr;              // [nl][3]
  int32_t* dn = nullptr;             // [1] = nl
  int32_t* cellStart = nullptr;      // [64 * 48 + 1]
  int32_t* cellItems = nullptr;      // [nl * 64]
};
