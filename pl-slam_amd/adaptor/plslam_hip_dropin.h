// plslam_hip_dropin.h -- the one header a maintainer force-includes into every translation unit of the reference build
// (`-include <repo>/pl-slam_amd/adaptor/plslam_hip_dropin.h`) to put the GPU front end behind the
// reference's own class names without touching a source file:
//   * ORB_SLAM2::ORBextractor / LINEextractor are declared here first, under the reference's own include guards, so the
//     `#include "ORBextractor.h"` / `"LineExtractor.h"` of include/Frame.h, KeyFrame.h and Tracking.h (which always resolve to
//     their sibling files, whatever the -I order) have nothing left to declare;
//   * ORB_SLAM2::ORBVocabulary becomes a class derived from the reference's DBoW2::TemplatedVocabulary whose
//     transform(features, BowVector&, FeatureVector&, levelsup) -- Frame::ComputeBoW, KeyFrame::ComputeBoW -- runs on the GPU
//     (ORBVocabulary.h, same guard trick; everything else of the vocabulary stays the reference's host code);
//   * ORB_SLAM2::ORBmatcher / LSDmatcher become classes derived from the reference's own (read once under the names
//     ORBmatcherCPU / LSDmatcherCPU), with the tracking-path searches re-declared on top of the C ABI.
// src/ORBextractor.cc and src/LineExtractor.cpp leave the build; src/ORBmatcher.cc and src/LSDmatcher.cpp stay in it, compiled
// with -DORBmatcher=ORBmatcherCPU -DLSDmatcher=LSDmatcherCPU (they are the base classes).
#ifndef PLSLAM_HIP_DROPIN_H
#define PLSLAM_HIP_DROPIN_H
#ifdef __cplusplus
#include "ORBextractor.h"
#include "LineExtractor.h"
#include "ORBVocabulary.h"
#if !defined(ORBmatcher) && !defined(LSDmatcher)
#include "HipORBmatcher.h"
#include "HipLSDmatcher.h"
#endif
#endif
#endif
