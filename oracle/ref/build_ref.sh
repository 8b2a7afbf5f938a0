#!/bin/bash
# Build oracle/_ref/*.so from the REFERENCE's own sources, where they lie under /root/reference (nothing is copied into the
# repo), against the stand-in OpenCV / Eigen headers of oracle/ref/stub (neither library is in the image; the OpenCV
# ALGORITHMS underneath -- resize, GaussianBlur, FAST, Sobel, LSD, remap, knnMatch -- are the oracle's restatements):
#   libdbow2_ref.so      Thirdparty/DBoW2 (FORB, BowVector, FeatureVector, ScoringObject, vocabulary I/O, transform)
#   liborb_ref.so        src/ORBextractor.cc
#   libmisc_ref.so       src/lineIterator.cpp
#   libline_ref.so       src/LineExtractor.cpp + Thirdparty/line_descriptor (LSDDetector_custom, binary_descriptor_custom)
#   libmatcher_ref.so    src/ORBmatcher.cc   against stand-ins for Frame / KeyFrame / MapPoint (slam_stub.h)
#   liblsdmatcher_ref.so src/LSDmatcher.cpp  against the same stand-ins + MapLine
#   libmapobj_ref.so     src/MapPoint.cc, src/MapLine.cpp with their own headers
#   libframe_ref.so      src/Frame.cc, KeyFrame.cc, ORBmatcher.cc, LSDmatcher.cpp, MapPoint.cc, MapLine.cpp, the extractors,
#                        DBoW2, lineIterator -- the reference's real classes throughout (stand-ins: Map, KeyFrameDatabase, Converter)
# (see the echo lines below for the exact list this script produced).  oracle/ref/build_adaptor.sh builds the same harness
# once more with the PRODUCT's adaptor classes in place of the extractors and matchers (libadaptor_hip.so / libadaptor_emu.so).
# Test infrastructure only.
set -e
REF=${1:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
D=$REF/Thirdparty/DBoW2
[ -f "$D/DBoW2/TemplatedVocabulary.h" ] || { echo "no reference at $REF: skipping oracle/_ref"; exit 0; }
OUT="$HERE/../_ref"
mkdir -p "$OUT"
g++ -O2 -std=c++14 -fPIC -shared -w -I "$HERE/stub" -I "$D" -o "$OUT/libdbow2_ref.so" \
  "$HERE/ref_dbow2.cc" "$D/DBoW2/FORB.cpp" "$D/DBoW2/BowVector.cpp" "$D/DBoW2/FeatureVector.cpp" "$D/DBoW2/ScoringObject.cpp" \
  "$D/DUtils/Random.cpp" "$D/DUtils/Timestamp.cpp"
echo "built $OUT/libdbow2_ref.so"
# The reference's ORB extractor on top of the oracle's restated OpenCV primitives (oracle/img_ops.cc); same float rules as
# the oracle build (no FMA contraction).
g++ -O2 -std=c++14 -fPIC -shared -w -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wl,-Bsymbolic -I "$HERE/stub" -I "$REF/include" \
  -o "$OUT/liborb_ref.so" \
  "$HERE/ref_orb.cc" "$REF/src/ORBextractor.cc" "$HERE/../img_ops.cc"
echo "built $OUT/liborb_ref.so"
g++ -O2 -std=c++14 -fPIC -shared -w -ffp-contract=off -I "$REF/include" -o "$OUT/libmisc_ref.so" "$HERE/ref_misc.cc" "$REF/src/lineIterator.cpp"
echo "built $OUT/libmisc_ref.so"
# The reference's line path (LINEextractor + the vendored twin of opencv_contrib's line_descriptor) on top of the oracle's
# restated LSD / GaussianBlur / Sobel; same float rules as the oracle build.
LD=$REF/Thirdparty/line_descriptor
g++ -O2 -std=c++14 -fPIC -shared -w -ffp-contract=off -fno-fast-math -I "$HERE/stub" -I "$REF/include" -I "$LD/include" \
  -o "$OUT/libline_ref.so" "$HERE/ref_line.cc" "$REF/src/LineExtractor.cpp" "$LD/src/LSDDetector_custom.cpp" \
  "$LD/src/binary_descriptor_custom.cpp" "$HERE/../img_ops.cc" "$HERE/../lsd.cc" -lquadmath
echo "built $OUT/libline_ref.so"
# The oracle's grid / descriptor helpers the matcher harnesses lean on, compiled apart (slam_stub.h is force-included
# into the reference's translation units only).
for f in frame_search match img_ops; do
  g++ -O2 -std=c++14 -fPIC -c -w -ffp-contract=off -fno-fast-math -o "$OUT/plo_$f.o" "$HERE/../$f.cc"
done
PLO_OBJS="$OUT/plo_frame_search.o $OUT/plo_match.o $OUT/plo_img_ops.o"
MFLAGS="-O2 -std=c++14 -fPIC -shared -w -pthread -ffp-contract=off -fno-fast-math -DMAPPOINT_H -DKEYFRAME_H -DFRAME_H"
MINC="-I $HERE/stub -I $LD/include -I $REF/include -I $REF -include $HERE/slam_stub.h"
# The reference's ORBmatcher.cc against stand-ins for Frame / KeyFrame / MapPoint (slam_stub.h replaces the three headers,
# whose include guards are pre-defined); grid lookups and descriptor helpers from the oracle.
g++ $MFLAGS $MINC -o "$OUT/libmatcher_ref.so" "$HERE/ref_matcher.cc" "$REF/src/ORBmatcher.cc" \
  "$D/DBoW2/FeatureVector.cpp" "$D/DBoW2/BowVector.cpp" $PLO_OBJS
echo "built $OUT/libmatcher_ref.so"
# The reference's LSDmatcher.cpp against the same stand-ins (+ MapLine); cv::BFMatcher::knnMatch = the oracle's knn2.
g++ $MFLAGS $MINC -o "$OUT/liblsdmatcher_ref.so" "$HERE/ref_lsdmatcher.cc" "$REF/src/LSDmatcher.cpp" "$D/DBoW2/FeatureVector.cpp" "$D/DBoW2/BowVector.cpp" $PLO_OBJS
echo "built $OUT/liblsdmatcher_ref.so"
# The reference's MapPoint.cc / MapLine.cpp (ComputeDistinctiveDescriptors) with their own headers, against stand-ins for
# KeyFrame / Frame / Map only (mapobj_stub.h).
g++ -O2 -std=c++14 -fPIC -shared -w -pthread -ffp-contract=off -fno-fast-math -DKEYFRAME_H -DFRAME_H -DMAP_H -I "$HERE/stub" \
  -I "$LD/include" -I "$REF/include" -I "$REF" -include "$HERE/mapobj_stub.h" -o "$OUT/libmapobj_ref.so" "$HERE/ref_mapobj.cc" \
  "$REF/src/MapPoint.cc" "$REF/src/MapLine.cpp" "$OUT/plo_match.o"
echo "built $OUT/libmapobj_ref.so"
# The reference's Frame.cc with its own header and the real MapPoint / MapLine / extractor / DBoW2 / lineIterator sources
# around it (real KeyFrame.cc too; stand-ins: Map, KeyFrameDatabase, Converter); the grid assignment and the window lookups are what the harness drives.
for f in lsd; do g++ -O2 -std=c++14 -fPIC -c -w -ffp-contract=off -fno-fast-math -o "$OUT/plo_$f.o" "$HERE/../$f.cc"; done
g++ -O2 -std=c++14 -fPIC -shared -w -pthread -ffp-contract=off -fno-fast-math -DPLO_REAL_FRAME -DPLO_REAL_KEYFRAME -DMAP_H -DCONVERTER_H \
  -DLOCALMAPPING_H -DKEYFRAMEDATABASE_H -I "$HERE/stub" -I "$HERE/stub/eigen3" -I "$LD/include" -I "$REF/include" -I "$REF" -include "$HERE/frame_stub.h" \
  -o "$OUT/libframe_ref.so" "$HERE/ref_frame.cc" "$REF/src/Frame.cc" "$REF/src/KeyFrame.cc" "$REF/src/ORBmatcher.cc" \
  "$REF/src/LSDmatcher.cpp" "$REF/src/MapPoint.cc" "$REF/src/MapLine.cpp" \
  "$REF/src/lineIterator.cpp" "$REF/src/ORBextractor.cc" "$REF/src/LineExtractor.cpp" "$LD/src/LSDDetector_custom.cpp" \
  "$LD/src/binary_descriptor_custom.cpp" "$D/DBoW2/FORB.cpp" "$D/DBoW2/BowVector.cpp" "$D/DBoW2/FeatureVector.cpp" \
  "$D/DBoW2/ScoringObject.cpp" "$D/DUtils/Random.cpp" "$D/DUtils/Timestamp.cpp" $PLO_OBJS "$OUT/plo_lsd.o" -lquadmath
echo "built $OUT/libframe_ref.so"
