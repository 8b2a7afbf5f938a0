// ORACLE (test infrastructure) -- CPU restatement of the Hamming matching on the hot path:
//   ORBmatcher::DescriptorDistance        reference src/ORBmatcher.cc:1764-1780
//   ORBmatcher::SearchByBoW(KF, F, ...)   reference src/ORBmatcher.cc:187-327 (+ ComputeThreeMaxima :1718-1759)
//   LSDmatcher::FrameBFMatch              reference src/LSDmatcher.cpp:462-486
//   LSDmatcher::lineDescriptorMAD         reference src/LSDmatcher.cpp:627-652
//   LSDmatcher::SearchDouble(F, F, ...)   reference src/LSDmatcher.cpp:427-460
//   cv::BFMatcher(NORM_HAMMING).knnMatch  (OpenCV, not in tree)
// Frame / KeyFrame / MapPoint objects are replaced by flat arrays (descriptors, angles, DBoW2 node id per
// feature, "has a live MapPoint" flags); the selection logic is restated statement by statement.
// PARITY UNPINNED, see oracle/plo.h.
#include "plo.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <map>
#include <vector>

namespace {
const int TH_LOW_ORB = 50, HISTO_LENGTH = 30;

// ORBmatcher::ComputeThreeMaxima, ORBmatcher.cc:1718-1759
void three_maxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = (int)histo[i].size();
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
    else if (s > max3) { max3 = s; ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}
}  // namespace

extern "C" {

// Bit-set count from the Stanford bithacks page, as the reference writes it.
int plo_descriptor_distance(const uint8_t* a, const uint8_t* b) {
  int32_t pa[8], pb[8];
  memcpy(pa, a, 32);
  memcpy(pb, b, 32);
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    unsigned int v = (unsigned)(pa[i] ^ pb[i]);
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return dist;
}

// knnMatch(q, t, k=2): exhaustive, insertion with strict '<' (ties keep the lower train index first).
// Empty slots (nt < 2): idx -1, dist INT_MAX.
void plo_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* dist) {
  for (int i = 0; i < nq; i++) {
    int b0 = INT_MAX, b1 = INT_MAX, i0 = -1, i1 = -1;
    for (int j = 0; j < nt; j++) {
      const int d = plo_descriptor_distance(q + (size_t)i * 32, t + (size_t)j * 32);
      if (d < b0) { b1 = b0; i1 = i0; b0 = d; i0 = j; }
      else if (d < b1) { b1 = d; i1 = j; }
    }
    idx[i * 2] = i0; idx[i * 2 + 1] = i1;
    dist[i * 2] = b0; dist[i * 2 + 1] = b1;
  }
}

// lineDescriptorMAD on a knn2 distance table (nq x 2, distances as the float the reference stores).
void plo_line_mad(const int32_t* dist, int nq, double* nn_mad, double* nn12_mad) {
  *nn_mad = 0; *nn12_mad = 0;
  if (nq <= 0) return;
  std::vector<float> d0(nq), gap(nq);
  for (int i = 0; i < nq; i++) { d0[i] = (float)dist[i * 2]; gap[i] = (float)dist[i * 2 + 1] - (float)dist[i * 2]; }
  std::vector<float> a = d0;
  std::sort(a.begin(), a.end());
  double nn_dist_median = a[nq / 2];
  for (int i = 0; i < nq; i++) a[i] = fabsf((float)(d0[i] - nn_dist_median));
  std::sort(a.begin(), a.end());
  *nn_mad = 1.4826 * a[nq / 2];
  std::vector<float> g = gap;
  std::sort(g.begin(), g.end(), [](float x, float y) { return x > y; });   // conpare_descriptor_by_NN12_dist: descending gap
  double nn12_dist_median = g[nq / 2];
  for (int i = 0; i < nq; i++) g[i] = fabsf((float)(gap[i] - nn12_dist_median));
  std::sort(g.begin(), g.end());
  *nn12_mad = 1.4826 * g[nq / 2];
}

// LSDmatcher::FrameBFMatch.  Guards: the reference indexes lmatches[i][1] (UB when ldesc2 has < 2 rows) and
// matches[size/2] (UB when empty) -> no matches in those cases.
void plo_line_bfmatch(const uint8_t* d1, int n1, const uint8_t* d2, int n2, float TH, float nnratio, int32_t* matches) {
  for (int i = 0; i < n1; i++) matches[i] = -1;
  if (n1 <= 0 || n2 < 2) return;
  std::vector<int32_t> idx((size_t)n1 * 2), dist((size_t)n1 * 2);
  plo_knn2(d1, n1, d2, n2, idx.data(), dist.data());
  double nn_mad, nn12_mad;
  plo_line_mad(dist.data(), n1, &nn_mad, &nn12_mad);
  const double nn12_dist_th = nn12_mad * 0.5;
  for (int i = 0; i < n1; i++) {
    const float m0 = (float)dist[i * 2], m1 = (float)dist[i * 2 + 1];
    const double dist_12 = m1 - m0;
    if (dist_12 > nn12_dist_th && m0 < TH && m0 < nnratio * m1) matches[i] = idx[i * 2];
  }
}

// LSDmatcher::SearchDouble(Frame&, Frame&, vector<int>&): both directions + mutual consistency.
int plo_line_search_double(const uint8_t* d1, int n1, const uint8_t* d2, int n2, float TH, float nnratio,
                           int32_t* matches12) {
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  if (n1 == 0 || n2 == 0) return 0;
  std::vector<int32_t> m1(std::max(n1, 1)), m2(std::max(n2, 1));
  plo_line_bfmatch(d1, n1, d2, n2, TH, nnratio, m1.data());
  plo_line_bfmatch(d2, n2, d1, n1, TH, nnratio, m2.data());
  int nmatches = 0;
  for (int i = 0; i < n1; i++) {
    const int j = m1[i];
    if (j >= 0) {
      if (m2[j] != i) m1[i] = -1;
      else nmatches++;
    }
  }
  for (int i = 0; i < n1; i++) matches12[i] = m1[i];
  return nmatches;
}

// LSDmatcher::mutualOverlap (reference src/LSDmatcher.cpp:550-625) for four points in homogeneous image coordinates (x, y, w):
// the distance of the two inner points over the distance of the two outer ones.  Mat - Mat is a float subtraction per element,
// cv::norm a double sum of squares in element order and a double sqrt, assigned to a float.
static float plo_overlap_dist(const float a[3], const float b[3]) {
  double s = 0;
  for (int k = 0; k < 3; k++) { const float d = a[k] - b[k]; s += (double)d * d; }
  return (float)std::sqrt(s);
}
static float plo_mutual_overlap(const float pts[4][3]) {
  float max_dist = 0.0f;
  int outer1 = 0, outer2 = 3, inner1 = 1, inner2 = 2;
  for (int i = 0; i < 3; i++)
    for (int j = i + 1; j < 4; j++) {
      const float dist = plo_overlap_dist(pts[i], pts[j]);
      if (dist > max_dist) { max_dist = dist; outer1 = i; outer2 = j; }
    }
  if (max_dist < 1.0f) return 0.0f;
  if (outer1 == 0) {
    if (outer2 == 1) { inner1 = 2; inner2 = 3; }
    else if (outer2 == 2) { inner1 = 1; inner2 = 3; }
    else { inner1 = 1; inner2 = 2; }
  } else if (outer1 == 1) {
    inner1 = 0;
    inner2 = outer2 == 2 ? 3 : 2;
  } else {
    inner1 = 0; inner2 = 1;
  }
  double s = 0;
  for (int k = 0; k < 3; k++) { const float d = pts[inner1][k] - pts[inner2][k]; s += (double)d * d; }
  return (float)(std::sqrt(s) / max_dist);
}

// LSDmatcher::FrameBFMatchNew (reference src/LSDmatcher.cpp:488-548): knnMatch k = 2, then for the nearest neighbour of every
// query (the loop over j runs to size() - 1 = 1) the end points of line 1 are carried over the fundamental matrix (epi = F p, a
// 3 x 3 float product accumulated in element order), intersected with line 2's equation (l2.cross(epi), l2 = the float casts of
// mvKeyLineFunctions), normalised by w (`Mat /= w`: a double division per element, back to float) and compared with line 2's own
// end points by mutualOverlap; a match needs w of both > 1e-12 in magnitude, distance < TH, overlap > 0.8 and the ratio test.
// seg = (startPointX, startPointY, endPointX, endPointY) per line; func2 = 3 doubles per line of set 2; F row-major.
// Guard: the reference's `size() - 1` wraps when ldesc2 has no rows (unsigned); fewer than 2 train rows -> no matches.
void plo_line_bfmatch_new(const uint8_t* d1, int n1, const uint8_t* d2, int n2, const float* seg1, const float* seg2,
                          const double* func2, const float* F, float TH, float nnratio, int32_t* matches) {
  for (int i = 0; i < n1; i++) matches[i] = -1;
  if (n1 <= 0 || n2 < 2) return;
  std::vector<int32_t> idx((size_t)n1 * 2), dist((size_t)n1 * 2);
  plo_knn2(d1, n1, d2, n2, idx.data(), dist.data());
  for (int q = 0; q < n1; q++) {
    const int t = idx[q * 2];
    const float m0 = (float)dist[q * 2], m1 = (float)dist[q * 2 + 1];
    float pts[4][3];
    bool ok = true;
    const float l2[3] = {(float)func2[3 * t], (float)func2[3 * t + 1], (float)func2[3 * t + 2]};
    for (int e = 0; e < 2; e++) {
      const float p[3] = {seg1[4 * q + 2 * e], seg1[4 * q + 2 * e + 1], 1.0f};
      float epi[3];
      for (int r = 0; r < 3; r++) {
        float acc = 0;
        for (int k = 0; k < 3; k++) acc += F[3 * r + k] * p[k];
        epi[r] = acc;
      }
      float c[3] = {l2[1] * epi[2] - l2[2] * epi[1], l2[2] * epi[0] - l2[0] * epi[2], l2[0] * epi[1] - l2[1] * epi[0]};
      if (!(std::fabs(c[2]) > 1e-12)) ok = false;
      const double w = c[2];
      for (int k = 0; k < 3; k++) pts[e][k] = (float)(c[k] / w);
    }
    if (!ok) continue;   // (`continue` to j = 1, where the loop ends)
    pts[2][0] = seg2[4 * t]; pts[2][1] = seg2[4 * t + 1]; pts[2][2] = 1.0f;
    pts[3][0] = seg2[4 * t + 2]; pts[3][1] = seg2[4 * t + 3]; pts[3][2] = 1.0f;
    const float score = plo_mutual_overlap(pts);
    if (m0 < TH && score > 0.8 && m0 < nnratio * m1) matches[q] = t;
  }
}

// LSDmatcher::SearchForTriangulationNew (reference src/LSDmatcher.cpp:780-832; its only call site, LocalMapping.cc:960, is commented
// out): FrameBFMatchNew both ways at TH_LOW -- F21 = ComputeF12(pKF2, pKF1) carries set 1's end points into image 2, F12 the other
// way -- the mutual check if isDouble, and only pairs of lines neither of which has a MapLine.  Returns nmatches.
int plo_line_search_for_triangulation_new(const uint8_t* d1, int n1, const uint8_t* d2, int n2, const float* seg1, const float* seg2,
                                          const double* func1, const double* func2, const float* F21, const float* F12,
                                          const uint8_t* has_ml1, const uint8_t* has_ml2, float TH, float nnratio, int is_double,
                                          int32_t* matches12) {
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  if (n1 == 0 || n2 == 0) return 0;
  std::vector<int32_t> m1(std::max(n1, 1)), m2(std::max(n2, 1));
  plo_line_bfmatch_new(d1, n1, d2, n2, seg1, seg2, func2, F21, TH, nnratio, m1.data());
  plo_line_bfmatch_new(d2, n2, d1, n1, seg2, seg1, func1, F12, TH, nnratio, m2.data());
  int nmatches = 0;
  for (int i = 0; i < n1; i++) {
    const int j = m1[i];
    if (j < 0) continue;
    if (is_double && m2[j] != i) continue;
    if (has_ml1[i] || has_ml2[j]) continue;
    matches12[i] = j;
    nmatches++;
  }
  return nmatches;
}

// ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches).
// set 1 = KeyFrame (desc1, angle1 = mvKeysUn[].angle, node1 = FeatureVector node of each feature,
// valid1 = feature has a non-bad MapPoint); set 2 = Frame.  matches21[j] = KeyFrame index whose MapPoint is
// assigned to Frame feature j, or -1.  Returns nmatches.
int plo_orb_search_by_bow(const uint8_t* desc1, const float* angle1, const int32_t* node1, const uint8_t* valid1, int n1,
                          const uint8_t* desc2, const float* angle2, const int32_t* node2, int n2, int th_low,
                          float nnratio, int check_ori, int32_t* matches21) {
  for (int j = 0; j < n2; j++) matches21[j] = -1;
  // DBoW2::FeatureVector = std::map<NodeId, std::vector<unsigned>>, features appended in index order
  std::map<int, std::vector<unsigned>> fv1, fv2;
  for (int i = 0; i < n1; i++) if (node1[i] >= 0) fv1[node1[i]].push_back((unsigned)i);
  for (int j = 0; j < n2; j++) if (node2[j] >= 0) fv2[node2[j]].push_back((unsigned)j);
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  auto KFit = fv1.begin(), KFend = fv1.end();
  auto Fit = fv2.begin(), Fend = fv2.end();
  while (KFit != KFend && Fit != Fend) {
    if (KFit->first == Fit->first) {
      const std::vector<unsigned>& vKF = KFit->second;
      const std::vector<unsigned>& vF = Fit->second;
      for (size_t iKF = 0; iKF < vKF.size(); iKF++) {
        const unsigned realIdxKF = vKF[iKF];
        if (!valid1[realIdxKF]) continue;
        const uint8_t* dKF = desc1 + (size_t)realIdxKF * 32;
        int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
        for (size_t iF = 0; iF < vF.size(); iF++) {
          const unsigned realIdxF = vF[iF];
          if (matches21[realIdxF] >= 0) continue;
          const int dist = plo_descriptor_distance(dKF, desc2 + (size_t)realIdxF * 32);
          if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = (int)realIdxF; }
          else if (dist < bestDist2) { bestDist2 = dist; }
        }
        if (bestDist1 <= th_low) {
          if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
            matches21[bestIdxF] = (int32_t)realIdxKF;
            if (check_ori) {
              float rot = angle1[realIdxKF] - angle2[bestIdxF];
              if (rot < 0.0) rot += 360.0f;
              int bin = (int)roundf(rot * factor);
              if (bin == HISTO_LENGTH) bin = 0;
              rotHist[bin].push_back(bestIdxF);
            }
            nmatches++;
          }
        }
      }
      ++KFit; ++Fit;
    } else if (KFit->first < Fit->first) {
      KFit = fv1.lower_bound(Fit->first);
    } else {
      Fit = fv2.lower_bound(KFit->first);
    }
  }
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0; j < rotHist[i].size(); j++) { matches21[rotHist[i][j]] = -1; nmatches--; }
    }
  }
  (void)TH_LOW_ORB;
  return nmatches;
}

// DBoW2 TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup), reference
// Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1217-1255, over a flat tree (children contiguous).
// nid = -1 / word = -1 when the word's weight is not > 0 (transform(features,...) skips those, :1158-1163).
void plo_bow_transform(const uint8_t* desc, int n, const uint8_t* node_desc, const int32_t* child_start,
                       const int32_t* child_count, const int32_t* word_id, const float* weight, int L, int levelsup,
                       int32_t* nid_out, int32_t* word_out) {
  const int nid_level = L - levelsup;
  for (int i = 0; i < n; i++) {
    const uint8_t* f = desc + (size_t)i * 32;
    int nid = nid_level <= 0 ? 0 : -1;
    int final_id = 0, current_level = 0;
    do {
      ++current_level;
      const int cs = child_start[final_id], cc = child_count[final_id];
      final_id = cs;
      double best_d = plo_descriptor_distance(f, node_desc + (size_t)final_id * 32);
      for (int k = 1; k < cc; k++) {
        const int id = cs + k;
        double d = plo_descriptor_distance(f, node_desc + (size_t)id * 32);
        if (d < best_d) { best_d = d; final_id = id; }
      }
      if (current_level == nid_level) nid = final_id;
    } while (child_count[final_id] > 0);
    const bool keep = weight[final_id] > 0;
    nid_out[i] = keep ? nid : -1;
    word_out[i] = keep ? word_id[final_id] : -1;
  }
}

}  // extern "C"

// ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12), reference
// src/ORBmatcher.cc:574-709.  valid1 / valid2 = feature carries a non-bad MapPoint; matches12[idx1] = feature of the second
// KeyFrame whose MapPoint the reference stores, or -1.  Note the strict `bestDist1 < TH_LOW` (:646) of this overload.
extern "C" int plo_orb_search_by_bow_kfkf(const uint8_t* desc1, const float* angle1, const int32_t* node1, const uint8_t* valid1,
                                          int n1, const uint8_t* desc2, const float* angle2, const int32_t* node2,
                                          const uint8_t* valid2, int n2, int th_low, float nnratio, int check_ori,
                                          int32_t* matches12) {
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  std::map<int, std::vector<unsigned>> fv1, fv2;
  for (int i = 0; i < n1; i++) if (node1[i] >= 0) fv1[node1[i]].push_back((unsigned)i);
  for (int j = 0; j < n2; j++) if (node2[j] >= 0) fv2[node2[j]].push_back((unsigned)j);
  std::vector<bool> vbMatched2(std::max(n2, 1), false);
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  int nmatches = 0;
  auto f1it = fv1.begin(), f1end = fv1.end();
  auto f2it = fv2.begin(), f2end = fv2.end();
  while (f1it != f1end && f2it != f2end) {
    if (f1it->first == f2it->first) {
      for (size_t i1 = 0; i1 < f1it->second.size(); i1++) {
        const size_t idx1 = f1it->second[i1];
        if (!valid1[idx1]) continue;
        const uint8_t* d1 = desc1 + idx1 * 32;
        int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
        for (size_t i2 = 0; i2 < f2it->second.size(); i2++) {
          const size_t idx2 = f2it->second[i2];
          if (vbMatched2[idx2] || !valid2[idx2]) continue;
          const int dist = plo_descriptor_distance(d1, desc2 + idx2 * 32);
          if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = (int)idx2; }
          else if (dist < bestDist2) { bestDist2 = dist; }
        }
        if (bestDist1 < th_low) {
          if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
            matches12[idx1] = bestIdx2;
            vbMatched2[bestIdx2] = true;
            if (check_ori) {
              float rot = angle1[idx1] - angle2[bestIdx2];
              if (rot < 0.0) rot += 360.0f;
              int bin = (int)roundf(rot * factor);
              if (bin == HISTO_LENGTH) bin = 0;
              rotHist[bin].push_back((int)idx1);
            }
            nmatches++;
          }
        }
      }
      ++f1it; ++f2it;
    } else if (f1it->first < f2it->first) {
      f1it = fv1.lower_bound(f2it->first);
    } else {
      f2it = fv2.lower_bound(f1it->first);
    }
  }
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0; j < rotHist[i].size(); j++) { matches12[rotHist[i][j]] = -1; nmatches--; }
    }
  }
  return nmatches;
}

// ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo = false), monocular, reference
// src/ORBmatcher.cc:720-912 (+ CheckDistEpipolarLine :154-173).  has_mp1 / has_mp2 = the feature already carries a MapPoint
// (those are skipped); F12 row-major 3x3 float; (ex, ey) = epipole of KF1's centre in KF2 (:731-735, computed by the caller
// from the poses); scale_factors2 / level_sigma2_2 = pKF2->mvScaleFactors / mvLevelSigma2.  Note that this fork never sets
// vbMatched2 (the upstream `vbMatched2[bestIdx2]=true` is missing at :856-859), so a feature of KF2 can be paired with
// several features of KF1 -- kept.  matches12[idx1] = idx2 or -1 (vMatchedPairs is its list of (i, matches12[i]) pairs).
extern "C" int plo_orb_search_for_triangulation(const plo_keypoint* kps1, const uint8_t* desc1, const int32_t* node1,
                                                const uint8_t* has_mp1, int n1, const plo_keypoint* kps2, const uint8_t* desc2,
                                                const int32_t* node2, const uint8_t* has_mp2, int n2, const float F12[9], float ex,
                                                float ey, const float* scale_factors2, const float* level_sigma2_2, int th_low,
                                                int check_ori, int32_t* matches12) {
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  std::map<int, std::vector<unsigned>> fv1, fv2;
  for (int i = 0; i < n1; i++) if (node1[i] >= 0) fv1[node1[i]].push_back((unsigned)i);
  for (int j = 0; j < n2; j++) if (node2[j] >= 0) fv2[node2[j]].push_back((unsigned)j);
  int nmatches = 0;
  std::vector<bool> vbMatched2(std::max(n2, 1), false);
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  auto f1it = fv1.begin(), f1end = fv1.end();
  auto f2it = fv2.begin(), f2end = fv2.end();
  while (f1it != f1end && f2it != f2end) {
    if (f1it->first == f2it->first) {
      for (size_t i1 = 0; i1 < f1it->second.size(); i1++) {
        const size_t idx1 = f1it->second[i1];
        if (has_mp1[idx1]) continue;
        const plo_keypoint& kp1 = kps1[idx1];
        const uint8_t* d1 = desc1 + idx1 * 32;
        int bestDist = th_low, bestIdx2 = -1;
        for (size_t i2 = 0; i2 < f2it->second.size(); i2++) {
          const size_t idx2 = f2it->second[i2];
          if (vbMatched2[idx2] || has_mp2[idx2]) continue;
          const int dist = plo_descriptor_distance(d1, desc2 + idx2 * 32);
          if (dist > th_low || dist > bestDist) continue;
          const plo_keypoint& kp2 = kps2[idx2];
          {   // !bStereo1 && !bStereo2
            const float distex = ex - kp2.x, distey = ey - kp2.y;
            if (distex * distex + distey * distey < 100 * scale_factors2[kp2.octave]) continue;
          }
          // CheckDistEpipolarLine(kp1, kp2, F12, pKF2)
          const float a = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
          const float b = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
          const float c = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
          const float num = a * kp2.x + b * kp2.y + c;
          const float den = a * a + b * b;
          if (den == 0) continue;
          const float dsqr = num * num / den;
          if (dsqr < 3.84 * level_sigma2_2[kp2.octave]) { bestIdx2 = (int)idx2; bestDist = dist; }
        }
        if (bestIdx2 >= 0) {
          matches12[idx1] = bestIdx2;
          nmatches++;
          if (check_ori) {
            float rot = kp1.angle - kps2[bestIdx2].angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            rotHist[bin].push_back((int)idx1);
          }
        }
      }
      ++f1it; ++f2it;
    } else if (f1it->first < f2it->first) {
      f1it = fv1.lower_bound(f2it->first);
    } else {
      f2it = fv2.lower_bound(f1it->first);
    }
  }
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0; j < rotHist[i].size(); j++) { matches12[rotHist[i][j]] = -1; nmatches--; }
    }
  }
  return nmatches;
}
