// ORACLE (test infrastructure) -- CPU restatement of the windowed (grid) searches of the tracking front end:
//   Frame::PosInGrid / AssignFeaturesToGrid            reference src/Frame.cc:278-293, 893-905
//   Frame::AssignFeaturesToGridForLine + LineIterator   reference src/Frame.cc:295-320, src/lineIterator.cpp:34-77
//   Frame::GetFeaturesInArea                            reference src/Frame.cc:713-766
//   Frame::GetFeaturesInAreaForLine                     reference src/Frame.cc:768-842
//   ORBmatcher::SearchForInitialization                 reference src/ORBmatcher.cc:455-572
//   ORBmatcher::SearchByProjection(F, MapPoints, th)    reference src/ORBmatcher.cc:56-144
//   ORBmatcher::SearchByProjection(Cur, Last, th, mono) reference src/ORBmatcher.cc:1441-1585
//   LSDmatcher::SearchByProjection(Cur, Last, th)       reference src/LSDmatcher.cpp:72-176
//   LSDmatcher::SearchByProjection(F, MapLines, th)     reference src/LSDmatcher.cpp:221-338
// Frame / MapPoint / MapLine objects are replaced by flat arrays.  The boundary sits after the projection: a
// query carries what the reference reads from the map element (mTrackProjX/Y, mnTrackScaleLevel, mTrackViewCos,
// descriptor, "Observations() > 0") or what it computes from the pose (u, v); pose algebra on cv::Mat stays
// with the caller.  The grid is CSR: cell (ix, iy) -> index ix*48 + iy, items in insertion order.
// PARITY UNPINNED, see oracle/plo.h.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <vector>

#include "plo.h"

namespace {
const int GRID_COLS = 64, GRID_ROWS = 48;   // Frame.h:44-45
const int HISTO_LENGTH = 30;
const int ORB_TH_HIGH = 100, ORB_TH_LOW = 50, LSD_TH_HIGH = 80;

struct GridP { float min_x, min_y, max_x, max_y, inv_w, inv_h; };

void three_maxima(const std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3) {   // ORBmatcher.cc:1718-1759
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

// Frame::GetFeaturesInArea, Frame.cc:713-766
void features_in_area(const plo_keypoint* kps, const GridP& g, const int32_t* cs, const int32_t* ci, float x, float y, float r,
                      int minLevel, int maxLevel, std::vector<int>& out) {
  out.clear();
  const int nMinCellX = std::max(0, (int)floorf((x - g.min_x - r) * g.inv_w));
  if (nMinCellX >= GRID_COLS) return;
  const int nMaxCellX = std::min(GRID_COLS - 1, (int)ceilf((x - g.min_x + r) * g.inv_w));
  if (nMaxCellX < 0) return;
  const int nMinCellY = std::max(0, (int)floorf((y - g.min_y - r) * g.inv_h));
  if (nMinCellY >= GRID_ROWS) return;
  const int nMaxCellY = std::min(GRID_ROWS - 1, (int)ceilf((y - g.min_y + r) * g.inv_h));
  if (nMaxCellY < 0) return;
  const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
  for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
    for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
      const int c = ix * GRID_ROWS + iy;
      for (int j = cs[c]; j < cs[c + 1]; j++) {
        const plo_keypoint& kp = kps[ci[j]];
        if (bCheckLevels) {
          if (kp.octave < minLevel) continue;
          if (maxLevel >= 0 && kp.octave > maxLevel) continue;
        }
        const float distx = kp.x - x, disty = kp.y - y;
        if (fabsf(distx) < r && fabsf(disty) < r) out.push_back(ci[j]);
      }
    }
}

// Frame::GetFeaturesInAreaForLine, Frame.cc:768-842 (minLevel / maxLevel are ignored there).
// `abs(float)` resolves to std::abs(float) (libstdc++ exports the overloads globally); sqrt on float operands.
void features_in_area_for_line(const plo_keyline* kl, const double* fn, const GridP& g, const int32_t* cs, const int32_t* ci,
                               float x1, float y1, float x2, float y2, float r, float TH, std::vector<int>& out,
                               std::vector<uint8_t>& seen) {
  out.clear();
  const float x[3] = {x1, (float)((x1 + x2) / 2.0), x2};
  const float y[3] = {y1, (float)((y1 + y2) / 2.0), y2};
  float delta1x = x1 - x2, delta1y = y1 - y2;
  const float norm_delta1 = sqrtf(delta1x * delta1x + delta1y * delta1y);
  delta1x /= norm_delta1;
  delta1y /= norm_delta1;
  for (int i = 0; i < 3; i++) {
    const int nMinCellX = std::max(0, (int)floorf((x[i] - g.min_x - r) * g.inv_w));
    if (nMinCellX >= GRID_COLS) continue;
    const int nMaxCellX = std::min(GRID_COLS - 1, (int)ceilf((x[i] - g.min_x + r) * g.inv_w));
    if (nMaxCellX < 0) continue;
    const int nMinCellY = std::max(0, (int)floorf((y[i] - g.min_y - r) * g.inv_h));
    if (nMinCellY >= GRID_ROWS) continue;
    const int nMaxCellY = std::min(GRID_ROWS - 1, (int)ceilf((y[i] - g.min_y + r) * g.inv_h));
    if (nMaxCellY < 0) continue;
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
      for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
        const int c = ix * GRID_ROWS + iy;
        for (int j = cs[c]; j < cs[c + 1]; j++) {
          const int id = ci[j];
          if (seen[id]) continue;
          const plo_keyline& k = kl[id];
          float delta2x = k.startPointX - k.endPointX, delta2y = k.startPointY - k.endPointY;
          const float norm_delta2 = sqrtf(delta2x * delta2x + delta2y * delta2y);
          delta2x /= norm_delta2;
          delta2y /= norm_delta2;
          const float CosSita = fabsf(delta1x * delta2x + delta1y * delta2y);
          if (CosSita < TH) continue;
          const float dist = (float)(fn[id * 3 + 0] * x[i] + fn[id * 3 + 1] * y[i] + fn[id * 3 + 2]);
          if (fabsf(dist) < r) {
            out.push_back(id);
            seen[id] = 1;
          }
        }
      }
  }
  for (int id : out) seen[id] = 0;
}
}  // namespace

extern "C" {

// Frame::AssignFeaturesToGrid.  cell_start[64*48+1], cell_items[n].  Returns the number of keypoints placed.
int plo_frame_assign_grid(const plo_keypoint* kps_un, int n, const float gp[6], int32_t* cell_start, int32_t* cell_items) {
  GridP g;
  memcpy(&g, gp, sizeof(g));
  std::vector<std::vector<int>> cells(GRID_COLS * GRID_ROWS);
  for (int i = 0; i < n; i++) {
    const int posX = (int)roundf((kps_un[i].x - g.min_x) * g.inv_w);   // PosInGrid: round(), not floor()
    const int posY = (int)roundf((kps_un[i].y - g.min_y) * g.inv_h);
    if (posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS) continue;
    cells[posX * GRID_ROWS + posY].push_back(i);
  }
  int k = 0;
  for (int c = 0; c < GRID_COLS * GRID_ROWS; c++) {
    cell_start[c] = k;
    for (int id : cells[c]) cell_items[k++] = id;
  }
  cell_start[GRID_COLS * GRID_ROWS] = k;
  return k;
}

// Frame::AssignFeaturesToGridForLine with the custom Bresenham LineIterator.  Returns the number of items (a
// line occupies every cell it crosses); items beyond `cap` are dropped (the return value still counts them).
int plo_frame_assign_grid_lines(const plo_keyline* kl, int nl, const float gp[6], int32_t* cell_start, int32_t* cell_items,
                                int cap) {
  GridP g;
  memcpy(&g, gp, sizeof(g));
  std::vector<std::vector<int>> cells(GRID_COLS * GRID_ROWS);
  for (int i = 0; i < nl; i++) {
    double x1 = kl[i].startPointX * g.inv_w, y1 = kl[i].startPointY * g.inv_h;   // float products, widened
    double x2 = kl[i].endPointX * g.inv_w, y2 = kl[i].endPointY * g.inv_h;
    const bool steep = std::abs(y2 - y1) > std::abs(x2 - x1);
    if (steep) { std::swap(x1, y1); std::swap(x2, y2); }
    if (x1 > x2) { std::swap(x1, x2); std::swap(y1, y2); }
    const double dx = x2 - x1, dy = std::abs(y2 - y1);
    double error = dx / 2.0;
    const int ystep = (y1 < y2) ? 1 : -1;
    int x = static_cast<int>(x1), y = static_cast<int>(y1);
    const int maxX = static_cast<int>(x2);
    while (x <= maxX) {
      const int px = steep ? y : x, py = steep ? x : y;
      error -= dy;
      if (error < 0) { y += ystep; error += dx; }
      x++;
      if (px >= 0 && px < GRID_COLS && py >= 0 && py < GRID_ROWS) cells[px * GRID_ROWS + py].push_back(i);
    }
  }
  int k = 0;
  for (int c = 0; c < GRID_COLS * GRID_ROWS; c++) {
    cell_start[c] = std::min(k, cap);
    for (int id : cells[c]) {
      if (k < cap) cell_items[k] = id;
      k++;
    }
  }
  cell_start[GRID_COLS * GRID_ROWS] = std::min(k, cap);
  return k;
}

int plo_features_in_area(const plo_keypoint* kps_un, const float gp[6], const int32_t* cs, const int32_t* ci, float x, float y,
                         float r, int min_level, int max_level, int32_t* out, int cap) {
  GridP g;
  memcpy(&g, gp, sizeof(g));
  std::vector<int> v;
  features_in_area(kps_un, g, cs, ci, x, y, r, min_level, max_level, v);
  for (size_t i = 0; i < v.size() && (int)i < cap; i++) out[i] = v[i];
  return (int)v.size();
}

int plo_features_in_area_for_line(const plo_keyline* kl, const double* fn, int nl, const float gp[6], const int32_t* cs,
                                  const int32_t* ci, float x1, float y1, float x2, float y2, float r, float TH, int32_t* out,
                                  int cap) {
  GridP g;
  memcpy(&g, gp, sizeof(g));
  std::vector<int> v;
  std::vector<uint8_t> seen(std::max(nl, 1), 0);
  features_in_area_for_line(kl, fn, g, cs, ci, x1, y1, x2, y2, r, TH, v, seen);
  for (size_t i = 0; i < v.size() && (int)i < cap; i++) out[i] = v[i];
  return (int)v.size();
}

// ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize).
// kps1 = F1.mvKeysUn, kps2 = F2.mvKeysUn (+ F2's grid); prev_matched[n1][2] is updated in place.
int plo_orb_search_for_initialization(const plo_keypoint* kps1, const uint8_t* desc1, int n1, const plo_keypoint* kps2,
                                      const uint8_t* desc2, int n2, const float gp2[6], const int32_t* cs2, const int32_t* ci2,
                                      float* prev_matched, int window_size, float nnratio, int check_ori, int32_t* matches12) {
  GridP g;
  memcpy(&g, gp2, sizeof(g));
  int nmatches = 0;
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  std::vector<int> vMatchedDistance(std::max(n2, 1), INT_MAX), vnMatches21(std::max(n2, 1), -1), vIndices2;
  for (int i1 = 0; i1 < n1; i1++) {
    const int level1 = kps1[i1].octave;
    if (level1 > 0) continue;
    features_in_area(kps2, g, cs2, ci2, prev_matched[i1 * 2], prev_matched[i1 * 2 + 1], (float)window_size, level1, level1,
                     vIndices2);
    if (vIndices2.empty()) continue;
    const uint8_t* d1 = desc1 + (size_t)i1 * 32;
    int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
    for (int i2 : vIndices2) {
      const int dist = plo_descriptor_distance(d1, desc2 + (size_t)i2 * 32);
      if (vMatchedDistance[i2] <= dist) continue;
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
      else if (dist < bestDist2) { bestDist2 = dist; }
    }
    if (bestDist <= ORB_TH_LOW) {
      if (bestDist < (float)bestDist2 * nnratio) {
        if (vnMatches21[bestIdx2] >= 0) { matches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
        matches12[i1] = bestIdx2;
        vnMatches21[bestIdx2] = i1;
        vMatchedDistance[bestIdx2] = bestDist;
        nmatches++;
        if (check_ori) {
          float rot = kps1[i1].angle - kps2[bestIdx2].angle;
          if (rot < 0.0) rot += 360.0f;
          int bin = (int)roundf(rot * factor);
          if (bin == HISTO_LENGTH) bin = 0;
          rotHist[bin].push_back(i1);
        }
      }
    }
  }
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int idx1 : rotHist[i])
        if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
    }
  }
  for (int i1 = 0; i1 < n1; i1++)
    if (matches12[i1] >= 0) {
      prev_matched[i1 * 2] = kps2[matches12[i1]].x;
      prev_matched[i1 * 2 + 1] = kps2[matches12[i1]].y;
    }
  return nmatches;
}

// ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>&, th), monocular (mvuRight < 0).
// Query iMP: q_valid = mbTrackInView && !isBad(); q_xy = (mTrackProjX, mTrackProjY); q_level = mnTrackScaleLevel;
// q_viewcos = mTrackViewCos; q_hasobs = Observations() > 0.  occupied[idx] = F.mvpMapPoints[idx] &&
// Observations() > 0 (updated as matches are assigned); assigned[idx] = query whose MapPoint now sits at idx.
int plo_orb_search_by_projection_mp(const plo_keypoint* kps_un, const uint8_t* desc, int n, const float gp[6], const int32_t* cs,
                                    const int32_t* ci, const float* scale_factors, uint8_t* occupied, int nq,
                                    const uint8_t* q_valid, const float* q_xy, const int32_t* q_level, const float* q_viewcos,
                                    const uint8_t* q_desc, const uint8_t* q_hasobs, float th, float nnratio, int32_t* assigned) {
  GridP g;
  memcpy(&g, gp, sizeof(g));
  int nmatches = 0;
  for (int i = 0; i < n; i++) assigned[i] = -1;
  const bool bFactor = th != 1.0;
  std::vector<int> vIndices;
  for (int iMP = 0; iMP < nq; iMP++) {
    if (!q_valid[iMP]) continue;
    const int nPredictedLevel = q_level[iMP];
    float r = q_viewcos[iMP] > 0.998 ? 2.5 : 4.0;   // RadiusByViewingCos, ORBmatcher.cc:146-152
    if (bFactor) r *= th;
    features_in_area(kps_un, g, cs, ci, q_xy[iMP * 2], q_xy[iMP * 2 + 1], r * scale_factors[nPredictedLevel], nPredictedLevel - 1,
                     nPredictedLevel, vIndices);
    if (vIndices.empty()) continue;
    const uint8_t* MPdescriptor = q_desc + (size_t)iMP * 32;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int idx : vIndices) {
      if (occupied[idx]) continue;
      const int dist = plo_descriptor_distance(MPdescriptor, desc + (size_t)idx * 32);
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = kps_un[idx].octave; bestIdx = idx; }
      else if (dist < bestDist2) { bestLevel2 = kps_un[idx].octave; bestDist2 = dist; }
    }
    if (bestDist <= ORB_TH_HIGH) {
      if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
      assigned[bestIdx] = iMP;
      occupied[bestIdx] = q_hasobs[iMP];
      nmatches++;
    }
  }
  return nmatches;
}

// ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono).
// Query i = LastFrame feature i: q_valid = MapPoint present && !mvbOutlier[i] && invzc >= 0; q_uv = projection
// (u, v) into the current frame; q_octave = LastFrame.mvKeys[i].octave; q_angle = LastFrame.mvKeysUn[i].angle;
// q_desc = pMP->GetDescriptor(); mode 0 = monocular / lateral (octave-1 .. octave+1), 1 = forward
// (>= octave), 2 = backward (0 .. octave).
static int search_by_projection_frame(const plo_keypoint* kps_un, const uint8_t* desc, int n, const float gp[6],
                                      const int32_t* cs, const int32_t* ci, const float* scale_factors, uint8_t* occupied,
                                      int nq, const uint8_t* q_valid, const float* q_uv, const int32_t* q_octave,
                                      const float* q_angle, const uint8_t* q_desc, const uint8_t* q_hasobs, float th, int mode,
                                      int check_ori, int dist_th, int32_t* assigned) {
  GridP g;
  memcpy(&g, gp, sizeof(g));
  int nmatches = 0;
  for (int i = 0; i < n; i++) assigned[i] = -1;
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  std::vector<int> vIndices2;
  for (int i = 0; i < nq; i++) {
    if (!q_valid[i]) continue;
    const float u = q_uv[i * 2], v = q_uv[i * 2 + 1];
    if (u < g.min_x || u > g.max_x) continue;
    if (v < g.min_y || v > g.max_y) continue;
    const int nLastOctave = q_octave[i];
    const float radius = th * scale_factors[nLastOctave];
    if (mode == 1) features_in_area(kps_un, g, cs, ci, u, v, radius, nLastOctave, -1, vIndices2);
    else if (mode == 2) features_in_area(kps_un, g, cs, ci, u, v, radius, 0, nLastOctave, vIndices2);
    else features_in_area(kps_un, g, cs, ci, u, v, radius, nLastOctave - 1, nLastOctave + 1, vIndices2);
    if (vIndices2.empty()) continue;
    const uint8_t* dMP = q_desc + (size_t)i * 32;
    int bestDist = 256, bestIdx2 = -1;
    for (int i2 : vIndices2) {
      if (occupied[i2]) continue;
      const int dist = plo_descriptor_distance(dMP, desc + (size_t)i2 * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    if (bestDist <= dist_th) {
      assigned[bestIdx2] = i;
      occupied[bestIdx2] = q_hasobs[i];
      nmatches++;
      if (check_ori) {
        float rot = q_angle[i] - kps_un[bestIdx2].angle;
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)roundf(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        rotHist[bin].push_back(bestIdx2);
      }
    }
  }
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++)
      if (i != ind1 && i != ind2 && i != ind3)
        for (int idx : rotHist[i]) { assigned[idx] = -1; occupied[idx] = 0; nmatches--; }   // mvpMapPoints[idx] = NULL
  }
  return nmatches;
}

int plo_orb_search_by_projection_frame(const plo_keypoint* kps_un, const uint8_t* desc, int n, const float gp[6],
                                       const int32_t* cs, const int32_t* ci, const float* scale_factors, uint8_t* occupied,
                                       int nq, const uint8_t* q_valid, const float* q_uv, const int32_t* q_octave,
                                       const float* q_angle, const uint8_t* q_desc, const uint8_t* q_hasobs, float th, int mode,
                                       int check_ori, int32_t* assigned) {
  return search_by_projection_frame(kps_un, desc, n, gp, cs, ci, scale_factors, occupied, nq, q_valid, q_uv, q_octave, q_angle,
                                    q_desc, q_hasobs, th, mode, check_ori, ORB_TH_HIGH, assigned);
}

// ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, th, ORBdist),
// reference src/ORBmatcher.cc:1587-1716 (relocalisation): same scan with the caller's distance threshold; q_level =
// pMP->PredictScale(...), band level-1..level+1; `occupied` = CurrentFrame.mvpMapPoints[i2] != NULL (q_hasobs all 1).
int plo_orb_search_by_projection_kf(const plo_keypoint* kps_un, const uint8_t* desc, int n, const float gp[6], const int32_t* cs,
                                    const int32_t* ci, const float* scale_factors, uint8_t* occupied, int nq,
                                    const uint8_t* q_valid, const float* q_uv, const int32_t* q_level, const float* q_angle,
                                    const uint8_t* q_desc, const uint8_t* q_hasobs, float th, int orb_dist, int check_ori,
                                    int32_t* assigned) {
  return search_by_projection_frame(kps_un, desc, n, gp, cs, ci, scale_factors, occupied, nq, q_valid, q_uv, q_level, q_angle,
                                    q_desc, q_hasobs, th, 0, check_ori, orb_dist, assigned);
}

// LSDmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th).
// Query i = LastFrame line i: q_valid = MapLine present && !mvbLineOutlier[i] && CurrentFrame.isInFrustum(pML, 0.5);
// q_seg = (mTrackProjX1, Y1, X2, Y2); q_length = LastFrame.mvKeylinesUn[i].lineLength.
int plo_line_search_by_projection_frame(const plo_keyline* kl, const uint8_t* ldesc, const double* fn, int nl, const float gp[6],
                                        const int32_t* cs, const int32_t* ci, uint8_t* occupied, int nq, const uint8_t* q_valid,
                                        const float* q_seg, const float* q_length, const uint8_t* q_desc,
                                        const uint8_t* q_hasobs, float th, int32_t* assigned) {
  GridP g;
  memcpy(&g, gp, sizeof(g));
  int nmatches = 0;
  for (int i = 0; i < nl; i++) assigned[i] = -1;
  std::vector<int> vIndices2;
  std::vector<uint8_t> seen(std::max(nl, 1), 0);
  for (int i = 0; i < nq; i++) {
    if (!q_valid[i]) continue;
    const float radius = th;
    features_in_area_for_line(kl, fn, g, cs, ci, q_seg[i * 4], q_seg[i * 4 + 1], q_seg[i * 4 + 2], q_seg[i * 4 + 3], radius, 0.96f,
                              vIndices2, seen);
    if (vIndices2.empty()) continue;
    const uint8_t* dML = q_desc + (size_t)i * 32;
    int bestDist = 256, bestIdx2 = -1;
    for (int i2 : vIndices2) {
      if (occupied[i2]) continue;
      const int dist = plo_descriptor_distance(dML, ldesc + (size_t)i2 * 32);
      const float max_ = std::max(q_length[i], kl[i2].lineLength), min_ = std::min(q_length[i], kl[i2].lineLength);
      if (min_ / max_ < 0.75) continue;
      if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
    }
    if (bestDist <= LSD_TH_HIGH) {
      assigned[bestIdx2] = i;
      occupied[bestIdx2] = q_hasobs[i];
      nmatches++;
    }
  }
  return nmatches;
}

// LSDmatcher::SearchByProjection(Frame& F, const vector<MapLine*>&, th).
int plo_line_search_by_projection_ml(const plo_keyline* kl, const uint8_t* ldesc, const double* fn, int nl, const float gp[6],
                                     const int32_t* cs, const int32_t* ci, uint8_t* occupied, int nq, const uint8_t* q_valid,
                                     const float* q_seg, const float* q_viewcos, const uint8_t* q_desc, const uint8_t* q_hasobs,
                                     float th, float nnratio, int32_t* assigned) {
  GridP g;
  memcpy(&g, gp, sizeof(g));
  int nmatches = 0;
  for (int i = 0; i < nl; i++) assigned[i] = -1;
  const bool bFactor = th != 1.0;
  std::vector<int> vIndices;
  std::vector<uint8_t> seen(std::max(nl, 1), 0);
  for (int iML = 0; iML < nq; iML++) {
    if (!q_valid[iML]) continue;
    float r = q_viewcos[iML] > 0.998 ? 5.0 : 8.0;   // LSDmatcher::RadiusByViewingCos, LSDmatcher.cpp:1004-1010
    if (bFactor) r *= th;
    features_in_area_for_line(kl, fn, g, cs, ci, q_seg[iML * 4], q_seg[iML * 4 + 1], q_seg[iML * 4 + 2], q_seg[iML * 4 + 3], r,
                              0.998f, vIndices, seen);
    if (vIndices.empty()) continue;
    const uint8_t* MLdescriptor = q_desc + (size_t)iML * 32;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int idx : vIndices) {
      if (occupied[idx]) continue;
      const int dist = plo_descriptor_distance(MLdescriptor, ldesc + (size_t)idx * 32);
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = kl[idx].octave; bestIdx = idx; }
      else if (dist < bestDist2) { bestLevel2 = kl[idx].octave; bestDist2 = dist; }
    }
    if (bestDist <= LSD_TH_HIGH) {
      if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
      assigned[bestIdx] = iML;
      occupied[bestIdx] = q_hasobs[iML];
      nmatches++;
    }
  }
  (void)ORB_TH_LOW;
  return nmatches;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Frame::UndistortKeyPoints (reference src/Frame.cc:915-945): cv::undistortPoints(mat, mat, mK, mDistCoef, Mat(), mK)
// on the keypoint coordinates, everything else of the KeyPoint copied; zero k1 -> plain copy (:917-921).
// cv::undistortPoints is OpenCV (not in tree): restated from the 3.2 cvUndistortPoints algorithm -- normalise with the
// reciprocal focal lengths, 5 fixed-point iterations of the inverse Brown model in double, re-project with P = K.
// ---------------------------------------------------------------------------------------------------------------
extern "C" void plo_undistort_keypoints(const plo_keypoint* kps, int n, const float K[4], const float D[5], plo_keypoint* out) {
  for (int i = 0; i < n; i++) out[i] = kps[i];
  if (D[0] == 0.0) return;
  const double fx = K[0], fy = K[1], cx = K[2], cy = K[3], ifx = 1. / fx, ify = 1. / fy;
  const double k1 = D[0], k2 = D[1], p1 = D[2], p2 = D[3], k3 = D[4];
  for (int i = 0; i < n; i++) {
    double x = kps[i].x, y = kps[i].y;
    const double x0 = x = (x - cx) * ifx, y0 = y = (y - cy) * ify;
    for (int j = 0; j < 5; j++) {
      const double r2 = x * x + y * y;
      const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
      const double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
      const double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
    const double xx = fx * x + 0 * y + cx, yy = 0 * x + fy * y + cy, ww = 1. / (0 * x + 0 * y + 1);
    out[i].x = (float)(xx * ww);
    out[i].y = (float)(yy * ww);
  }
}

// MapPoint::ComputeDistinctiveDescriptors (reference src/MapPoint.cc:249-314) / MapLine twin (src/MapLine.cpp:256-330):
// of N observed descriptors pick the one with the least median Hamming distance to the others
// (median = sorted row[(size_t)(0.5*(N-1))], first minimum wins).  Returns the index, -1 for N == 0.
extern "C" int plo_distinctive_descriptor(const uint8_t* desc, int n) {
  if (n <= 0) return -1;
  int BestMedian = INT_MAX, BestIdx = 0;
  std::vector<int> vDists(n);
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) vDists[j] = i == j ? 0 : plo_descriptor_distance(desc + (size_t)i * 32, desc + (size_t)j * 32);
    std::sort(vDists.begin(), vDists.end());
    const int median = vDists[(size_t)(0.5 * (n - 1))];
    if (median < BestMedian) { BestMedian = median; BestIdx = i; }
  }
  return BestIdx;
}

// The search inside ORBmatcher::Fuse (reference src/ORBmatcher.cc:914-1061; the Sim3 overload :1063-1197 has the same
// core): KeyFrame::GetFeaturesInArea(u, v, th*scale[l]) (src/KeyFrame.cc:606-645: no level filter), then kpLevel in
// [l-1, l], the monocular chi-square gate e2 * mvInvLevelSigma2[kpLevel] > 5.99, best Hamming <= TH_LOW.
// best_idx[q] = keypoint or -1; returns the number of queries with a result.
extern "C" int plo_orb_fuse_search(const plo_keypoint* kps_un, const uint8_t* desc, int n, const float gp[6], const int32_t* cs,
                                   const int32_t* ci, const float* scale_factors, const float* inv_level_sigma2, int nq,
                                   const uint8_t* q_valid, const float* q_uv, const int32_t* q_level, const uint8_t* q_desc, float th,
                                   int th_low, int32_t* best_idx) {
  GridP g;
  memcpy(&g, gp, sizeof(g));
  int nfound = 0;
  std::vector<int> vIndices;
  (void)n;
  for (int i = 0; i < nq; i++) {
    best_idx[i] = -1;
    if (!q_valid[i]) continue;
    const float u = q_uv[i * 2], v = q_uv[i * 2 + 1];
    const int nPredictedLevel = q_level[i];
    const float radius = th * scale_factors[nPredictedLevel];
    features_in_area(kps_un, g, cs, ci, u, v, radius, -1, -1, vIndices);
    if (vIndices.empty()) continue;
    const uint8_t* dMP = q_desc + (size_t)i * 32;
    int bestDist = 256, bestIdx = -1;
    for (int idx : vIndices) {
      const plo_keypoint& kp = kps_un[idx];
      const int kpLevel = kp.octave;
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      const float ex = u - kp.x, ey = v - kp.y;
      const float e2 = ex * ex + ey * ey;
      if (e2 * inv_level_sigma2[kpLevel] > 5.99) continue;
      const int dist = plo_descriptor_distance(dMP, desc + (size_t)idx * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    if (bestDist <= th_low) { best_idx[i] = bestIdx; nfound++; }
  }
  return nfound;
}

// ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, vpPoints, vpMatched, th), reference src/ORBmatcher.cc:329-453
// (loop closing).  occupied[idx] = vpMatched[idx] != NULL (in/out); assigned[idx] = query stored there, or -1.
extern "C" int plo_orb_search_by_projection_sim3(const plo_keypoint* kps_un, const uint8_t* desc, int n, const float gp[6],
                                                 const int32_t* cs, const int32_t* ci, const float* scale_factors,
                                                 uint8_t* occupied, int nq, const uint8_t* q_valid, const float* q_uv,
                                                 const int32_t* q_level, const uint8_t* q_desc, float th, int th_low,
                                                 int32_t* assigned) {
  GridP g;
  memcpy(&g, gp, sizeof(g));
  int nmatches = 0;
  for (int i = 0; i < n; i++) assigned[i] = -1;
  std::vector<int> vIndices;
  for (int iMP = 0; iMP < nq; iMP++) {
    if (!q_valid[iMP]) continue;
    const int nPredictedLevel = q_level[iMP];
    const float radius = th * scale_factors[nPredictedLevel];
    features_in_area(kps_un, g, cs, ci, q_uv[iMP * 2], q_uv[iMP * 2 + 1], radius, -1, -1, vIndices);
    if (vIndices.empty()) continue;
    const uint8_t* dMP = q_desc + (size_t)iMP * 32;
    int bestDist = 256, bestIdx = -1;
    for (int idx : vIndices) {
      if (occupied[idx]) continue;
      const int kpLevel = kps_un[idx].octave;
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      const int dist = plo_descriptor_distance(dMP, desc + (size_t)idx * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    if (bestDist <= th_low) { assigned[bestIdx] = iMP; occupied[bestIdx] = 1; nmatches++; }
  }
  return nmatches;
}

// ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th), reference src/ORBmatcher.cc:1199-1439 (loop closing,
// LoopClosing.cc:330).  The caller hands over the projections: q12_* = KeyFrame 1's map points transformed into KeyFrame 2
// (valid = pMP && !vbAlreadyMatched1 && !isBad() && depth >= 0 && IsInImage && distance inside the invariance region;
// uv, level = PredictScale, desc = pMP->GetDescriptor()), q21_* the other way round; query i belongs to keypoint slot i.
// Each direction: KeyFrame::GetFeaturesInArea(u, v, th*scale[l]), octave in [l-1, l], best Hamming (first minimum),
// accepted when <= TH_HIGH (:1283-1313, :1363-1393); then the agreement check (:1396-1412).
// match1[n1] / match2[n2] = vnMatch1 / vnMatch2; match12[i1] = agreed index in KeyFrame 2 or -1; returns nFound.
static void sim3_one_way(const plo_keypoint* kps, const uint8_t* desc, const GridP& g, const int32_t* cs, const int32_t* ci,
                         const float* scale_factors, int nq, const uint8_t* q_valid, const float* q_uv, const int32_t* q_level,
                         const uint8_t* q_desc, float th, int th_high, int32_t* vnMatch) {
  std::vector<int> vIndices;
  for (int i = 0; i < nq; i++) {
    vnMatch[i] = -1;
    if (!q_valid[i]) continue;
    const int nPredictedLevel = q_level[i];
    const float radius = th * scale_factors[nPredictedLevel];
    features_in_area(kps, g, cs, ci, q_uv[i * 2], q_uv[i * 2 + 1], radius, -1, -1, vIndices);
    if (vIndices.empty()) continue;
    const uint8_t* dMP = q_desc + (size_t)i * 32;
    int bestDist = INT_MAX, bestIdx = -1;
    for (int idx : vIndices) {
      const plo_keypoint& kp = kps[idx];
      if (kp.octave < nPredictedLevel - 1 || kp.octave > nPredictedLevel) continue;
      const int dist = plo_descriptor_distance(dMP, desc + (size_t)idx * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
    }
    if (bestDist <= th_high) vnMatch[i] = bestIdx;
  }
}

extern "C" int plo_orb_search_by_sim3(const plo_keypoint* kps1, const uint8_t* desc1, int n1, const int32_t* cs1, const int32_t* ci1,
                                      const plo_keypoint* kps2, const uint8_t* desc2, int n2, const int32_t* cs2, const int32_t* ci2,
                                      const float gp[6], const float* scale_factors, const uint8_t* q12_valid, const float* q12_uv,
                                      const int32_t* q12_level, const uint8_t* q12_desc, const uint8_t* q21_valid,
                                      const float* q21_uv, const int32_t* q21_level, const uint8_t* q21_desc, float th, int th_high,
                                      int32_t* match1, int32_t* match2, int32_t* match12) {
  GridP g;
  memcpy(&g, gp, sizeof(g));
  sim3_one_way(kps2, desc2, g, cs2, ci2, scale_factors, n1, q12_valid, q12_uv, q12_level, q12_desc, th, th_high, match1);
  sim3_one_way(kps1, desc1, g, cs1, ci1, scale_factors, n2, q21_valid, q21_uv, q21_level, q21_desc, th, th_high, match2);
  int nFound = 0;
  for (int i1 = 0; i1 < n1; i1++) {
    match12[i1] = -1;
    const int idx2 = match1[i1];
    if (idx2 >= 0) {
      const int idx1 = match2[idx2];
      if (idx1 == i1) { match12[i1] = idx2; nFound++; }
    }
  }
  return nFound;
}

// The search inside LSDmatcher::Fuse(pKF, vpMapLines, th) (reference src/LSDmatcher.cpp:860-1002) with
// KeyFrame::GetLinesInArea (src/KeyFrame.cc:647-683: brute force over the KeyFrame's lines, midpoint distance <= r^2 and
// |cos| of the direction angle >= TH = 0.998).  Reference quirk kept: the candidate rows are read from
// pKF->mDescriptors (the ORB matrix, :963) with the LINE index -- `cand_desc` is whatever matrix the caller passes there.
// best_idx[q] = line index or -1; returns the number of queries with a result.
// KeyFrame::GetLinesInArea (reference src/KeyFrame.cc:647-683): brute force over the KeyFrame's lines -- squared distance
// between the query's midpoint and the line's midpoint (kl.pt) <= r^2, |cos| of the angle between the directions >= TH.
static void keyframe_lines_in_area(const plo_keyline* kl, int nl, float x1, float y1, float x2, float y2, float r, float TH,
                                   std::vector<int>& out) {
  out.clear();
  float delta1x = x1 - x2, delta1y = y1 - y2;
  const float norm_delta1 = sqrtf(delta1x * delta1x + delta1y * delta1y);
  delta1x /= norm_delta1;
  delta1y /= norm_delta1;
  for (int j = 0; j < nl; j++) {
    const plo_keyline& k = kl[j];
    const float distance = (float)((0.5 * (x1 + x2) - k.pt_x) * (0.5 * (x1 + x2) - k.pt_x) +
                                   (0.5 * (y1 + y2) - k.pt_y) * (0.5 * (y1 + y2) - k.pt_y));
    if (distance > r * r) continue;
    float delta2x = k.startPointX - k.endPointX, delta2y = k.startPointY - k.endPointY;
    const float norm_delta2 = sqrtf(delta2x * delta2x + delta2y * delta2y);
    delta2x /= norm_delta2;
    delta2y /= norm_delta2;
    const float CosSita = fabsf(delta1x * delta2x + delta1y * delta2y);
    if (CosSita < TH) continue;
    out.push_back(j);
  }
}
extern "C" int plo_keyframe_lines_in_area(const plo_keyline* kl, int nl, float x1, float y1, float x2, float y2, float r, float TH,
                                          int32_t* out, int cap) {
  std::vector<int> v;
  keyframe_lines_in_area(kl, nl, x1, y1, x2, y2, r, TH, v);
  for (size_t i = 0; i < v.size() && (int)i < cap; i++) out[i] = v[i];
  return (int)v.size();
}

extern "C" int plo_line_fuse_search(const plo_keyline* kl, const uint8_t* cand_desc, int nl, const float* scale_factors_line, int nq,
                                    const uint8_t* q_valid, const float* q_seg, const int32_t* q_level, const uint8_t* q_desc,
                                    float th, float TH, int th_low, int32_t* best_idx) {
  int nfound = 0;
  std::vector<int> vIndices;
  for (int i = 0; i < nq; i++) {
    best_idx[i] = -1;
    if (!q_valid[i]) continue;
    const float x1 = q_seg[i * 4], y1 = q_seg[i * 4 + 1], x2 = q_seg[i * 4 + 2], y2 = q_seg[i * 4 + 3];
    const int nPredictedLevel = q_level[i];
    const float r = th * scale_factors_line[nPredictedLevel];
    keyframe_lines_in_area(kl, nl, x1, y1, x2, y2, r, TH, vIndices);
    int bestDist = 256, bestIdx = -1;
    for (int j : vIndices) {
      const plo_keyline& k = kl[j];
      if (k.octave < nPredictedLevel - 1 || k.octave > nPredictedLevel) continue;
      const int dist = plo_descriptor_distance(q_desc + (size_t)i * 32, cand_desc + (size_t)j * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx = j; }
    }
    if (bestDist <= th_low) { best_idx[i] = bestIdx; nfound++; }
  }
  return nfound;
}

// ------------------------------------------------------------------------------------------------------------------
// Frame::isInFrustum for map points and map lines (reference src/Frame.cc:560-623, 625-711) + PredictScale.
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct View { float R[9], t[3], Ow[3], fx, fy, cx, cy, minX, minY, maxX, maxY, logScale; };
// `mRcw*P + mtcw`: cv::gemm(A, B, 1, C, 1) on CV_32F accumulates in double and rounds once (pinned definition)
inline void to_camera(const View& v, const float* P, float* Pc) {
  for (int i = 0; i < 3; i++) {
    double s = (double)v.R[i * 3] * (double)P[0];
    s += (double)v.R[i * 3 + 1] * (double)P[1];
    s += (double)v.R[i * 3 + 2] * (double)P[2];
    Pc[i] = (float)(s + (double)v.t[i]);
  }
}
inline float norm3(const float* a) {   // cv::norm(NORM_L2) of a CV_32F vector: double accumulation
  double s = (double)a[0] * (double)a[0];
  s += (double)a[1] * (double)a[1];
  s += (double)a[2] * (double)a[2];
  return (float)std::sqrt(s);
}
inline double dot3(const float* a, const float* b) {   // Mat::dot: double accumulation, returns double
  double s = (double)a[0] * (double)b[0];
  s += (double)a[1] * (double)b[1];
  s += (double)a[2] * (double)b[2];
  return s;
}
}  // namespace

extern "C" void plo_frame_is_in_frustum_points(const float view[24], int nlevels, int n, const float* pos, const float* normal,
                                               const float* min_dist, const float* max_dist, float viewing_cos_limit,
                                               uint8_t* valid, float* uv, int32_t* level, float* viewcos) {
  View v;
  memcpy(&v, view, sizeof(v));
  for (int i = 0; i < n; i++) {
    valid[i] = 0; uv[2 * i] = uv[2 * i + 1] = 0.f; level[i] = 0; viewcos[i] = 0.f;
    const float* P = pos + 3 * i;
    float Pc[3];
    to_camera(v, P, Pc);
    if (Pc[2] < 0.0f) continue;
    const float invz = 1.0f / Pc[2];
    const float u = v.fx * Pc[0] * invz + v.cx;
    const float w = v.fy * Pc[1] * invz + v.cy;
    if (u < v.minX || u > v.maxX) continue;
    if (w < v.minY || w > v.maxY) continue;
    const float maxDistance = 1.2f * max_dist[i], minDistance = 0.8f * min_dist[i];   // Get{Max,Min}DistanceInvariance
    const float PO[3] = {P[0] - v.Ow[0], P[1] - v.Ow[1], P[2] - v.Ow[2]};
    const float dist = norm3(PO);
    if (dist < minDistance || dist > maxDistance) continue;
    const float viewCos = (float)(dot3(PO, normal + 3 * i) / dist);
    if (viewCos < viewing_cos_limit) continue;
    const float ratio = max_dist[i] / dist;   // MapPoint::PredictScale(dist, Frame*)
    int nScale = (int)ceilf(logf(ratio) / v.logScale);
    if (nScale < 0) nScale = 0;
    else if (nScale >= nlevels) nScale = nlevels - 1;
    valid[i] = 1; uv[2 * i] = u; uv[2 * i + 1] = w; level[i] = nScale; viewcos[i] = viewCos;
  }
}

extern "C" void plo_frame_is_in_frustum_lines(const float view[24], int n, const float* pos6, const float* normal,
                                              const float* min_dist, const float* max_dist, float viewing_cos_limit,
                                              uint8_t* valid, float* seg, int32_t* level, float* viewcos) {
  View v;
  memcpy(&v, view, sizeof(v));
  for (int i = 0; i < n; i++) {
    valid[i] = 0; seg[4 * i] = seg[4 * i + 1] = seg[4 * i + 2] = seg[4 * i + 3] = 0.f; level[i] = 0; viewcos[i] = 0.f;
    const float *SP = pos6 + 6 * i, *EP = SP + 3;
    float SPc[3], EPc[3];
    to_camera(v, SP, SPc);
    to_camera(v, EP, EPc);
    if (SPc[2] < 0.0f || EPc[2] < 0.0f) continue;
    const float invz1 = 1.0f / SPc[2];
    const float u1 = v.fx * SPc[0] * invz1 + v.cx;
    const float v1 = v.fy * SPc[1] * invz1 + v.cy;
    if (u1 < v.minX || u1 > v.maxX) continue;
    if (v1 < v.minY || v1 > v.maxY) continue;
    const float invz2 = 1.0f / EPc[2];
    const float u2 = v.fx * EPc[0] * invz2 + v.cx;
    const float v2 = v.fy * EPc[1] * invz2 + v.cy;
    if (u2 < v.minX || u2 > v.maxX) continue;
    if (v2 < v.minY || v2 > v.maxY) continue;
    const float maxDistance = 1.2f * max_dist[i], minDistance = 0.8f * min_dist[i];
    float OM[3];
    for (int k = 0; k < 3; k++) OM[k] = 0.5f * (SP[k] + EP[k]) - v.Ow[k];   // 0.5*(SP+EP) - mOw
    const float dist = norm3(OM);
    if (dist < minDistance || dist > maxDistance) continue;
    const float viewCos = (float)(dot3(OM, normal + 3 * i) / dist);
    if (viewCos < viewing_cos_limit) continue;
    const float ratio = max_dist[i] / dist;   // MapLine::PredictScale(dist, mfLogScaleFactor): no clamping
    const int nScale = (int)ceilf(logf(ratio) / v.logScale);
    valid[i] = 1; seg[4 * i] = u1; seg[4 * i + 1] = v1; seg[4 * i + 2] = u2; seg[4 * i + 3] = v2; level[i] = nScale; viewcos[i] = viewCos;
  }
}

extern "C" void plo_frame_project_points(const float view[24], int form, int n, const float* pos, uint8_t* front, float* uv) {
  View v;
  memcpy(&v, view, sizeof(v));
  for (int i = 0; i < n; i++) {
    float Pc[3];
    to_camera(v, pos + 3 * i, Pc);
    if (form == 0) {
      const float xc = Pc[0], yc = Pc[1];
      const float invzc = 1.0 / Pc[2];
      front[i] = invzc < 0 ? 0 : 1;
      uv[2 * i] = v.fx * xc * invzc + v.cx;
      uv[2 * i + 1] = v.fy * yc * invzc + v.cy;
    } else {
      front[i] = Pc[2] < 0.0f ? 0 : 1;
      const float invz = form == 2 ? (float)(1.0 / Pc[2]) : 1 / Pc[2];
      const float x = Pc[0] * invz;
      const float y = Pc[1] * invz;
      uv[2 * i] = v.fx * x + v.cx;
      uv[2 * i + 1] = v.fy * y + v.cy;
    }
  }
}

// The gates in front of the back end's pose-driven searches, one restatement with the differences between the five loops as flags
// (reference src/ORBmatcher.cc: relocalisation SearchByProjection :1591-1640 = flags 2; loop-closing SearchByProjection :337-395 =
// 1|4|8|32; Fuse(pKF, vpMapPoints) :945-975 = 1|4|8|32; Fuse(pKF, Scw, ..) :1096-1128 = 1|2|4|8|32; SearchBySim3 :1206-1290 /
// :1313-1365 = 1|2|4|8|16|64).  Flag values = PLH_GATE_* of include/plslam_hip.h.  valid: in = the caller's map-side gates.
extern "C" void plo_map_point_gates(const float view[24], int nlevels, const float R2t2[12], int flags, int n, const float* pos,
                                    const float* normal, const float* min_dist_inv, const float* max_dist_inv, const float* max_dist,
                                    uint8_t* valid, float* uv, float* dist_out, int32_t* level) {
  View v;
  memcpy(&v, view, sizeof(v));
  for (int i = 0; i < n; i++) {
    const bool pre = valid[i] != 0;
    valid[i] = 0; uv[2 * i] = uv[2 * i + 1] = 0.f; level[i] = 0; dist_out[i] = 0.f;
    if (!pre) continue;
    const float* P = pos + 3 * i;
    float Pc[3];
    to_camera(v, P, Pc);                       // cv::Mat p3Dc = Rcw*p3Dw + tcw
    if (flags & 64) {                          // cv::Mat p3Dc2 = sR21*p3Dc1 + t21
      View w2;
      memcpy(w2.R, R2t2, 9 * sizeof(float));
      memcpy(w2.t, R2t2 + 9, 3 * sizeof(float));
      float Q[3];
      to_camera(w2, Pc, Q);
      Pc[0] = Q[0]; Pc[1] = Q[1]; Pc[2] = Q[2];
    }
    if ((flags & 1) && Pc[2] < 0.0f) continue;
    const float invz = (flags & 2) ? (float)(1.0 / Pc[2]) : 1 / Pc[2];
    float u, w;
    if (flags & 4) {
      const float x = Pc[0] * invz;
      const float y = Pc[1] * invz;
      u = v.fx * x + v.cx;
      w = v.fy * y + v.cy;
    } else {
      u = v.fx * Pc[0] * invz + v.cx;
      w = v.fy * Pc[1] * invz + v.cy;
    }
    if (flags & 8) {
      if (!(u >= v.minX && u < v.maxX && w >= v.minY && w < v.maxY)) continue;   // KeyFrame::IsInImage
    } else {
      if (u < v.minX || u > v.maxX) continue;
      if (w < v.minY || w > v.maxY) continue;
    }
    const float PO[3] = {P[0] - v.Ow[0], P[1] - v.Ow[1], P[2] - v.Ow[2]};
    const float dist = (flags & 16) ? norm3(Pc) : norm3(PO);
    if (dist < min_dist_inv[i] || dist > max_dist_inv[i]) continue;   // Get{Min,Max}DistanceInvariance()
    if ((flags & 32) && dot3(PO, normal + 3 * i) < 0.5 * dist) continue;
    int nScale = 0;
    if (max_dist) {
      const float ratio = max_dist[i] / dist;    // MapPoint::PredictScale(dist, pKF | pF) (MapPoint.cc:394-424: both clamp)
      nScale = (int)ceilf(logf(ratio) / v.logScale);
      if (nScale < 0) nScale = 0;
      else if (nScale >= nlevels) nScale = nlevels - 1;
    }
    valid[i] = 1; uv[2 * i] = u; uv[2 * i + 1] = w; level[i] = nScale; dist_out[i] = dist;
  }
}
