// Windowed (grid) searches of the tracking front end, batch-first (gfx950 / CDNA4, wave64).
//
//   k_grid_points / k_grid_lines   Frame::AssignFeaturesToGrid / AssignFeaturesToGridForLine (+ LineIterator)
//                                  reference src/Frame.cc:278-320, 893-905; src/lineIterator.cpp:34-77
//   k_search_init                  ORBmatcher::SearchForInitialization            src/ORBmatcher.cc:455-572
//   k_search_proj_points           ORBmatcher::SearchByProjection(F, MapPoints)    src/ORBmatcher.cc:56-144
//                                  ORBmatcher::SearchByProjection(Cur, Last)       src/ORBmatcher.cc:1441-1585
//   k_search_proj_lines            LSDmatcher::SearchByProjection(Cur, Last)       src/LSDmatcher.cpp:72-176
//                                  LSDmatcher::SearchByProjection(F, MapLines)     src/LSDmatcher.cpp:221-338
//   (candidate generation = Frame::GetFeaturesInArea / GetFeaturesInAreaForLine, src/Frame.cc:713-842)
//
// Every search is greedy and order dependent (a keypoint taken by an earlier query is skipped by later ones), so
// one wavefront walks the queries of a frame pair in the reference's order; inside a query the 64 lanes gather
// the candidates of the grid window (one grid column = one contiguous CSR range), compute the 256-bit Hamming
// distances in parallel, and the best / second-best replay runs over the candidate list in the reference's
// enumeration order.  Per-frame match state lives in LDS.  Throughput comes from the batch (one wave per pair).
#include <algorithm>
#include <vector>

#include "plh_common.h"
#include "plh_stage.h"
#include "frame_resident.h"

namespace plh {

#define FS_WAVE_SYNC() PLH_WAVE_SYNC()

constexpr int GCOLS = PLH_GRID_COLS, GROWS = PLH_GRID_ROWS, GCELLS = PLH_GRID_CELLS;

struct ScaleTab {
  float v[16];
};

__device__ __forceinline__ int hamming_rows(const uint8_t* a, const uint8_t* b) {
  const unsigned long long* x = reinterpret_cast<const unsigned long long*>(a);
  const unsigned long long* y = reinterpret_cast<const unsigned long long*>(b);
  return __popcll(x[0] ^ y[0]) + __popcll(x[1] ^ y[1]) + __popcll(x[2] ^ y[2]) + __popcll(x[3] ^ y[3]);
}

// ------------------------------------------------------------------------------------------------------------
// Grid construction: count -> exclusive scan -> scatter (atomics) -> per-cell ascending sort (= insertion order,
// because the reference appends feature indices in increasing order).  One 256-thread block per frame.
// ------------------------------------------------------------------------------------------------------------
__device__ void grid_scan_and_publish(int* cnt, int* part, int32_t* cellStart) {
  const int tid = threadIdx.x;
  constexpr int PER = GCELLS / 256;   // 12
  int s = 0;
  for (int k = 0; k < PER; k++) s += cnt[tid * PER + k];
  part[tid] = s;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const int v = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = part[tid] - s;
  for (int k = 0; k < PER; k++) {
    const int c = cnt[tid * PER + k];
    cellStart[tid * PER + k] = run;
    cnt[tid * PER + k] = run;   // becomes the scatter cursor
    run += c;
  }
  if (tid == 255) cellStart[GCELLS] = run;
  __syncthreads();
}

__device__ void grid_sort_cells(const int32_t* cellStart, int32_t* items) {
  for (int c = threadIdx.x; c < GCELLS; c += 256) {
    const int s = cellStart[c], e = cellStart[c + 1];
    for (int i = s + 1; i < e; i++) {
      const int v = items[i];
      int j = i - 1;
      while (j >= s && items[j] > v) { items[j + 1] = items[j]; j--; }
      items[j + 1] = v;
    }
  }
}

__global__ void __launch_bounds__(256) k_grid_points(const plh_keypoint* kps, const int* nArr, int cap, plh_grid_params g,
                                                     int32_t* cellStartAll, int32_t* itemsAll) {
  __shared__ int cnt[GCELLS];
  __shared__ int part[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = min(nArr[b], cap);
  const plh_keypoint* K = kps + (long long)b * cap;
  int32_t* cellStart = cellStartAll + (long long)b * (GCELLS + 1);
  int32_t* items = itemsAll + (long long)b * cap;
  for (int i = tid; i < GCELLS; i += 256) cnt[i] = 0;
  __syncthreads();
  auto cellOf = [&](int i) -> int {   // Frame::PosInGrid: round(), not floor()
    const int posX = (int)roundf((K[i].x - g.min_x) * g.inv_w);
    const int posY = (int)roundf((K[i].y - g.min_y) * g.inv_h);
    if (posX < 0 || posX >= GCOLS || posY < 0 || posY >= GROWS) return -1;
    return posX * GROWS + posY;
  };
  for (int i = tid; i < n; i += 256) {
    const int c = cellOf(i);
    if (c >= 0) atomicAdd(&cnt[c], 1);
  }
  __syncthreads();
  grid_scan_and_publish(cnt, part, cellStart);
  for (int i = tid; i < n; i += 256) {
    const int c = cellOf(i);
    if (c >= 0) items[atomicAdd(&cnt[c], 1)] = i;
  }
  __syncthreads();
  __threadfence_block();
  grid_sort_cells(cellStart, items);
}

// The reference's Bresenham-style LineIterator over grid coordinates; calls f(cell) for every in-range cell.
template <typename F>
__device__ __forceinline__ void line_cells(const plh_keyline& kl, const plh_grid_params& g, F f) {
  double x1 = (double)(kl.startPointX * g.inv_w), y1 = (double)(kl.startPointY * g.inv_h);
  double x2 = (double)(kl.endPointX * g.inv_w), y2 = (double)(kl.endPointY * g.inv_h);
  const bool steep = fabs(y2 - y1) > fabs(x2 - x1);
  if (steep) { double t = x1; x1 = y1; y1 = t; t = x2; x2 = y2; y2 = t; }
  if (x1 > x2) { double t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; }
  const double dx = x2 - x1, dy = fabs(y2 - y1);
  double error = dx / 2.0;
  const int ystep = (y1 < y2) ? 1 : -1;
  int x = (int)x1, y = (int)y1;
  const int maxX = (int)x2;
  while (x <= maxX) {
    const int px = steep ? y : x, py = steep ? x : y;
    error -= dy;
    if (error < 0) { y += ystep; error += dx; }
    x++;
    if (px >= 0 && px < GCOLS && py >= 0 && py < GROWS) f(px * GROWS + py);
    if (x > 4096) break;   // NaN / absurd coordinates: bounded walk
  }
}

__global__ void __launch_bounds__(256) k_grid_lines(const plh_keyline* kls, const int* nArr, int cap, plh_grid_params g,
                                                    int32_t* cellStartAll, int32_t* itemsAll, int itemCap) {
  __shared__ int cnt[GCELLS];
  __shared__ int part[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = min(nArr[b], cap);
  const plh_keyline* K = kls + (long long)b * cap;
  int32_t* cellStart = cellStartAll + (long long)b * (GCELLS + 1);
  int32_t* items = itemsAll + (long long)b * itemCap;
  for (int i = tid; i < GCELLS; i += 256) cnt[i] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += 256) line_cells(K[i], g, [&](int c) { atomicAdd(&cnt[c], 1); });
  __syncthreads();
  grid_scan_and_publish(cnt, part, cellStart);
  for (int i = tid; i < n; i += 256)
    line_cells(K[i], g, [&](int c) {
      const int pos = atomicAdd(&cnt[c], 1);
      if (pos < itemCap) items[pos] = i;
    });
  __syncthreads();
  __threadfence_block();
  grid_sort_cells(cellStart, items);
}

// ------------------------------------------------------------------------------------------------------------
// Candidate generation
// ------------------------------------------------------------------------------------------------------------
struct CellWin {
  int x0, x1, y0, y1;
  bool ok;
};
__device__ __forceinline__ CellWin cell_window(const plh_grid_params& g, float x, float y, float r) {
  CellWin w;
  w.ok = false;
  w.x0 = max(0, (int)floorf((x - g.min_x - r) * g.inv_w));
  if (w.x0 >= GCOLS) return w;
  w.x1 = min(GCOLS - 1, (int)ceilf((x - g.min_x + r) * g.inv_w));
  if (w.x1 < 0) return w;
  w.y0 = max(0, (int)floorf((y - g.min_y - r) * g.inv_h));
  if (w.y0 >= GROWS) return w;
  w.y1 = min(GROWS - 1, (int)ceilf((y - g.min_y + r) * g.inv_h));
  if (w.y1 < 0) return w;
  w.ok = true;
  return w;
}

// Frame::GetFeaturesInArea: candidate indices in the reference's order (ix, then iy, then insertion order) into
// list[]; returns their number (uniform).
__device__ int collect_points(const plh_keypoint* K, const plh_grid_params& g, const int32_t* cs, const int32_t* ci, float x,
                              float y, float r, int minLevel, int maxLevel, int* list, int lane) {
  const CellWin w = cell_window(g, x, y, r);
  if (!w.ok) return 0;
  const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
  int cnt = 0;
  for (int ix = w.x0; ix <= w.x1; ix++) {
    const int s = cs[ix * GROWS + w.y0], e = cs[ix * GROWS + w.y1 + 1];
    for (int base = s; base < e; base += 64) {
      const int j = base + lane;
      bool ok = false;
      int id = 0;
      if (j < e) {
        id = ci[j];
        const plh_keypoint kp = K[id];
        ok = true;
        if (bCheckLevels) {
          if (kp.octave < minLevel) ok = false;
          if (maxLevel >= 0 && kp.octave > maxLevel) ok = false;
        }
        const float distx = kp.x - x, disty = kp.y - y;
        if (!(fabsf(distx) < r && fabsf(disty) < r)) ok = false;
      }
      const unsigned long long m = __ballot(ok);
      if (ok) list[cnt + __popcll(m & lanemask_lt())] = id;
      cnt += __popcll(m);
    }
  }
  return cnt;
}

// Frame::GetFeaturesInAreaForLine (minLevel / maxLevel are ignored by the reference).  seen[] must be all zero on
// entry and is restored to zero on exit.
__device__ int collect_lines(const plh_keyline* K, const double* fn, const plh_grid_params& g, const int32_t* cs,
                             const int32_t* ci, float x1, float y1, float x2, float y2, float r, float TH, int* list,
                             unsigned char* seen, int lane) {
  const float xs[3] = {x1, (float)((x1 + x2) / 2.0), x2};
  const float ys[3] = {y1, (float)((y1 + y2) / 2.0), y2};
  float delta1x = x1 - x2, delta1y = y1 - y2;
  const float norm_delta1 = sqrtf(delta1x * delta1x + delta1y * delta1y);
  delta1x /= norm_delta1;
  delta1y /= norm_delta1;
  int cnt = 0;
  for (int i = 0; i < 3; i++) {
    const CellWin w = cell_window(g, xs[i], ys[i], r);
    if (!w.ok) continue;
    for (int ix = w.x0; ix <= w.x1; ix++) {
      const int s = cs[ix * GROWS + w.y0], e = cs[ix * GROWS + w.y1 + 1];
      for (int base = s; base < e; base += 64) {
        const int j = base + lane;
        bool ok = false;
        int id = 0;
        FS_WAVE_SYNC();
        if (j < e) {
          id = ci[j];
          if (!seen[id]) {
            const plh_keyline k = K[id];
            float delta2x = k.startPointX - k.endPointX, delta2y = k.startPointY - k.endPointY;
            const float norm_delta2 = sqrtf(delta2x * delta2x + delta2y * delta2y);
            delta2x /= norm_delta2;
            delta2y /= norm_delta2;
            const float CosSita = fabsf(delta1x * delta2x + delta1y * delta2y);
            if (!(CosSita < TH)) {
              const float dist = (float)(fn[id * 3 + 0] * (double)xs[i] + fn[id * 3 + 1] * (double)ys[i] + fn[id * 3 + 2]);
              ok = fabsf(dist) < r;
            }
          }
        }
        // the same line sits in several cells of the window: only its first occurrence is appended
        unsigned long long m = __ballot(ok);
        while (m) {
          const int l = __ffsll((long long)m) - 1;
          m &= m - 1;
          const int idl = __shfl(id, l);
          FS_WAVE_SYNC();
          if (!seen[idl]) {
            FS_WAVE_SYNC();
            if (lane == 0) { list[cnt] = idl; seen[idl] = 1; }
            cnt++;
          }
        }
      }
    }
  }
  FS_WAVE_SYNC();
  for (int t = lane; t < cnt; t += 64) seen[list[t]] = 0;
  FS_WAVE_SYNC();
  return cnt;
}

__device__ __forceinline__ void three_maxima_lanes(int myHist, int& ind1, int& ind2, int& ind3) {
  // ComputeThreeMaxima's insertion cascade as selects on locals (written through the reference parameters the compiler
  // turned the three indices into a scratch array addressed by a selected pointer)
  int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
  for (int b = 0; b < 30; b++) {
    const int s = __shfl(myHist, b);
    const bool g1 = s > max1, g2 = s > max2, g3 = s > max3;
    max3 = g2 ? max2 : (g3 ? s : max3);
    i3 = g2 ? i2 : (g3 ? b : i3);
    max2 = g1 ? max1 : (g2 ? s : max2);
    i2 = g1 ? i1 : (g2 ? b : i2);
    max1 = g1 ? s : max1;
    i1 = g1 ? b : i1;
  }
  if (max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
  else if (max3 < 0.1f * (float)max1) { i3 = -1; }
  ind1 = i1; ind2 = i2; ind3 = i3;
}

__device__ __forceinline__ int rot_bin(float a1, float a2) {
  const float factor = 1.0f / 30;
  float rot = a1 - a2;
  if (rot < 0.0) rot += 360.0f;
  int bin = (int)roundf(rot * factor);
  if (bin == 30) bin = 0;
  return bin;
}

// ------------------------------------------------------------------------------------------------------------
// ORBmatcher::SearchForInitialization.  LDS: list[cap] dist[cap] m12[cap] m21[cap] mdist[cap] (int) + bin1[cap] (u8)
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_search_init(const plh_keypoint* kps1, const uint8_t* desc1, const int* n1Arr,
                                                    const plh_keypoint* kps2, const uint8_t* desc2, const int* n2Arr, int cap,
                                                    plh_grid_params g, const int32_t* csAll, const int32_t* ciAll,
                                                    float* prevMatched, int windowSize, float nnratio, int checkOri,
                                                    int32_t* matches12, int32_t* nmatchesOut) {
  HIP_DYNAMIC_SHARED(unsigned char, smem)
  int* list = (int*)smem;
  int* dist = list + cap;
  int* m12 = dist + cap;
  int* m21 = m12 + cap;
  int* mdist = m21 + cap;
  unsigned char* bin1 = (unsigned char*)(mdist + cap);
  const int pair = blockIdx.x, lane = threadIdx.x;
  const long long o = (long long)pair * cap;
  const plh_keypoint *K1 = kps1 + o, *K2 = kps2 + o;
  const uint8_t *D1 = desc1 + o * 32, *D2 = desc2 + o * 32;
  const int32_t* cs = csAll + (long long)pair * (GCELLS + 1);
  const int32_t* ci = ciAll + o;
  float* PM = prevMatched + o * 2;
  const int n1 = min(n1Arr[pair], cap), n2 = min(n2Arr[pair], cap);
  for (int i = lane; i < cap; i += 64) { m12[i] = -1; m21[i] = -1; mdist[i] = 0x7fffffff; bin1[i] = 255; }
  FS_WAVE_SYNC();
  int nmatches = 0, myHist = 0;
  for (int i1 = 0; i1 < n1; i1++) {
    const int level1 = K1[i1].octave;
    if (level1 > 0) continue;
    FS_WAVE_SYNC();
    const int K = collect_points(K2, g, cs, ci, PM[i1 * 2], PM[i1 * 2 + 1], (float)windowSize, level1, level1, list, lane);
    if (K == 0) continue;
    FS_WAVE_SYNC();
    for (int t = lane; t < K; t += 64) dist[t] = hamming_rows(D1 + (long long)i1 * 32, D2 + (long long)list[t] * 32);
    FS_WAVE_SYNC();
    int bestDist = 0x7fffffff, bestDist2 = 0x7fffffff, bestIdx2 = -1;
    for (int t = 0; t < K; t++) {
      const int i2 = list[t], d = dist[t];
      if (mdist[i2] <= d) continue;
      if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestIdx2 = i2; }
      else if (d < bestDist2) { bestDist2 = d; }
    }
    if (bestDist <= 50) {
      if ((float)bestDist < (float)bestDist2 * nnratio) {
        FS_WAVE_SYNC();
        const int prev = m21[bestIdx2];
        FS_WAVE_SYNC();
        if (prev >= 0) { m12[prev] = -1; nmatches--; }
        m12[i1] = bestIdx2;
        m21[bestIdx2] = i1;
        mdist[bestIdx2] = bestDist;
        nmatches++;
        if (checkOri) {
          const int bin = rot_bin(K1[i1].angle, K2[bestIdx2].angle);
          bin1[i1] = (unsigned char)bin;
          if (lane == bin) myHist++;
        }
      }
    }
  }
  FS_WAVE_SYNC();
  if (checkOri) {
    int ind1, ind2, ind3;
    three_maxima_lanes(myHist, ind1, ind2, ind3);
    int removed = 0;
    for (int i = lane; i < n1; i += 64) {
      const int b = bin1[i];
      if (b != 255 && b != ind1 && b != ind2 && b != ind3 && m12[i] >= 0) { m12[i] = -1; removed++; }
    }
    nmatches -= wave_sum(removed);
  }
  FS_WAVE_SYNC();
  for (int i = lane; i < cap; i += 64) {
    const int m = i < n1 ? m12[i] : -1;
    matches12[o + i] = m;
    if (m >= 0) { PM[i * 2] = K2[m].x; PM[i * 2 + 1] = K2[m].y; }
  }
  if (lane == 0) nmatchesOut[pair] = nmatches;
  (void)n2;
}

// ------------------------------------------------------------------------------------------------------------
// ORBmatcher::SearchByProjection, both forms.  variant 0: (F, MapPoints, th) -- radius from the viewing cosine,
// levels (l-1, l), best + second with the same-level ratio test; variant 1: (Cur, Last, th, mono) -- image-bounds
// test, radius th*scale[octave], level band by `mode`, best only, rotation histogram;
// variant 2: the search inside ORBmatcher::Fuse (ORBmatcher.cc:914-1061, 1063-1197) -- no occupancy, levels (l-1, l), the
// monocular chi-square gate e2 * invLevelSigma2 > 5.99, result per QUERY (assigned[q] = best keypoint or -1);
// variant 3: loop-closing SearchByProjection(KF, Scw, points, vpMatched, th) (:329-453) -- occupancy, levels (l-1, l), best.
// LDS: list[cap] dist[cap] asg[cap] (int) + occ[cap] (u8) ; per query (variant 1): pushBin[qcap] (u8), pushIdx[qcap] (int)
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_search_proj_points(int variant, const plh_keypoint* kps, const uint8_t* desc,
                                                           const int* nArr, int cap, plh_grid_params g, const int32_t* csAll,
                                                           const int32_t* ciAll, ScaleTab sf, ScaleTab invSig2, uint8_t* occupiedAll,
                                                           const int* nqArr, int qcap, const uint8_t* qValid, const float* qXY,
                                                           const int32_t* qLevel, const float* qAux, const uint8_t* qDesc,
                                                           const uint8_t* qHasObs, float th, float nnratio, int mode,
                                                           int checkOri, int distTh, int nlevels, int qLds, int32_t* assignedAll,
                                                           int32_t* nmatchesOut) {
  HIP_DYNAMIC_SHARED(unsigned char, smem)
  // qLds = qcap when the rotation histogram needs its per-query records (variant 1 with checkOri), else 0: the queries
  // themselves are streamed from global memory, so the other variants take any number of them (a local map of
  // Tracking::SearchLocalPoints routinely holds more than 6000 points).
  int* list = (int*)smem;
  int* dist = list + cap;
  int* asg = dist + cap;
  int* pushIdx = asg + cap;
  unsigned char* occ = (unsigned char*)(pushIdx + qLds);
  unsigned char* pushBin = occ + cap;
  const int pair = blockIdx.x, lane = threadIdx.x;
  const long long o = (long long)pair * cap, qo = (long long)pair * qcap;
  const plh_keypoint* K = kps + o;
  const uint8_t* D = desc + o * 32;
  const int32_t* cs = csAll + (long long)pair * (GCELLS + 1);
  const int32_t* ci = ciAll + o;
  const int n = min(nArr[pair], cap), nq = min(nqArr[pair], qcap);
  for (int i = lane; i < cap; i += 64) { asg[i] = -1; occ[i] = (i < n && variant != 2) ? occupiedAll[o + i] : (variant == 2 ? 0 : 1); }
  for (int i = lane; i < qLds; i += 64) pushBin[i] = 255;
  if (variant == 2)
    for (int i = lane; i < qcap; i += 64) assignedAll[qo + i] = -1;
  FS_WAVE_SYNC();
  int nmatches = 0, myHist = 0;
  const bool bFactor = th != 1.0;
  for (int q = 0; q < nq; q++) {
    if (!qValid[qo + q]) continue;
    const float x = qXY[(qo + q) * 2], y = qXY[(qo + q) * 2 + 1];
    const int lvl = qLevel[qo + q];
    if (lvl < 0 || lvl >= nlevels) continue;   // outside mvScaleFactors: the reference would read past the vector; skipped
    float radius;
    int minL, maxL;
    if (variant == 0) {
      float r = qAux[qo + q] > 0.998 ? 2.5 : 4.0;   // RadiusByViewingCos
      if (bFactor) r *= th;
      radius = r * sf.v[lvl];
      minL = lvl - 1; maxL = lvl;
    } else if (variant == 1) {
      if (x < g.min_x || x > g.max_x) continue;
      if (y < g.min_y || y > g.max_y) continue;
      radius = th * sf.v[lvl];
      if (mode == 1) { minL = lvl; maxL = -1; }
      else if (mode == 2) { minL = 0; maxL = lvl; }
      else { minL = lvl - 1; maxL = lvl + 1; }
    } else {   // Fuse / loop closing: KeyFrame::GetFeaturesInArea(u, v, radius), then kpLevel in [l-1, l]
      radius = th * sf.v[lvl];
      minL = lvl - 1; maxL = lvl;
    }
    FS_WAVE_SYNC();
    const int Kc = collect_points(K, g, cs, ci, x, y, radius, minL, maxL, list, lane);
    if (Kc == 0) continue;
    FS_WAVE_SYNC();
    for (int t = lane; t < Kc; t += 64) dist[t] = hamming_rows(qDesc + (qo + q) * 32, D + (long long)list[t] * 32);
    FS_WAVE_SYNC();
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int t = 0; t < Kc; t++) {
      const int idx = list[t];
      if (occ[idx]) continue;
      if (variant == 2) {   // reprojection error gate (monocular): e2 * mvInvLevelSigma2[kpLevel] > 5.99
        const float ex = x - K[idx].x, ey = y - K[idx].y;
        const float e2 = ex * ex + ey * ey;
        if (e2 * invSig2.v[K[idx].octave & 15] > 5.99) continue;
      }
      const int d = dist[t];
      if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestLevel2 = bestLevel; bestLevel = K[idx].octave; bestIdx = idx; }
      else if (variant == 0 && d < bestDist2) { bestLevel2 = K[idx].octave; bestDist2 = d; }
    }
    if (bestDist <= distTh) {
      if (variant == 0 && bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;
      if (variant == 2) {
        if (lane == 0) assignedAll[qo + q] = bestIdx;
        nmatches++;
        continue;
      }
      FS_WAVE_SYNC();
      asg[bestIdx] = q;
      occ[bestIdx] = qHasObs[qo + q];
      nmatches++;
      if (variant == 1 && checkOri && qLds) {
        const int bin = rot_bin(qAux[qo + q], K[bestIdx].angle);
        pushBin[q] = (unsigned char)bin;
        pushIdx[q] = bestIdx;
        if (lane == bin) myHist++;
      }
    }
  }
  FS_WAVE_SYNC();
  if (variant == 1 && checkOri && qLds) {
    int ind1, ind2, ind3;
    three_maxima_lanes(myHist, ind1, ind2, ind3);
    int removed = 0;
    for (int q = lane; q < nq; q += 64) {
      const int b = pushBin[q];
      if (b != 255 && b != ind1 && b != ind2 && b != ind3) { asg[pushIdx[q]] = -1; occ[pushIdx[q]] = 0; removed++; }   // mvpMapPoints[..] = NULL
    }
    nmatches -= wave_sum(removed);
  }
  FS_WAVE_SYNC();
  if (variant != 2)
    for (int i = lane; i < cap; i += 64) {
      assignedAll[o + i] = i < n ? asg[i] : -1;
      if (i < n) occupiedAll[o + i] = occ[i];
    }
  if (lane == 0) nmatchesOut[pair] = nmatches;
}

// ------------------------------------------------------------------------------------------------------------
// LSDmatcher::SearchByProjection, both forms.  variant 0: (F, MapLines, th); variant 1: (Cur, Last, th).
// LDS: list[cap] dist[cap] asg[cap] (int) + occ[cap] seen[cap] (u8)
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_search_proj_lines(int variant, const plh_keyline* kls, const uint8_t* ldesc,
                                                          const double* fnAll, const int* nArr, int cap, plh_grid_params g,
                                                          const int32_t* csAll, const int32_t* ciAll, int itemCap,
                                                          uint8_t* occupiedAll, const int* nqArr, int qcap,
                                                          const uint8_t* qValid, const float* qSeg, const float* qAux,
                                                          const uint8_t* qDesc, const uint8_t* qHasObs, float th, float nnratio,
                                                          int32_t* assignedAll, int32_t* nmatchesOut) {
  HIP_DYNAMIC_SHARED(unsigned char, smem)
  int* list = (int*)smem;
  int* dist = list + cap;
  int* asg = dist + cap;
  unsigned char* occ = (unsigned char*)(asg + cap);
  unsigned char* seen = occ + cap;
  const int pair = blockIdx.x, lane = threadIdx.x;
  const long long o = (long long)pair * cap, qo = (long long)pair * qcap;
  const plh_keyline* K = kls + o;
  const uint8_t* D = ldesc + o * 32;
  const double* fn = fnAll + o * 3;
  const int32_t* cs = csAll + (long long)pair * (GCELLS + 1);
  const int32_t* ci = ciAll + (long long)pair * itemCap;
  const int n = min(nArr[pair], cap), nq = min(nqArr[pair], qcap);
  for (int i = lane; i < cap; i += 64) { asg[i] = -1; seen[i] = 0; occ[i] = i < n ? occupiedAll[o + i] : 1; }
  FS_WAVE_SYNC();
  int nmatches = 0;
  const bool bFactor = th != 1.0;
  for (int q = 0; q < nq; q++) {
    if (!qValid[qo + q]) continue;
    const float* sg = qSeg + (qo + q) * 4;
    float r, TH;
    if (variant == 0) {
      r = qAux[qo + q] > 0.998 ? 5.0 : 8.0;   // LSDmatcher::RadiusByViewingCos
      if (bFactor) r *= th;
      TH = 0.998f;
    } else {
      r = th;
      TH = 0.96f;
    }
    FS_WAVE_SYNC();
    const int Kc = collect_lines(K, fn, g, cs, ci, sg[0], sg[1], sg[2], sg[3], r, TH, list, seen, lane);
    if (Kc == 0) continue;
    for (int t = lane; t < Kc; t += 64) dist[t] = hamming_rows(qDesc + (qo + q) * 32, D + (long long)list[t] * 32);
    FS_WAVE_SYNC();
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int t = 0; t < Kc; t++) {
      const int idx = list[t];
      if (occ[idx]) continue;
      const int d = dist[t];
      if (variant == 1) {
        const float la = qAux[qo + q], lb = K[idx].lineLength;
        const float max_ = fmaxf(la, lb), min_ = fminf(la, lb);
        if (min_ / max_ < 0.75) continue;
        if (d < bestDist) { bestDist = d; bestIdx = idx; }
      } else {
        if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestLevel2 = bestLevel; bestLevel = K[idx].octave; bestIdx = idx; }
        else if (d < bestDist2) { bestLevel2 = K[idx].octave; bestDist2 = d; }
      }
    }
    if (bestDist <= 80) {
      if (variant == 0 && bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;
      FS_WAVE_SYNC();
      asg[bestIdx] = q;
      occ[bestIdx] = qHasObs[qo + q];
      nmatches++;
    }
  }
  FS_WAVE_SYNC();
  for (int i = lane; i < cap; i += 64) {
    assignedAll[o + i] = i < n ? asg[i] : -1;
    if (i < n) occupiedAll[o + i] = occ[i];
  }
  if (lane == 0) nmatchesOut[pair] = nmatches;
}


// ------------------------------------------------------------------------------------------------------------
// Round 6: the projection searches as PREPASS + ORDERED RESOLVE.
//
// k_search_proj_points / k_search_proj_lines above walk a frame's queries with ONE wavefront, each query a chain of five dependent
// global round trips (query -> grid cells -> items -> keypoints -> descriptors): 2.7 us per query, 16 ms for the 6000 points of a local
// map -- a frame searched by a live tracker (Tracking.cc:1792-1800) took eight times as long as on the CPU, and a resident batch
// ran one wavefront per SIMD.  What is order dependent in the reference loops is only WHICH candidates are still free when a query
// takes its turn; the candidates of a query and their Hamming distances are not.  So:
//   * prepass (k_proj_points_prepass / k_proj_lines_prepass): one LANE per query, every query of every frame in parallel: enumerate the
//     window in the reference's order, apply every test that depends on the query and the candidate alone (level band, window,
//     direction, length ratio, chi-square gate, the INITIAL occupancy), and keep the PROJ_TOPK best candidates in (distance,
//     enumeration order) order.  The reference's (best, second best) are the first two FREE entries of that order (strict `<` in its
//     scan: the first of equal distances wins -- a stable insertion reproduces it), so a short list is all a query needs unless
//     nearly all of it has been taken by earlier queries (flag `truncated`: more candidates existed than the list holds).
//   * resolve (k_proj_resolve): one wavefront per frame takes the queries 64 at a time, one per lane: every lane evaluates its list
//     against the current occupancy; a lane whose best or second best is claimed by an EARLIER accepting lane of the same round (or
//     whose truncated list ran out) must wait; everything in front of the first such lane is committed at once, then the round
//     repeats from there.  A list that ran out is re-evaluated exactly by the one-wavefront code (slow path: rare).
// Same assignments as the sequential kernels, which stay as the oracle's peers for A/B (plh_debug_set_proj_serial) and serve
// frames whose capacity does not fit the 13-bit index of a list entry.
// ------------------------------------------------------------------------------------------------------------
#ifndef PLH_PROJ_TOPK
#define PLH_PROJ_TOPK 8   // (plh_debug_set_proj_serial(2) cuts the lists to 2 at run time: nearly every contended query then takes the slow path)
#endif
constexpr int PROJ_TOPK = PLH_PROJ_TOPK;
constexpr uint32_t PROJ_EMPTY = 0xffffffffu, PROJ_TRUNC = 0x40000000u;
struct ProjTop {
  uint32_t e[PROJ_TOPK];   // idx | dist << 13 | level << 22 (| PROJ_TRUNC in e[0]); PROJ_EMPTY = no entry
};
__device__ __forceinline__ uint32_t proj_entry(int idx, int dist, int level) { return (uint32_t)idx | ((uint32_t)dist << 13) | ((uint32_t)(level & 15) << 22); }
__device__ __forceinline__ int proj_idx(uint32_t e) { return (int)(e & 0x1fffu); }
__device__ __forceinline__ int proj_dist(uint32_t e) { return (int)((e >> 13) & 0x1ffu); }
__device__ __forceinline__ int proj_level(uint32_t e) { return (int)((e >> 22) & 15u); }
// stable insertion: an entry goes in front of the first one with a LARGER distance (equal distances keep their order of arrival)
__device__ __forceinline__ void proj_insert(uint32_t (&top)[PROJ_TOPK], bool& trunc, int idx, int dist, int level) {
  const uint32_t ne = proj_entry(idx, dist, level);
  if (top[PROJ_TOPK - 1] != PROJ_EMPTY && dist >= proj_dist(top[PROJ_TOPK - 1])) { trunc = true; return; }
  if (top[PROJ_TOPK - 1] != PROJ_EMPTY) trunc = true;   // the last entry is pushed out
  uint32_t carry = ne;
  bool placed = false;
#pragma unroll
  for (int k = 0; k < PROJ_TOPK; k++) {
    const uint32_t cur = top[k];
    if (!placed && (cur == PROJ_EMPTY || dist < proj_dist(cur))) placed = true;
    if (placed) { top[k] = carry; carry = cur; }
  }
}

__global__ void __launch_bounds__(256) k_proj_points_prepass(int variant, const plh_keypoint* kps, const uint8_t* desc, const int* nArr, int cap,
                                                             plh_grid_params g, const int32_t* csAll, const int32_t* ciAll, ScaleTab sf,
                                                             ScaleTab invSig2, const uint8_t* occupiedAll, const int* nqArr, int qcap,
                                                             const uint8_t* qValid, const float* qXY, const int32_t* qLevel, const float* qAux,
                                                             const uint8_t* qDesc, float th, int mode, int nlevels, ProjTop* out, int keep) {
  const int pair = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
  if (q >= qcap) return;
  const long long o = (long long)pair * cap, qo = (long long)pair * qcap;
  uint32_t top[PROJ_TOPK];
#pragma unroll
  for (int k = 0; k < PROJ_TOPK; k++) top[k] = PROJ_EMPTY;
  bool trunc = false;
  const int n = min(nArr[pair], cap), nq = min(nqArr[pair], qcap);
  bool live = q < nq && qValid[qo + q] != 0;
  int lvl = 0;
  float x = 0.f, y = 0.f, radius = 0.f;
  int minL = 0, maxL = 0;
  if (live) {
    x = qXY[(qo + q) * 2]; y = qXY[(qo + q) * 2 + 1];
    lvl = qLevel[qo + q];
    live = lvl >= 0 && lvl < nlevels;
  }
  if (live) {
    if (variant == 0) {
      float r = qAux[qo + q] > 0.998 ? 2.5 : 4.0;   // RadiusByViewingCos
      if (th != 1.0) r *= th;
      radius = r * sf.v[lvl];
      minL = lvl - 1; maxL = lvl;
    } else if (variant == 1) {
      if (x < g.min_x || x > g.max_x || y < g.min_y || y > g.max_y) live = false;
      radius = th * sf.v[lvl];
      if (mode == 1) { minL = lvl; maxL = -1; }
      else if (mode == 2) { minL = 0; maxL = lvl; }
      else { minL = lvl - 1; maxL = lvl + 1; }
    } else {
      radius = th * sf.v[lvl];
      minL = lvl - 1; maxL = lvl;
    }
  }
  if (live) {
    const CellWin w = cell_window(g, x, y, radius);
    if (w.ok) {
      const plh_keypoint* K = kps + o;
      const uint8_t* D = desc + o * 32;
      const int32_t* cs = csAll + (long long)pair * (GCELLS + 1);
      const int32_t* ci = ciAll + o;
      const unsigned long long* qd = reinterpret_cast<const unsigned long long*>(qDesc + (qo + q) * 32);
      const unsigned long long q0 = qd[0], q1 = qd[1], q2 = qd[2], q3 = qd[3];
      const bool bCheckLevels = (minL > 0) || (maxL >= 0);
      for (int ix = w.x0; ix <= w.x1; ix++) {
        const int s = cs[ix * GROWS + w.y0], e = cs[ix * GROWS + w.y1 + 1];
        for (int j = s; j < e; j++) {
          const int id = ci[j];
          if (id < 0 || id >= n) continue;
          const plh_keypoint kp = K[id];
          if (bCheckLevels) {
            if (kp.octave < minL) continue;
            if (maxL >= 0 && kp.octave > maxL) continue;
          }
          const float distx = kp.x - x, disty = kp.y - y;
          if (!(fabsf(distx) < radius && fabsf(disty) < radius)) continue;
          if (variant == 2) {   // reprojection error gate (monocular): e2 * mvInvLevelSigma2[kpLevel] > 5.99
            const float ex = x - kp.x, ey = y - kp.y;
            const float e2 = ex * ex + ey * ey;
            if (e2 * invSig2.v[kp.octave & 15] > 5.99) continue;
          } else if (occupiedAll[o + id]) {
            continue;   // taken before the call: never a candidate (occupancy only ever grows while the queries are walked)
          }
          const unsigned long long* dd = reinterpret_cast<const unsigned long long*>(D + (long long)id * 32);
          const int d = __popcll(q0 ^ dd[0]) + __popcll(q1 ^ dd[1]) + __popcll(q2 ^ dd[2]) + __popcll(q3 ^ dd[3]);
          proj_insert(top, trunc, id, d, kp.octave);
        }
      }
    }
  }
#pragma unroll
  for (int k = 1; k < PROJ_TOPK; k++)   // (test switch plh_debug_set_proj_serial(2): a list of `keep` entries, so that lists run out)
    if (k >= keep && top[k] != PROJ_EMPTY) { top[k] = PROJ_EMPTY; trunc = true; }
  if (trunc && top[0] != PROJ_EMPTY) top[0] |= PROJ_TRUNC;
  ProjTop t;
#pragma unroll
  for (int k = 0; k < PROJ_TOPK; k++) t.e[k] = top[k];
  out[qo + q] = t;
}

__global__ void __launch_bounds__(256) k_proj_lines_prepass(int variant, const plh_keyline* kls, const uint8_t* ldesc, const double* fnAll,
                                                            const int* nArr, int cap, plh_grid_params g, const int32_t* csAll,
                                                            const int32_t* ciAll, int itemCap, const uint8_t* occupiedAll, const int* nqArr,
                                                            int qcap, const uint8_t* qValid, const float* qSeg, const float* qAux,
                                                            const uint8_t* qDesc, float th, ProjTop* out, int keep) {
  const int pair = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
  if (q >= qcap) return;
  const long long o = (long long)pair * cap, qo = (long long)pair * qcap;
  uint32_t top[PROJ_TOPK];
#pragma unroll
  for (int k = 0; k < PROJ_TOPK; k++) top[k] = PROJ_EMPTY;
  bool trunc = false;
  const int n = min(nArr[pair], cap), nq = min(nqArr[pair], qcap);
  if (q < nq && qValid[qo + q]) {
    const plh_keyline* K = kls + o;
    const uint8_t* D = ldesc + o * 32;
    const double* fn = fnAll + o * 3;
    const int32_t* cs = csAll + (long long)pair * (GCELLS + 1);
    const int32_t* ci = ciAll + (long long)pair * itemCap;
    const float* sg = qSeg + (qo + q) * 4;
    const float x1 = sg[0], y1 = sg[1], x2 = sg[2], y2 = sg[3];
    float r, TH;
    if (variant == 0) {
      r = qAux[qo + q] > 0.998 ? 5.0 : 8.0;   // LSDmatcher::RadiusByViewingCos
      if (th != 1.0) r *= th;
      TH = 0.998f;
    } else {
      r = th;
      TH = 0.96f;
    }
    const float xs[3] = {x1, (float)((x1 + x2) / 2.0), x2};
    const float ys[3] = {y1, (float)((y1 + y2) / 2.0), y2};
    float delta1x = x1 - x2, delta1y = y1 - y2;
    const float norm_delta1 = sqrtf(delta1x * delta1x + delta1y * delta1y);
    delta1x /= norm_delta1;
    delta1y /= norm_delta1;
    const unsigned long long* qd = reinterpret_cast<const unsigned long long*>(qDesc + (qo + q) * 32);
    const unsigned long long q0 = qd[0], q1 = qd[1], q2 = qd[2], q3 = qd[3];
    const float la = qAux[qo + q];
    // the same line sits in several cells and is probed from three points: GetFeaturesInAreaForLine lists it once, at its first
    // passing probe.  A line that is in the top list is recognised by its index; one that is not (it was rejected for its distance, or
    // pushed out) would be rejected again, so that test is all the de-duplication the list needs -- except for the `truncated` flag,
    // which a second visit must not set on its own: the small bitmap remembers the lines already counted.
    uint32_t seenLo[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};   // lines 0 .. 255 (a frame keeps nLSDFeature + 1 = 201); beyond: conservative
    for (int i = 0; i < 3; i++) {
      const CellWin w = cell_window(g, xs[i], ys[i], r);
      if (!w.ok) continue;
      for (int ix = w.x0; ix <= w.x1; ix++) {
        const int s = cs[ix * GROWS + w.y0], e = cs[ix * GROWS + w.y1 + 1];
        for (int j = s; j < e; j++) {
          const int id = ci[j];
          if (id < 0 || id >= n) continue;
          if (id < 256 && ((seenLo[id >> 5] >> (id & 31)) & 1u)) continue;
          bool dup = false;
#pragma unroll
          for (int k = 0; k < PROJ_TOPK; k++) dup = dup || (top[k] != PROJ_EMPTY && proj_idx(top[k]) == id);
          if (dup) continue;
          const plh_keyline k = K[id];
          float delta2x = k.startPointX - k.endPointX, delta2y = k.startPointY - k.endPointY;
          const float norm_delta2 = sqrtf(delta2x * delta2x + delta2y * delta2y);
          delta2x /= norm_delta2;
          delta2y /= norm_delta2;
          const float CosSita = fabsf(delta1x * delta2x + delta1y * delta2y);
          if (CosSita < TH) continue;
          const float dist = (float)(fn[id * 3 + 0] * (double)xs[i] + fn[id * 3 + 1] * (double)ys[i] + fn[id * 3 + 2]);
          if (!(fabsf(dist) < r)) continue;
          if (id < 256) {
#pragma unroll
            for (int wd = 0; wd < 8; wd++)
              if (wd == (id >> 5)) seenLo[wd] |= 1u << (id & 31);
          }
          if (occupiedAll[o + id]) continue;
          if (variant == 1) {
            const float lb = k.lineLength;
            const float max_ = fmaxf(la, lb), min_ = fminf(la, lb);
            if (min_ / max_ < 0.75) continue;
          }
          const unsigned long long* dd = reinterpret_cast<const unsigned long long*>(D + (long long)id * 32);
          const int d = __popcll(q0 ^ dd[0]) + __popcll(q1 ^ dd[1]) + __popcll(q2 ^ dd[2]) + __popcll(q3 ^ dd[3]);
          proj_insert(top, trunc, id, d, k.octave);
        }
      }
    }
  }
#pragma unroll
  for (int k = 1; k < PROJ_TOPK; k++)   // (test switch plh_debug_set_proj_serial(2): a list of `keep` entries, so that lists run out)
    if (k >= keep && top[k] != PROJ_EMPTY) { top[k] = PROJ_EMPTY; trunc = true; }
  if (trunc && top[0] != PROJ_EMPTY) top[0] |= PROJ_TRUNC;
  ProjTop t;
#pragma unroll
  for (int k = 0; k < PROJ_TOPK; k++) t.e[k] = top[k];
  out[qo + q] = t;
}

// Lines, small launches (at most PROJ_G8_MAX queries in all -- a tracker's call on one frame): EIGHT lanes per query.  A line query
// probes three points with windows of many cells, and every occurrence of a keyline in them costs a 68-byte record, its line equation
// and a descriptor row: one lane per query walks ~250 occurrences one after the other -- 250 us for the 200 map lines of ONE frame,
// what made LSDmatcher::SearchByProjection through the adaptor as slow as the CPU.  (A resident batch has queries enough to fill the
// GPU with one lane each, and there the eight-lane form's eightfold index scan costs 3 x the time: measured, 0.99 -> 3.06 ms per 1024
// frames.)  The LINES are dealt to the query's eight lanes (keyline index mod 8): every lane walks all occurrences
// (one index load each) but pays for its own lines only, so that all occurrences of a line meet in one lane -- "listed once, at its
// first passing probe" stays a per-lane matter -- and keeps its best PROJ_TOPK by (distance, occurrence number); the occurrence
// number IS the enumeration order of the reference's loops.  A lane's list is exact for its lines, so the PROJ_TOPK least heads of
// the eight lists, taken one by one, are the one-lane form's list.
constexpr int PROJ_LG = 8;   // lanes per line query
constexpr long long PROJ_G8_MAX = 4096;   // queries of a launch up to which the eight-lane form is used
__device__ __forceinline__ unsigned long long proj_key(int dist, unsigned seq, int id, int level) {
  return ((unsigned long long)(unsigned)dist << 44) | ((unsigned long long)(seq & 0x7ffffffu) << 17) | ((unsigned long long)(unsigned)id << 4) |
         (unsigned long long)(level & 15);
}
__device__ __forceinline__ int pkey_id(unsigned long long k) { return (int)((k >> 4) & 0x1fffu); }
constexpr unsigned long long PKEY_EMPTY = ~0ull;
__global__ void __launch_bounds__(256) k_proj_lines_prepass_g8(int variant, const plh_keyline* kls, const uint8_t* ldesc, const double* fnAll,
                                                            const int* nArr, int cap, plh_grid_params g, const int32_t* csAll,
                                                            const int32_t* ciAll, int itemCap, const uint8_t* occupiedAll, const int* nqArr,
                                                            int qcap, const uint8_t* qValid, const float* qSeg, const float* qAux,
                                                            const uint8_t* qDesc, float th, ProjTop* out, int keep) {
  const int pair = blockIdx.y, q = blockIdx.x * (256 / PROJ_LG) + (int)(threadIdx.x / PROJ_LG), sub = (int)(threadIdx.x % PROJ_LG);
  // (a group's eight lanes stay together through the shuffles of the merge: no early return)
  const long long o = (long long)pair * cap, qo = (long long)pair * qcap;
  unsigned long long top[PROJ_TOPK];
#pragma unroll
  for (int k = 0; k < PROJ_TOPK; k++) top[k] = PKEY_EMPTY;
  bool trunc = false;
  const int n = min(nArr[pair], cap), nq = min(nqArr[pair], qcap);
  if (q < qcap && q < nq && qValid[qo + q]) {
    const plh_keyline* K = kls + o;
    const uint8_t* D = ldesc + o * 32;
    const double* fn = fnAll + o * 3;
    const int32_t* cs = csAll + (long long)pair * (GCELLS + 1);
    const int32_t* ci = ciAll + (long long)pair * itemCap;
    const float* sg = qSeg + (qo + q) * 4;
    const float x1 = sg[0], y1 = sg[1], x2 = sg[2], y2 = sg[3];
    float r, TH;
    if (variant == 0) {
      r = qAux[qo + q] > 0.998 ? 5.0 : 8.0;   // LSDmatcher::RadiusByViewingCos
      if (th != 1.0) r *= th;
      TH = 0.998f;
    } else {
      r = th;
      TH = 0.96f;
    }
    const float xs[3] = {x1, (float)((x1 + x2) / 2.0), x2};
    const float ys[3] = {y1, (float)((y1 + y2) / 2.0), y2};
    float delta1x = x1 - x2, delta1y = y1 - y2;
    const float norm_delta1 = sqrtf(delta1x * delta1x + delta1y * delta1y);
    delta1x /= norm_delta1;
    delta1y /= norm_delta1;
    const unsigned long long* qd = reinterpret_cast<const unsigned long long*>(qDesc + (qo + q) * 32);
    const unsigned long long q0 = qd[0], q1 = qd[1], q2 = qd[2], q3 = qd[3];
    const float la = qAux[qo + q];
    // The same line sits in several cells and is probed from three points: GetFeaturesInAreaForLine lists it once, at its first
    // passing probe.  A line that is in the lane's list is recognised by its index, and one that is not (rejected for its distance, or
    // pushed out) would be rejected again -- a later occurrence has the same distance and a larger number.  The small bitmap only keeps
    // a second visit from setting `truncated` on its own.
    uint32_t seenLo[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};   // lines 0 .. 255 (a frame keeps nLSDFeature + 1 = 201); beyond: conservative
    unsigned seqBase = 0u;   // occurrences in front of the current column: the same in the eight lanes
    for (int i = 0; i < 3; i++) {
      const CellWin w = cell_window(g, xs[i], ys[i], r);
      if (!w.ok) continue;
      for (int ix = w.x0; ix <= w.x1; ix++) {
        const int s = cs[ix * GROWS + w.y0], e = cs[ix * GROWS + w.y1 + 1];
        for (int j = s; j < e; j++) {
          const int id = ci[j];
          if (id < 0 || id >= n || (id & (PROJ_LG - 1)) != sub) continue;
          if (id < 256 && ((seenLo[id >> 5] >> (id & 31)) & 1u)) continue;
          bool dup = false;
#pragma unroll
          for (int k = 0; k < PROJ_TOPK; k++) dup = dup || (top[k] != PKEY_EMPTY && pkey_id(top[k]) == id);
          if (dup) continue;
          const plh_keyline k = K[id];
          float delta2x = k.startPointX - k.endPointX, delta2y = k.startPointY - k.endPointY;
          const float norm_delta2 = sqrtf(delta2x * delta2x + delta2y * delta2y);
          delta2x /= norm_delta2;
          delta2y /= norm_delta2;
          const float CosSita = fabsf(delta1x * delta2x + delta1y * delta2y);
          if (CosSita < TH) continue;
          const float dist = (float)(fn[id * 3 + 0] * (double)xs[i] + fn[id * 3 + 1] * (double)ys[i] + fn[id * 3 + 2]);
          if (!(fabsf(dist) < r)) continue;
          if (id < 256) {
#pragma unroll
            for (int wd = 0; wd < 8; wd++)
              if (wd == (id >> 5)) seenLo[wd] |= 1u << (id & 31);
          }
          if (occupiedAll[o + id]) continue;
          if (variant == 1) {
            const float lb = k.lineLength;
            const float max_ = fmaxf(la, lb), min_ = fminf(la, lb);
            if (min_ / max_ < 0.75) continue;
          }
          const unsigned long long* dd = reinterpret_cast<const unsigned long long*>(D + (long long)id * 32);
          const int d = __popcll(q0 ^ dd[0]) + __popcll(q1 ^ dd[1]) + __popcll(q2 ^ dd[2]) + __popcll(q3 ^ dd[3]);
          unsigned long long ne = proj_key(d, seqBase + (unsigned)(j - s), id, k.octave);
          if (top[PROJ_TOPK - 1] != PKEY_EMPTY) trunc = true;   // one entry will not fit
#pragma unroll
          for (int kk = 0; kk < PROJ_TOPK; kk++)
            if (ne < top[kk]) { const unsigned long long t = top[kk]; top[kk] = ne; ne = t; }
        }
        seqBase += (unsigned)max(e - s, 0);
      }
    }
  }
  // ---- merge of the group's eight lists: PROJ_TOPK times the least head (the trip count is the same for every group of the wavefront)
  uint32_t fin[PROJ_TOPK];
#pragma unroll
  for (int k = 0; k < PROJ_TOPK; k++) {
    unsigned long long m = top[0];
#pragma unroll
    for (int d = 1; d < PROJ_LG; d <<= 1) {
      const unsigned long long other = __shfl_xor(m, d);
      m = other < m ? other : m;
    }
    if (m != PKEY_EMPTY && top[0] == m) {    // keys are unique: exactly one lane of the group pops
#pragma unroll
      for (int kk = 0; kk + 1 < PROJ_TOPK; kk++) top[kk] = top[kk + 1];
      top[PROJ_TOPK - 1] = PKEY_EMPTY;
    }
    fin[k] = m == PKEY_EMPTY ? PROJ_EMPTY : proj_entry(pkey_id(m), (int)(m >> 44), (int)(m & 15ull));
  }
  {   // more candidates than the list holds: a lane's own list overflowed, or something is left after the merge
    bool anyTrunc = trunc || top[0] != PKEY_EMPTY;
#pragma unroll
    for (int d = 1; d < PROJ_LG; d <<= 1) anyTrunc = anyTrunc || (__shfl_xor((int)anyTrunc, d) != 0);
    trunc = anyTrunc;
  }
#pragma unroll
  for (int k = 1; k < PROJ_TOPK; k++)   // (test switch plh_debug_set_proj_serial(2): a list of `keep` entries, so that lists run out)
    if (k >= keep && fin[k] != PROJ_EMPTY) { fin[k] = PROJ_EMPTY; trunc = true; }
  if (trunc && fin[0] != PROJ_EMPTY) fin[0] |= PROJ_TRUNC;
  if (sub == 0 && q < qcap) {
    ProjTop t;
#pragma unroll
    for (int k = 0; k < PROJ_TOPK; k++) t.e[k] = fin[k];
    out[qo + q] = t;
  }
}

// The exact scan of ONE query against the current occupancy, by the whole wavefront (the loops of k_search_proj_points /
// k_search_proj_lines): the slow path of the resolve.  Returns the packed best / second (PROJ_EMPTY = none) in all lanes.
struct ProjPick { uint32_t best, second; };
__device__ ProjPick proj_slow_points(int variant, const plh_keypoint* K, const uint8_t* D, const plh_grid_params& g, const int32_t* cs,
                                     const int32_t* ci, const ScaleTab& sf, const ScaleTab& invSig2, const unsigned char* occ, int* list,
                                     int* dist, float x, float y, int lvl, float aux, const uint8_t* qd, float th, int mode, int lane) {
  float radius;
  int minL, maxL;
  if (variant == 0) {
    float r = aux > 0.998 ? 2.5 : 4.0;
    if (th != 1.0) r *= th;
    radius = r * sf.v[lvl];
    minL = lvl - 1; maxL = lvl;
  } else if (variant == 1) {
    radius = th * sf.v[lvl];
    if (mode == 1) { minL = lvl; maxL = -1; }
    else if (mode == 2) { minL = 0; maxL = lvl; }
    else { minL = lvl - 1; maxL = lvl + 1; }
  } else {
    radius = th * sf.v[lvl];
    minL = lvl - 1; maxL = lvl;
  }
  ProjPick p{PROJ_EMPTY, PROJ_EMPTY};
  FS_WAVE_SYNC();
  const int Kc = collect_points(K, g, cs, ci, x, y, radius, minL, maxL, list, lane);
  if (Kc == 0) return p;
  FS_WAVE_SYNC();
  for (int t = lane; t < Kc; t += 64) dist[t] = hamming_rows(qd, D + (long long)list[t] * 32);
  FS_WAVE_SYNC();
  int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1, idx2 = -1;
  for (int t = 0; t < Kc; t++) {
    const int idx = list[t];
    if (occ[idx]) continue;
    if (variant == 2) {
      const float ex = x - K[idx].x, ey = y - K[idx].y;
      const float e2 = ex * ex + ey * ey;
      if (e2 * invSig2.v[K[idx].octave & 15] > 5.99) continue;
    }
    const int d = dist[t];
    if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestLevel2 = bestLevel; idx2 = bestIdx; bestLevel = K[idx].octave; bestIdx = idx; }
    else if (d < bestDist2) { bestLevel2 = K[idx].octave; bestDist2 = d; idx2 = idx; }
  }
  FS_WAVE_SYNC();
  if (bestIdx >= 0) p.best = proj_entry(bestIdx, bestDist, bestLevel);
  if (idx2 >= 0) p.second = proj_entry(idx2, bestDist2, bestLevel2);
  return p;
}
__device__ ProjPick proj_slow_lines(int variant, const plh_keyline* K, const uint8_t* D, const double* fn, const plh_grid_params& g,
                                    const int32_t* cs, const int32_t* ci, const unsigned char* occ, unsigned char* seen, int* list, int* dist,
                                    const float* sg, float aux, const uint8_t* qd, float th, int lane) {
  float r, TH;
  if (variant == 0) {
    r = aux > 0.998 ? 5.0 : 8.0;
    if (th != 1.0) r *= th;
    TH = 0.998f;
  } else {
    r = th;
    TH = 0.96f;
  }
  ProjPick p{PROJ_EMPTY, PROJ_EMPTY};
  FS_WAVE_SYNC();
  const int Kc = collect_lines(K, fn, g, cs, ci, sg[0], sg[1], sg[2], sg[3], r, TH, list, seen, lane);
  if (Kc == 0) return p;
  for (int t = lane; t < Kc; t += 64) dist[t] = hamming_rows(qd, D + (long long)list[t] * 32);
  FS_WAVE_SYNC();
  int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1, idx2 = -1;
  for (int t = 0; t < Kc; t++) {
    const int idx = list[t];
    if (occ[idx]) continue;
    const int d = dist[t];
    if (variant == 1) {
      const float la = aux, lb = K[idx].lineLength;
      const float max_ = fmaxf(la, lb), min_ = fminf(la, lb);
      if (min_ / max_ < 0.75) continue;
      if (d < bestDist) { bestDist = d; bestIdx = idx; bestLevel = K[idx].octave; }
    } else {
      if (d < bestDist) { bestDist2 = bestDist; bestDist = d; bestLevel2 = bestLevel; idx2 = bestIdx; bestLevel = K[idx].octave; bestIdx = idx; }
      else if (d < bestDist2) { bestLevel2 = K[idx].octave; bestDist2 = d; idx2 = idx; }
    }
  }
  FS_WAVE_SYNC();
  if (bestIdx >= 0) p.best = proj_entry(bestIdx, bestDist, bestLevel);
  if (idx2 >= 0) p.second = proj_entry(idx2, bestDist2, bestLevel2);
  return p;
}

// The ordered resolve.  kind 0: points, kind 1: lines (what the slow path needs); variant as in the sequential kernels.
// LDS: list[cap] claim/dist[cap] asg[cap] pushIdx[qLds] (int) + occ[cap] seen[cap] pushBin[qLds] (u8) + hist[32] (int)
struct ProjFrame {
  const plh_keypoint* kps; const uint8_t* desc;             // points
  const plh_keyline* kls; const double* fn; int itemCap;    // lines (desc = LBD rows)
  const int32_t* cs; const int32_t* ci;
};
__global__ void __launch_bounds__(64) k_proj_resolve(int kind, int variant, ProjFrame F, const int* nArr, int cap, plh_grid_params g, ScaleTab sf,
                                                     ScaleTab invSig2, uint8_t* occupiedAll, const int* nqArr, int qcap, const uint8_t* qValid,
                                                     const float* qPos, const int32_t* qLevel, const float* qAux, const uint8_t* qDesc,
                                                     const uint8_t* qHasObs, float th, float nnratio, int mode, int checkOri, int distTh,
                                                     int nlevels, int qLds, const ProjTop* tops, int32_t* assignedAll, int32_t* nmatchesOut) {
  HIP_DYNAMIC_SHARED(unsigned char, smem)
  int* list = (int*)smem;
  int* claim = list + cap;          // doubles as dist[] of the slow path (never live at the same time)
  int* asg = claim + cap;
  int* pushIdx = asg + cap;
  int* hist = pushIdx + qLds;
  unsigned char* occ = (unsigned char*)(hist + 32);
  unsigned char* seen = occ + cap;
  unsigned char* pushBin = seen + cap;
  const int pair = blockIdx.x, lane = threadIdx.x;
  const long long o = (long long)pair * cap, qo = (long long)pair * qcap;
  const int n = min(nArr[pair], cap), nq = min(nqArr[pair], qcap);
  const bool perQuery = kind == 0 && variant == 2;   // Fuse: no occupancy, one answer per query
  const bool needSecond = variant == 0;
  const int posW = kind == 0 ? 2 : 4;
  for (int i = lane; i < cap; i += 64) {
    asg[i] = -1; claim[i] = 0x7fffffff; seen[i] = 0;
    occ[i] = perQuery ? 0 : (i < n ? occupiedAll[o + i] : 1);
  }
  for (int i = lane; i < qLds; i += 64) pushBin[i] = 255;
  if (lane < 32) hist[lane] = 0;
  FS_WAVE_SYNC();
  const plh_keypoint* K = F.kps ? F.kps + o : nullptr;
  const plh_keyline* KL = F.kls ? F.kls + o : nullptr;
  const uint8_t* D = F.desc + o * 32;
  const double* fn = F.fn ? F.fn + o * 3 : nullptr;
  const int32_t* cs = F.cs + (long long)pair * (GCELLS + 1);
  const int32_t* ci = F.ci + (kind == 0 ? o : (long long)pair * F.itemCap);
  int nmatches = 0;
  for (int base = 0; base < nq; base += 64) {
    const int q = base + lane;
    uint32_t top[PROJ_TOPK];
    bool trunc = false;
    unsigned char hasobs = 0;
    if (q < nq) {
      const ProjTop t = tops[qo + q];
#pragma unroll
      for (int k = 0; k < PROJ_TOPK; k++) top[k] = t.e[k];
      if (top[0] != PROJ_EMPTY) { trunc = (top[0] & PROJ_TRUNC) != 0u; top[0] &= ~PROJ_TRUNC; }
      hasobs = qHasObs[qo + q];
    } else {
#pragma unroll
      for (int k = 0; k < PROJ_TOPK; k++) top[k] = PROJ_EMPTY;
    }
    if (perQuery) {   // the prepass applied every test: the first entry is the answer
      const bool hit = top[0] != PROJ_EMPTY && proj_dist(top[0]) <= distTh;
      if (q < qcap && q < nq) assignedAll[qo + q] = hit ? proj_idx(top[0]) : -1;
      nmatches += __popcll(__ballot(hit));
      continue;
    }
    unsigned long long active = __ballot(top[0] != PROJ_EMPTY);
    uint32_t slowBest = PROJ_EMPTY, slowSecond = PROJ_EMPTY;   // a lane whose list ran out gets its picks from the slow path
    bool haveSlow = false;
    while (active) {
      const bool on = ((active >> lane) & 1ull) != 0ull;
      // first two free entries of my list (or the slow path's picks)
      uint32_t b = PROJ_EMPTY, s2 = PROJ_EMPTY;
      bool ranOut = false;
      if (on) {
        if (haveSlow) {
          b = slowBest; s2 = slowSecond;
        } else {
#pragma unroll
          for (int k = 0; k < PROJ_TOPK; k++) {
            const uint32_t e = top[k];
            if (e == PROJ_EMPTY) continue;
            if (occ[proj_idx(e)]) continue;
            if (b == PROJ_EMPTY) b = e;
            else if (s2 == PROJ_EMPTY) s2 = e;
          }
          ranOut = trunc && (b == PROJ_EMPTY || (needSecond && s2 == PROJ_EMPTY));
        }
      }
      bool accept = on && !ranOut && b != PROJ_EMPTY && proj_dist(b) <= distTh;
      if (accept && needSecond && s2 != PROJ_EMPTY && proj_level(b) == proj_level(s2) && (float)proj_dist(b) > nnratio * (float)proj_dist(s2)) accept = false;
      // an accepting lane whose map element has observations takes its keypoint away from the later lanes of this round
      FS_WAVE_SYNC();
      if (accept && hasobs) atomicMin(&claim[proj_idx(b)], lane);
      FS_WAVE_SYNC();
      bool blocked = on && ranOut;
      if (on && !ranOut) {
        if (b != PROJ_EMPTY && claim[proj_idx(b)] < lane) blocked = true;
        if (needSecond && s2 != PROJ_EMPTY && claim[proj_idx(s2)] < lane) blocked = true;
      }
      FS_WAVE_SYNC();
      if (accept && hasobs) claim[proj_idx(b)] = 0x7fffffff;
      FS_WAVE_SYNC();
      const unsigned long long blockedM = __ballot(blocked) & active;
      const int first = blockedM ? __ffsll((long long)blockedM) - 1 : 64;
      const unsigned long long commitM = active & (first >= 64 ? ~0ull : ((1ull << first) - 1ull));
      const bool mine = ((commitM >> lane) & 1ull) != 0ull;
      if (mine && accept) {
        const int bi = proj_idx(b);
        atomicMax(&asg[bi], q);               // a later query overwrites an earlier one whose map element has no observation
        if (hasobs) occ[bi] = 1;
        if (kind == 0 && variant == 1 && checkOri && qLds) {
          const int bin = rot_bin(qAux[qo + q], K[bi].angle);
          pushBin[q] = (unsigned char)bin;
          pushIdx[q] = bi;
          atomicAdd(&hist[bin], 1);
        }
      }
      nmatches += __popcll(__ballot(mine && accept));
      active &= ~commitM;
      haveSlow = haveSlow && !mine;
      FS_WAVE_SYNC();
      if (first < 64 && commitM == 0ull) {
        // the lowest waiting lane has nothing in front of it: what blocks it is its own list, which ran out.  Scan its window again,
        // exactly, against the occupancy as it stands (the whole wavefront works for that one query).
        const int qs = base + first;
        ProjPick pk;
        if (kind == 0)
          pk = proj_slow_points(variant, K, D, g, cs, ci, sf, invSig2, occ, list, claim, qPos[(qo + qs) * 2], qPos[(qo + qs) * 2 + 1],
                                qLevel[qo + qs], qAux[qo + qs], qDesc + (qo + qs) * 32, th, mode, lane);
        else
          pk = proj_slow_lines(variant, KL, D, fn, g, cs, ci, occ, seen, list, claim, qPos + (qo + qs) * 4, qAux[qo + qs], qDesc + (qo + qs) * 32, th,
                               lane);
        FS_WAVE_SYNC();
        for (int i = lane; i < cap; i += 64) claim[i] = 0x7fffffff;   // (the slow path used it as dist[])
        FS_WAVE_SYNC();
        if (lane == first) { slowBest = pk.best; slowSecond = pk.second; haveSlow = true; trunc = false; }
        if (pk.best == PROJ_EMPTY) {   // nothing free in its window: the query is through
          active &= ~(1ull << first);
          if (lane == first) haveSlow = false;
        }
      }
    }
    (void)posW;
  }
  FS_WAVE_SYNC();
  if (kind == 0 && variant == 1 && checkOri && qLds) {
    int ind1, ind2, ind3;
    const int myHist = lane < 30 ? hist[lane] : 0;
    three_maxima_lanes(myHist, ind1, ind2, ind3);
    int removed = 0;
    for (int q = lane; q < nq; q += 64) {
      const int b = pushBin[q];
      if (b != 255 && b != ind1 && b != ind2 && b != ind3) { asg[pushIdx[q]] = -1; occ[pushIdx[q]] = 0; removed++; }   // mvpMapPoints[..] = NULL
    }
    nmatches -= wave_sum(removed);
  }
  FS_WAVE_SYNC();
  if (!perQuery)
    for (int i = lane; i < cap; i += 64) {
      assignedAll[o + i] = i < n ? asg[i] : -1;
      if (i < n) occupiedAll[o + i] = occ[i];
    }
  if (lane == 0) nmatchesOut[pair] = nmatches;
}

// ------------------------------------------------------------------------------------------------------------
// The search inside LSDmatcher::Fuse (reference src/LSDmatcher.cpp:860-1002) with KeyFrame::GetLinesInArea
// (src/KeyFrame.cc:647-683): brute force over the KeyFrame's lines -- lanes take the lines, the minimum of
// (distance, line index) over the wave is the reference's first best.  One wave per frame walks the queries.
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_line_fuse_search(const plh_keyline* kls, const uint8_t* candDesc, const int* nArr, int cap,
                                                         ScaleTab sfl, int nlevels, const int* nqArr, int qcap, const uint8_t* qValid,
                                                         const float* qSeg, const int32_t* qLevel, const uint8_t* qDesc, float th,
                                                         float TH, int thLow, int32_t* bestAll, int32_t* nfoundOut) {
  const int pair = blockIdx.x, lane = threadIdx.x;
  const long long o = (long long)pair * cap, qo = (long long)pair * qcap;
  const plh_keyline* K = kls + o;
  const uint8_t* D = candDesc + o * 32;
  const int n = min(nArr[pair], cap), nq = min(nqArr[pair], qcap);
  int nfound = 0;
  for (int q = 0; q < qcap; q++) {
    int best = -1;
    // a predicted level outside the KeyFrame's scale table (MapLine::PredictScale does not clamp) is skipped: the reference
    // would index mvScaleFactorsLine out of bounds there
    if (q < nq && qValid[qo + q] && qLevel[qo + q] >= 0 && qLevel[qo + q] < nlevels) {
      const float* sg = qSeg + (qo + q) * 4;
      const float x1 = sg[0], y1 = sg[1], x2 = sg[2], y2 = sg[3];
      const int lvl = qLevel[qo + q];
      const float r = th * sfl.v[lvl];
      float delta1x = x1 - x2, delta1y = y1 - y2;
      const float norm_delta1 = sqrtf(delta1x * delta1x + delta1y * delta1y);
      delta1x /= norm_delta1;
      delta1y /= norm_delta1;
      int key = 0x7fffffff;
      for (int j = lane; j < n; j += 64) {
        const plh_keyline k = K[j];
        const float distance = (float)((0.5 * (x1 + x2) - k.pt_x) * (0.5 * (x1 + x2) - k.pt_x) +
                                       (0.5 * (y1 + y2) - k.pt_y) * (0.5 * (y1 + y2) - k.pt_y));
        if (distance > r * r) continue;
        float delta2x = k.startPointX - k.endPointX, delta2y = k.startPointY - k.endPointY;
        const float norm_delta2 = sqrtf(delta2x * delta2x + delta2y * delta2y);
        delta2x /= norm_delta2;
        delta2y /= norm_delta2;
        const float CosSita = fabsf(delta1x * delta2x + delta1y * delta2y);
        if (CosSita < TH) continue;
        if (k.octave < lvl - 1 || k.octave > lvl) continue;
        const int d = hamming_rows(qDesc + (qo + q) * 32, D + (long long)j * 32);
        key = min(key, (d << 16) | j);
      }
      for (int s = 32; s >= 1; s >>= 1) key = min(key, __shfl_xor(key, s));
      if (key != 0x7fffffff && (key >> 16) <= thLow) { best = key & 0xffff; nfound++; }
    }
    if (lane == 0) bestAll[qo + q] = best;
  }
  if (lane == 0) nfoundOut[pair] = nfound;
}

}  // namespace plh

using namespace plh;

// ORBmatcher::SearchBySim3, agreement check (ORBmatcher.cc:1396-1412): match12[i1] = vnMatch1[i1] when
// vnMatch2[vnMatch1[i1]] == i1.
__global__ void __launch_bounds__(64) k_sim3_agree(const int* n1Arr, const int* n2Arr, int cap, const int32_t* match1,
                                                   const int32_t* match2, int32_t* match12, int32_t* nfoundOut) {
  const int pair = blockIdx.x, lane = threadIdx.x;
  const long long o = (long long)pair * cap;
  const int n1 = min(n1Arr[pair], cap), n2 = min(n2Arr[pair], cap);
  int found = 0;
  for (int i1 = lane; i1 < cap; i1 += 64) {
    int r = -1;
    if (i1 < n1) {
      const int idx2 = match1[o + i1];
      if (idx2 >= 0 && idx2 < n2 && match2[o + idx2] == i1) { r = idx2; found++; }
    }
    match12[o + i1] = r;
  }
  found = wave_sum(found);
  if (lane == 0) nfoundOut[pair] = found;
}

namespace {
// A/B and test switch: 1 = the one-wavefront-per-frame kernels of rounds 1-5 (k_search_proj_points / k_search_proj_lines), 0 = prepass +
// ordered resolve (default).  Same assignments either way (tests/test_frame_search.py runs both against the oracle).
// 2 = prepass + resolve with the candidate lists cut to two entries: contended queries run out of list and take the slow path (tests).
bool g_proj_serial = false;
int g_proj_keep = PROJ_TOPK;
bool scale_tab(const float* sf, int nlevels, ScaleTab* t) {
  if (!sf || nlevels <= 0 || nlevels > 16) return false;
  for (int i = 0; i < 16; i++) t->v[i] = i < nlevels ? sf[i] : 0.f;
  return true;
}
}  // namespace

extern "C" {

plh_status plh_debug_set_proj_serial(int on) {
  g_proj_serial = on == 1;
  g_proj_keep = on == 2 ? 2 : PROJ_TOPK;
  return PLH_OK;
}

plh_status plh_frame_assign_grid_batch_dev(const plh_keypoint* d_kps_un, const int32_t* d_n, int cap, int batch,
                                           const plh_grid_params* gp, int32_t* d_cell_start, int32_t* d_cell_items,
                                           void* stream) {
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  if (!d_kps_un || !d_n || !gp || !d_cell_start || !d_cell_items || cap <= 0 || batch <= 0) {
    set_error("plh_frame_assign_grid_batch_dev: invalid argument");
    return PLH_ERR_INVALID;
  }
  hipLaunchKernelGGL(k_grid_points, dim3(batch), dim3(256), 0, (hipStream_t)stream, d_kps_un, (const int*)d_n, cap, *gp,
                     d_cell_start, d_cell_items);
  PLH_LAUNCH_CHECK();
  return PLH_OK;
}

plh_status plh_frame_assign_grid_lines_batch_dev(const plh_keyline* d_kl, const int32_t* d_nl, int cap, int batch,
                                                 const plh_grid_params* gp, int32_t* d_cell_start, int32_t* d_cell_items,
                                                 int item_cap, void* stream) {
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  if (!d_kl || !d_nl || !gp || !d_cell_start || !d_cell_items || cap <= 0 || batch <= 0 || item_cap < cap * PLH_GRID_COLS) {
    set_error("plh_frame_assign_grid_lines_batch_dev: invalid argument (item_cap must be >= cap * 64)");
    return PLH_ERR_INVALID;
  }
  hipLaunchKernelGGL(k_grid_lines, dim3(batch), dim3(256), 0, (hipStream_t)stream, d_kl, (const int*)d_nl, cap, *gp,
                     d_cell_start, d_cell_items, item_cap);
  PLH_LAUNCH_CHECK();
  return PLH_OK;
}

plh_status plh_orb_search_for_initialization_batch_dev(const plh_keypoint* d_kps1, const uint8_t* d_desc1, const int32_t* d_n1,
                                                       const plh_keypoint* d_kps2, const uint8_t* d_desc2, const int32_t* d_n2,
                                                       int cap, int pairs, const plh_grid_params* gp2,
                                                       const int32_t* d_cell_start2, const int32_t* d_cell_items2,
                                                       float* d_prev_matched, int window_size, float nnratio, int check_ori,
                                                       int32_t* d_matches12, int32_t* d_nmatches, void* stream) {
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  if (!d_kps1 || !d_desc1 || !d_n1 || !d_kps2 || !d_desc2 || !d_n2 || !gp2 || !d_cell_start2 || !d_cell_items2 ||
      !d_prev_matched || !d_matches12 || !d_nmatches || cap <= 0 || cap > 6000 || pairs <= 0) {
    set_error("plh_orb_search_for_initialization_batch_dev: invalid argument (cap must be in 1..6000)");
    return PLH_ERR_INVALID;
  }
  const size_t lds = (size_t)cap * (5 * 4 + 1) + 64;
  if (lds_request(k_search_init, lds, "plh_orb_search_for_initialization_batch_dev") != PLH_OK) return PLH_ERR_INVALID;
  hipLaunchKernelGGL(k_search_init, dim3(pairs), dim3(64), lds, (hipStream_t)stream, d_kps1, d_desc1, (const int*)d_n1, d_kps2,
                     d_desc2, (const int*)d_n2, cap, *gp2, d_cell_start2, d_cell_items2, d_prev_matched, window_size, nnratio,
                     check_ori, d_matches12, d_nmatches);
  PLH_LAUNCH_CHECK();
  return PLH_OK;
}

static plh_status launch_proj_points(int variant, const plh_keypoint* d_kps_un, const uint8_t* d_desc, const int32_t* d_n, int cap,
                                     int pairs, const plh_grid_params* gp, const int32_t* d_cs, const int32_t* d_ci,
                                     const float* scale_factors, int nlevels, uint8_t* d_occupied, const int32_t* d_nq, int qcap,
                                     const uint8_t* d_q_valid, const float* d_q_xy, const int32_t* d_q_level, const float* d_q_aux,
                                     const uint8_t* d_q_desc, const uint8_t* d_q_hasobs, float th, float nnratio, int mode,
                                     int check_ori, int32_t* d_assigned, int32_t* d_nmatches, void* stream, const char* who,
                                     int dist_th = 100 /* ORBmatcher::TH_HIGH */, const float* inv_level_sigma2 = nullptr) {
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  ScaleTab sf, is2;
  for (int i = 0; i < 16; i++) is2.v[i] = (inv_level_sigma2 && i < nlevels) ? inv_level_sigma2[i] : 0.f;
  if (!d_kps_un || !d_desc || !d_n || !gp || !d_cs || !d_ci || !scale_tab(scale_factors, nlevels, &sf) || !d_occupied || !d_nq ||
      !d_q_valid || !d_q_xy || !d_q_level || !d_q_aux || !d_q_desc || !d_q_hasobs || !d_assigned || !d_nmatches || cap <= 0 ||
      cap > 6000 || qcap <= 0 || pairs <= 0 || mode < 0 || mode > 2) {
    set_error("%s: invalid argument (cap in 1..6000; 1..16 levels)", who);
    return PLH_ERR_INVALID;
  }
  // per-query LDS records exist only for the rotation histogram of the (Cur, Last) / (Cur, KeyFrame) forms
  const int q_lds = (variant == 1 && check_ori) ? qcap : 0;
  if (q_lds > 12000) {
    set_error("%s: invalid argument (qcap <= 12000 with check_ori)", who);
    return PLH_ERR_INVALID;
  }
  if (!g_proj_serial) {
    // round 6: prepass (a lane per query, all frames in parallel) + ordered resolve (a wavefront per frame): see k_proj_resolve
    ProjTop* tops = nullptr;
    const size_t bytes = (size_t)pairs * qcap * sizeof(ProjTop);
    PLH_HIP(hipMallocAsync((void**)&tops, bytes, (hipStream_t)stream));
    hipLaunchKernelGGL(k_proj_points_prepass, dim3((qcap + 255) / 256, pairs), dim3(256), 0, (hipStream_t)stream, variant, d_kps_un, d_desc,
                       (const int*)d_n, cap, *gp, d_cs, d_ci, sf, is2, (const uint8_t*)d_occupied, (const int*)d_nq, qcap, d_q_valid, d_q_xy,
                       d_q_level, d_q_aux, d_q_desc, th, mode, nlevels, tops, g_proj_keep);
    const size_t lds = (size_t)cap * (3 * 4 + 2) + (size_t)q_lds * (4 + 1) + 128 + 64;
    ProjFrame F{d_kps_un, d_desc, nullptr, nullptr, 0, d_cs, d_ci};
    plh_status st = lds_request(k_proj_resolve, lds, who);
    if (st == PLH_OK) {
      hipLaunchKernelGGL(k_proj_resolve, dim3(pairs), dim3(64), lds, (hipStream_t)stream, 0, variant, F, (const int*)d_n, cap, *gp, sf, is2,
                         d_occupied, (const int*)d_nq, qcap, d_q_valid, d_q_xy, d_q_level, d_q_aux, d_q_desc, d_q_hasobs, th, nnratio, mode,
                         check_ori, dist_th, nlevels, q_lds, (const ProjTop*)tops, d_assigned, d_nmatches);
    }
    const hipError_t le = hipGetLastError();
    (void)hipFreeAsync(tops, (hipStream_t)stream);
    if (st != PLH_OK) return st;
    if (le != hipSuccess) { set_error("%s: kernel launch -> %s", who, hipGetErrorString(le)); return PLH_ERR_HIP; }
    return PLH_OK;
  }
  const size_t lds = (size_t)cap * (3 * 4 + 1) + (size_t)q_lds * (4 + 1) + 64;
  if (lds_request(k_search_proj_points, lds, who) != PLH_OK) return PLH_ERR_INVALID;
  hipLaunchKernelGGL(k_search_proj_points, dim3(pairs), dim3(64), lds, (hipStream_t)stream, variant, d_kps_un, d_desc,
                     (const int*)d_n, cap, *gp, d_cs, d_ci, sf, is2, d_occupied, (const int*)d_nq, qcap, d_q_valid, d_q_xy, d_q_level,
                     d_q_aux, d_q_desc, d_q_hasobs, th, nnratio, mode, check_ori, dist_th, nlevels, q_lds, d_assigned, d_nmatches);
  PLH_LAUNCH_CHECK();
  return PLH_OK;
}

plh_status plh_orb_search_by_projection_mp_batch_dev(const plh_keypoint* d_kps_un, const uint8_t* d_desc, const int32_t* d_n,
                                                     int cap, int pairs, const plh_grid_params* gp, const int32_t* d_cell_start,
                                                     const int32_t* d_cell_items, const float* scale_factors, int nlevels,
                                                     uint8_t* d_occupied, const int32_t* d_nq, int qcap, const uint8_t* d_q_valid,
                                                     const float* d_q_xy, const int32_t* d_q_level, const float* d_q_viewcos,
                                                     const uint8_t* d_q_desc, const uint8_t* d_q_hasobs, float th, float nnratio,
                                                     int32_t* d_assigned, int32_t* d_nmatches, void* stream) {
  return launch_proj_points(0, d_kps_un, d_desc, d_n, cap, pairs, gp, d_cell_start, d_cell_items, scale_factors, nlevels,
                            d_occupied, d_nq, qcap, d_q_valid, d_q_xy, d_q_level, d_q_viewcos, d_q_desc, d_q_hasobs, th, nnratio, 0,
                            0, d_assigned, d_nmatches, stream, "plh_orb_search_by_projection_mp_batch_dev");
}

plh_status plh_orb_search_by_projection_frame_batch_dev(const plh_keypoint* d_kps_un, const uint8_t* d_desc, const int32_t* d_n,
                                                        int cap, int pairs, const plh_grid_params* gp,
                                                        const int32_t* d_cell_start, const int32_t* d_cell_items,
                                                        const float* scale_factors, int nlevels, uint8_t* d_occupied,
                                                        const int32_t* d_nq, int qcap, const uint8_t* d_q_valid,
                                                        const float* d_q_uv, const int32_t* d_q_octave, const float* d_q_angle,
                                                        const uint8_t* d_q_desc, const uint8_t* d_q_hasobs, float th, int mode,
                                                        int check_ori, int32_t* d_assigned, int32_t* d_nmatches, void* stream) {
  return launch_proj_points(1, d_kps_un, d_desc, d_n, cap, pairs, gp, d_cell_start, d_cell_items, scale_factors, nlevels,
                            d_occupied, d_nq, qcap, d_q_valid, d_q_uv, d_q_octave, d_q_angle, d_q_desc, d_q_hasobs, th, 0.f, mode,
                            check_ori, d_assigned, d_nmatches, stream, "plh_orb_search_by_projection_frame_batch_dev");
}

// ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, th, ORBdist)
// (ORBmatcher.cc:1587-1716, relocalisation): the last-frame form with the caller's distance threshold; queries are the
// KeyFrame's map points (valid = pMP && !isBad() && !sAlreadyFound.count(pMP) && depth inside the scale pyramid,
// level = PredictScale, angle = pKF->mvKeysUn[i].angle), occupied = CurrentFrame.mvpMapPoints[i2] != NULL, hasobs = 1.
plh_status plh_orb_search_by_projection_kf_batch_dev(const plh_keypoint* d_kps_un, const uint8_t* d_desc, const int32_t* d_n,
                                                     int cap, int pairs, const plh_grid_params* gp, const int32_t* d_cell_start,
                                                     const int32_t* d_cell_items, const float* scale_factors, int nlevels,
                                                     uint8_t* d_occupied, const int32_t* d_nq, int qcap, const uint8_t* d_q_valid,
                                                     const float* d_q_uv, const int32_t* d_q_level, const float* d_q_angle,
                                                     const uint8_t* d_q_desc, const uint8_t* d_q_hasobs, float th, int orb_dist,
                                                     int check_ori, int32_t* d_assigned, int32_t* d_nmatches, void* stream) {
  return launch_proj_points(1, d_kps_un, d_desc, d_n, cap, pairs, gp, d_cell_start, d_cell_items, scale_factors, nlevels,
                            d_occupied, d_nq, qcap, d_q_valid, d_q_uv, d_q_level, d_q_angle, d_q_desc, d_q_hasobs, th, 0.f, 0,
                            check_ori, d_assigned, d_nmatches, stream, "plh_orb_search_by_projection_kf_batch_dev", orb_dist);
}

// The search inside ORBmatcher::Fuse(pKF, vpMapPoints, th) and Fuse(pKF, Scw, vpPoints, th, vpReplacePoint)
// (ORBmatcher.cc:914-1061, 1063-1197): per query the best keypoint of the KeyFrame (Hamming <= TH_LOW) among those of
// level l-1..l inside the window whose reprojection error passes the chi-square gate.  The MapPoint replace / add logic
// that follows mutates the map and stays with the caller.
plh_status plh_orb_fuse_search_batch_dev(const plh_keypoint* d_kps_un, const uint8_t* d_desc, const int32_t* d_n, int cap, int pairs,
                                         const plh_grid_params* gp, const int32_t* d_cell_start, const int32_t* d_cell_items,
                                         const float* scale_factors, const float* inv_level_sigma2, int nlevels,
                                         const int32_t* d_nq, int qcap, const uint8_t* d_q_valid, const float* d_q_uv,
                                         const int32_t* d_q_level, const uint8_t* d_q_desc, float th, int th_low,
                                         int32_t* d_best_idx, int32_t* d_nfound, void* stream) {
  if (!inv_level_sigma2) return PLH_ERR_INVALID;
  // occupancy / hasobs / aux are unused by this variant: any valid device pointers do
  return launch_proj_points(2, d_kps_un, d_desc, d_n, cap, pairs, gp, d_cell_start, d_cell_items, scale_factors, nlevels,
                            const_cast<uint8_t*>(d_q_valid), d_nq, qcap, d_q_valid, d_q_uv, d_q_level, d_q_uv, d_q_desc, d_q_valid, th,
                            0.f, 0, 0, d_best_idx, d_nfound, stream, "plh_orb_fuse_search_batch_dev", th_low, inv_level_sigma2);
}

// ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:329-453, loop closing).
// Query iMP: valid = !isBad() && !spAlreadyFound.count(pMP) && depth > 0 && IsInImage && distance range && viewing angle;
// occupied = vpMatched[idx] != NULL (in/out), hasobs all 1.
plh_status plh_orb_search_by_projection_sim3_batch_dev(const plh_keypoint* d_kps_un, const uint8_t* d_desc, const int32_t* d_n,
                                                       int cap, int pairs, const plh_grid_params* gp, const int32_t* d_cell_start,
                                                       const int32_t* d_cell_items, const float* scale_factors, int nlevels,
                                                       uint8_t* d_occupied, const int32_t* d_nq, int qcap,
                                                       const uint8_t* d_q_valid, const float* d_q_uv, const int32_t* d_q_level,
                                                       const uint8_t* d_q_desc, const uint8_t* d_q_hasobs, float th, int th_low,
                                                       int32_t* d_assigned, int32_t* d_nmatches, void* stream) {
  return launch_proj_points(3, d_kps_un, d_desc, d_n, cap, pairs, gp, d_cell_start, d_cell_items, scale_factors, nlevels,
                            d_occupied, d_nq, qcap, d_q_valid, d_q_uv, d_q_level, d_q_uv, d_q_desc, d_q_hasobs, th, 0.f, 0, 0,
                            d_assigned, d_nmatches, stream, "plh_orb_search_by_projection_sim3_batch_dev", th_low);
}

// ORBmatcher::SearchBySim3 (ORBmatcher.cc:1199-1439): the two one-way searches are the per-query best-candidate search
// (variant 2 with the chi-square gate open and TH_HIGH), the agreement check (:1396-1412) is k_sim3_agree.
plh_status plh_orb_search_by_sim3_batch_dev(const plh_keypoint* d_kps1_un, const uint8_t* d_desc1, const int32_t* d_n1,
                                            const int32_t* d_cell_start1, const int32_t* d_cell_items1,
                                            const plh_keypoint* d_kps2_un, const uint8_t* d_desc2, const int32_t* d_n2,
                                            const int32_t* d_cell_start2, const int32_t* d_cell_items2, int cap, int pairs,
                                            const plh_grid_params* gp, const float* scale_factors, int nlevels,
                                            const uint8_t* d_q12_valid, const float* d_q12_uv, const int32_t* d_q12_level,
                                            const uint8_t* d_q12_desc, const uint8_t* d_q21_valid, const float* d_q21_uv,
                                            const int32_t* d_q21_level, const uint8_t* d_q21_desc, float th, int th_high,
                                            int32_t* d_match1, int32_t* d_match2, int32_t* d_match12, int32_t* d_nfound,
                                            void* stream) {
  static const char* who = "plh_orb_search_by_sim3_batch_dev";
  if (!d_match1 || !d_match2 || !d_match12 || !d_nfound || !d_q12_valid || !d_q21_valid || !d_n1 || !d_n2) {
    set_error("%s: invalid argument", who);
    return PLH_ERR_INVALID;
  }
  // KeyFrame 1's map points searched in KeyFrame 2 (one query per keypoint slot of KeyFrame 1), then the reverse;
  // d_nfound doubles as the per-direction counter until the agreement kernel overwrites it
  plh_status st = launch_proj_points(2, d_kps2_un, d_desc2, d_n2, cap, pairs, gp, d_cell_start2, d_cell_items2, scale_factors, nlevels,
                                     const_cast<uint8_t*>(d_q12_valid), d_n1, cap, d_q12_valid, d_q12_uv, d_q12_level, d_q12_uv,
                                     d_q12_desc, d_q12_valid, th, 0.f, 0, 0, d_match1, d_nfound, stream, who, th_high, nullptr);
  if (st != PLH_OK) return st;
  st = launch_proj_points(2, d_kps1_un, d_desc1, d_n1, cap, pairs, gp, d_cell_start1, d_cell_items1, scale_factors, nlevels,
                          const_cast<uint8_t*>(d_q21_valid), d_n2, cap, d_q21_valid, d_q21_uv, d_q21_level, d_q21_uv, d_q21_desc,
                          d_q21_valid, th, 0.f, 0, 0, d_match2, d_nfound, stream, who, th_high, nullptr);
  if (st != PLH_OK) return st;
  hipLaunchKernelGGL(k_sim3_agree, dim3(pairs), dim3(64), 0, (hipStream_t)stream, (const int*)d_n1, (const int*)d_n2, cap,
                     (const int32_t*)d_match1, (const int32_t*)d_match2, d_match12, d_nfound);
  PLH_LAUNCH_CHECK();
  return PLH_OK;
}

plh_status plh_line_fuse_search_batch_dev(const plh_keyline* d_kl, const uint8_t* d_cand_desc, const int32_t* d_nl, int cap, int pairs,
                                          const float* scale_factors_line, int nlevels, const int32_t* d_nq, int qcap,
                                          const uint8_t* d_q_valid, const float* d_q_seg, const int32_t* d_q_level,
                                          const uint8_t* d_q_desc, float th, float cos_th, int th_low, int32_t* d_best_idx,
                                          int32_t* d_nfound, void* stream) {
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  ScaleTab sfl;
  if (!d_kl || !d_cand_desc || !d_nl || !scale_tab(scale_factors_line, nlevels, &sfl) || !d_nq || !d_q_valid || !d_q_seg ||
      !d_q_level || !d_q_desc || !d_best_idx || !d_nfound || cap <= 0 || cap > 65535 || qcap <= 0 || pairs <= 0) {
    set_error("plh_line_fuse_search_batch_dev: invalid argument");
    return PLH_ERR_INVALID;
  }
  hipLaunchKernelGGL(k_line_fuse_search, dim3(pairs), dim3(64), 0, (hipStream_t)stream, d_kl, d_cand_desc, (const int*)d_nl, cap, sfl,
                     nlevels, (const int*)d_nq, qcap, d_q_valid, d_q_seg, d_q_level, d_q_desc, th, cos_th, th_low, d_best_idx, d_nfound);
  PLH_LAUNCH_CHECK();
  return PLH_OK;
}

static plh_status launch_proj_lines(int variant, const plh_keyline* d_kl, const uint8_t* d_ldesc, const double* d_linefn,
                                    const int32_t* d_nl, int cap, int pairs, const plh_grid_params* gp, const int32_t* d_cs,
                                    const int32_t* d_ci, int item_cap, uint8_t* d_occupied, const int32_t* d_nq, int qcap,
                                    const uint8_t* d_q_valid, const float* d_q_seg, const float* d_q_aux, const uint8_t* d_q_desc,
                                    const uint8_t* d_q_hasobs, float th, float nnratio, int32_t* d_assigned, int32_t* d_nmatches,
                                    void* stream, const char* who) {
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  if (!d_kl || !d_ldesc || !d_linefn || !d_nl || !gp || !d_cs || !d_ci || !d_occupied || !d_nq || !d_q_valid || !d_q_seg ||
      !d_q_aux || !d_q_desc || !d_q_hasobs || !d_assigned || !d_nmatches || cap <= 0 || cap > 8000 || qcap <= 0 || pairs <= 0 ||
      item_cap < cap) {
    set_error("%s: invalid argument (cap in 1..8000)", who);
    return PLH_ERR_INVALID;
  }
  if (!g_proj_serial) {
    ProjTop* tops = nullptr;
    PLH_HIP(hipMallocAsync((void**)&tops, (size_t)pairs * qcap * sizeof(ProjTop), (hipStream_t)stream));
    if ((long long)pairs * qcap <= PROJ_G8_MAX)
      hipLaunchKernelGGL(k_proj_lines_prepass_g8, dim3((qcap * PROJ_LG + 255) / 256, pairs), dim3(256), 0, (hipStream_t)stream, variant, d_kl, d_ldesc,
                         d_linefn, (const int*)d_nl, cap, *gp, d_cs, d_ci, item_cap, (const uint8_t*)d_occupied, (const int*)d_nq, qcap, d_q_valid,
                         d_q_seg, d_q_aux, d_q_desc, th, tops, g_proj_keep);
    else
      hipLaunchKernelGGL(k_proj_lines_prepass, dim3((qcap + 255) / 256, pairs), dim3(256), 0, (hipStream_t)stream, variant, d_kl, d_ldesc, d_linefn,
                         (const int*)d_nl, cap, *gp, d_cs, d_ci, item_cap, (const uint8_t*)d_occupied, (const int*)d_nq, qcap, d_q_valid, d_q_seg,
                         d_q_aux, d_q_desc, th, tops, g_proj_keep);
    const size_t lds2 = (size_t)cap * (3 * 4 + 2) + 128 + 64;
    ScaleTab none;
    for (int i = 0; i < 16; i++) none.v[i] = 0.f;
    ProjFrame F{nullptr, d_ldesc, d_kl, d_linefn, item_cap, d_cs, d_ci};
    plh_status st = lds_request(k_proj_resolve, lds2, who);
    if (st == PLH_OK) {
      hipLaunchKernelGGL(k_proj_resolve, dim3(pairs), dim3(64), lds2, (hipStream_t)stream, 1, variant, F, (const int*)d_nl, cap, *gp, none, none,
                         d_occupied, (const int*)d_nq, qcap, d_q_valid, d_q_seg, (const int32_t*)nullptr, d_q_aux, d_q_desc, d_q_hasobs, th, nnratio,
                         0, 0, 80, 16, 0, (const ProjTop*)tops, d_assigned, d_nmatches);
    }
    const hipError_t le = hipGetLastError();
    (void)hipFreeAsync(tops, (hipStream_t)stream);
    if (st != PLH_OK) return st;
    if (le != hipSuccess) { set_error("%s: kernel launch -> %s", who, hipGetErrorString(le)); return PLH_ERR_HIP; }
    return PLH_OK;
  }
  const size_t lds = (size_t)cap * (3 * 4 + 2) + 64;
  if (lds_request(k_search_proj_lines, lds, who) != PLH_OK) return PLH_ERR_INVALID;
  hipLaunchKernelGGL(k_search_proj_lines, dim3(pairs), dim3(64), lds, (hipStream_t)stream, variant, d_kl, d_ldesc, d_linefn,
                     (const int*)d_nl, cap, *gp, d_cs, d_ci, item_cap, d_occupied, (const int*)d_nq, qcap, d_q_valid, d_q_seg,
                     d_q_aux, d_q_desc, d_q_hasobs, th, nnratio, d_assigned, d_nmatches);
  PLH_LAUNCH_CHECK();
  return PLH_OK;
}

plh_status plh_line_search_by_projection_frame_batch_dev(const plh_keyline* d_kl, const uint8_t* d_ldesc, const double* d_linefn,
                                                         const int32_t* d_nl, int cap, int pairs, const plh_grid_params* gp,
                                                         const int32_t* d_cell_start, const int32_t* d_cell_items, int item_cap,
                                                         uint8_t* d_occupied, const int32_t* d_nq, int qcap,
                                                         const uint8_t* d_q_valid, const float* d_q_seg, const float* d_q_length,
                                                         const uint8_t* d_q_desc, const uint8_t* d_q_hasobs, float th,
                                                         int32_t* d_assigned, int32_t* d_nmatches, void* stream) {
  return launch_proj_lines(1, d_kl, d_ldesc, d_linefn, d_nl, cap, pairs, gp, d_cell_start, d_cell_items, item_cap, d_occupied, d_nq,
                           qcap, d_q_valid, d_q_seg, d_q_length, d_q_desc, d_q_hasobs, th, 0.f, d_assigned, d_nmatches, stream,
                           "plh_line_search_by_projection_frame_batch_dev");
}

plh_status plh_line_search_by_projection_ml_batch_dev(const plh_keyline* d_kl, const uint8_t* d_ldesc, const double* d_linefn,
                                                      const int32_t* d_nl, int cap, int pairs, const plh_grid_params* gp,
                                                      const int32_t* d_cell_start, const int32_t* d_cell_items, int item_cap,
                                                      uint8_t* d_occupied, const int32_t* d_nq, int qcap, const uint8_t* d_q_valid,
                                                      const float* d_q_seg, const float* d_q_viewcos, const uint8_t* d_q_desc,
                                                      const uint8_t* d_q_hasobs, float th, float nnratio, int32_t* d_assigned,
                                                      int32_t* d_nmatches, void* stream) {
  return launch_proj_lines(0, d_kl, d_ldesc, d_linefn, d_nl, cap, pairs, gp, d_cell_start, d_cell_items, item_cap, d_occupied, d_nq,
                           qcap, d_q_valid, d_q_seg, d_q_viewcos, d_q_desc, d_q_hasobs, th, nnratio, d_assigned, d_nmatches, stream,
                           "plh_line_search_by_projection_ml_batch_dev");
}

// ---- host-buffer conveniences: one call = one reference call on ONE frame (stage over PCIe, block until done).
// The frame's grid is rebuilt on the device inside the call (two tiny kernels); the reference builds it once in the
// Frame constructor, a caller that keeps frames resident uses the *_batch_dev entry points instead.
extern "C++" {
namespace {
// Staging of a call's host arrays through the calling thread's arena (plh_stage.h): every array is packed into the pinned mirror,
// ONE copy takes them up, the kernels run on the thread's own stream, one copy brings the results back.  (Rounds 1-5: a hipMalloc
// and a blocking copy per array, the null stream, a device-wide synchronisation, a hipFree per array.)
struct Stage {
  Stager st;
  plh_status rc;
  explicit Stage(int device) : rc(st.begin(device)) {}
  template <typename T> T* up(const T* host, size_t count, size_t alloc_count = 0) {
    if (rc != PLH_OK) return nullptr;
    return (count && host) ? st.in(host, count, alloc_count) : st.scratch_zero<T>(std::max(std::max(count, alloc_count), (size_t)1));
  }
  template <typename T> T* alloc(size_t count) { return rc == PLH_OK ? st.scratch_zero<T>(std::max(count, (size_t)1)) : nullptr; }
  void fetch_bytes(void* dst, const void* d, size_t bytes) { if (bytes) st.fetch(static_cast<uint8_t*>(dst), static_cast<const uint8_t*>(d), bytes); }
  hipStream_t stream() const { return st.stream(); }
  plh_status ready() { return rc != PLH_OK ? rc : st.upload(); }
  plh_status finish() { return st.download(); }
};
#define STAGE_OK(s)                          \
  do {                                       \
    const plh_status rc__ = (s).ready();     \
    if (rc__ != PLH_OK) return rc__;         \
  } while (0)
}  // namespace
}  // extern "C++"

plh_status plh_orb_search_for_initialization(const plh_keypoint* kps1, const uint8_t* desc1, int n1, const plh_keypoint* kps2,
                                             const uint8_t* desc2, int n2, const plh_grid_params* gp2, float* prev_matched,
                                             int window_size, float nnratio, int check_ori, int32_t* matches12, int* nmatches,
                                             int device) {
  if (n1 < 0 || n2 < 0 || !nmatches || !gp2 || (n1 > 0 && (!kps1 || !desc1 || !prev_matched || !matches12)) ||
      (n2 > 0 && (!kps2 || !desc2)))
    return PLH_ERR_INVALID;
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  *nmatches = 0;
  if (n1 == 0 || n2 == 0) return PLH_OK;
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  const int cap = std::max(n1, n2);
  Stage s(device);
  plh_keypoint* dk1 = s.up(kps1, n1, cap); uint8_t* dd1 = s.up(desc1, (size_t)n1 * 32, (size_t)cap * 32);
  plh_keypoint* dk2 = s.up(kps2, n2, cap); uint8_t* dd2 = s.up(desc2, (size_t)n2 * 32, (size_t)cap * 32);
  const int32_t ns[2] = {n1, n2};
  int32_t* dn = s.up(ns, 2);
  float* dpm = s.up(prev_matched, (size_t)n1 * 2, (size_t)cap * 2);
  int32_t* dcs = s.alloc<int32_t>(PLH_GRID_CELLS + 1); int32_t* dci = s.alloc<int32_t>(cap);
  int32_t* dm = s.alloc<int32_t>(cap); int32_t* dc = s.alloc<int32_t>(1);
  STAGE_OK(s);
  plh_status st = plh_frame_assign_grid_batch_dev(dk2, dn + 1, cap, 1, gp2, dcs, dci, s.stream());
  if (st == PLH_OK)
    st = plh_orb_search_for_initialization_batch_dev(dk1, dd1, dn, dk2, dd2, dn + 1, cap, 1, gp2, dcs, dci, dpm, window_size, nnratio,
                                                     check_ori, dm, dc, s.stream());
  if (st != PLH_OK) return st;
  s.fetch_bytes(matches12, dm, (size_t)n1 * 4);
  s.fetch_bytes(prev_matched, dpm, (size_t)n1 * 8);
  s.fetch_bytes(nmatches, dc, 4);
  return s.finish();
}

static plh_status host_proj_points(int variant, const plh_keypoint* kps_un, const uint8_t* desc, int n, const plh_grid_params* gp,
                                   const float* scale_factors, int nlevels, uint8_t* occupied, int nq, const uint8_t* q_valid,
                                   const float* q_xy, const int32_t* q_level, const float* q_aux, const uint8_t* q_desc,
                                   const uint8_t* q_hasobs, float th, float nnratio, int mode, int check_ori, int32_t* assigned,
                                   int* nmatches, int device) {
  if (n < 0 || nq < 0 || !nmatches || !gp || !scale_factors || (n > 0 && (!kps_un || !desc || !occupied || !assigned)) ||
      (nq > 0 && (!q_valid || !q_xy || !q_level || !q_aux || !q_desc || !q_hasobs)))
    return PLH_ERR_INVALID;
  for (int i = 0; i < n; i++) assigned[i] = -1;
  *nmatches = 0;
  if (n == 0 || nq == 0) return PLH_OK;
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  Stage s(device);
  plh_keypoint* dk = s.up(kps_un, n); uint8_t* dd = s.up(desc, (size_t)n * 32); uint8_t* docc = s.up(occupied, n);
  const int32_t ns[2] = {n, nq};
  int32_t* dn = s.up(ns, 2);
  uint8_t* qv = s.up(q_valid, nq); float* qx = s.up(q_xy, (size_t)nq * 2); int32_t* ql = s.up(q_level, nq);
  float* qa = s.up(q_aux, nq); uint8_t* qd = s.up(q_desc, (size_t)nq * 32); uint8_t* qh = s.up(q_hasobs, nq);
  int32_t* dcs = s.alloc<int32_t>(PLH_GRID_CELLS + 1); int32_t* dci = s.alloc<int32_t>(n);
  int32_t* da = s.alloc<int32_t>(n); int32_t* dc = s.alloc<int32_t>(1);
  STAGE_OK(s);
  plh_status st = plh_frame_assign_grid_batch_dev(dk, dn, n, 1, gp, dcs, dci, s.stream());
  if (st == PLH_OK)
    st = launch_proj_points(variant, dk, dd, dn, n, 1, gp, dcs, dci, scale_factors, nlevels, docc, dn + 1, nq, qv, qx, ql, qa, qd, qh,
                            th, nnratio, mode, check_ori, da, dc, s.stream(), "plh_orb_search_by_projection");
  if (st != PLH_OK) return st;
  s.fetch_bytes(assigned, da, (size_t)n * 4);
  s.fetch_bytes(occupied, docc, (size_t)n);
  s.fetch_bytes(nmatches, dc, 4);
  return s.finish();
}

plh_status plh_orb_search_by_projection_mp(const plh_keypoint* kps_un, const uint8_t* desc, int n, const plh_grid_params* gp,
                                           const float* scale_factors, int nlevels, uint8_t* occupied, int nq,
                                           const uint8_t* q_valid, const float* q_xy, const int32_t* q_level,
                                           const float* q_viewcos, const uint8_t* q_desc, const uint8_t* q_hasobs, float th,
                                           float nnratio, int32_t* assigned, int* nmatches, int device) {
  return host_proj_points(0, kps_un, desc, n, gp, scale_factors, nlevels, occupied, nq, q_valid, q_xy, q_level, q_viewcos, q_desc,
                          q_hasobs, th, nnratio, 0, 0, assigned, nmatches, device);
}

plh_status plh_orb_search_by_projection_frame(const plh_keypoint* kps_un, const uint8_t* desc, int n, const plh_grid_params* gp,
                                              const float* scale_factors, int nlevels, uint8_t* occupied, int nq,
                                              const uint8_t* q_valid, const float* q_uv, const int32_t* q_octave,
                                              const float* q_angle, const uint8_t* q_desc, const uint8_t* q_hasobs, float th,
                                              int mode, int check_ori, int32_t* assigned, int* nmatches, int device) {
  if (mode < 0 || mode > 2) return PLH_ERR_INVALID;
  return host_proj_points(1, kps_un, desc, n, gp, scale_factors, nlevels, occupied, nq, q_valid, q_uv, q_octave, q_angle, q_desc,
                          q_hasobs, th, 0.f, mode, check_ori, assigned, nmatches, device);
}

// ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (ORBmatcher.cc:1199-1439) on two KeyFrames, host buffers: see
// plh_orb_search_by_sim3_batch_dev.  q12_* has n1 rows (KeyFrame 1's points seen from KeyFrame 2), q21_* n2 rows.
plh_status plh_orb_search_by_sim3(const plh_keypoint* kps1_un, const uint8_t* desc1, int n1, const plh_keypoint* kps2_un,
                                  const uint8_t* desc2, int n2, const plh_grid_params* gp, const float* scale_factors, int nlevels,
                                  const uint8_t* q12_valid, const float* q12_uv, const int32_t* q12_level, const uint8_t* q12_desc,
                                  const uint8_t* q21_valid, const float* q21_uv, const int32_t* q21_level, const uint8_t* q21_desc,
                                  float th, int th_high, int32_t* match12, int* nfound, int device) {
  if (n1 < 0 || n2 < 0 || !nfound || !gp || !scale_factors || (n1 > 0 && (!kps1_un || !desc1 || !match12 || !q12_valid || !q12_uv ||
      !q12_level || !q12_desc)) || (n2 > 0 && (!kps2_un || !desc2 || !q21_valid || !q21_uv || !q21_level || !q21_desc)))
    return PLH_ERR_INVALID;
  for (int i = 0; i < n1; i++) match12[i] = -1;
  *nfound = 0;
  if (n1 == 0 || n2 == 0) return PLH_OK;
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  const size_t cap = (size_t)std::max(n1, n2);
  Stage s(device);
  plh_keypoint* k1 = s.up(kps1_un, n1, cap); uint8_t* d1 = s.up(desc1, (size_t)n1 * 32, cap * 32);
  plh_keypoint* k2 = s.up(kps2_un, n2, cap); uint8_t* d2 = s.up(desc2, (size_t)n2 * 32, cap * 32);
  const int32_t ns[2] = {n1, n2};
  int32_t* dn = s.up(ns, 2);
  uint8_t* v12 = s.up(q12_valid, n1, cap); float* u12 = s.up(q12_uv, (size_t)n1 * 2, cap * 2); int32_t* l12 = s.up(q12_level, n1, cap);
  uint8_t* e12 = s.up(q12_desc, (size_t)n1 * 32, cap * 32);
  uint8_t* v21 = s.up(q21_valid, n2, cap); float* u21 = s.up(q21_uv, (size_t)n2 * 2, cap * 2); int32_t* l21 = s.up(q21_level, n2, cap);
  uint8_t* e21 = s.up(q21_desc, (size_t)n2 * 32, cap * 32);
  int32_t* cs1 = s.alloc<int32_t>(PLH_GRID_CELLS + 1); int32_t* ci1 = s.alloc<int32_t>(cap);
  int32_t* cs2 = s.alloc<int32_t>(PLH_GRID_CELLS + 1); int32_t* ci2 = s.alloc<int32_t>(cap);
  int32_t* m1 = s.alloc<int32_t>(cap); int32_t* m2 = s.alloc<int32_t>(cap); int32_t* m12 = s.alloc<int32_t>(cap);
  int32_t* dc = s.alloc<int32_t>(1);
  STAGE_OK(s);
  plh_status st = plh_frame_assign_grid_batch_dev(k1, dn, (int)cap, 1, gp, cs1, ci1, s.stream());
  if (st == PLH_OK) st = plh_frame_assign_grid_batch_dev(k2, dn + 1, (int)cap, 1, gp, cs2, ci2, s.stream());
  if (st == PLH_OK)
    st = plh_orb_search_by_sim3_batch_dev(k1, d1, dn, cs1, ci1, k2, d2, dn + 1, cs2, ci2, (int)cap, 1, gp, scale_factors, nlevels, v12, u12,
                                          l12, e12, v21, u21, l21, e21, th, th_high, m1, m2, m12, dc, s.stream());
  if (st != PLH_OK) return st;
  s.fetch_bytes(match12, m12, (size_t)n1 * 4);
  s.fetch_bytes(nfound, dc, 4);
  return s.finish();
}

// ORBmatcher::SearchByProjection(Frame& Cur, KeyFrame* pKF, sAlreadyFound, th, ORBdist) (ORBmatcher.cc:1587-1716, Tracking::Relocalization)
// on one frame, host buffers: see plh_orb_search_by_projection_kf_batch_dev.
plh_status plh_orb_search_by_projection_kf(const plh_keypoint* kps_un, const uint8_t* desc, int n, const plh_grid_params* gp,
                                           const float* scale_factors, int nlevels, uint8_t* occupied, int nq, const uint8_t* q_valid,
                                           const float* q_uv, const int32_t* q_level, const float* q_angle, const uint8_t* q_desc,
                                           float th, int orb_dist, int check_ori, int32_t* assigned, int* nmatches, int device) {
  if (n < 0 || nq < 0 || !nmatches || !gp || !scale_factors || (n > 0 && (!kps_un || !desc || !occupied || !assigned)) ||
      (nq > 0 && (!q_valid || !q_uv || !q_level || !q_angle || !q_desc)))
    return PLH_ERR_INVALID;
  for (int i = 0; i < n; i++) assigned[i] = -1;
  *nmatches = 0;
  if (n == 0 || nq == 0) return PLH_OK;
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  Stage s(device);
  plh_keypoint* dk = s.up(kps_un, n); uint8_t* dd = s.up(desc, (size_t)n * 32); uint8_t* docc = s.up(occupied, n);
  const int32_t ns[2] = {n, nq};
  int32_t* dn = s.up(ns, 2);
  std::vector<uint8_t> ones((size_t)nq, 1);
  uint8_t* qv = s.up(q_valid, nq); float* qx = s.up(q_uv, (size_t)nq * 2); int32_t* ql = s.up(q_level, nq);
  float* qa = s.up(q_angle, nq); uint8_t* qd = s.up(q_desc, (size_t)nq * 32); uint8_t* qh = s.up(ones.data(), nq);
  int32_t* dcs = s.alloc<int32_t>(PLH_GRID_CELLS + 1); int32_t* dci = s.alloc<int32_t>(n);
  int32_t* da = s.alloc<int32_t>(n); int32_t* dc = s.alloc<int32_t>(1);
  STAGE_OK(s);
  plh_status st = plh_frame_assign_grid_batch_dev(dk, dn, n, 1, gp, dcs, dci, s.stream());
  if (st == PLH_OK)
    st = plh_orb_search_by_projection_kf_batch_dev(dk, dd, dn, n, 1, gp, dcs, dci, scale_factors, nlevels, docc, dn + 1, nq, qv, qx, ql, qa,
                                                   qd, qh, th, orb_dist, check_ori, da, dc, s.stream());
  if (st != PLH_OK) return st;
  s.fetch_bytes(assigned, da, (size_t)n * 4);
  s.fetch_bytes(occupied, docc, (size_t)n);
  s.fetch_bytes(nmatches, dc, 4);
  return s.finish();
}

// ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:329-453) on one KeyFrame, host
// buffers: see plh_orb_search_by_projection_sim3_batch_dev.
plh_status plh_orb_search_by_projection_sim3(const plh_keypoint* kps_un, const uint8_t* desc, int n, const plh_grid_params* gp,
                                             const float* scale_factors, int nlevels, uint8_t* occupied, int nq,
                                             const uint8_t* q_valid, const float* q_uv, const int32_t* q_level, const uint8_t* q_desc,
                                             const uint8_t* q_hasobs, float th, int th_low, int32_t* assigned, int* nmatches,
                                             int device) {
  if (n < 0 || nq < 0 || !nmatches || !gp || !scale_factors || (n > 0 && (!kps_un || !desc || !occupied || !assigned)) ||
      (nq > 0 && (!q_valid || !q_uv || !q_level || !q_desc || !q_hasobs)))
    return PLH_ERR_INVALID;
  for (int i = 0; i < n; i++) assigned[i] = -1;
  *nmatches = 0;
  if (n == 0 || nq == 0) return PLH_OK;
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  Stage s(device);
  plh_keypoint* dk = s.up(kps_un, n); uint8_t* dd = s.up(desc, (size_t)n * 32); uint8_t* docc = s.up(occupied, n);
  const int32_t ns[2] = {n, nq};
  int32_t* dn = s.up(ns, 2);
  uint8_t* qv = s.up(q_valid, nq); float* qx = s.up(q_uv, (size_t)nq * 2); int32_t* ql = s.up(q_level, nq);
  uint8_t* qd = s.up(q_desc, (size_t)nq * 32); uint8_t* qh = s.up(q_hasobs, nq);
  int32_t* dcs = s.alloc<int32_t>(PLH_GRID_CELLS + 1); int32_t* dci = s.alloc<int32_t>(n);
  int32_t* da = s.alloc<int32_t>(n); int32_t* dc = s.alloc<int32_t>(1);
  STAGE_OK(s);
  plh_status st = plh_frame_assign_grid_batch_dev(dk, dn, n, 1, gp, dcs, dci, s.stream());
  if (st == PLH_OK)
    st = plh_orb_search_by_projection_sim3_batch_dev(dk, dd, dn, n, 1, gp, dcs, dci, scale_factors, nlevels, docc, dn + 1, nq, qv, qx, ql, qd,
                                                     qh, th, th_low, da, dc, s.stream());
  if (st != PLH_OK) return st;
  s.fetch_bytes(assigned, da, (size_t)n * 4);
  s.fetch_bytes(occupied, docc, (size_t)n);
  s.fetch_bytes(nmatches, dc, 4);
  return s.finish();
}

// The search inside ORBmatcher::Fuse(pKF, vpMapPoints, th) / Fuse(pKF, Scw, ...) (ORBmatcher.cc:914-1197) on one KeyFrame, host
// buffers: see plh_orb_fuse_search_batch_dev.  best_idx[nq] = keypoint the query settles on, or -1.
plh_status plh_orb_fuse_search(const plh_keypoint* kps_un, const uint8_t* desc, int n, const plh_grid_params* gp,
                               const float* scale_factors, const float* inv_level_sigma2, int nlevels, int nq, const uint8_t* q_valid,
                               const float* q_uv, const int32_t* q_level, const uint8_t* q_desc, float th, int th_low,
                               int32_t* best_idx, int* nfound, int device) {
  if (n < 0 || nq < 0 || !nfound || !gp || !scale_factors || (n > 0 && (!kps_un || !desc)) ||
      (nq > 0 && (!q_valid || !q_uv || !q_level || !q_desc || !best_idx)))
    return PLH_ERR_INVALID;
  for (int i = 0; i < nq; i++) best_idx[i] = -1;
  *nfound = 0;
  if (n == 0 || nq == 0) return PLH_OK;
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  Stage s(device);
  plh_keypoint* dk = s.up(kps_un, n); uint8_t* dd = s.up(desc, (size_t)n * 32);
  const int32_t ns[2] = {n, nq};
  int32_t* dn = s.up(ns, 2);
  uint8_t* qv = s.up(q_valid, nq); float* qx = s.up(q_uv, (size_t)nq * 2); int32_t* ql = s.up(q_level, nq);
  uint8_t* qd = s.up(q_desc, (size_t)nq * 32);
  int32_t* dcs = s.alloc<int32_t>(PLH_GRID_CELLS + 1); int32_t* dci = s.alloc<int32_t>(n);
  int32_t* db = s.alloc<int32_t>(nq); int32_t* dc = s.alloc<int32_t>(1);
  STAGE_OK(s);
  plh_status st = plh_frame_assign_grid_batch_dev(dk, dn, n, 1, gp, dcs, dci, s.stream());
  const float noGate[16] = {0};   // inv_level_sigma2 == NULL: no chi-square gate (the Sim3 overload of Fuse): e2 * 0 never exceeds it
  if (st == PLH_OK)
    st = plh_orb_fuse_search_batch_dev(dk, dd, dn, n, 1, gp, dcs, dci, scale_factors, inv_level_sigma2 ? inv_level_sigma2 : noGate, nlevels,
                                       dn + 1, nq, qv, qx, ql, qd, th, th_low, db, dc, s.stream());
  if (st != PLH_OK) return st;
  s.fetch_bytes(best_idx, db, (size_t)nq * 4);
  s.fetch_bytes(nfound, dc, 4);
  return s.finish();
}

static plh_status host_proj_lines(int variant, const plh_keyline* kl, const uint8_t* ldesc, const double* linefn, int nl,
                                  const plh_grid_params* gp, uint8_t* occupied, int nq, const uint8_t* q_valid, const float* q_seg,
                                  const float* q_aux, const uint8_t* q_desc, const uint8_t* q_hasobs, float th, float nnratio,
                                  int32_t* assigned, int* nmatches, int device) {
  if (nl < 0 || nq < 0 || !nmatches || !gp || (nl > 0 && (!kl || !ldesc || !linefn || !occupied || !assigned)) ||
      (nq > 0 && (!q_valid || !q_seg || !q_aux || !q_desc || !q_hasobs)))
    return PLH_ERR_INVALID;
  for (int i = 0; i < nl; i++) assigned[i] = -1;
  *nmatches = 0;
  if (nl == 0 || nq == 0) return PLH_OK;
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  Stage s(device);
  const int itemCap = nl * PLH_GRID_COLS;
  plh_keyline* dk = s.up(kl, nl); uint8_t* dd = s.up(ldesc, (size_t)nl * 32); double* dfn = s.up(linefn, (size_t)nl * 3);
  uint8_t* docc = s.up(occupied, nl);
  const int32_t ns[2] = {nl, nq};
  int32_t* dn = s.up(ns, 2);
  uint8_t* qv = s.up(q_valid, nq); float* qs = s.up(q_seg, (size_t)nq * 4); float* qa = s.up(q_aux, nq);
  uint8_t* qd = s.up(q_desc, (size_t)nq * 32); uint8_t* qh = s.up(q_hasobs, nq);
  int32_t* dcs = s.alloc<int32_t>(PLH_GRID_CELLS + 1); int32_t* dci = s.alloc<int32_t>(itemCap);
  int32_t* da = s.alloc<int32_t>(nl); int32_t* dc = s.alloc<int32_t>(1);
  STAGE_OK(s);
  plh_status st = plh_frame_assign_grid_lines_batch_dev(dk, dn, nl, 1, gp, dcs, dci, itemCap, s.stream());
  if (st == PLH_OK)
    st = launch_proj_lines(variant, dk, dd, dfn, dn, nl, 1, gp, dcs, dci, itemCap, docc, dn + 1, nq, qv, qs, qa, qd, qh, th, nnratio, da,
                           dc, s.stream(), "plh_line_search_by_projection");
  if (st != PLH_OK) return st;
  s.fetch_bytes(assigned, da, (size_t)nl * 4);
  s.fetch_bytes(occupied, docc, (size_t)nl);
  s.fetch_bytes(nmatches, dc, 4);
  return s.finish();
}

plh_status plh_line_search_by_projection_frame(const plh_keyline* kl, const uint8_t* ldesc, const double* linefn, int nl,
                                               const plh_grid_params* gp, uint8_t* occupied, int nq, const uint8_t* q_valid,
                                               const float* q_seg, const float* q_length, const uint8_t* q_desc,
                                               const uint8_t* q_hasobs, float th, int32_t* assigned, int* nmatches, int device) {
  return host_proj_lines(1, kl, ldesc, linefn, nl, gp, occupied, nq, q_valid, q_seg, q_length, q_desc, q_hasobs, th, 0.f, assigned,
                         nmatches, device);
}

plh_status plh_line_search_by_projection_ml(const plh_keyline* kl, const uint8_t* ldesc, const double* linefn, int nl,
                                            const plh_grid_params* gp, uint8_t* occupied, int nq, const uint8_t* q_valid,
                                            const float* q_seg, const float* q_viewcos, const uint8_t* q_desc,
                                            const uint8_t* q_hasobs, float th, float nnratio, int32_t* assigned, int* nmatches,
                                            int device) {
  return host_proj_lines(0, kl, ldesc, linefn, nl, gp, occupied, nq, q_valid, q_seg, q_viewcos, q_desc, q_hasobs, th, nnratio,
                         assigned, nmatches, device);
}

// The search inside LSDmatcher::Fuse(pKF, vpMapLines, th) on one KeyFrame, host buffers: see plh_line_fuse_search_batch_dev.
plh_status plh_line_fuse_search(const plh_keyline* kl, const uint8_t* cand_desc, int nl, const float* scale_factors_line, int nlevels,
                                int nq, const uint8_t* q_valid, const float* q_seg, const int32_t* q_level, const uint8_t* q_desc,
                                float th, float cos_th, int th_low, int32_t* best_idx, int* nfound, int device) {
  if (nl < 0 || nq < 0 || !nfound || !scale_factors_line || (nl > 0 && (!kl || !cand_desc)) ||
      (nq > 0 && (!q_valid || !q_seg || !q_level || !q_desc || !best_idx)))
    return PLH_ERR_INVALID;
  for (int i = 0; i < nq; i++) best_idx[i] = -1;
  *nfound = 0;
  if (nl == 0 || nq == 0) return PLH_OK;
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  Stage s(device);
  plh_keyline* dk = s.up(kl, nl); uint8_t* dd = s.up(cand_desc, (size_t)nl * 32);
  const int32_t ns[2] = {nl, nq};
  int32_t* dn = s.up(ns, 2);
  uint8_t* qv = s.up(q_valid, nq); float* qs = s.up(q_seg, (size_t)nq * 4); int32_t* ql = s.up(q_level, nq);
  uint8_t* qd = s.up(q_desc, (size_t)nq * 32);
  int32_t* db = s.alloc<int32_t>(nq); int32_t* dc = s.alloc<int32_t>(1);
  STAGE_OK(s);
  plh_status st = plh_line_fuse_search_batch_dev(dk, dd, dn, nl, 1, scale_factors_line, nlevels, dn + 1, nq, qv, qs, ql, qd, th, cos_th,
                                                 th_low, db, dc, s.stream());
  if (st != PLH_OK) return st;
  s.fetch_bytes(best_idx, db, (size_t)nq * 4);
  s.fetch_bytes(nfound, dc, 4);
  return s.finish();
}


// ---------------------------------------------------------------------------------------------------------------------
// Resident frames (round 6): frame_resident.h.  create = one upload + the grid kernel; the *_resident searches stage the
// queries only (the calling thread's arena) and read keypoints, descriptors and grid where they already lie.
// ---------------------------------------------------------------------------------------------------------------------
plh_status plh_frame_points_create(const plh_keypoint* kps_un, const uint8_t* desc, int n, const plh_grid_params* gp, int device,
                                   plh_frame_points** out) {
  if (!out || n < 0 || !gp || (n > 0 && (!kps_un || !desc))) { set_error("plh_frame_points_create: invalid argument"); return PLH_ERR_INVALID; }
  if (n > 6000) { set_error("plh_frame_points_create: %d keypoints (the searches hold at most 6000 per frame)", n); return PLH_ERR_CAPACITY; }
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  Stager st;
  plh_status rc = st.begin(device);
  if (rc != PLH_OK) return rc;
  plh_frame_points* f = new plh_frame_points();
  f->device = device; f->n = n; f->gp = *gp;
  const size_t N = (size_t)std::max(n, 1);
  const size_t oK = 0, oD = Stager::padded(N * sizeof(plh_keypoint)), oN = oD + Stager::padded(N * 32), oS = oN + 256,
               oI = oS + Stager::padded((PLH_GRID_CELLS + 1) * 4), oO = oI + Stager::padded(N * 4), total = oO + Stager::padded(N * 4);
  if (hipMalloc((void**)&f->block, total) != hipSuccess) {
    (void)hipGetLastError();
    delete f;
    set_error("plh_frame_points_create: cannot allocate %zu bytes", total);
    return PLH_ERR_ALLOC;
  }
  f->kps = (plh_keypoint*)(f->block + oK); f->desc = f->block + oD; f->dn = (int32_t*)(f->block + oN);
  f->cellStart = (int32_t*)(f->block + oS); f->cellItems = (int32_t*)(f->block + oI); f->node = (int32_t*)(f->block + oO);
  hipStream_t s = st.stream();
  const int32_t n32 = n;
  hipError_t e = hipMemcpyAsync(f->dn, &n32, 4, hipMemcpyHostToDevice, s);
  if (e == hipSuccess && n) e = hipMemcpyAsync(f->kps, kps_un, (size_t)n * sizeof(plh_keypoint), hipMemcpyHostToDevice, s);
  if (e == hipSuccess && n) e = hipMemcpyAsync(f->desc, desc, (size_t)n * 32, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemsetAsync(f->cellStart, 0, (PLH_GRID_CELLS + 1) * 4, s);
  if (e == hipSuccess) rc = plh_frame_assign_grid_batch_dev(f->kps, f->dn, (int)N, 1, gp, f->cellStart, f->cellItems, s);
  if (e == hipSuccess && rc == PLH_OK) e = hipStreamSynchronize(s);   // (the host arrays are the caller's: copied when this returns)
  if (e != hipSuccess || rc != PLH_OK) {
    if (e != hipSuccess) { set_error("plh_frame_points_create: %s", hipGetErrorString(e)); rc = PLH_ERR_HIP; }
    (void)hipFree(f->block);
    delete f;
    return rc;
  }
  *out = f;
  return PLH_OK;
}
// Frame::ComputeBoW's FeatureVector as one node id per feature (-1: the feature is in no node -- a stopped word): what
// SearchByBoW walks.  (Immutable otherwise: set it before the frame is searched by plh_orb_search_by_bow_resident.)
plh_status plh_frame_points_set_nodes(plh_frame_points* f, const int32_t* node) {
  if (!f || (f->n > 0 && !node)) return PLH_ERR_INVALID;
  PLH_HIP(hipSetDevice(f->device));
  if (f->n) PLH_HIP(hipMemcpy(f->node, node, (size_t)f->n * 4, hipMemcpyHostToDevice));
  f->hasNodes = true;
  return PLH_OK;
}
plh_status plh_frame_points_destroy(plh_frame_points* f) {
  if (!f) return PLH_OK;
  (void)hipSetDevice(f->device);
  (void)hipFree(f->block);
  delete f;
  return PLH_OK;
}
int plh_frame_points_count(const plh_frame_points* f) { return f ? f->n : 0; }

plh_status plh_frame_lines_create(const plh_keyline* kl, const uint8_t* ldesc, const double* linefn, int nl, const plh_grid_params* gp,
                                  int device, plh_frame_lines** out) {
  if (!out || nl < 0 || !gp || (nl > 0 && (!kl || !ldesc || !linefn))) { set_error("plh_frame_lines_create: invalid argument"); return PLH_ERR_INVALID; }
  if (nl > 8000) { set_error("plh_frame_lines_create: %d lines (the searches hold at most 8000 per frame)", nl); return PLH_ERR_CAPACITY; }
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  Stager st;
  plh_status rc = st.begin(device);
  if (rc != PLH_OK) return rc;
  plh_frame_lines* f = new plh_frame_lines();
  const size_t N = (size_t)std::max(nl, 1);
  f->device = device; f->nl = nl; f->gp = *gp; f->itemCap = (int)N * PLH_GRID_COLS;
  const size_t oK = 0, oD = Stager::padded(N * sizeof(plh_keyline)), oF = oD + Stager::padded(N * 32), oN = oF + Stager::padded(N * 24),
               oS = oN + 256, oI = oS + Stager::padded((PLH_GRID_CELLS + 1) * 4), total = oI + Stager::padded((size_t)f->itemCap * 4);
  if (hipMalloc((void**)&f->block, total) != hipSuccess) {
    (void)hipGetLastError();
    delete f;
    set_error("plh_frame_lines_create: cannot allocate %zu bytes", total);
    return PLH_ERR_ALLOC;
  }
  f->kl = (plh_keyline*)(f->block + oK); f->ldesc = f->block + oD; f->fn = (double*)(f->block + oF); f->dn = (int32_t*)(f->block + oN);
  f->cellStart = (int32_t*)(f->block + oS); f->cellItems = (int32_t*)(f->block + oI);
  hipStream_t s = st.stream();
  const int32_t n32 = nl;
  hipError_t e = hipMemcpyAsync(f->dn, &n32, 4, hipMemcpyHostToDevice, s);
  if (e == hipSuccess && nl) e = hipMemcpyAsync(f->kl, kl, (size_t)nl * sizeof(plh_keyline), hipMemcpyHostToDevice, s);
  if (e == hipSuccess && nl) e = hipMemcpyAsync(f->ldesc, ldesc, (size_t)nl * 32, hipMemcpyHostToDevice, s);
  if (e == hipSuccess && nl) e = hipMemcpyAsync(f->fn, linefn, (size_t)nl * 24, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemsetAsync(f->cellStart, 0, (PLH_GRID_CELLS + 1) * 4, s);
  if (e == hipSuccess) rc = plh_frame_assign_grid_lines_batch_dev(f->kl, f->dn, (int)N, 1, gp, f->cellStart, f->cellItems, f->itemCap, s);
  if (e == hipSuccess && rc == PLH_OK) e = hipStreamSynchronize(s);
  if (e != hipSuccess || rc != PLH_OK) {
    if (e != hipSuccess) { set_error("plh_frame_lines_create: %s", hipGetErrorString(e)); rc = PLH_ERR_HIP; }
    (void)hipFree(f->block);
    delete f;
    return rc;
  }
  *out = f;
  return PLH_OK;
}
plh_status plh_frame_lines_destroy(plh_frame_lines* f) {
  if (!f) return PLH_OK;
  (void)hipSetDevice(f->device);
  (void)hipFree(f->block);
  delete f;
  return PLH_OK;
}
int plh_frame_lines_count(const plh_frame_lines* f) { return f ? f->nl : 0; }

static plh_status resident_proj_points(int variant, const plh_frame_points* f, const float* scale_factors, int nlevels, uint8_t* occupied,
                                       int nq, const uint8_t* q_valid, const float* q_xy, const int32_t* q_level, const float* q_aux,
                                       const uint8_t* q_desc, const uint8_t* q_hasobs, float th, float nnratio, int mode, int check_ori,
                                       int32_t* assigned, int* nmatches, const char* who) {
  if (!f || nq < 0 || !nmatches || !scale_factors || (f->n > 0 && (!occupied || !assigned)) ||
      (nq > 0 && (!q_valid || !q_xy || !q_level || !q_aux || !q_desc || !q_hasobs)))
    return PLH_ERR_INVALID;
  const int n = f->n;
  for (int i = 0; i < n; i++) assigned[i] = -1;
  *nmatches = 0;
  if (n == 0 || nq == 0) return PLH_OK;
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  Stager st;
  plh_status rc = st.begin(f->device);
  if (rc != PLH_OK) return rc;
  uint8_t* docc = st.inout(occupied, (size_t)n);
  int32_t* da = st.out(assigned, (size_t)n);
  int32_t nm = 0;
  int32_t* dc = st.out(&nm, 1);
  const int32_t nq32 = nq;
  const int32_t* dnq = st.in(&nq32, 1);
  const uint8_t* qv = st.in(q_valid, (size_t)nq); const float* qx = st.in(q_xy, (size_t)nq * 2); const int32_t* ql = st.in(q_level, (size_t)nq);
  const float* qa = st.in(q_aux, (size_t)nq); const uint8_t* qd = st.in(q_desc, (size_t)nq * 32); const uint8_t* qh = st.in(q_hasobs, (size_t)nq);
  if ((rc = st.upload()) != PLH_OK) return rc;
  rc = launch_proj_points(variant, f->kps, f->desc, f->dn, n, 1, &f->gp, f->cellStart, f->cellItems, scale_factors, nlevels, docc, dnq, nq, qv,
                          qx, ql, qa, qd, qh, th, nnratio, mode, check_ori, da, dc, st.stream(), who);
  if (rc != PLH_OK) return rc;
  if ((rc = st.download()) != PLH_OK) return rc;
  *nmatches = nm;
  return PLH_OK;
}
plh_status plh_orb_search_by_projection_mp_resident(const plh_frame_points* f, const float* scale_factors, int nlevels, uint8_t* occupied,
                                                    int nq, const uint8_t* q_valid, const float* q_xy, const int32_t* q_level,
                                                    const float* q_viewcos, const uint8_t* q_desc, const uint8_t* q_hasobs, float th,
                                                    float nnratio, int32_t* assigned, int* nmatches) {
  return resident_proj_points(0, f, scale_factors, nlevels, occupied, nq, q_valid, q_xy, q_level, q_viewcos, q_desc, q_hasobs, th, nnratio,
                              0, 0, assigned, nmatches, "plh_orb_search_by_projection_mp_resident");
}
plh_status plh_orb_search_by_projection_frame_resident(const plh_frame_points* f, const float* scale_factors, int nlevels,
                                                       uint8_t* occupied, int nq, const uint8_t* q_valid, const float* q_uv,
                                                       const int32_t* q_octave, const float* q_angle, const uint8_t* q_desc,
                                                       const uint8_t* q_hasobs, float th, int mode, int check_ori, int32_t* assigned,
                                                       int* nmatches) {
  if (mode < 0 || mode > 2) return PLH_ERR_INVALID;
  return resident_proj_points(1, f, scale_factors, nlevels, occupied, nq, q_valid, q_uv, q_octave, q_angle, q_desc, q_hasobs, th, 0.f, mode,
                              check_ori, assigned, nmatches, "plh_orb_search_by_projection_frame_resident");
}
// ... with the projection of :1474-1484 on the device as well: the queries arrive as world positions (MapPoint::GetWorldPos of the last
// frame's points) and the current pose; q_valid = the caller's map-side gates (pMP && !mvbOutlier[i] && a descriptor), to which the
// projection adds `!(invzc < 0)`.
__global__ void __launch_bounds__(256) k_and_bytes(uint8_t* a, const uint8_t* b, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] = a[i] && b[i];
}
plh_status plh_orb_search_by_projection_frame_resident_world(const plh_frame_points* f, const float* scale_factors, int nlevels,
                                                             uint8_t* occupied, int nq, const plh_frame_view* view, const uint8_t* q_valid,
                                                             const float* q_world, const int32_t* q_octave, const float* q_angle,
                                                             const uint8_t* q_desc, const uint8_t* q_hasobs, float th, int mode,
                                                             int check_ori, int32_t* assigned, int* nmatches) {
  const char* who = "plh_orb_search_by_projection_frame_resident_world";
  if (!f || nq < 0 || !nmatches || !scale_factors || !view || mode < 0 || mode > 2 || (f->n > 0 && (!occupied || !assigned)) ||
      (nq > 0 && (!q_valid || !q_world || !q_octave || !q_angle || !q_desc || !q_hasobs))) {
    set_error("%s: invalid argument", who);
    return PLH_ERR_INVALID;
  }
  const int n = f->n;
  for (int i = 0; i < n; i++) assigned[i] = -1;
  *nmatches = 0;
  if (n == 0 || nq == 0) return PLH_OK;
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  Stager st;
  plh_status rc = st.begin(f->device);
  if (rc != PLH_OK) return rc;
  uint8_t* docc = st.inout(occupied, (size_t)n);
  int32_t* da = st.out(assigned, (size_t)n);
  int32_t nm = 0;
  int32_t* dc = st.out(&nm, 1);
  const int32_t nq32 = nq;
  const int32_t* dnq = st.in(&nq32, 1);
  const plh_frame_view* dview = st.in(view, 1);
  uint8_t* qv = const_cast<uint8_t*>(st.in(q_valid, (size_t)nq));   // (the arena's copy: the projection's verdict is folded into it)
  const float* qw = st.in(q_world, (size_t)nq * 3);
  const int32_t* ql = st.in(q_octave, (size_t)nq);
  const float* qa = st.in(q_angle, (size_t)nq); const uint8_t* qd = st.in(q_desc, (size_t)nq * 32); const uint8_t* qh = st.in(q_hasobs, (size_t)nq);
  float* quv = st.scratch<float>((size_t)nq * 2);
  uint8_t* qfront = st.scratch<uint8_t>((size_t)nq);
  if ((rc = st.upload()) != PLH_OK) return rc;
  if ((rc = plh_frame_project_points_batch_dev(dview, 1, dnq, nq, qw, 0, qfront, quv, st.stream())) != PLH_OK) return rc;
  hipLaunchKernelGGL(k_and_bytes, dim3((nq + 255) / 256), dim3(256), 0, (hipStream_t)st.stream(), qv, (const uint8_t*)qfront, nq);
  PLH_LAUNCH_CHECK();
  rc = launch_proj_points(1, f->kps, f->desc, f->dn, n, 1, &f->gp, f->cellStart, f->cellItems, scale_factors, nlevels, docc, dnq, nq, qv, quv, ql,
                          qa, qd, qh, th, 0.f, mode, check_ori, da, dc, st.stream(), who);
  if (rc != PLH_OK) return rc;
  if ((rc = st.download()) != PLH_OK) return rc;
  *nmatches = nm;
  return PLH_OK;
}
// ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) on two resident frames.
plh_status plh_orb_search_for_initialization_resident(const plh_frame_points* f1, const plh_frame_points* f2, float* prev_matched,
                                                      int window_size, float nnratio, int check_ori, int32_t* matches12, int* nmatches) {
  if (!f1 || !f2 || !nmatches || (f1->n > 0 && (!prev_matched || !matches12)) || f1->device != f2->device) return PLH_ERR_INVALID;
  const int n1 = f1->n, n2 = f2->n;
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  *nmatches = 0;
  if (n1 == 0 || n2 == 0) return PLH_OK;
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  const int cap = std::max(n1, n2);
  Stager st;
  plh_status rc = st.begin(f1->device);
  if (rc != PLH_OK) return rc;
  float* dpm = st.inout(prev_matched, (size_t)n1 * 2, (size_t)cap * 2);
  int32_t* dm = st.out(matches12, (size_t)n1, (size_t)cap);
  int32_t nm = 0;
  int32_t* dc = st.out(&nm, 1);
  if ((rc = st.upload()) != PLH_OK) return rc;
  rc = plh_orb_search_for_initialization_batch_dev(f1->kps, f1->desc, f1->dn, f2->kps, f2->desc, f2->dn, cap, 1, &f2->gp, f2->cellStart,
                                                   f2->cellItems, dpm, window_size, nnratio, check_ori, dm, dc, st.stream());
  if (rc != PLH_OK) return rc;
  if ((rc = st.download()) != PLH_OK) return rc;
  *nmatches = nm;
  return PLH_OK;
}

static plh_status resident_proj_lines(int variant, const plh_frame_lines* f, uint8_t* occupied, int nq, const uint8_t* q_valid,
                                      const float* q_seg, const float* q_aux, const uint8_t* q_desc, const uint8_t* q_hasobs, float th,
                                      float nnratio, int32_t* assigned, int* nmatches, const char* who) {
  if (!f || nq < 0 || !nmatches || (f->nl > 0 && (!occupied || !assigned)) || (nq > 0 && (!q_valid || !q_seg || !q_aux || !q_desc || !q_hasobs)))
    return PLH_ERR_INVALID;
  const int nl = f->nl;
  for (int i = 0; i < nl; i++) assigned[i] = -1;
  *nmatches = 0;
  if (nl == 0 || nq == 0) return PLH_OK;
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  Stager st;
  plh_status rc = st.begin(f->device);
  if (rc != PLH_OK) return rc;
  uint8_t* docc = st.inout(occupied, (size_t)nl);
  int32_t* da = st.out(assigned, (size_t)nl);
  int32_t nm = 0;
  int32_t* dc = st.out(&nm, 1);
  const int32_t nq32 = nq;
  const int32_t* dnq = st.in(&nq32, 1);
  const uint8_t* qv = st.in(q_valid, (size_t)nq); const float* qs = st.in(q_seg, (size_t)nq * 4); const float* qa = st.in(q_aux, (size_t)nq);
  const uint8_t* qd = st.in(q_desc, (size_t)nq * 32); const uint8_t* qh = st.in(q_hasobs, (size_t)nq);
  if ((rc = st.upload()) != PLH_OK) return rc;
  rc = launch_proj_lines(variant, f->kl, f->ldesc, f->fn, f->dn, nl, 1, &f->gp, f->cellStart, f->cellItems, f->itemCap, docc, dnq, nq, qv, qs,
                         qa, qd, qh, th, nnratio, da, dc, st.stream(), who);
  if (rc != PLH_OK) return rc;
  if ((rc = st.download()) != PLH_OK) return rc;
  *nmatches = nm;
  return PLH_OK;
}
plh_status plh_line_search_by_projection_frame_resident(const plh_frame_lines* f, uint8_t* occupied, int nq, const uint8_t* q_valid,
                                                        const float* q_seg, const float* q_length, const uint8_t* q_desc,
                                                        const uint8_t* q_hasobs, float th, int32_t* assigned, int* nmatches) {
  return resident_proj_lines(1, f, occupied, nq, q_valid, q_seg, q_length, q_desc, q_hasobs, th, 0.f, assigned, nmatches,
                             "plh_line_search_by_projection_frame_resident");
}
plh_status plh_line_search_by_projection_ml_resident(const plh_frame_lines* f, uint8_t* occupied, int nq, const uint8_t* q_valid,
                                                     const float* q_seg, const float* q_viewcos, const uint8_t* q_desc,
                                                     const uint8_t* q_hasobs, float th, float nnratio, int32_t* assigned, int* nmatches) {
  return resident_proj_lines(0, f, occupied, nq, q_valid, q_seg, q_viewcos, q_desc, q_hasobs, th, nnratio, assigned, nmatches,
                             "plh_line_search_by_projection_ml_resident");
}

}  // extern "C"
