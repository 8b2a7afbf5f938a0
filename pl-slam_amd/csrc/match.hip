// Hamming matching kernels (gfx950, wave64) + C ABI.  No MFMA: 256-bit XOR + popcount on the VALU.
//
//   k_knn2            cv::BFMatcher(NORM_HAMMING).knnMatch(k=2)        (LSDmatcher.cpp:468-469)
//   k_line_bfmatch    LSDmatcher::FrameBFMatch + lineDescriptorMAD      (LSDmatcher.cpp:462-486, 627-652)
//   k_line_mutual     LSDmatcher::SearchDouble mutual check             (LSDmatcher.cpp:445-455)
//   k_search_by_bow   ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...)   (ORBmatcher.cc:187-327, 1718-1759)
//
// All results are integer / index valued and bit-exact against the oracle.
#include <climits>
#include <vector>

#include "plh_common.h"
#include "plh_stage.h"
#include "frame_resident.h"

namespace plh {

struct Desc256 {
  unsigned long long w[4];
};

__device__ __forceinline__ Desc256 load_desc(const uint8_t* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 a = q[0], b = q[1];
  Desc256 d;
  d.w[0] = (unsigned long long)a.x | ((unsigned long long)a.y << 32);
  d.w[1] = (unsigned long long)a.z | ((unsigned long long)a.w << 32);
  d.w[2] = (unsigned long long)b.x | ((unsigned long long)b.y << 32);
  d.w[3] = (unsigned long long)b.z | ((unsigned long long)b.w << 32);
  return d;
}

// ---------------------------------------------------------------------------------------------
// Brute-force 2-NN.  grid (ceil(q_cap/256), pairs); each thread owns one query row in registers;
// train rows stream through an 8 KiB LDS tile and are read as wave-wide broadcasts.
// Scan order is ascending train index with strict '<', i.e. ties keep the lower index (BFMatcher).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_knn2(const uint8_t* q, const int* nqArr, int nqConst, int qCap, const uint8_t* t,
                                              const int* ntArr, int ntConst, int tCap, int32_t* idx, int32_t* dist) {
  __shared__ unsigned long long tile[256 * 4];
  const int pair = blockIdx.y;
  // counts are clamped to the row capacities like everywhere else (a count from another handle must not overrun the sets)
  const int nq = max(0, min(nqArr ? nqArr[pair] : nqConst, qCap));
  const int nt = max(0, min(ntArr ? ntArr[pair] : ntConst, tCap));
  const int tid = threadIdx.x;
  const int i = blockIdx.x * 256 + tid;
  if (blockIdx.x * 256 >= max(nq, 1) && blockIdx.x > 0) return;
  const uint8_t* qp = q + (long long)pair * qCap * 32;
  const uint8_t* tp = t + (long long)pair * tCap * 32;
  Desc256 me;
  me.w[0] = me.w[1] = me.w[2] = me.w[3] = 0;
  if (i < nq) me = load_desc(qp + (long long)i * 32);
  int b0 = INT_MAX, b1 = INT_MAX, i0 = -1, i1 = -1;
  for (int base = 0; base < nt; base += 256) {
    const int cnt = min(256, nt - base);
    __syncthreads();
    if (tid < cnt) {
      const Desc256 d = load_desc(tp + (long long)(base + tid) * 32);
      tile[tid * 4 + 0] = d.w[0]; tile[tid * 4 + 1] = d.w[1]; tile[tid * 4 + 2] = d.w[2]; tile[tid * 4 + 3] = d.w[3];
    }
    __syncthreads();
    for (int j = 0; j < cnt; j++) {
      const int d = __popcll(me.w[0] ^ tile[j * 4]) + __popcll(me.w[1] ^ tile[j * 4 + 1]) +
                    __popcll(me.w[2] ^ tile[j * 4 + 2]) + __popcll(me.w[3] ^ tile[j * 4 + 3]);
      if (d < b0) { b1 = b0; i1 = i0; b0 = d; i0 = base + j; }
      else if (d < b1) { b1 = d; i1 = base + j; }
    }
  }
  if (i < nq) {
    const long long o = ((long long)pair * qCap + i) * 2;
    idx[o] = i0; idx[o + 1] = i1;
    dist[o] = b0; dist[o + 1] = b1;
  }
}

// ---------------------------------------------------------------------------------------------
// FrameBFMatch on a knn2 table.  One block per pair.  The two medians of lineDescriptorMAD are
// order statistics of integer gaps in [0,256] -> exact via 257-bin histograms, no sort needed.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_line_bfmatch(const int32_t* idx, const int32_t* dist, const int* nqArr,
                                                      const int* ntArr, int qCap, float TH, float nnratio,
                                                      int32_t* matches) {
  __shared__ int hist[260];
  __shared__ int s_med;
  __shared__ double s_th;
  const int pair = blockIdx.x, tid = threadIdx.x;
  const int nq = max(0, min(nqArr[pair], qCap)), nt = max(0, ntArr[pair]);
  const int32_t* D = dist + (long long)pair * qCap * 2;
  const int32_t* I = idx + (long long)pair * qCap * 2;
  int32_t* M = matches + (long long)pair * qCap;
  const bool live = nq > 0 && nt >= 2;   // the reference is UB otherwise (lmatches[i][1], matches[size/2])
  for (int k = tid; k < 260; k += 256) hist[k] = 0;
  __syncthreads();
  if (live)
    for (int i = tid; i < nq; i += 256) atomicAdd(&hist[D[i * 2 + 1] - D[i * 2]], 1);
  __syncthreads();
  if (tid == 0 && live) {   // element [nq/2] of the gaps sorted DESCENDING
    int cum = 0, g = 256;
    for (; g >= 0; g--) { cum += hist[g]; if (cum > nq / 2) break; }
    s_med = g;
  }
  __syncthreads();
  for (int k = tid; k < 260; k += 256) hist[k] = 0;
  __syncthreads();
  if (live) {
    const int med = s_med;
    for (int i = tid; i < nq; i += 256) atomicAdd(&hist[abs(D[i * 2 + 1] - D[i * 2] - med)], 1);
  }
  __syncthreads();
  if (tid == 0 && live) {   // element [nq/2] of the absolute deviations sorted ASCENDING
    int cum = 0, g = 0;
    for (; g <= 256; g++) { cum += hist[g]; if (cum > nq / 2) break; }
    const double nn12_mad = 1.4826 * (double)(float)g;
    s_th = nn12_mad * 0.5;
  }
  __syncthreads();
  for (int i = tid; i < qCap; i += 256) {
    int m = -1;
    if (live && i < nq) {
      const float m0 = (float)D[i * 2], m1 = (float)D[i * 2 + 1];
      const double dist_12 = m1 - m0;
      if (dist_12 > s_th && m0 < TH && m0 < nnratio * m1) m = I[i * 2];
    }
    M[i] = m;
  }
}

__global__ void __launch_bounds__(256) k_line_mutual(const int32_t* m1, const int32_t* m2, const int* n1Arr, const int* n2Arr,
                                                     int cap, int32_t* out, int32_t* nmatches) {
  __shared__ int s_cnt;
  const int pair = blockIdx.x, tid = threadIdx.x;
  const int n1 = max(0, min(n1Arr[pair], cap)), n2 = max(0, min(n2Arr[pair], cap));
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  int c = 0;
  for (int i = tid; i < cap; i += 256) {
    int j = -1;
    if (i < n1 && n1 > 0 && n2 > 0) {
      j = m1[(long long)pair * cap + i];
      if (j >= 0) {
        if (m2[(long long)pair * cap + j] != i) j = -1;
        else c++;
      }
    }
    out[(long long)pair * cap + i] = j;
  }
  if (c) atomicAdd(&s_cnt, c);
  __syncthreads();
  if (tid == 0) nmatches[pair] = s_cnt;
}

// ---------------------------------------------------------------------------------------------
// SearchByBoW.  One block per pair.  Phase A (all threads): stable rank-sort of both feature sets
// by (node id, feature index) == iteration order of DBoW2::FeatureVector (std::map<node, vector<idx>>).
// Phase B: KeyFrame features in that order, sequential because of the greedy "already matched" skip
// (ORBmatcher.cc:240) -- but only WITHIN a node: a KeyFrame feature's candidates are the Frame features under the same
// node, so a claim made under one node is never read under another.  The nodes of a pair (some hundred, a dozen features
// each) are dealt to the block's wavefronts round-robin (round 6; rounds 1 - 5 walked all of a pair's features on wave 0:
// 3.6 ms for one pair of 1000 features, which is what a tracker pays in TrackReferenceKeyFrame); the 64 lanes scan the
// Frame candidates of the node.  The rotation histogram only counts (ComputeThreeMaxima reads the bins' sizes), so it is
// summed over the wavefronts at the end.
// ---------------------------------------------------------------------------------------------
// (wave_min_i32: DPP row steps on the hardware, plh_shims.h)
constexpr int BOW_K = 4;        // candidates kept per KeyFrame feature of a small node group
constexpr int BOW_SMALL = 64;   // a node group is small if its candidate range is no wider than this
__global__ void __launch_bounds__(1024) k_search_by_bow(const uint8_t* desc1, const float* angle1, const int32_t* node1,
                                                       const uint8_t* valid1, const int* n1Arr, const uint8_t* desc2,
                                                       const float* angle2, const int32_t* node2, const int* n2Arr, int cap,
                                                       int angStride, int thLow, float nnratio, int checkOri, int32_t* matches21,
                                                       int32_t* nmatchesOut, const uint8_t* valid2, int kfkf) {
  HIP_DYNAMIC_SHARED(unsigned char, smem)
  unsigned* tops = (unsigned*)smem;      // [cap][BOW_K] the best candidates of a KeyFrame feature (small node groups); before that, the
  unsigned long long* skey = (unsigned long long*)smem;   // sort keys of phase A (at most 2 cap - 1 of them: the same 16 cap bytes)
  int* nd1 = (int*)(tops + (size_t)cap * BOW_K);   // node id per feature
  int* nd2 = nd1 + cap;
  int* snode2 = nd2 + cap;               // node id by sorted position (set 2)
  int* m2 = snode2 + cap;                // Frame feature -> KeyFrame feature
  float* ang2 = (float*)(m2 + cap);      // angle of the Frame's keypoints (read at every accepted match)
  unsigned short* ord1 = (unsigned short*)(ang2 + cap);
  unsigned short* ord2 = ord1 + cap;
  unsigned short* rlo = ord2 + cap;      // per sorted position of set 1: [rlo, rhi) = sorted positions of set 2 with
  unsigned short* rhi = rlo + cap;       // the same node (empty for invalid / unusable features)
  unsigned short* gs = rhi + cap;        // [cap + 1] first sorted position of every node group of set 1
  unsigned char* bin2 = (unsigned char*)(gs + cap + 2);
  unsigned char* gbig = bin2 + cap;      // per node group: its candidate range is wider than a wavefront (BOW_SMALL)
  __shared__ int s_part[1024];
  __shared__ int s_groups, s_nm, s_hist[32];

  const int pair = blockIdx.x, tid = threadIdx.x, T = (int)blockDim.x;
  const int n1 = max(0, min(n1Arr[pair], cap)), n2 = max(0, min(n2Arr[pair], cap));
  const long long o = (long long)pair * cap;
  if (tid < 32) s_hist[tid] = 0;
  if (tid == 0) { s_nm = 0; s_groups = 0; }
  for (int i = tid; i < cap; i += T) {
    nd1[i] = i < n1 ? node1[o + i] : -1;
    nd2[i] = i < n2 ? node2[o + i] : -1;
    m2[i] = -1;
    bin2[i] = 255;
    ang2[i] = (checkOri && i < n2) ? angle2[(o + i) * angStride] : 0.f;
  }
  __syncthreads();
  // (node, feature index) order of both sets: bitonic sort of 64-bit keys node << 16 | index in LDS; invalid nodes (< 0) sort to the
  // end (key INT_MAX).  Rounds 1 - 5 ranked every feature against all others -- O(n^2) on one CU: 350 of the 430 us a lone pair of
  // 2000 features cost.
  for (int set = 0; set < 2; set++) {
    const int n = set ? n2 : n1;
    const int* nd = set ? nd2 : nd1;
    int np = 1;
    while (np < n) np <<= 1;
    for (int i = tid; i < np; i += T)
      skey[i] = i < n ? ((unsigned long long)(unsigned)(nd[i] < 0 ? INT_MAX : nd[i]) << 16) | (unsigned long long)i : ~0ull;
    __syncthreads();
    for (int k = 2; k <= np; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = tid; t < (np >> 1); t += T) {
          const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
          const unsigned long long a = skey[i], b = skey[l];
          if (((i & k) == 0) ? a > b : a < b) { skey[i] = b; skey[l] = a; }
        }
        __syncthreads();
      }
    for (int r = tid; r < n; r += T) {
      const unsigned long long e = skey[r];
      if (set) { ord2[r] = (unsigned short)(e & 0xffffu); snode2[r] = (int)(e >> 16); }
      else ord1[r] = (unsigned short)(e & 0xffffu);
    }
    __syncthreads();
  }
  // candidate ranges of every KeyFrame feature, in parallel: the sequential walk below then only reads them
  for (int r1 = tid; r1 < n1; r1 += T) {
    const int i = ord1[r1];
    const int nd = nd1[i];
    int lo = 0, hi2 = 0;
    if (nd >= 0 && valid1[o + i]) {
      int hi = n2;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (snode2[mid] < nd) lo = mid + 1; else hi = mid; }
      int top = n2;
      hi2 = lo;
      while (hi2 < top) { const int mid = (hi2 + top) >> 1; if (snode2[mid] <= nd) hi2 = mid + 1; else top = mid; }
    }
    rlo[r1] = (unsigned short)lo;
    rhi[r1] = (unsigned short)hi2;
  }
  // the node groups of set 1: sorted positions where the node changes (each thread a contiguous chunk, counts scanned by wave 0)
  const int chunk = (n1 + T - 1) / T, c0 = min(tid * chunk, n1), c1 = min(c0 + chunk, n1);
  {
    int cnt = 0;
    for (int r1 = c0; r1 < c1; r1++) cnt += r1 == 0 || nd1[ord1[r1]] != nd1[ord1[r1 - 1]];
    s_part[tid] = cnt;
  }
  __syncthreads();
  if (tid < 64) {
    const int per = (T + 63) / 64;
    int sum = 0;
    for (int k = 0; k < per; k++) sum += tid * per + k < T ? s_part[tid * per + k] : 0;
    int incl = sum;
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d);
      if (tid >= d) incl += v;
    }
    int run = incl - sum;
    for (int k = 0; k < per; k++)
      if (tid * per + k < T) { const int c = s_part[tid * per + k]; s_part[tid * per + k] = run; run += c; }
    if (tid == 63) s_groups = incl;
  }
  __syncthreads();
  {
    int g = s_part[tid];
    for (int r1 = c0; r1 < c1; r1++)
      if (r1 == 0 || nd1[ord1[r1]] != nd1[ord1[r1 - 1]]) gs[g++] = (unsigned short)r1;
    if (tid == 0) gs[s_groups] = (unsigned short)n1;
  }
  __syncthreads();
  const int nGroups = s_groups;
  const float factor = 1.0f / 30;
  const uint8_t* D1 = desc1 + o * 32;
  const uint8_t* D2 = desc2 + o * 32;
  int nmatches = 0;
  // ---- small node groups (candidate range <= BOW_SMALL: all of them with a real vocabulary -- 10^2 nodes four levels up, a dozen
  // features each): the distances do not depend on the matching state, so every KeyFrame feature first gets the BOW_K best candidates
  // of its node in (distance, position) order, all features in parallel; the reference's (best, second best) are the first two FREE
  // entries of that order (strict `<` in its scan: the first of equal distances wins).  Then ONE LANE per group replays the group's
  // features in order on the lists -- LDS reads only; a list that was cut and has fewer than two free entries left is re-evaluated
  // exactly over the node's candidates.
  for (int g = tid; g < nGroups; g += T) {
    int big = 0;
    for (int r1 = gs[g]; r1 < gs[g + 1]; r1++) big |= (int)rhi[r1] - (int)rlo[r1] > BOW_SMALL;
    gbig[g] = (unsigned char)big;
  }
  for (int r1 = tid; r1 < n1; r1 += T) {
    const int lo = rlo[r1], hi2 = rhi[r1];
    unsigned top[BOW_K];
#pragma unroll
    for (int k = 0; k < BOW_K; k++) top[k] = 0xffffffffu;
    bool cut = false;
    if (lo < hi2 && hi2 - lo <= BOW_SMALL) {
      const Desc256 dk = load_desc(D1 + (long long)ord1[r1] * 32);
      for (int c = lo; c < hi2; c++) {
        const int f = ord2[c];
        if (valid2 && !valid2[o + f]) continue;
        const Desc256 df = load_desc(D2 + (long long)f * 32);
        unsigned e = ((unsigned)hamming256(dk.w, df.w) << 16) | (unsigned)(c - lo);
        if (top[BOW_K - 1] != 0xffffffffu) cut = true;   // one entry will not fit
#pragma unroll
        for (int k = 0; k < BOW_K; k++)
          if (e < top[k]) { const unsigned t = top[k]; top[k] = e; e = t; }
      }
      if (cut && top[0] != 0xffffffffu) top[0] |= 0x80000000u;
    }
#pragma unroll
    for (int k = 0; k < BOW_K; k++) tops[(size_t)r1 * BOW_K + k] = top[k];
  }
  __syncthreads();
  for (int g = tid; g < nGroups; g += T) {
    if (gbig[g]) continue;
    for (int r1 = gs[g]; r1 < gs[g + 1]; r1++) {
      const int lo = rlo[r1], hi2 = rhi[r1];
      if (lo >= hi2) continue;
      const int i = ord1[r1];
      int b1 = 256, b2 = 256, bestC = -1, nfree = 0;
      bool cut = false;
#pragma unroll
      for (int k = 0; k < BOW_K; k++) {
        unsigned e = tops[(size_t)r1 * BOW_K + k];
        if (e == 0xffffffffu || nfree == 2) continue;
        if (k == 0) { cut = (e & 0x80000000u) != 0u; e &= 0x7fffffffu; }
        const int c = lo + (int)(e & 0xffffu);
        if (m2[ord2[c]] >= 0) continue;
        if (nfree == 0) { b1 = (int)(e >> 16); bestC = c; }
        else b2 = (int)(e >> 16);
        nfree++;
      }
      if (cut && nfree < 2) {   // the list ran out: the reference's scan over what is still free under the node
        const Desc256 dk = load_desc(D1 + (long long)i * 32);
        b1 = 256; b2 = 256; bestC = -1;
        for (int c = lo; c < hi2; c++) {
          const int f = ord2[c];
          if (m2[f] >= 0) continue;
          if (valid2 && !valid2[o + f]) continue;
          const Desc256 df = load_desc(D2 + (long long)f * 32);
          const int d = hamming256(dk.w, df.w);
          if (d < b1) { b2 = b1; b1 = d; bestC = c; }
          else if (d < b2) { b2 = d; }
        }
      }
      if (bestC < 0) continue;
      if ((kfkf ? b1 < thLow : b1 <= thLow) && (float)b1 < nnratio * (float)b2) {
        const int bestF = ord2[bestC];
        int bin = 255;
        if (checkOri) {
          float rot = angle1[(o + i) * angStride] - ang2[bestF];
          if (rot < 0.0f) rot += 360.0f;
          bin = (int)roundf(rot * factor);
          if (bin == 30) bin = 0;
          atomicAdd(&s_hist[bin], 1);
        }
        m2[bestF] = i;
        bin2[bestF] = (unsigned char)bin;
        nmatches++;
      }
    }
  }
  if (nmatches) atomicAdd(&s_nm, nmatches);
  nmatches = 0;
  // ---- wide node groups (a vocabulary cut near its root: hundreds of features under one node), a wavefront per group

  const int lane = tid & 63, wv = tid >> 6, nWaves = T >> 6;
  int myHist = 0;      // lane b < 30 holds rotHist[b].size()
  // The walk is sequential (a match removes its Frame feature from every later search), but what a step READS from
  // global memory does not depend on the matching state: the KeyFrame descriptor, its angle and the first 64 candidate
  // descriptors of the NEXT feature are requested before the current one is decided.  Two named register sets take
  // turns (no copies), all loads are unconditional (indices clamped).
  struct Pre {
    int i, lo, hi, f;
    float a1;
    Desc256 dk, df;
  };
  auto fetch = [&](int r1, int rEnd, Pre& p) {   // position r1 of the group that ends at rEnd
    const int rr = min(r1, max(n1 - 1, 0));
    p.i = ord1[rr];
    p.lo = rlo[rr];
    p.hi = r1 < rEnd ? (int)rhi[rr] : 0;
    if (r1 >= rEnd) p.lo = 0;
    p.f = ord2[min(p.lo + lane, max(n2 - 1, 0))];
    p.dk = load_desc(D1 + (long long)p.i * 32);
    p.df = load_desc(D2 + (long long)p.f * 32);
    p.a1 = angle1[(o + p.i) * angStride];
  };
  auto step = [&](const Pre& p) {
    const int lo = p.lo, hi2 = p.hi;
    if (lo >= hi2) return;
    int b1 = 256, b2 = 256, p1 = 0x7fffffff;
    {
      const int c = lo + lane;
      bool ok = c < hi2 && m2[p.f] < 0;
      if (ok && valid2) ok = valid2[o + p.f] != 0;   // KeyFrame-KeyFrame form: pMP2 missing or bad (ORBmatcher.cc:628-632)
      if (ok) { b1 = hamming256(p.dk.w, p.df.w); p1 = c; }
    }
    for (int c = lo + 64 + lane; c < hi2; c += 64) {   // more than 64 features under one node (rare)
      const int f = ord2[c];
      if (m2[f] >= 0) continue;
      if (valid2 && !valid2[o + f]) continue;
      const Desc256 df = load_desc(D2 + (long long)f * 32);
      const int d = hamming256(p.dk.w, df.w);
      if (d < b1) { b2 = b1; b1 = d; p1 = c; }
      else if (d < b2) { b2 = d; }
    }
    // wave reduction: best = min (dist, position); second = min over everything except the winning element
    const int key = b1 < 256 ? ((b1 << 20) | (p1 - lo)) : 0x7fffffff;
    const int kmin = wave_min_i32(key);
    if (kmin == 0x7fffffff) return;
    const bool winner = (key == kmin) && b1 < 256;
    const int bestDist2 = wave_min_i32(winner ? b2 : b1);
    const int bestDist1 = kmin >> 20, bestPos = (kmin & 0xfffff) + lo;
    if ((kfkf ? bestDist1 < thLow : bestDist1 <= thLow) && (float)bestDist1 < nnratio * (float)bestDist2) {
      const int bestF = ord2[bestPos];
      int bin = 255;
      if (checkOri) {
        float rot = p.a1 - ang2[bestF];
        if (rot < 0.0f) rot += 360.0f;
        bin = (int)roundf(rot * factor);
        if (bin == 30) bin = 0;
        if (lane == bin) myHist++;
      }
      // every lane performs the same store (uniform values): later reads by any lane see it
      PLH_WAVE_SYNC();
      m2[bestF] = p.i;
      bin2[bestF] = (unsigned char)bin;
      PLH_WAVE_SYNC();
      nmatches++;
    }
  };
  if (n1 > 0 && n2 > 0) {
    int nb = 0;   // wide groups so far
    for (int g = 0; g < nGroups; g++) {
      if (!gbig[g]) continue;
      if (nb++ % nWaves != wv) continue;
      const int rBeg = gs[g], rEnd = gs[g + 1];
      Pre A, B;
      fetch(rBeg, rEnd, A);
      for (int r1 = rBeg; r1 < rEnd; r1 += 2) {
        fetch(r1 + 1, rEnd, B);
        step(A);
        fetch(r1 + 2, rEnd, A);
        step(B);
      }
    }
  }
  PLH_WAVE_SYNC();
  // the wavefronts' counts into one histogram; wave 0 finishes
  if (checkOri && lane < 30 && myHist) atomicAdd(&s_hist[lane], myHist);
  if (lane == 0 && nmatches) atomicAdd(&s_nm, nmatches);
  __syncthreads();
  if (wv != 0) return;
  myHist = lane < 30 ? s_hist[lane] : 0;
  nmatches = s_nm;
  if (checkOri) {   // ComputeThreeMaxima over the 30 bins
    int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
    for (int b = 0; b < 30; b++) {
      const int s = __shfl(myHist, b);
      if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = b; }
      else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = b; }
      else if (s > max3) { max3 = s; ind3 = b; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
    int removed = 0;
    for (int f = lane; f < n2; f += 64) {
      const int b = bin2[f];
      if (m2[f] >= 0 && b != ind1 && b != ind2 && b != ind3) { m2[f] = -1; removed++; }
    }
    removed = wave_sum(removed);
    nmatches -= removed;
  }
  if (!kfkf) {
    for (int f = lane; f < cap; f += 64) matches21[o + f] = f < n2 ? m2[f] : -1;
  } else {   // vpMatches12: indexed by the first KeyFrame's features (the match relation is a partial bijection)
    for (int f = lane; f < cap; f += 64) matches21[o + f] = -1;
    PLH_WAVE_SYNC();
    for (int f = lane; f < n2; f += 64)
      if (m2[f] >= 0) matches21[o + m2[f]] = f;
  }
  if (lane == 0) nmatchesOut[pair] = nmatches;
}

// ---------------------------------------------------------------------------------------------
// ORBmatcher::SearchForTriangulation (monocular), reference src/ORBmatcher.cc:720-912 + CheckDistEpipolarLine :154-173.
// Phase A as in k_search_by_bow.  Phase B: this fork never marks KF2 features as matched, so the KF1 features are
// independent: the four waves take them round-robin, lanes scan the node's candidates; the reference's scan
// ("dist <= bestDist" before the geometric tests) ends on the eligible candidate of least distance, LAST one among ties.
// ---------------------------------------------------------------------------------------------
struct TriGeom {
  float F[9];
  float ex, ey;
  float sf[16], sig2[16];
};

__global__ void __launch_bounds__(256) k_search_triangulation(const plh_keypoint* kps1, const uint8_t* desc1, const int32_t* node1,
                                                              const uint8_t* hasMp1, const int* n1Arr, const plh_keypoint* kps2,
                                                              const uint8_t* desc2, const int32_t* node2, const uint8_t* hasMp2,
                                                              const int* n2Arr, int cap, TriGeom g, int thLow, int checkOri,
                                                              int32_t* matches12, int32_t* nmatchesOut) {
  HIP_DYNAMIC_SHARED(unsigned char, smem)
  int* nd1 = (int*)smem;
  int* nd2 = nd1 + cap;
  int* snode2 = nd2 + cap;
  int* m12 = snode2 + cap;
  unsigned short* ord1 = (unsigned short*)(m12 + cap);
  unsigned short* ord2 = ord1 + cap;
  unsigned char* bin1 = (unsigned char*)(ord2 + cap);
  __shared__ int s_hist[32];
  __shared__ int s_cnt;

  const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n1 = min(n1Arr[pair], cap), n2 = min(n2Arr[pair], cap);
  const long long o = (long long)pair * cap;
  for (int i = tid; i < cap; i += 256) {
    nd1[i] = i < n1 ? node1[o + i] : -1;
    nd2[i] = i < n2 ? node2[o + i] : -1;
    m12[i] = -1;
    bin1[i] = 255;
  }
  if (tid < 32) s_hist[tid] = 0;
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  for (int i = tid; i < n1; i += 256) {
    const int k = nd1[i] < 0 ? INT_MAX : nd1[i];
    int r = 0;
    for (int j = 0; j < n1; j++) {
      const int kj = nd1[j] < 0 ? INT_MAX : nd1[j];
      r += (kj < k) || (kj == k && j < i);
    }
    ord1[r] = (unsigned short)i;
  }
  for (int i = tid; i < n2; i += 256) {
    const int k = nd2[i] < 0 ? INT_MAX : nd2[i];
    int r = 0;
    for (int j = 0; j < n2; j++) {
      const int kj = nd2[j] < 0 ? INT_MAX : nd2[j];
      r += (kj < k) || (kj == k && j < i);
    }
    ord2[r] = (unsigned short)i;
    snode2[r] = k;
  }
  __syncthreads();

  const plh_keypoint *K1 = kps1 + o, *K2 = kps2 + o;
  const uint8_t *D1 = desc1 + o * 32, *D2 = desc2 + o * 32;
  int myMatches = 0;
  for (int r1 = wv; r1 < n1; r1 += 4) {
    const int i = ord1[r1];
    const int nd = nd1[i];
    if (nd < 0) break;
    if (hasMp1[o + i]) continue;
    int lo = 0, hi = n2;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (snode2[mid] < nd) lo = mid + 1; else hi = mid; }
    int hi2 = lo, top = n2;
    while (hi2 < top) { const int mid = (hi2 + top) >> 1; if (snode2[mid] <= nd) hi2 = mid + 1; else top = mid; }
    if (lo >= hi2) continue;
    const plh_keypoint kp1 = K1[i];
    const Desc256 dk = load_desc(D1 + (long long)i * 32);
    const float a = kp1.x * g.F[0] + kp1.y * g.F[3] + g.F[6];
    const float b = kp1.x * g.F[1] + kp1.y * g.F[4] + g.F[7];
    const float c = kp1.x * g.F[2] + kp1.y * g.F[5] + g.F[8];
    const float den = a * a + b * b;
    int key = 0x7fffffff;
    for (int p = lo + lane; p < hi2; p += 64) {
      const int f = ord2[p];
      if (hasMp2[o + f]) continue;
      const Desc256 df = load_desc(D2 + (long long)f * 32);
      const int d = hamming256(dk.w, df.w);
      if (d > thLow) continue;
      const plh_keypoint kp2 = K2[f];
      const float distex = g.ex - kp2.x, distey = g.ey - kp2.y;
      if (distex * distex + distey * distey < 100 * g.sf[kp2.octave & 15]) continue;
      const float num = a * kp2.x + b * kp2.y + c;
      if (den == 0) continue;
      const float dsqr = num * num / den;
      if (!((double)dsqr < 3.84 * (double)g.sig2[kp2.octave & 15])) continue;
      key = min(key, (d << 16) | (0xffff - (p - lo)));
    }
    for (int s = 32; s >= 1; s >>= 1) key = min(key, __shfl_xor(key, s));
    if (key == 0x7fffffff) continue;
    const int bestIdx2 = ord2[lo + (0xffff - (key & 0xffff))];
    if (lane == 0) {
      m12[i] = bestIdx2;
      myMatches++;
      if (checkOri) {
        float rot = kp1.angle - K2[bestIdx2].angle;
        if (rot < 0.0f) rot += 360.0f;
        int bin = (int)roundf(rot * (1.0f / 30));
        if (bin == 30) bin = 0;
        bin1[i] = (unsigned char)bin;
        atomicAdd(&s_hist[bin], 1);
      }
    }
  }
  if (lane == 0 && myMatches) atomicAdd(&s_cnt, myMatches);
  __syncthreads();
  if (checkOri) {
    int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
    for (int b = 0; b < 30; b++) {
      const int sz = s_hist[b];
      if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = b; }
      else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = b; }
      else if (sz > max3) { max3 = sz; ind3 = b; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
    int removed = 0;
    for (int i = tid; i < n1; i += 256) {
      const int b = bin1[i];
      if (b != 255 && b != ind1 && b != ind2 && b != ind3) { m12[i] = -1; removed++; }
    }
    if (removed) atomicAdd(&s_cnt, -removed);
  }
  __syncthreads();
  for (int i = tid; i < cap; i += 256) matches12[o + i] = i < n1 ? m12[i] : -1;
  if (tid == 0) nmatchesOut[pair] = s_cnt;
}

static size_t tri_lds_bytes(int cap) { return (size_t)cap * (4 * 4 + 2 * 2 + 1) + 64; }

static size_t bow_lds_bytes(int cap) { return (size_t)cap * (5 * 4 + BOW_K * 4 + 5 * 2 + 2) + 64 + 8; }
// wavefronts per pair: a lone pair (a tracker's TrackReferenceKeyFrame) gets sixteen to deal its nodes to, a resident batch four
static int bow_threads(int pairs) { return pairs >= 1024 ? 256 : (pairs >= 128 ? 512 : 1024); }


}  // namespace plh

using namespace plh;

extern "C" {

int plh_descriptor_distance(const uint8_t* a, const uint8_t* b) {
  int d = 0;
  for (int i = 0; i < 4; i++) {
    unsigned long long x, y;
    memcpy(&x, a + 8 * i, 8);
    memcpy(&y, b + 8 * i, 8);
    d += __builtin_popcountll(x ^ y);
  }
  return d;
}

plh_status plh_hamming_knn2_batch_dev(const uint8_t* d_q, const int32_t* d_nq, int q_cap, const uint8_t* d_t,
                                      const int32_t* d_nt, int t_cap, int pairs, int32_t* d_idx, int32_t* d_dist,
                                      void* stream) {
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  if (!d_q || !d_t || !d_idx || !d_dist || !d_nq || !d_nt || q_cap <= 0 || t_cap <= 0 || pairs <= 0) {
    set_error("plh_hamming_knn2_batch_dev: invalid argument");
    return PLH_ERR_INVALID;
  }
  dim3 grid((q_cap + 255) / 256, pairs), block(256);
  hipLaunchKernelGGL(k_knn2, grid, block, 0, (hipStream_t)stream, d_q, (const int*)d_nq, 0, q_cap, d_t, (const int*)d_nt, 0,
                     t_cap, d_idx, d_dist);
  PLH_LAUNCH_CHECK();
  return PLH_OK;
}

plh_status plh_hamming_knn2_dev(const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int32_t* d_idx, int32_t* d_dist,
                                void* stream) {
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  if (nq < 0 || nt < 0 || (nq > 0 && (!d_q || !d_idx || !d_dist)) || (nt > 0 && !d_t)) {
    set_error("plh_hamming_knn2_dev: invalid argument");
    return PLH_ERR_INVALID;
  }
  if (nq == 0) return PLH_OK;
  dim3 grid((nq + 255) / 256, 1), block(256);
  hipLaunchKernelGGL(k_knn2, grid, block, 0, (hipStream_t)stream, d_q, (const int*)nullptr, nq, nq, d_t, (const int*)nullptr,
                     nt, std::max(nt, 1), d_idx, d_dist);
  PLH_LAUNCH_CHECK();
  return PLH_OK;
}

plh_status plh_hamming_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* dist, int device) {
  if (nq < 0 || nt < 0 || (nq > 0 && (!q || !idx || !dist)) || (nt > 0 && !t)) return PLH_ERR_INVALID;
  if (nq == 0) return PLH_OK;
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  Stager st;
  plh_status rc = st.begin(device);
  if (rc != PLH_OK) return rc;
  const uint8_t* dq = st.in(q, (size_t)nq * 32);
  const uint8_t* dt = nt ? st.in(t, (size_t)nt * 32) : st.scratch<uint8_t>(32);
  int32_t* di = st.out(idx, (size_t)nq * 2);
  int32_t* dd = st.out(dist, (size_t)nq * 2);
  if ((rc = st.upload()) != PLH_OK) return rc;
  if ((rc = plh_hamming_knn2_dev(dq, nq, dt, nt, di, dd, st.stream())) != PLH_OK) return rc;
  return st.download();
}

plh_status plh_line_bfmatch_batch_dev(const int32_t* d_idx, const int32_t* d_dist, const int32_t* d_nq, const int32_t* d_nt,
                                      int q_cap, int pairs, float th, float nnratio, int32_t* d_matches, void* stream) {
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  if (!d_idx || !d_dist || !d_nq || !d_nt || !d_matches || q_cap <= 0 || pairs <= 0) return PLH_ERR_INVALID;
  hipLaunchKernelGGL(k_line_bfmatch, dim3(pairs), dim3(256), 0, (hipStream_t)stream, d_idx, d_dist, (const int*)d_nq,
                     (const int*)d_nt, q_cap, th, nnratio, d_matches);
  PLH_LAUNCH_CHECK();
  return PLH_OK;
}

size_t plh_line_search_double_workspace(int cap, int pairs) {
  // 2 x (idx + dist tables: cap x 2 int32) + 2 x matches (cap int32), per pair
  return (size_t)pairs * cap * (2 * 2 * 2 * 4 + 2 * 4) + 256;
}

plh_status plh_line_search_double_batch_dev(const uint8_t* d_desc1, const int32_t* d_n1, const uint8_t* d_desc2,
                                            const int32_t* d_n2, int cap, int pairs, float th, float nnratio,
                                            int32_t* d_matches12, int32_t* d_nmatches, void* d_workspace,
                                            size_t workspace_bytes, void* stream) {
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  if (!d_desc1 || !d_desc2 || !d_n1 || !d_n2 || !d_matches12 || !d_nmatches || !d_workspace || cap <= 0 || pairs <= 0 ||
      workspace_bytes < plh_line_search_double_workspace(cap, pairs)) {
    set_error("plh_line_search_double_batch_dev: invalid argument / workspace too small");
    return PLH_ERR_INVALID;
  }
  const size_t tab = (size_t)pairs * cap * 2;
  int32_t* idx12 = (int32_t*)d_workspace;
  int32_t* dist12 = idx12 + tab;
  int32_t* idx21 = dist12 + tab;
  int32_t* dist21 = idx21 + tab;
  int32_t* m1 = dist21 + tab;
  int32_t* m2 = m1 + (size_t)pairs * cap;
  plh_status st;
  if ((st = plh_hamming_knn2_batch_dev(d_desc1, d_n1, cap, d_desc2, d_n2, cap, pairs, idx12, dist12, stream)) != PLH_OK) return st;
  if ((st = plh_hamming_knn2_batch_dev(d_desc2, d_n2, cap, d_desc1, d_n1, cap, pairs, idx21, dist21, stream)) != PLH_OK) return st;
  if ((st = plh_line_bfmatch_batch_dev(idx12, dist12, d_n1, d_n2, cap, pairs, th, nnratio, m1, stream)) != PLH_OK) return st;
  if ((st = plh_line_bfmatch_batch_dev(idx21, dist21, d_n2, d_n1, cap, pairs, th, nnratio, m2, stream)) != PLH_OK) return st;
  hipLaunchKernelGGL(k_line_mutual, dim3(pairs), dim3(256), 0, (hipStream_t)stream, (const int32_t*)m1, (const int32_t*)m2,
                     (const int*)d_n1, (const int*)d_n2, cap, d_matches12, d_nmatches);
  PLH_LAUNCH_CHECK();
  return PLH_OK;
}

plh_status plh_orb_search_by_bow_batch_dev(const uint8_t* d_desc1, const float* d_angle1, const int32_t* d_node1,
                                           const uint8_t* d_valid1, const int32_t* d_n1, const uint8_t* d_desc2,
                                           const float* d_angle2, const int32_t* d_node2, const int32_t* d_n2, int cap,
                                           int pairs, int th_low, float nnratio, int check_ori, int32_t* d_matches21,
                                           int32_t* d_nmatches, void* stream) {
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  if (!d_desc1 || !d_angle1 || !d_node1 || !d_valid1 || !d_n1 || !d_desc2 || !d_angle2 || !d_node2 || !d_n2 ||
      !d_matches21 || !d_nmatches || cap <= 0 || cap > 6000 || pairs <= 0) {
    set_error("plh_orb_search_by_bow_batch_dev: invalid argument (cap must be in 1..6000)");
    return PLH_ERR_INVALID;
  }
  if (lds_request(k_search_by_bow, bow_lds_bytes(cap), "SearchByBoW") != PLH_OK) return PLH_ERR_INVALID;
  hipLaunchKernelGGL(k_search_by_bow, dim3(pairs), dim3(bow_threads(pairs)), bow_lds_bytes(cap), (hipStream_t)stream, d_desc1, d_angle1,
                     d_node1, d_valid1, (const int*)d_n1, d_desc2, d_angle2, d_node2, (const int*)d_n2, cap, 1, th_low, nnratio,
                     check_ori, d_matches21, d_nmatches, (const uint8_t*)nullptr, 0);
  PLH_LAUNCH_CHECK();
  return PLH_OK;
}

// Same search with the angles read from plh_keypoint records (kp.angle, as the reference does from mvKeysUn / mvKeys).
plh_status plh_orb_search_by_bow_kp_batch_dev(const uint8_t* d_desc1, const plh_keypoint* d_kps1, const int32_t* d_node1,
                                              const uint8_t* d_valid1, const int32_t* d_n1, const uint8_t* d_desc2,
                                              const plh_keypoint* d_kps2, const int32_t* d_node2, const int32_t* d_n2, int cap,
                                              int pairs, int th_low, float nnratio, int check_ori, int32_t* d_matches21,
                                              int32_t* d_nmatches, void* stream) {
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  if (!d_desc1 || !d_kps1 || !d_node1 || !d_valid1 || !d_n1 || !d_desc2 || !d_kps2 || !d_node2 || !d_n2 || !d_matches21 ||
      !d_nmatches || cap <= 0 || cap > 6000 || pairs <= 0) {
    set_error("plh_orb_search_by_bow_kp_batch_dev: invalid argument (cap must be in 1..6000)");
    return PLH_ERR_INVALID;
  }
  static_assert(sizeof(plh_keypoint) == 28, "plh_keypoint layout");
  if (lds_request(k_search_by_bow, bow_lds_bytes(cap), "SearchByBoW") != PLH_OK) return PLH_ERR_INVALID;
  hipLaunchKernelGGL(k_search_by_bow, dim3(pairs), dim3(bow_threads(pairs)), bow_lds_bytes(cap), (hipStream_t)stream, d_desc1,
                     reinterpret_cast<const float*>(d_kps1) + 3, d_node1, d_valid1, (const int*)d_n1, d_desc2,
                     reinterpret_cast<const float*>(d_kps2) + 3, d_node2, (const int*)d_n2, cap, 7, th_low, nnratio, check_ori,
                     d_matches21, d_nmatches, (const uint8_t*)nullptr, 0);
  PLH_LAUNCH_CHECK();
  return PLH_OK;
}

// ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12) (ORBmatcher.cc:574-709).
plh_status plh_orb_search_by_bow_kfkf_batch_dev(const uint8_t* d_desc1, const plh_keypoint* d_kps1, const int32_t* d_node1,
                                                const uint8_t* d_valid1, const int32_t* d_n1, const uint8_t* d_desc2,
                                                const plh_keypoint* d_kps2, const int32_t* d_node2, const uint8_t* d_valid2,
                                                const int32_t* d_n2, int cap, int pairs, int th_low, float nnratio, int check_ori,
                                                int32_t* d_matches12, int32_t* d_nmatches, void* stream) {
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  if (!d_desc1 || !d_kps1 || !d_node1 || !d_valid1 || !d_n1 || !d_desc2 || !d_kps2 || !d_node2 || !d_valid2 || !d_n2 ||
      !d_matches12 || !d_nmatches || cap <= 0 || cap > 6000 || pairs <= 0) {
    set_error("plh_orb_search_by_bow_kfkf_batch_dev: invalid argument (cap must be in 1..6000)");
    return PLH_ERR_INVALID;
  }
  if (lds_request(k_search_by_bow, bow_lds_bytes(cap), "SearchByBoW") != PLH_OK) return PLH_ERR_INVALID;
  hipLaunchKernelGGL(k_search_by_bow, dim3(pairs), dim3(bow_threads(pairs)), bow_lds_bytes(cap), (hipStream_t)stream, d_desc1,
                     reinterpret_cast<const float*>(d_kps1) + 3, d_node1, d_valid1, (const int*)d_n1, d_desc2,
                     reinterpret_cast<const float*>(d_kps2) + 3, d_node2, (const int*)d_n2, cap, 7, th_low, nnratio, check_ori,
                     d_matches12, d_nmatches, d_valid2, 1);
  PLH_LAUNCH_CHECK();
  return PLH_OK;
}

// ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, false) (ORBmatcher.cc:720-912), monocular.
plh_status plh_orb_search_for_triangulation_batch_dev(const plh_keypoint* d_kps1, const uint8_t* d_desc1, const int32_t* d_node1,
                                                      const uint8_t* d_has_mp1, const int32_t* d_n1, const plh_keypoint* d_kps2,
                                                      const uint8_t* d_desc2, const int32_t* d_node2, const uint8_t* d_has_mp2,
                                                      const int32_t* d_n2, int cap, int pairs, const float F12[9], float ex, float ey,
                                                      const float* scale_factors2, const float* level_sigma2_2, int nlevels,
                                                      int th_low, int check_ori, int32_t* d_matches12, int32_t* d_nmatches,
                                                      void* stream) {
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  if (!d_kps1 || !d_desc1 || !d_node1 || !d_has_mp1 || !d_n1 || !d_kps2 || !d_desc2 || !d_node2 || !d_has_mp2 || !d_n2 || !F12 ||
      !scale_factors2 || !level_sigma2_2 || nlevels <= 0 || nlevels > 16 || !d_matches12 || !d_nmatches || cap <= 0 || cap > 6000 ||
      pairs <= 0) {
    set_error("plh_orb_search_for_triangulation_batch_dev: invalid argument (cap in 1..6000, 1..16 levels)");
    return PLH_ERR_INVALID;
  }
  TriGeom g;
  for (int i = 0; i < 9; i++) g.F[i] = F12[i];
  g.ex = ex; g.ey = ey;
  for (int i = 0; i < 16; i++) { g.sf[i] = i < nlevels ? scale_factors2[i] : 0.f; g.sig2[i] = i < nlevels ? level_sigma2_2[i] : 0.f; }
  if (lds_request(k_search_triangulation, tri_lds_bytes(cap), "SearchForTriangulation") != PLH_OK) return PLH_ERR_INVALID;
  hipLaunchKernelGGL(k_search_triangulation, dim3(pairs), dim3(256), tri_lds_bytes(cap), (hipStream_t)stream, d_kps1, d_desc1, d_node1,
                     d_has_mp1, (const int*)d_n1, d_kps2, d_desc2, d_node2, d_has_mp2, (const int*)d_n2, cap, g, th_low, check_ori,
                     d_matches12, d_nmatches);
  PLH_LAUNCH_CHECK();
  return PLH_OK;
}

// ---- host-buffer conveniences: one call = one reference call.  The arrays go through the calling thread's staging arena
// (plh_stage.h: one packed copy up, the thread's own stream, one copy back) ----

// LSDmatcher::SearchDouble(Frame&, Frame&, vector<int>&) on two mLdesc matrices (LSDmatcher.cpp:427-460).
plh_status plh_line_search_double(const uint8_t* ldesc1, int n1, const uint8_t* ldesc2, int n2, float th, float nnratio,
                                  int32_t* matches12, int* nmatches, int device) {
  if (n1 < 0 || n2 < 0 || !nmatches || (n1 > 0 && (!ldesc1 || !matches12)) || (n2 > 0 && !ldesc2)) return PLH_ERR_INVALID;
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  *nmatches = 0;
  if (n1 == 0 || n2 == 0) return PLH_OK;   // reference: `if(ldesc1.rows == 0 || ldesc2.rows == 0) return 0;`
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  const int cap = std::max(n1, n2);
  const size_t wsb = plh_line_search_double_workspace(cap, 1);
  Stager st;
  plh_status rc = st.begin(device);
  if (rc != PLH_OK) return rc;
  const uint8_t* d1 = st.in(ldesc1, (size_t)n1 * 32);
  const uint8_t* d2 = st.in(ldesc2, (size_t)n2 * 32);
  const int32_t ns[2] = {n1, n2};
  const int32_t* dn = st.in(ns, 2);
  int32_t* dm = st.out(matches12, (size_t)n1, (size_t)cap);
  int32_t nm = 0;
  int32_t* dc = st.out(&nm, 1);
  void* ws = st.scratch<uint8_t>(wsb);
  if ((rc = st.upload()) != PLH_OK) return rc;
  rc = plh_line_search_double_batch_dev(d1, dn, d2, dn + 1, cap, 1, th, nnratio, dm, dc, ws, wsb, st.stream());
  if (rc != PLH_OK) return rc;
  if ((rc = st.download()) != PLH_OK) return rc;
  *nmatches = nm;
  return PLH_OK;
}

// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) on flat host arrays (see plh_orb_search_by_bow_batch_dev).
plh_status plh_orb_search_by_bow(const uint8_t* desc1, const float* angle1, const int32_t* node1, const uint8_t* valid1, int n1,
                                 const uint8_t* desc2, const float* angle2, const int32_t* node2, int n2, int th_low,
                                 float nnratio, int check_ori, int32_t* matches21, int* nmatches, int device) {
  if (n1 < 0 || n2 < 0 || !nmatches || (n2 > 0 && !matches21)) return PLH_ERR_INVALID;
  for (int j = 0; j < n2; j++) matches21[j] = -1;
  *nmatches = 0;
  if (n1 == 0 || n2 == 0) return PLH_OK;
  if (!desc1 || !angle1 || !node1 || !valid1 || !desc2 || !angle2 || !node2) return PLH_ERR_INVALID;
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  const int cap = std::max(n1, n2);
  Stager st;
  plh_status rc = st.begin(device);
  if (rc != PLH_OK) return rc;
  const uint8_t* d1 = st.in(desc1, (size_t)n1 * 32); const float* a1 = st.in(angle1, (size_t)n1); const int32_t* k1 = st.in(node1, (size_t)n1);
  const uint8_t* v1 = st.in(valid1, (size_t)n1);
  const uint8_t* d2 = st.in(desc2, (size_t)n2 * 32); const float* a2 = st.in(angle2, (size_t)n2); const int32_t* k2 = st.in(node2, (size_t)n2);
  const int32_t ns[2] = {n1, n2};
  const int32_t* dn = st.in(ns, 2);
  int32_t* dm = st.out(matches21, (size_t)n2, (size_t)cap);
  int32_t nm = 0;
  int32_t* dc = st.out(&nm, 1);
  if ((rc = st.upload()) != PLH_OK) return rc;
  rc = plh_orb_search_by_bow_batch_dev(d1, a1, k1, v1, dn, d2, a2, k2, dn + 1, cap, 1, th_low, nnratio, check_ori, dm, dc, st.stream());
  if (rc != PLH_OK) return rc;
  if ((rc = st.download()) != PLH_OK) return rc;
  *nmatches = nm;
  return PLH_OK;
}

// ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vpMatches12) on flat host arrays (LoopClosing.cc:ComputeSim3's call;
// see plh_orb_search_by_bow_kfkf_batch_dev): kps = mvKeysUn, node = FeatureVector node per feature, valid = carries a non-bad
// MapPoint.  matches12[n1] = feature of KeyFrame 2 whose MapPoint is paired with feature idx1 of KeyFrame 1, or -1.
plh_status plh_orb_search_by_bow_kfkf(const plh_keypoint* kps1, const uint8_t* desc1, const int32_t* node1, const uint8_t* valid1,
                                      int n1, const plh_keypoint* kps2, const uint8_t* desc2, const int32_t* node2,
                                      const uint8_t* valid2, int n2, int th_low, float nnratio, int check_ori, int32_t* matches12,
                                      int* nmatches, int device) {
  if (n1 < 0 || n2 < 0 || !nmatches || (n1 > 0 && !matches12)) return PLH_ERR_INVALID;
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  *nmatches = 0;
  if (n1 == 0 || n2 == 0) return PLH_OK;
  if (!kps1 || !desc1 || !node1 || !valid1 || !kps2 || !desc2 || !node2 || !valid2) return PLH_ERR_INVALID;
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  const int cap = std::max(n1, n2);
  Stager st;
  plh_status rc = st.begin(device);
  if (rc != PLH_OK) return rc;
  const plh_keypoint* k1 = st.in(kps1, (size_t)n1); const uint8_t* d1 = st.in(desc1, (size_t)n1 * 32); const int32_t* o1 = st.in(node1, (size_t)n1);
  const uint8_t* v1 = st.in(valid1, (size_t)n1);
  const plh_keypoint* k2 = st.in(kps2, (size_t)n2); const uint8_t* d2 = st.in(desc2, (size_t)n2 * 32); const int32_t* o2 = st.in(node2, (size_t)n2);
  const uint8_t* v2 = st.in(valid2, (size_t)n2);
  const int32_t ns[2] = {n1, n2};
  const int32_t* dn = st.in(ns, 2);
  int32_t* dm = st.out(matches12, (size_t)n1, (size_t)cap);
  int32_t nm = 0;
  int32_t* dc = st.out(&nm, 1);
  if ((rc = st.upload()) != PLH_OK) return rc;
  rc = plh_orb_search_by_bow_kfkf_batch_dev(d1, k1, o1, v1, dn, d2, k2, o2, v2, dn + 1, cap, 1, th_low, nnratio, check_ori, dm, dc, st.stream());
  if (rc != PLH_OK) return rc;
  if ((rc = st.download()) != PLH_OK) return rc;
  *nmatches = nm;
  return PLH_OK;
}

// LSDmatcher::FrameBFMatch(ldesc1, ldesc2, LineMatches, TH) (LSDmatcher.cpp:462-486): the one-directional matcher on host buffers.
plh_status plh_line_frame_bfmatch(const uint8_t* ldesc1, int n1, const uint8_t* ldesc2, int n2, float th, float nnratio,
                                  int32_t* matches12, int device) {
  if (n1 < 0 || n2 < 0 || (n1 > 0 && (!ldesc1 || !matches12)) || (n2 > 0 && !ldesc2)) return PLH_ERR_INVALID;
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  if (n1 == 0 || n2 == 0) return PLH_OK;
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  const int cap = std::max(n1, n2);
  Stager st;
  plh_status rc = st.begin(device);
  if (rc != PLH_OK) return rc;
  const uint8_t* d1 = st.in(ldesc1, (size_t)n1 * 32);
  const uint8_t* d2 = st.in(ldesc2, (size_t)n2 * 32);
  const int32_t ns[2] = {n1, n2};
  const int32_t* dn = st.in(ns, 2);
  int32_t* di = st.scratch<int32_t>((size_t)cap * 2);
  int32_t* dd = st.scratch<int32_t>((size_t)cap * 2);
  int32_t* dm = st.out(matches12, (size_t)n1, (size_t)cap);
  if ((rc = st.upload()) != PLH_OK) return rc;
  rc = plh_hamming_knn2_batch_dev(d1, dn, cap, d2, dn + 1, cap, 1, di, dd, st.stream());
  if (rc == PLH_OK) rc = plh_line_bfmatch_batch_dev(di, dd, dn, dn + 1, cap, 1, th, nnratio, dm, st.stream());
  if (rc != PLH_OK) return rc;
  return st.download();
}

// ---------------------------------------------------------------------------------------------------------------------
// LSDmatcher::FrameBFMatchNew + mutualOverlap (LSDmatcher.cpp:488-625) and SearchForTriangulationNew (:780-832) on a knn2 table.
// One lane per query: the nearest neighbour's line is intersected with the epipolar lines of the query's two end points and
// the overlap of the carried-over segment with the neighbour's own decides, beside the distance and ratio tests.  The arithmetic
// is the reference's, operation by operation (float products and sums in its element order, the normalisation `Mat /= w` as a double
// division per element, cv::norm as a double sum of squares and a double sqrt assigned to a float); the build has no contraction.
// ---------------------------------------------------------------------------------------------------------------------
struct F33 { float m[9]; };

__device__ __forceinline__ double bfnew_norm(const float* a, const float* b) {   // cv::norm(a - b), 3 x 1 CV_32F: a double
  double s = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) { const float d = a[k] - b[k]; s += (double)d * (double)d; }
  return sqrt(s);
}

// mutualOverlap (:550-625): pts[0..1] = the carried-over end points, pts[2..3] = the neighbour's.  The reference finds the outer pair
// (largest `float dist = norm(..)`, first wins) and divides the norm of the two others by it; the two others of pair k of the loop order
// (0,1) (0,2) (0,3) (1,2) (1,3) (2,3) are pair 5 - k, lower index first as in the reference's inner1 / inner2: all six norms once, no
// run-time index.
__device__ __forceinline__ float bfnew_overlap(const float (*pts)[3]) {
  const double n[6] = {bfnew_norm(pts[0], pts[1]), bfnew_norm(pts[0], pts[2]), bfnew_norm(pts[0], pts[3]),
                       bfnew_norm(pts[1], pts[2]), bfnew_norm(pts[1], pts[3]), bfnew_norm(pts[2], pts[3])};
  float maxDist = 0.0f;
  double inner = 0.0;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const float d = (float)n[k];
    if (d > maxDist) { maxDist = d; inner = n[5 - k]; }
  }
  if (maxDist < 1.0f) return 0.0f;
  return (float)(inner / (double)maxDist);
}

__global__ void __launch_bounds__(256) k_line_bfmatch_new(const int32_t* idx, const int32_t* dist, int nq, int nt, const float* seg1,
                                                          const float* seg2, const double* func2, F33 F, float TH, float nnratio,
                                                          int32_t* M) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  int m = -1;
  if (nt >= 2) {   // (the reference's loop bound size() - 1 leaves nothing to look at with one train row)
    const int t = idx[q * 2];
    const float m0 = (float)dist[q * 2], m1 = (float)dist[q * 2 + 1];
    const float l2[3] = {(float)func2[3 * t], (float)func2[3 * t + 1], (float)func2[3 * t + 2]};
    float pts[4][3];
    bool ok = true;
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const float p[3] = {seg1[4 * q + 2 * e], seg1[4 * q + 2 * e + 1], 1.0f};
      float epi[3];
#pragma unroll
      for (int r = 0; r < 3; r++) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 3; k++) acc += F.m[3 * r + k] * p[k];
        epi[r] = acc;
      }
      const float c[3] = {l2[1] * epi[2] - l2[2] * epi[1], l2[2] * epi[0] - l2[0] * epi[2], l2[0] * epi[1] - l2[1] * epi[0]};
      if (!((double)fabsf(c[2]) > 1e-12)) ok = false;
      const double w = (double)c[2];
#pragma unroll
      for (int k = 0; k < 3; k++) pts[e][k] = (float)((double)c[k] / w);
    }
    if (ok) {
      pts[2][0] = seg2[4 * t]; pts[2][1] = seg2[4 * t + 1]; pts[2][2] = 1.0f;
      pts[3][0] = seg2[4 * t + 2]; pts[3][1] = seg2[4 * t + 3]; pts[3][2] = 1.0f;
      const float score = bfnew_overlap(pts);
      if (m0 < TH && (double)score > 0.8 && m0 < nnratio * m1) m = t;
    }
  }
  M[q] = m;
}

// SearchForTriangulationNew's loop (:812-820): the mutual check if isDouble, then only pairs neither line of which has a MapLine
__global__ void __launch_bounds__(256) k_line_tri_new_resolve(const int32_t* m1, const int32_t* m2, int n1, const uint8_t* ml1,
                                                              const uint8_t* ml2, int isDouble, int32_t* out, int32_t* nmatches) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n1) return;
  int j = m1[i];
  if (j >= 0 && ((isDouble && m2[j] != i) || ml1[i] || ml2[j])) j = -1;
  out[i] = j;
  if (j >= 0) atomicAdd(nmatches, 1);
}

// LSDmatcher::FrameBFMatchNew(ldesc1, ldesc2, LineMatches, kls1, kls2, kls2func, F, TH) (LSDmatcher.cpp:488-548) on host buffers.
plh_status plh_line_frame_bfmatch_new(const uint8_t* ldesc1, int n1, const uint8_t* ldesc2, int n2, const float* seg1, const float* seg2,
                                      const double* func2, const float F[9], float th, float nnratio, int32_t* matches12, int device) {
  if (n1 < 0 || n2 < 0 || (n1 > 0 && (!ldesc1 || !matches12 || !seg1)) || (n2 > 0 && (!ldesc2 || !seg2 || !func2)) || !F) return PLH_ERR_INVALID;
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  if (n1 == 0 || n2 == 0) return PLH_OK;
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  const int cap = std::max(n1, n2);
  Stager st;
  plh_status rc = st.begin(device);
  if (rc != PLH_OK) return rc;
  const uint8_t* d1 = st.in(ldesc1, (size_t)n1 * 32);
  const uint8_t* d2 = st.in(ldesc2, (size_t)n2 * 32);
  const float* s1 = st.in(seg1, (size_t)n1 * 4);
  const float* s2 = st.in(seg2, (size_t)n2 * 4);
  const double* f2 = st.in(func2, (size_t)n2 * 3);
  const int32_t ns[2] = {n1, n2};
  const int32_t* dn = st.in(ns, 2);
  int32_t* di = st.scratch<int32_t>((size_t)cap * 2);
  int32_t* dd = st.scratch<int32_t>((size_t)cap * 2);
  int32_t* dm = st.out(matches12, (size_t)n1, (size_t)cap);
  if ((rc = st.upload()) != PLH_OK) return rc;
  rc = plh_hamming_knn2_batch_dev(d1, dn, cap, d2, dn + 1, cap, 1, di, dd, st.stream());
  if (rc != PLH_OK) return rc;
  F33 Fm;
  for (int i = 0; i < 9; i++) Fm.m[i] = F[i];
  hipLaunchKernelGGL(k_line_bfmatch_new, dim3((n1 + 255) / 256), dim3(256), 0, st.stream(), (const int32_t*)di, (const int32_t*)dd, n1, n2, s1,
                     s2, f2, Fm, th, nnratio, dm);
  PLH_LAUNCH_CHECK();
  return st.download();
}

// LSDmatcher::SearchForTriangulationNew(pKF1, pKF2, vMatchedPairs, isDouble) (LSDmatcher.cpp:780-832) on host buffers: F21 =
// ComputeF12(pKF2, pKF1) carries set 1's end points into image 2, F12 = ComputeF12(pKF1, pKF2) the other way; has_ml = GetMapLine(i) != 0.
plh_status plh_line_search_for_triangulation_new(const uint8_t* ldesc1, int n1, const uint8_t* ldesc2, int n2, const float* seg1,
                                                 const float* seg2, const double* func1, const double* func2, const float F21[9],
                                                 const float F12[9], const uint8_t* has_ml1, const uint8_t* has_ml2, float th,
                                                 float nnratio, int is_double, int32_t* matches12, int* nmatches, int device) {
  if (n1 < 0 || n2 < 0 || !nmatches || (n1 > 0 && !matches12)) return PLH_ERR_INVALID;
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  *nmatches = 0;
  if (n1 == 0 || n2 == 0) return PLH_OK;
  if (!ldesc1 || !ldesc2 || !seg1 || !seg2 || !func1 || !func2 || !F21 || !F12 || !has_ml1 || !has_ml2) return PLH_ERR_INVALID;
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  const int cap = std::max(n1, n2);
  Stager st;
  plh_status rc = st.begin(device);
  if (rc != PLH_OK) return rc;
  const uint8_t* d1 = st.in(ldesc1, (size_t)n1 * 32);
  const uint8_t* d2 = st.in(ldesc2, (size_t)n2 * 32);
  const float* s1 = st.in(seg1, (size_t)n1 * 4);
  const float* s2 = st.in(seg2, (size_t)n2 * 4);
  const double* f1 = st.in(func1, (size_t)n1 * 3);
  const double* f2 = st.in(func2, (size_t)n2 * 3);
  const uint8_t* ml1 = st.in(has_ml1, (size_t)n1);
  const uint8_t* ml2 = st.in(has_ml2, (size_t)n2);
  const int32_t ns[2] = {n1, n2};
  const int32_t* dn = st.in(ns, 2);
  int32_t* i12 = st.scratch<int32_t>((size_t)cap * 2);
  int32_t* e12 = st.scratch<int32_t>((size_t)cap * 2);
  int32_t* i21 = st.scratch<int32_t>((size_t)cap * 2);
  int32_t* e21 = st.scratch<int32_t>((size_t)cap * 2);
  int32_t* m1 = st.scratch<int32_t>((size_t)cap);
  int32_t* m2 = st.scratch<int32_t>((size_t)cap);
  int32_t* dcnt = st.scratch_zero<int32_t>(1);
  int32_t* dm = st.out(matches12, (size_t)n1, (size_t)cap);
  if ((rc = st.upload()) != PLH_OK) return rc;
  if ((rc = plh_hamming_knn2_batch_dev(d1, dn, cap, d2, dn + 1, cap, 1, i12, e12, st.stream())) != PLH_OK) return rc;
  if ((rc = plh_hamming_knn2_batch_dev(d2, dn + 1, cap, d1, dn, cap, 1, i21, e21, st.stream())) != PLH_OK) return rc;
  F33 Fa, Fb;
  for (int i = 0; i < 9; i++) { Fa.m[i] = F21[i]; Fb.m[i] = F12[i]; }
  hipLaunchKernelGGL(k_line_bfmatch_new, dim3((n1 + 255) / 256), dim3(256), 0, st.stream(), (const int32_t*)i12, (const int32_t*)e12, n1, n2,
                     s1, s2, f2, Fa, th, nnratio, m1);
  PLH_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_line_bfmatch_new, dim3((n2 + 255) / 256), dim3(256), 0, st.stream(), (const int32_t*)i21, (const int32_t*)e21, n2, n1,
                     s2, s1, f1, Fb, th, nnratio, m2);
  PLH_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_line_tri_new_resolve, dim3((n1 + 255) / 256), dim3(256), 0, st.stream(), (const int32_t*)m1, (const int32_t*)m2, n1,
                     ml1, ml2, is_double, dm, dcnt);
  PLH_LAUNCH_CHECK();
  int32_t nm = 0;
  st.fetch(&nm, (const int32_t*)dcnt, 1);
  if ((rc = st.download()) != PLH_OK) return rc;
  *nmatches = nm;
  return PLH_OK;
}

// ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, false) on host buffers: see
// plh_orb_search_for_triangulation_batch_dev.
plh_status plh_orb_search_for_triangulation(const plh_keypoint* kps1, const uint8_t* desc1, const int32_t* node1, const uint8_t* has_mp1,
                                            int n1, const plh_keypoint* kps2, const uint8_t* desc2, const int32_t* node2,
                                            const uint8_t* has_mp2, int n2, const float F12[9], float ex, float ey,
                                            const float* scale_factors2, const float* level_sigma2_2, int nlevels, int th_low,
                                            int check_ori, int32_t* matches12, int* nmatches, int device) {
  if (n1 < 0 || n2 < 0 || !nmatches || (n1 > 0 && !matches12)) return PLH_ERR_INVALID;
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  *nmatches = 0;
  if (n1 == 0 || n2 == 0) return PLH_OK;
  if (!kps1 || !desc1 || !node1 || !has_mp1 || !kps2 || !desc2 || !node2 || !has_mp2 || !F12 || !scale_factors2 || !level_sigma2_2)
    return PLH_ERR_INVALID;
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  const int cap = std::max(n1, n2);
  Stager st;
  plh_status rc = st.begin(device);
  if (rc != PLH_OK) return rc;
  const plh_keypoint* k1 = st.in(kps1, (size_t)n1); const uint8_t* d1 = st.in(desc1, (size_t)n1 * 32); const int32_t* o1 = st.in(node1, (size_t)n1);
  const uint8_t* v1 = st.in(has_mp1, (size_t)n1);
  const plh_keypoint* k2 = st.in(kps2, (size_t)n2); const uint8_t* d2 = st.in(desc2, (size_t)n2 * 32); const int32_t* o2 = st.in(node2, (size_t)n2);
  const uint8_t* v2 = st.in(has_mp2, (size_t)n2);
  const int32_t ns[2] = {n1, n2};
  const int32_t* dn = st.in(ns, 2);
  int32_t* dm = st.out(matches12, (size_t)n1, (size_t)cap);
  int32_t nm = 0;
  int32_t* dc = st.out(&nm, 1);
  if ((rc = st.upload()) != PLH_OK) return rc;
  rc = plh_orb_search_for_triangulation_batch_dev(k1, d1, o1, v1, dn, k2, d2, o2, v2, dn + 1, cap, 1, F12, ex, ey, scale_factors2,
                                                  level_sigma2_2, nlevels, th_low, check_ori, dm, dc, st.stream());
  if (rc != PLH_OK) return rc;
  if ((rc = st.download()) != PLH_OK) return rc;
  *nmatches = nm;
  return PLH_OK;
}

// ---- resident frames (frame_resident.h): the two-frame matchers of the tracking path on frames that already lie on the device ----
// ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vpMapPointMatches) (ORBmatcher.cc:187-327): kf / f carry their FeatureVector
// nodes (plh_frame_points_set_nodes); valid1[i] = the KeyFrame's feature i holds a non-bad MapPoint.
plh_status plh_orb_search_by_bow_resident(const plh_frame_points* kf, const uint8_t* valid1, const plh_frame_points* f, int th_low,
                                          float nnratio, int check_ori, int32_t* matches21, int* nmatches) {
  if (!kf || !f || !nmatches || (f->n > 0 && !matches21) || kf->device != f->device) return PLH_ERR_INVALID;
  const int n1 = kf->n, n2 = f->n;
  for (int j = 0; j < n2; j++) matches21[j] = -1;
  *nmatches = 0;
  if (n1 == 0 || n2 == 0) return PLH_OK;
  if (!valid1 || !kf->hasNodes || !f->hasNodes) { set_error("plh_orb_search_by_bow_resident: a frame has no FeatureVector nodes (plh_frame_points_set_nodes)"); return PLH_ERR_INVALID; }
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  const int cap = std::max(n1, n2);
  Stager st;
  plh_status rc = st.begin(f->device);
  if (rc != PLH_OK) return rc;
  const uint8_t* v1 = st.in(valid1, (size_t)n1);
  int32_t* dm = st.out(matches21, (size_t)n2, (size_t)cap);
  int32_t nm = 0;
  int32_t* dc = st.out(&nm, 1);
  if ((rc = st.upload()) != PLH_OK) return rc;
  rc = plh_orb_search_by_bow_kp_batch_dev(kf->desc, kf->kps, kf->node, v1, kf->dn, f->desc, f->kps, f->node, f->dn, cap, 1, th_low, nnratio,
                                          check_ori, dm, dc, st.stream());
  if (rc != PLH_OK) return rc;
  if ((rc = st.download()) != PLH_OK) return rc;
  *nmatches = nm;
  return PLH_OK;
}
// LSDmatcher::SearchDouble(Frame&, Frame&, LineMatches) / (KeyFrame*, Frame&) (LSDmatcher.cpp:375-460) on two resident line sets.
plh_status plh_line_search_double_resident(const plh_frame_lines* l1, const plh_frame_lines* l2, float th, float nnratio, int32_t* matches12,
                                           int* nmatches) {
  if (!l1 || !l2 || !nmatches || (l1->nl > 0 && !matches12) || l1->device != l2->device) return PLH_ERR_INVALID;
  const int n1 = l1->nl, n2 = l2->nl;
  for (int i = 0; i < n1; i++) matches12[i] = -1;
  *nmatches = 0;
  if (n1 == 0 || n2 == 0) return PLH_OK;
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  const int cap = std::max(n1, n2);
  const size_t wsb = plh_line_search_double_workspace(cap, 1);
  Stager st;
  plh_status rc = st.begin(l1->device);
  if (rc != PLH_OK) return rc;
  int32_t* dm = st.out(matches12, (size_t)n1, (size_t)cap);
  int32_t nm = 0;
  int32_t* dc = st.out(&nm, 1);
  void* ws = st.scratch<uint8_t>(wsb);
  if ((rc = st.upload()) != PLH_OK) return rc;
  rc = plh_line_search_double_batch_dev(l1->ldesc, l1->dn, l2->ldesc, l2->dn, cap, 1, th, nnratio, dm, dc, ws, wsb, st.stream());
  if (rc != PLH_OK) return rc;
  if ((rc = st.download()) != PLH_OK) return rc;
  *nmatches = nm;
  return PLH_OK;
}

}  // extern "C"
