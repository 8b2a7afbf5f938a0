// Line kernels, stage 2b: the rectangles of the line-support regions that region growing kept.
//
// cv::LineSegmentDetector evaluates region2rect() (centroid, inertia, get_theta, extents) right after region_grow(), but what
// region growing needs from it is the density decision, which k_lsd_grow* takes from a float bracket of the density whenever
// the bracket is clear (lsd_density_screen, lsd_grow.hip).  The rectangle of a region that is kept changes no mark and no later
// decision, so it does not have to be computed by the wavefront that owns the frame -- where its sequential double sums ran on
// 3 of 64 lanes -- nor in the frame's order.  k_lsd_grow* leaves an LsdRegionEntry per kept region (lsd_rect_dev.h: where its
// pixels lie in the frame's log, how many, its region angle) in the region's segment slot; this kernel evaluates them ONE LANE
// PER REGION: every lane runs the reference's sequential sums for its own region, in region order (bit-identical doubles), 64
// regions per wavefront instead of one.  Regions are handed to the lanes sorted by size class so that the lanes of a wavefront
// loop about equally long.
//
// LSD_REFINE_ADV (rect_improve / rect_nfa / nfa, lsd_rect_dev.h) reads the immutable level-line field only and decides only
// whether the segment is kept: k_lsd_rects_adv leaves the rectangle in an LsdAdvRec, and lsd_adv.hip runs rect_nfa() / nfa() /
// rect_improve() on them, followed by a stable compaction of the surviving segments.
#include <algorithm>
#include <cstdlib>

#include "lsd_rect_dev.h"

namespace plh {

constexpr int RC_BINS = 128;

// floor(log2) of the size and its next two bits: sizes within a class differ by less than a quarter
__device__ __forceinline__ int rc_size_class(unsigned cnt) {
  const unsigned c = cnt < 4u ? 4u : cnt;
  const int l = 31 - __clz((int)c);
  return min(4 * (l - 2) + (int)((c >> (l - 2)) & 3u), RC_BINS - 1);
}

// ---------------------------------------------------------------------------------------------
// The kernels (k_lsd_rects_sort, k_lsd_rects[_adv] below).
//   1. the entries are sorted by size class (largest first), 64 consecutive ones go to the lanes of a wavefront;
//   2. the wavefront stages its 64 regions' pixels through LDS, RC_K per lane at a time: packed coordinates from the frame's log
//      and gx^2 + gy^2 from the array beside it (written by region growing, which had the record in hand), loaded 8 regions x 8
//      consecutive entries per instruction -- a lane per region reading its own stream touches 64 cache lines per load and left
//      the first version of this file bound by the texture addresser (5.3 ms per 1536 frames, profiles/r04_screen_ab_v1...);
//   3. every lane adds its own region's terms from LDS in region order.  Rows of the staging tiles are padded with zero terms,
//      which the sums absorb exactly (+0.0; the accumulators are never -0.0), as lsd_chain_add does.
// No record, no table entry is read here: the level-line field stays with region growing.
//
// Round 6, second half -- the log is read in whole 64-byte sectors, once per pass.  Rounds 4 - 5 read a region's stream 8 entries
// (32 bytes, anywhere in the log) per round trip and kept the weights of the first pass in a plane of doubles for the second: PMC
// traffic 7.3 MB per frame for 0.12 M region pixels (profiles/hbm_traffic.json of build b6c019a1) -- every sector of the log came
// in twice per pass (its two halves one round trip apart, with 6 000 wavefronts' streams through a 4 MB L2 in between) and the
// weight plane cost 8 bytes written + 8 read per pixel.  Now a region's chunks are cut at ABSOLUTE multiples of 16 log entries (the
// first chunk of a region starts up to 15 padding entries early), two chunks = one sector per stream are requested in the same
// load phase, and the second pass takes gx^2 + gy^2 from the log again and evaluates the weight a second time -- the expression
// the gradient table was filled with, q_modgrad(), on the lane that adds it -- instead of reading it back: 8 + 8 + 4 bytes per pixel
// fetched once, nothing written but the rectangles, and 5.1 KB of LDS per wavefront instead of 7.4.
// ---------------------------------------------------------------------------------------------
constexpr int RC_K = 8;
constexpr int RC_PITCH = RC_K + 1;

struct RcStage {   // one wavefront's part of the block's LDS
  uint32_t* p;     // [64][RC_PITCH] packed pixels
  uint32_t* q;     // [64][RC_PITCH] gx^2 + gy^2
  uint32_t* off;   // [64] log offset of lane r's region, rounded down to a multiple of 16 entries (a sector of the log)
  uint32_t* lohi;  // [64] the region's entries inside its padded stream: [lo, hi) = lo | hi << 16 (0: the lane has no region)
};
struct RcLoad {    // a pair of chunks (one sector per region and stream) on its way from global memory to the staging tiles
  uint32_t p[16], q[16];
};

// MODE 0: coordinates only (extents); 1: coordinates + gx^2 + gy^2.  Chunks c and c + 1 (c even) of all 64 regions.
template <int MODE>
__device__ __forceinline__ void rc_load(const RcStage& st, const uint32_t* log, const uint32_t* logq, int lane, int c, RcLoad& L) {
#pragma unroll
  for (int g = 0; g < 8; g++) {
    const int r = 8 * g + (lane >> 3);
    const uint32_t lh = st.lohi[r], base = st.off[r];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const uint32_t j = (uint32_t)((c + h) * RC_K + (lane & 7));
      const bool valid = j >= (lh & 0xffffu) && j < (lh >> 16);
      const uint32_t a = base + j;
      L.p[2 * g + h] = valid ? log[a] : 0u;
      if (MODE == 1) L.q[2 * g + h] = valid ? logq[a] : 0u;
    }
  }
}
template <int MODE>
__device__ __forceinline__ void rc_put(const RcStage& st, int lane, int h, const RcLoad& L) {
#pragma unroll
  for (int g = 0; g < 8; g++) {
    const int r = 8 * g + (lane >> 3), k = lane & 7;
    st.p[r * RC_PITCH + k] = L.p[2 * g + h];
    if (MODE == 1) st.q[r * RC_PITCH + k] = L.q[2 * g + h];   // (q = 0 for the padding: weight +0.0)
  }
}
// One pass over the 64 regions' streams: `sum(c)` adds chunk c from the lane's row of the staging tiles.
template <int MODE, typename F>
__device__ __forceinline__ void rc_pass(const RcStage& st, const uint32_t* log, const uint32_t* logq, int lane, int chunks, F&& sum) {
  RcLoad L;
  rc_load<MODE>(st, log, logq, lane, 0, L);
  for (int c = 0; c < chunks; c += 2) {
    rc_put<MODE>(st, lane, 0, L);
    PLH_WAVE_SYNC();
    sum(c);
    PLH_WAVE_SYNC();
    if (c + 1 >= chunks) break;
    rc_put<MODE>(st, lane, 1, L);
    PLH_WAVE_SYNC();
    if (c + 2 < chunks) rc_load<MODE>(st, log, logq, lane, c + 2, L);   // the next sectors are in flight while this chunk is added up
    sum(c + 1);
    PLH_WAVE_SYNC();
  }
}

// region2rect() + get_theta() of the 64 regions of a wavefront (lane = region; cnt 0 = none): oracle/lsd.cc region2rect, the same
// expressions in the same order as lsd_region2rect (lsd_grow.hip).  rec = x1 y1 x2 y2 width theta dx dy.
__device__ __forceinline__ void rc_wave_region2rect(const RcStage& st, const uint32_t* log, const uint32_t* logq, int lane,
                                                    uint32_t myOff, int myCnt, double reg_angle, double prec, double rec[8]) {
  PLH_WAVE_SYNC();
  const int myLo = myCnt > 0 ? (int)(myOff & 15u) : 0, myHi = myLo + myCnt;   // (regions of this path are < 512 pixels: hi < 2^16)
  st.off[lane] = myOff & ~15u; st.lohi[lane] = (uint32_t)myLo | ((uint32_t)myHi << 16);
  int maxHi = myHi;
  for (int m = 32; m >= 1; m >>= 1) maxHi = max(maxHi, __shfl_xor(maxHi, m));
  PLH_WAVE_SYNC();
  const int chunks = (maxHi + RC_K - 1) / RC_K;
  const uint32_t* sp = st.p + lane * RC_PITCH;
  const uint32_t* sq = st.q + lane * RC_PITCH;
  double sx = 0, sy = 0, sum = 0;
  rc_pass<1>(st, log, logq, lane, chunks, [&](int) {
#pragma unroll
    for (int k = 0; k < RC_K; k++) {
      const uint32_t p = sp[k];
      const double w = q_modgrad(sq[k]);
      sx += (double)(int)(p & 0xffffu) * w;
      sy += (double)(int)(p >> 16) * w;
      sum += w;
    }
  });
  const double x = sx / sum, y = sy / sum;
  double Ixx = 0, Iyy = 0, Ixy = 0;
  rc_pass<1>(st, log, logq, lane, chunks, [&](int) {
#pragma unroll
    for (int k = 0; k < RC_K; k++) {
      const uint32_t p = sp[k];
      const double w = q_modgrad(sq[k]);
      const double ddx = (double)(int)(p & 0xffffu) - x, ddy = (double)(int)(p >> 16) - y;
      Ixx += ddy * ddy * w;
      Iyy += ddx * ddx * w;
      Ixy += -(ddx * ddy * w);   // Ixy -= v  ==  Ixy += -v
    }
  });
  const double theta = lsd_rect_theta(Ixx, Iyy, Ixy, reg_angle, prec);
  const D2 cs = lsd_sincos_inl(theta);
  const double dx = cs.x, dy = cs.y;
  double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
  rc_pass<0>(st, log, logq, lane, chunks, [&](int c) {
#pragma unroll
    for (int k = 0; k < RC_K; k++) {
      const int j = c * RC_K + k;
      if (j >= myLo && j < myHi) {
        const uint32_t p = sp[k];
        const double rdx = (double)(int)(p & 0xffffu) - x, rdy = (double)(int)(p >> 16) - y;
        const double l = rdx * dx + rdy * dy;
        const double w = -rdx * dy + rdy * dx;
        l_max = fmax(l_max, l); l_min = fmin(l_min, l);
        w_max = fmax(w_max, w); w_min = fmin(w_min, w);
      }
    }
  });
  rec[0] = x + l_min * dx; rec[1] = y + l_min * dy;
  rec[2] = x + l_max * dx; rec[3] = y + l_max * dy;
  const double width = w_max - w_min;
  rec[4] = width < 1.0 ? 1.0 : width;
  rec[5] = theta; rec[6] = dx; rec[7] = dy;
}

// The same rectangle for ONE region by a whole wavefront -- for the few regions of a frame that are hundreds or thousands of pixels:
// in the lane-per-region form above such a region keeps its lane (and with it the wavefront) busy for cnt / RC_K dependent
// trips to global memory per pass, and the kernel's duration is that one wavefront's.  Here 64 consecutive log entries are
// loaded at once, every lane forms its element's terms (the products of the reference, rounded as there), and lanes 0 .. 2 add
// the three series from LDS in element order -- lsd_region2rect's scheme (lsd_grow.hip), the same doubles as one lane adding them.
// T: [3][64] doubles of LDS.  The result is uniform over the wavefront.
__device__ __forceinline__ double rc_chain_add(const double* T, int lane, double acc, int n) {
  const int ch = min(lane, 2);
  const D2* row = reinterpret_cast<const D2*>(T + ch * 64);
  const int n2 = ((n + 7) >> 3) << 2;   // (the terms behind the end are +0.0: exact, the accumulators are never -0.0)
  for (int l = 0; l < n2; l += 4) {
    const D2 v0 = row[l], v1 = row[l + 1], v2 = row[l + 2], v3 = row[l + 3];
    acc += v0.x; acc += v0.y; acc += v1.x; acc += v1.y;
    acc += v2.x; acc += v2.y; acc += v3.x; acc += v3.y;
  }
  return acc;
}
__device__ __forceinline__ double rc_wave_max(double v) {
  for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m));
  return v;
}
__device__ __forceinline__ void rc_region2rect_by_wave(double* T, const uint32_t* log, const uint32_t* logq, int lane, uint32_t off, int cnt,
                                                       double reg_angle, double prec, double rec[8]) {
  double acc = 0;
  for (int base = 0; base < cnt; base += 64) {
    const int i = base + lane;
    double w = 0, wx = 0, wy = 0;
    if (i < cnt) {
      const uint32_t p = log[off + (uint32_t)i];
      w = q_modgrad(logq[off + (uint32_t)i]);
      wx = (double)(int)(p & 0xffffu) * w;
      wy = (double)(int)(p >> 16) * w;
    }
    PLH_WAVE_SYNC();
    T[lane] = wx; T[64 + lane] = wy; T[128 + lane] = w;
    PLH_WAVE_SYNC();
    acc = rc_chain_add(T, lane, acc, min(64, cnt - base));
  }
  const double sum = bcast_f64(acc, 2);
  const double x = bcast_f64(acc, 0) / sum, y = bcast_f64(acc, 1) / sum;
  acc = 0;
  for (int base = 0; base < cnt; base += 64) {
    const int i = base + lane;
    double a = 0, b = 0, cc = 0;
    if (i < cnt) {
      const uint32_t p = log[off + (uint32_t)i];
      const double w = q_modgrad(logq[off + (uint32_t)i]);
      const double ddx = (double)(int)(p & 0xffffu) - x, ddy = (double)(int)(p >> 16) - y;
      a = ddy * ddy * w;
      b = ddx * ddx * w;
      cc = -(ddx * ddy * w);   // Ixy -= v  ==  Ixy += -v
    }
    PLH_WAVE_SYNC();
    T[lane] = a; T[64 + lane] = b; T[128 + lane] = cc;
    PLH_WAVE_SYNC();
    acc = rc_chain_add(T, lane, acc, min(64, cnt - base));
  }
  const double Ixx = bcast_f64(acc, 0), Iyy = bcast_f64(acc, 1), Ixy = bcast_f64(acc, 2);
  const double theta = lsd_rect_theta(Ixx, Iyy, Ixy, reg_angle, prec);
  const D2 cs = lsd_sincos_inl(theta);
  const double dx = cs.x, dy = cs.y;
  double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
  for (int i = lane; i < cnt; i += 64) {
    const uint32_t p = log[off + (uint32_t)i];
    const double rdx = (double)(int)(p & 0xffffu) - x, rdy = (double)(int)(p >> 16) - y;
    const double l = rdx * dx + rdy * dy;
    const double w = -rdx * dy + rdy * dx;
    l_max = fmax(l_max, l); l_min = fmin(l_min, l);
    w_max = fmax(w_max, w); w_min = fmin(w_min, w);
  }
  l_max = rc_wave_max(l_max); l_min = -rc_wave_max(-l_min);   // (extremes: exact in any order)
  w_max = rc_wave_max(w_max); w_min = -rc_wave_max(-w_min);
  rec[0] = x + l_min * dx; rec[1] = y + l_min * dy;
  rec[2] = x + l_max * dx; rec[3] = y + l_max * dy;
  const double width = w_max - w_min;
  rec[4] = width < 1.0 ? 1.0 : width;
  rec[5] = theta; rec[6] = dx; rec[7] = dy;
}

// ADV = false: LSD_REFINE_STD, every rectangle is a segment.  ADV = true: the rectangle goes to its slot's LsdAdvRec; lsd_adv.hip
// takes it from there.
// Round 5: two kernels.  The one-block-per-frame form (256 threads, 46.6 KB of LDS, 170 registers) could not be placed on a CU
// whose LDS the region-growing wavefronts of the other sub-batches hold (24 x 5 KiB of 160): alone 2.0 ms per 1536 frames, inside
// the pipeline 16 ms on the line chain's critical path.  k_lsd_rects_sort leaves the size-class order in the frame's park area
// (a light block per frame, 512 bytes of LDS); k_lsd_rects_sums runs ONE WAVEFRONT per block (7.4 KB of LDS) on 64 consecutive
// entries of that order, a few blocks per frame (launch_lsd_rects) taking the groups round robin (the largest regions first, so the blocks of a
// frame end together).  Which lane evaluates a region changes nothing about its sums.
constexpr int RC_GROUPS_MAX = 32, RC_GROUPS_MIN = 4;   // blocks per frame: launch_lsd_rects picks by batch size (below)
// regions of this size class and above (>= 512 pixels: a few dozen per busy frame) are evaluated one per wavefront, the rest one
// per lane (64 regions of similar size per wavefront)
constexpr int RC_BIG_CLASS = 4 * (9 - 2);

__global__ void __launch_bounds__(256) k_lsd_rects_sort(LineDeviceArgs a) {
  __shared__ int s_hist[RC_BINS];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = min(a.nSegs[b], a.segCap);
  const uint4* ent = reinterpret_cast<const uint4*>(a.segs + (long long)b * a.arenaStride);
  uint32_t* order = a.park + (long long)b * a.arenaStride + 2;   // (line_plan.h: park = [list length, -, order[segCap], work list[segCap]])
  if (tid == 0) order[-2] = 0u;   // park[0]: LSD_REFINE_ADV's work list starts empty (k_adv_first appends, lsd_adv.hip)
  for (int i = tid; i < RC_BINS; i += 256) s_hist[i] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += 256) atomicAdd(&s_hist[rc_size_class(ent[i].y)], 1);
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int k = RC_BINS - 1; k >= 0; k--) {
      if (k == RC_BIG_CLASS - 1) order[-1] = (uint32_t)acc;   // park[1]: how many entries lie in the classes above (the order's head)
      const int c = s_hist[k]; s_hist[k] = acc; acc += c;
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += 256) order[atomicAdd(&s_hist[rc_size_class(ent[i].y)], 1)] = (uint32_t)i;
}

template <bool ADV>
__device__ __forceinline__ void lsd_rects_group(const LineDeviceArgs& a) {
  __shared__ __attribute__((aligned(16))) uint32_t s_pq[2 * 64 * RC_PITCH];   // the staging tiles; rc_region2rect_by_wave's three
  uint32_t* const s_p = s_pq;                                                  // series (3 x 64 doubles) lie over them
  uint32_t* const s_q = s_pq + 64 * RC_PITCH;
  double* const s_t = reinterpret_cast<double*>(s_pq);
  static_assert(2 * 64 * RC_PITCH * 4 >= 3 * 64 * 8, "series buffer");
  __shared__ uint32_t s_off[64];
  __shared__ uint32_t s_lohi[64];
  const int b = blockIdx.y, lane = threadIdx.x;
  const int n = min(a.nSegs[b], a.segCap);
  uint4* ent = reinterpret_cast<uint4*>(a.segs + (long long)b * a.arenaStride);
  const uint32_t* log = a.reg + (long long)b * a.arenaStride;
  const uint32_t* logq = a.regq + (long long)b * a.arenaStride;
  const uint32_t* order = a.park + (long long)b * a.arenaStride + 2;
  RcStage st;
  st.p = s_p; st.q = s_q; st.off = s_off; st.lohi = s_lohi;
  auto put = [&](int slot, const double* rec) {
    if constexpr (!ADV) {
      lsd_store_segment(&ent[slot], rec);
    } else {
      LsdAdvRec* ar = a.adv + (long long)b * a.segCap + slot;
#pragma unroll
      for (int k2 = 0; k2 < 8; k2++) ar->r[k2] = rec[k2];
      ar->r[8] = a.prec; ar->r[9] = a.p;
    }
  };
  // the head of the order: one big region per wavefront at a time (the blocks of a frame take them round robin) ...
  const int nBig = min((int)order[-1], n);
  for (int k = (int)blockIdx.x; k < nBig; k += (int)gridDim.x) {
    const int slot = (int)order[k];
    const uint4 e = ent[slot];   // LsdRegionEntry (uniform)
    double rec[8];
    rc_region2rect_by_wave(s_t, log, logq, lane, e.x, (int)e.y, (double)__uint_as_float(e.z) * kDegToRads, a.prec, rec);
    if (lane == 0) put(slot, rec);
  }
  // ... then the rest, 64 consecutive entries of the order per wavefront, one region per lane
  for (int base = nBig + (int)blockIdx.x * 64; base < n; base += 64 * (int)gridDim.x) {
    const int k = base + lane;
    int slot = -1;
    uint4 e = uint4{0u, 0u, 0u, 0u};
    if (k < n) { slot = (int)order[k]; e = ent[slot]; }   // LsdRegionEntry
    double rec[8];
    rc_wave_region2rect(st, log, logq, lane, e.x, (int)e.y, (double)__uint_as_float(e.z) * kDegToRads, a.prec, rec);
    if (slot >= 0) put(slot, rec);
  }
}
#if defined(PLH_RECTS_WAVES) && !defined(HIPEMU)   // (A/B builds)
#define PLH_RECTS_ATTR __attribute__((amdgpu_waves_per_eu(PLH_RECTS_WAVES)))
#else
#define PLH_RECTS_ATTR
#endif
__global__ void __launch_bounds__(64) PLH_RECTS_ATTR k_lsd_rects(LineDeviceArgs a) { lsd_rects_group<false>(a); }
__global__ void __launch_bounds__(64) PLH_RECTS_ATTR k_lsd_rects_adv(LineDeviceArgs a) { lsd_rects_group<true>(a); }

void launch_lsd_adv(const LineDeviceArgs& a, hipStream_t s);   // lsd_adv.hip
// One-wavefront blocks per frame for the rectangle / LSD_REFINE_ADV kernels: as many as it takes to put ~ 4096 wavefronts on the GPU
// (a lone frame: `hi` blocks, its regions side by side), but no more -- every resident block holds its LDS tile and a wave slot while
// it waits for memory, and LDS is what the two halves of the front end share on a CU: at 1536 frames per launch 32 blocks per frame
// instead of 4 cost k_lsd_rects 2.1 -> 4.7 ms alone and THE WHOLE FRONT END 3.5 % (profiles/r05_blocks_per_frame_ab.txt).
int lsd_blocks_per_frame(int batch, int lo, int hi) {
#if defined(HIPEMU) || defined(PLH_GROW_PROF)
  // the CPU emulator build (and the counter build of tools/) can ask for the large-batch shape on a small batch, so that the tests
  // walk the several-chunks-per-block paths; the product build has no such knob
  if (getenv("PLH_BLOCKS_PER_FRAME_MIN")) return lo;
#endif
  return std::max(lo, std::min(hi, 4096 / std::max(batch, 1)));
}
void launch_lsd_rects(const LineDeviceArgs& a, hipStream_t s) {
  const int groups = lsd_blocks_per_frame(a.batch, RC_GROUPS_MIN, RC_GROUPS_MAX);
  hipLaunchKernelGGL(k_lsd_rects_sort, dim3(a.batch), dim3(256), 0, s, a);
  if (!a.refineAdv) {
    hipLaunchKernelGGL(k_lsd_rects, dim3(groups, a.batch), dim3(64), 0, s, a);
    return;
  }
  hipLaunchKernelGGL(k_lsd_rects_adv, dim3(groups, a.batch), dim3(64), 0, s, a);
  launch_lsd_adv(a, s);
}

}  // namespace plh
