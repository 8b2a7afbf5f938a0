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
// whether the segment is kept: it runs here as well, in two kernels -- the first rect_nfa() of every rectangle in the lane that
// computed it (k_lsd_rects_adv); the rectangles it does not pass (about one in ten) are parked and improved by k_lsd_improve,
// the five variants of each rect_improve() stage side by side -- followed by a stable compaction of the surviving segments.
#include "lsd_rect_dev.h"

namespace plh {

constexpr int RC_CHUNK = 4096;   // entries sorted and evaluated at a time
constexpr int RC_BINS = 128;
constexpr uint32_t RC_DROPPED = 0xffffffffu;   // first word of a slot whose rectangle LSD_REFINE_ADV rejected (a NaN: never a coordinate)

struct RcFrame {
  const uint32_t* P;
  const LsdAngleEntry* A;
  int spitch, sw, sh;
};

// floor(log2) of the size and its next two bits: sizes within a class differ by less than a quarter
__device__ __forceinline__ int rc_size_class(unsigned cnt) {
  const unsigned c = cnt < 4u ? 4u : cnt;
  const int l = 31 - __clz((int)c);
  return min(4 * (l - 2) + (int)((c >> (l - 2)) & 3u), RC_BINS - 1);
}

// rect_nfa() (oracle/lsd.cc rect_nfa, with the published code's quirks: integer scan-line steps, the tail point's x where a y is
// meant).  The scan-line bounds advance by integer steps from an integer start, so row y's span is a closed form of the number
// of rows walked before it; rows outside the image are skipped before the step update (`continue`), so they do not count.
__device__ __attribute__((noinline)) double rc_rect_nfa(const RcFrame& f, const LsdAdvRect& r, double logNT) {
  const double hw = r.width / 2.0, dyhw = r.dy * hw, dxhw = r.dx * hw;
  int ox[4] = {(int)(r.x1 - dyhw), (int)(r.x2 - dyhw), (int)(r.x2 + dyhw), (int)(r.x1 + dyhw)};
  int oy[4] = {(int)(r.y1 + dxhw), (int)(r.y2 + dxhw), (int)(r.y2 - dxhw), (int)(r.y1 - dxhw)};
  // std::sort by (x, y) ascending: a sorting network on four elements
#define LSD_CSWAP(i, j)                                                          \
  if (ox[j] < ox[i] || (ox[j] == ox[i] && oy[j] < oy[i])) {                      \
    const int tx = ox[i], ty = oy[i];                                            \
    ox[i] = ox[j]; oy[i] = oy[j]; ox[j] = tx; oy[j] = ty;                        \
  }
  LSD_CSWAP(0, 1) LSD_CSWAP(2, 3) LSD_CSWAP(0, 2) LSD_CSWAP(1, 3) LSD_CSWAP(1, 2)
#undef LSD_CSWAP
  int iMin = 0, iMax = 0;
  for (int i = 1; i < 4; ++i) {
    if (oy[iMin] > oy[i]) iMin = i;
    if (oy[iMax] < oy[i]) iMax = i;
  }
  unsigned taken = 1u << iMin;
  int iL = -1, iR = -1, iT = -1;
  for (int i = 0; i < 4; ++i)
    if (!((taken >> i) & 1u)) { if (iL < 0) iL = i; else if (ox[iL] > ox[i]) iL = i; }
  taken |= 1u << iL;
  for (int i = 0; i < 4; ++i)
    if (!((taken >> i) & 1u)) { if (iR < 0) iR = i; else if (ox[iR] < ox[i]) iR = i; }
  taken |= 1u << iR;
  for (int i = 0; i < 4; ++i)
    if (!((taken >> i) & 1u)) { if (iT < 0) iT = i; else if (ox[iT] > ox[i]) iT = i; }
  const int mx = ox[iMin], my = oy[iMin], lx = ox[iL], ly = oy[iL], rx = ox[iR], ry = oy[iR], tx = ox[iT];
  // integer divisions, and the tail point's x where a y is meant: as published
  const long long fl = (my != ly) ? (mx - lx) / (my - ly) : 0, sl = (ly != tx) ? (lx - tx) / (ly - tx) : 0;
  const long long fr = (my != ry) ? (mx - rx) / (my - ry) : 0, sr = (ry != tx) ? (rx - tx) / (ry - tx) : 0;
  const int yA = max(my, 0), yB = min(oy[iMax], f.sh - 1);   // the scan lines inside the image
  int total = 0, alg = 0;
  for (int y = yA; y <= yB; ++y) {
    // steps taken in front of row y: one per row of [yA, y), the first kind for the rows above the left (right) corner
    const long long j = (long long)(y - yA);
    long long nl = (long long)min(y, ly) - yA, nr = (long long)min(y, ry) - yA;
    nl = nl < 0 ? 0 : nl; nr = nr < 0 ? 0 : nr;
    const long long left = mx + fl * nl + sl * (j - nl), right = mx + fr * nr + sr * (j - nr);
    const int xa = (int)(left < 0 ? 0 : left), xb = (int)(right > f.sw - 1 ? f.sw - 1 : right);
    const uint32_t* row = f.P + __umul24((unsigned)y, (unsigned)f.spitch);
    for (int x = xa; x <= xb; ++x) {
      ++total;
      const unsigned rec = row[x];
      if ((rec & LSD_REC_DEF) && lsd_aligned(r.theta, (double)f.A[rec & LSD_REC_IDX].angf * kDegToRads, r.prec)) ++alg;
    }
  }
  return lsd_nfa(total, alg, r.p, logNT);
}

// ---------------------------------------------------------------------------------------------
// The kernels.  One block (4 wavefronts) per frame.
//   1. weights: one coalesced pass over the frame's log computes every kept pixel's modgrad -- sqrt((gx^2 + gy^2) / 4.0) from the
//      record's table index, the expression the table itself was filled with (line_kernels.hip k_lsd_angle_table) -- into W
//      (doubles, the seed-list + scratch areas of the frame's block: both are free after region growing);
//   2. the entries are sorted by size class (largest first), 64 consecutive ones go to the lanes of a wavefront;
//   3. the wavefront stages its 64 regions' pixels and weights through LDS, RC_K per lane at a time -- loaded 8 regions x 8
//      consecutive pixels per instruction (a lane per region reading its own stream would touch 64 cache lines per load and
//      leave the kernel bound by the texture addresser: 5.3 ms per 1536 frames in the first version of this file, r04_screen_ab_v1)
//      -- and every lane adds its own region's terms from LDS in region order.  Rows of the staging tiles are padded with
//      zero terms, which the sums absorb exactly (+0.0; the accumulators are never -0.0), as lsd_chain_add does.
// ---------------------------------------------------------------------------------------------
constexpr int RC_K = 8;
constexpr int RC_PITCH = RC_K + 1;

struct RcStage {   // one wavefront's part of the block's LDS
  uint32_t* p;     // [64][RC_PITCH] packed pixels
  double* w;       // [64][RC_PITCH] weights
  uint32_t* off;   // [64] log offset of lane r's region
  int* cnt;        // [64] its pixel count (0: the lane has no region)
};

template <bool WITH_W>
__device__ __forceinline__ void rc_stage(const RcStage& st, const uint32_t* log, const double* W, int lane, int c) {
  uint32_t pv[8];
  double wv[8];
#pragma unroll
  for (int g = 0; g < 8; g++) {
    const int r = 8 * g + (lane >> 3), idx = c * RC_K + (lane & 7);
    const bool valid = idx < st.cnt[r];
    const uint32_t a = st.off[r] + (uint32_t)idx;
    pv[g] = valid ? log[a] : 0u;
    if (WITH_W) wv[g] = valid ? W[a] : 0.0;
  }
#pragma unroll
  for (int g = 0; g < 8; g++) {
    const int r = 8 * g + (lane >> 3), k = lane & 7;
    st.p[r * RC_PITCH + k] = pv[g];
    if (WITH_W) st.w[r * RC_PITCH + k] = wv[g];
  }
}

// region2rect() + get_theta() of the 64 regions of a wavefront (lane = region; cnt 0 = none): oracle/lsd.cc region2rect, the same
// expressions in the same order as lsd_region2rect (lsd_grow.hip).  rec = x1 y1 x2 y2 width theta dx dy.
__device__ __forceinline__ void rc_wave_region2rect(const RcStage& st, const uint32_t* log, const double* W, int lane, uint32_t myOff, int myCnt,
                                                    double reg_angle, double prec, double rec[8]) {
  PLH_WAVE_SYNC();
  st.off[lane] = myOff; st.cnt[lane] = myCnt;
  int maxCnt = myCnt;
  for (int m = 32; m >= 1; m >>= 1) maxCnt = max(maxCnt, __shfl_xor(maxCnt, m));
  PLH_WAVE_SYNC();
  const int chunks = (maxCnt + RC_K - 1) / RC_K;
  const uint32_t* sp = st.p + lane * RC_PITCH;
  const double* sw = st.w + lane * RC_PITCH;
  double sx = 0, sy = 0, sum = 0;
  for (int c = 0; c < chunks; c++) {
    rc_stage<true>(st, log, W, lane, c);
    PLH_WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < RC_K; k++) {
      const uint32_t p = sp[k];
      const double w = sw[k];
      sx += (double)(int)(p & 0xffffu) * w;
      sy += (double)(int)(p >> 16) * w;
      sum += w;
    }
    PLH_WAVE_SYNC();
  }
  const double x = sx / sum, y = sy / sum;
  double Ixx = 0, Iyy = 0, Ixy = 0;
  for (int c = 0; c < chunks; c++) {
    rc_stage<true>(st, log, W, lane, c);
    PLH_WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < RC_K; k++) {
      const uint32_t p = sp[k];
      const double w = sw[k];
      const double ddx = (double)(int)(p & 0xffffu) - x, ddy = (double)(int)(p >> 16) - y;
      Ixx += ddy * ddy * w;
      Iyy += ddx * ddx * w;
      Ixy += -(ddx * ddy * w);   // Ixy -= v  ==  Ixy += -v
    }
    PLH_WAVE_SYNC();
  }
  const double theta = lsd_rect_theta(Ixx, Iyy, Ixy, reg_angle, prec);
  const D2 cs = lsd_sincos_inl(theta);
  const double dx = cs.x, dy = cs.y;
  double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
  for (int c = 0; c < chunks; c++) {
    rc_stage<false>(st, log, W, lane, c);
    PLH_WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < RC_K; k++) {
      if (c * RC_K + k < myCnt) {
        const uint32_t p = sp[k];
        const double rdx = (double)(int)(p & 0xffffu) - x, rdy = (double)(int)(p >> 16) - y;
        const double l = rdx * dx + rdy * dy;
        const double w = -rdx * dy + rdy * dx;
        l_max = fmax(l_max, l); l_min = fmin(l_min, l);
        w_max = fmax(w_max, w); w_min = fmin(w_min, w);
      }
    }
    PLH_WAVE_SYNC();
  }
  rec[0] = x + l_min * dx; rec[1] = y + l_min * dy;
  rec[2] = x + l_max * dx; rec[3] = y + l_max * dy;
  const double width = w_max - w_min;
  rec[4] = width < 1.0 ? 1.0 : width;
  rec[5] = theta; rec[6] = dx; rec[7] = dy;
}

__device__ __forceinline__ void rc_store_segment(uint4* slot, const double* rec) {
  float sg[4];
  lsd_segment_of(rec, sg);
  *slot = uint4{__float_as_uint(sg[0]), __float_as_uint(sg[1]), __float_as_uint(sg[2]), __float_as_uint(sg[3])};
}

// ADV = false: LSD_REFINE_STD, every rectangle is a segment.  ADV = true: the rectangle's first rect_nfa() as well; a rectangle
// that passes is a segment, the others are parked for k_lsd_improve: slot -- which keeps its entry -- and log_nfa in the frame's
// park list.
template <bool ADV>
__device__ __forceinline__ void lsd_rects_frame(const LineDeviceArgs& a) {
  __shared__ uint32_t s_order[RC_CHUNK];
  __shared__ int s_hist[RC_BINS];
  __shared__ int s_nPark;
  __shared__ __attribute__((aligned(16))) double s_w[4 * 64 * RC_PITCH];
  __shared__ uint32_t s_p[4 * 64 * RC_PITCH];
  __shared__ uint32_t s_off[4 * 64];
  __shared__ int s_cnt[4 * 64];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = min(a.nSegs[b], a.segCap);
  uint4* ent = reinterpret_cast<uint4*>(a.segs + (long long)b * a.arenaStride);
  const uint32_t* log = a.reg + (long long)b * a.arenaStride;
  double* W = reinterpret_cast<double*>(a.ordered + (long long)b * a.arenaStride);   // `ordered` and `scr` are adjacent: one double per pixel
  RcFrame f;
  f.P = a.pix + (long long)b * a.arenaStride; f.A = a.angleTab; f.spitch = a.spitch; f.sw = a.sw; f.sh = a.sh;
  uint32_t* park = a.park + (long long)b * a.arenaStride;   // [0] count, then {slot, log_nfa (2 words)} per parked rectangle: segCap of them fit
  if (tid == 0) s_nPark = 0;
  if (n > 0) {   // 1. the weights of all kept pixels, one coalesced pass over the log
    const uint4 last = ent[n - 1];
    const int logTotal = (int)(last.x + last.y);
    for (int j = tid; j < logTotal; j += 256) {
      const uint32_t p = log[j];
      W[j] = q_modgrad(lsd_rec_q(f.P[__umul24(p >> 16, (unsigned)f.spitch) + (p & 0xffffu)]));
    }
  }
  __syncthreads();
  RcStage st;
  st.p = s_p + wv * 64 * RC_PITCH; st.w = s_w + wv * 64 * RC_PITCH; st.off = s_off + wv * 64; st.cnt = s_cnt + wv * 64;
  for (int c0 = 0; c0 < n; c0 += RC_CHUNK) {
    const int m = min(RC_CHUNK, n - c0);
    // 2. counting sort of the chunk's entries by size class, the largest first (they set the pace of their wavefront)
    for (int i = tid; i < RC_BINS; i += 256) s_hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < m; i += 256) atomicAdd(&s_hist[rc_size_class(ent[c0 + i].y)], 1);
    __syncthreads();
    if (tid == 0) {
      int acc = 0;
      for (int k = RC_BINS - 1; k >= 0; k--) { const int c = s_hist[k]; s_hist[k] = acc; acc += c; }
    }
    __syncthreads();
    for (int i = tid; i < m; i += 256) s_order[atomicAdd(&s_hist[rc_size_class(ent[c0 + i].y)], 1)] = (uint32_t)(c0 + i);
    __syncthreads();
    // 3. 64 sorted entries per wavefront at a time
    for (int base = wv * 64; base < m; base += 256) {
      const int k = base + lane;
      int slot = -1;
      uint4 e = uint4{0u, 0u, 0u, 0u};
      if (k < m) { slot = (int)s_order[k]; e = ent[slot]; }   // LsdRegionEntry
      double rec[8];
      rc_wave_region2rect(st, log, W, lane, e.x, (int)e.y, (double)__uint_as_float(e.z) * kDegToRads, a.prec, rec);
      if (slot < 0) continue;
      if constexpr (!ADV) {
        rc_store_segment(&ent[slot], rec);
      } else {
        LsdAdvRect R;
        R.x1 = rec[0]; R.y1 = rec[1]; R.x2 = rec[2]; R.y2 = rec[3]; R.width = rec[4]; R.theta = rec[5]; R.dx = rec[6]; R.dy = rec[7];
        R.prec = a.prec; R.p = a.p;
        const double log_nfa = rc_rect_nfa(f, R, a.logNT);
        if (log_nfa > 0.0) {
          rc_store_segment(&ent[slot], rec);
        } else {   // rect_improve(): k_lsd_improve, five variants at a time (the slot keeps its entry, the rectangle is evaluated again there)
          const int q = atomicAdd(&s_nPark, 1);
          uint32_t lw[2];
          __builtin_memcpy(lw, &log_nfa, 8);
          park[2 + 3 * q] = (uint32_t)slot;
          park[2 + 3 * q + 1] = lw[0];
          park[2 + 3 * q + 2] = lw[1];
        }
      }
    }
    __syncthreads();
  }
  if constexpr (ADV) {
    if (tid == 0) park[0] = (uint32_t)s_nPark;
  }
}
__global__ void __launch_bounds__(256) k_lsd_rects(LineDeviceArgs a) { lsd_rects_frame<false>(a); }
__global__ void __launch_bounds__(256) k_lsd_rects_adv(LineDeviceArgs a) { lsd_rects_frame<true>(a); }

// One variant of rect_improve(): the rectangle R after `m` iterations (1 .. 5) of stage `stage`'s loop body.  The loops modify
// their rectangle whether or not the variant is accepted, so the five variants of a stage follow from the stage's starting
// rectangle alone and can be evaluated side by side.  Returns false when the loop's width gate stops before iteration m.
__device__ __forceinline__ bool rc_variant(int stage, int m, LsdAdvRect& r) {
  const double delta = 0.5, delta_2 = delta / 2.0;
  if (stage == 0) {          // finer precision (no gate)
    for (int j = 0; j < m; j++) { r.p /= 2; r.prec = r.p * kPI; }
    return true;
  }
  for (int j = 0; j < m; j++) {
    if (!((r.width - delta) >= 0.5)) return false;
    if (stage == 1) {        // reduce width
      r.width -= delta;
    } else if (stage == 2) { // reduce one side
      r.x1 += -r.dy * delta_2; r.y1 += r.dx * delta_2;
      r.x2 += -r.dy * delta_2; r.y2 += r.dx * delta_2;
      r.width -= delta;
    } else if (stage == 3) { // reduce the other side
      r.x1 -= -r.dy * delta_2; r.y1 -= r.dx * delta_2;
      r.x2 -= -r.dy * delta_2; r.y2 -= r.dx * delta_2;
      r.width -= delta;
    } else {                 // finer precision again
      r.p /= 2;
      r.prec = r.p * kPI;
    }
  }
  return true;
}

// LSD_REFINE_ADV, second half: rect_improve() of the parked rectangles (oracle/lsd.cc rect_improve), then a stable compaction
// of the frame's surviving segments.  One block per frame; per stage every (rectangle, iteration) pair is one lane's task --
// 5 x the rectangles' parallelism, and a lane never runs more than one rect_nfa() in a row -- and the rectangle's owner then
// replays the loop's `if (v > log_nfa)` over the five values in order.
constexpr int RC_IMP = 256;   // rectangles improved at a time
__global__ void __launch_bounds__(256) k_lsd_improve(LineDeviceArgs a) {
  __shared__ int s_wave[4];
  __shared__ __attribute__((aligned(16))) double s_R[RC_IMP * 10];   // x1 y1 x2 y2 width theta dx dy prec p
  __shared__ double s_nfa[RC_IMP];
  __shared__ double s_v[RC_IMP * 5];
  __shared__ unsigned char s_ok[RC_IMP * 5], s_active[RC_IMP];
  __shared__ __attribute__((aligned(16))) double s_w[4 * 64 * RC_PITCH];
  __shared__ uint32_t s_p[4 * 64 * RC_PITCH];
  __shared__ uint32_t s_off[4 * 64];
  __shared__ int s_cnt[4 * 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = min(a.nSegs[b], a.segCap);
  uint4* ent = reinterpret_cast<uint4*>(a.segs + (long long)b * a.arenaStride);
  RcFrame f;
  f.P = a.pix + (long long)b * a.arenaStride; f.A = a.angleTab; f.spitch = a.spitch; f.sw = a.sw; f.sh = a.sh;
  const double* W = reinterpret_cast<const double*>(a.ordered + (long long)b * a.arenaStride);   // (k_lsd_rects_adv's weights)
  const uint32_t* log = a.reg + (long long)b * a.arenaStride;
  const uint32_t* park = a.park + (long long)b * a.arenaStride;
  const int nPark = n > 0 ? (int)park[0] : 0;
  const int lane = tid & 63, wv = tid >> 6;
  RcStage st;
  st.p = s_p + wv * 64 * RC_PITCH; st.w = s_w + wv * 64 * RC_PITCH; st.off = s_off + wv * 64; st.cnt = s_cnt + wv * 64;
  for (int base = 0; base < nPark; base += RC_IMP) {
    const int na = min(RC_IMP, nPark - base);
    int slot = -1;
    {   // the parked rectangles again (64 per wavefront, as k_lsd_rects_adv evaluated them), into LDS
      uint4 e = uint4{0u, 0u, 0u, 0u};
      double lnfa = 0;
      if (tid < na) {
        const uint32_t* pk = park + 2 + 3 * (base + tid);
        slot = (int)pk[0];
        const uint32_t lw[2] = {pk[1], pk[2]};
        __builtin_memcpy(&lnfa, lw, 8);
        e = ent[slot];
      }
      if (wv * 64 < na) {
        double rec[8];
        rc_wave_region2rect(st, log, W, lane, e.x, (int)e.y, (double)__uint_as_float(e.z) * kDegToRads, a.prec, rec);
        if (tid < na) {
#pragma unroll
          for (int k = 0; k < 8; k++) s_R[tid * 10 + k] = rec[k];
          s_R[tid * 10 + 8] = a.prec; s_R[tid * 10 + 9] = a.p;
          s_nfa[tid] = lnfa;
          s_active[tid] = 1;
        }
      }
    }
    __syncthreads();
    for (int stage = 0; stage < 5; stage++) {
      for (int t = tid; t < na * 5; t += 256) {
        const int i = t / 5, m = t - 5 * i + 1;
        bool ok = false;
        double v = 0;
        if (s_active[i]) {
          LsdAdvRect r;
          const double* q = s_R + i * 10;
          r.x1 = q[0]; r.y1 = q[1]; r.x2 = q[2]; r.y2 = q[3]; r.width = q[4]; r.theta = q[5]; r.dx = q[6]; r.dy = q[7]; r.prec = q[8]; r.p = q[9];
          ok = rc_variant(stage, m, r);
          if (ok) v = rc_rect_nfa(f, r, a.logNT);
        }
        s_v[t] = v;
        s_ok[t] = ok ? 1 : 0;
      }
      __syncthreads();
      if (tid < na && s_active[tid]) {   // the loop's accept rule over the stage's variants, in order
        double log_nfa = s_nfa[tid];
        int best = 0;
        for (int m = 1; m <= 5; m++)
          if (s_ok[tid * 5 + m - 1] && s_v[tid * 5 + m - 1] > log_nfa) { log_nfa = s_v[tid * 5 + m - 1]; best = m; }
        if (best) {
          LsdAdvRect r;
          double* q = s_R + tid * 10;
          r.x1 = q[0]; r.y1 = q[1]; r.x2 = q[2]; r.y2 = q[3]; r.width = q[4]; r.theta = q[5]; r.dx = q[6]; r.dy = q[7]; r.prec = q[8]; r.p = q[9];
          (void)rc_variant(stage, best, r);
          q[0] = r.x1; q[1] = r.y1; q[2] = r.x2; q[3] = r.y2; q[4] = r.width; q[8] = r.prec; q[9] = r.p;
          s_nfa[tid] = log_nfa;
        }
        if (log_nfa > 0.0) s_active[tid] = 0;   // LOG_EPS: meaningful, the remaining stages are skipped
      }
      __syncthreads();
    }
    if (tid < na) {
      if (s_nfa[tid] > 0.0) rc_store_segment(&ent[slot], s_R + tid * 10);
      else ent[slot] = uint4{RC_DROPPED, 0u, 0u, 0u};
    }
    __syncthreads();
  }
  // stable compaction of the surviving segments (a slot moves down or stays: chunks in order never overwrite unread input)
  int outBase = 0;
  for (int c0 = 0; c0 < n; c0 += 256) {
    const int i = c0 + tid;
    uint4 v = uint4{RC_DROPPED, 0u, 0u, 0u};
    if (i < n) v = ent[i];
    const bool valid = v.x != RC_DROPPED;
    const unsigned long long bm = __ballot(valid);
    if (lane == 0) s_wave[wv] = __popcll(bm);
    __syncthreads();
    int off = __popcll(bm & ((1ull << lane) - 1ull));
    for (int w = 0; w < wv; w++) off += s_wave[w];
    const int tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    if (valid) ent[outBase + off] = v;
    outBase += tot;
    __syncthreads();
  }
  if (tid == 0) a.nSegs[b] = outBase;
}

void launch_lsd_rects(const LineDeviceArgs& a, hipStream_t s) {
  if (!a.refineAdv) {
    hipLaunchKernelGGL(k_lsd_rects, dim3(a.batch), dim3(256), 0, s, a);
    return;
  }
  hipLaunchKernelGGL(k_lsd_rects_adv, dim3(a.batch), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_lsd_improve, dim3(a.batch), dim3(256), 0, s, a);
}

}  // namespace plh
