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
// whether the segment is kept: it runs here as well, in two phases -- the first rect_nfa() of every rectangle in the lanes that
// computed it; the rectangles it does not pass (about one in ten) are collected and improved densely packed, one lane each --
// followed by a stable compaction of the surviving segments.
#include "lsd_rect_dev.h"

namespace plh {

constexpr int RC_CHUNK = 4096;   // entries sorted and evaluated at a time
constexpr int RC_BINS = 128;
constexpr int RC_IMP_WORDS = 20;   // a rectangle waiting for rect_improve(): 9 doubles (x1 y1 x2 y2 width theta dx dy log_nfa) + its slot;
                                   // the list starts 2 words into the frame's scratch area (word 0 = the count)
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

__device__ __forceinline__ double rc_weight(const RcFrame& f, uint32_t p) {   // modgrad of a packed pixel
  return f.A[f.P[__umul24(p >> 16, (unsigned)f.spitch) + (p & 0xffffu)] & LSD_REC_IDX].modgrad;
}

// region2rect() + get_theta() for one region, one lane: oracle/lsd.cc region2rect; the same expressions in the same order as
// lsd_region2rect (lsd_grow.hip), whose three-lane chains add exactly these terms in exactly this order.
// rec = x1 y1 x2 y2 width theta dx dy.
__device__ __forceinline__ void rc_region2rect(const RcFrame& f, const uint32_t* reg, int cnt, double reg_angle, double prec, double rec[8]) {
  double sx = 0, sy = 0, sum = 0;
  int i = 0;
  for (; i + 4 <= cnt; i += 4) {   // four gathers in flight, the adds in order
    uint32_t p[4];
    double w[4];
#pragma unroll
    for (int k = 0; k < 4; k++) p[k] = reg[i + k];
#pragma unroll
    for (int k = 0; k < 4; k++) w[k] = rc_weight(f, p[k]);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      sx += (double)(int)(p[k] & 0xffffu) * w[k];
      sy += (double)(int)(p[k] >> 16) * w[k];
      sum += w[k];
    }
  }
  for (; i < cnt; i++) {
    const uint32_t p = reg[i];
    const double w = rc_weight(f, p);
    sx += (double)(int)(p & 0xffffu) * w;
    sy += (double)(int)(p >> 16) * w;
    sum += w;
  }
  const double x = sx / sum, y = sy / sum;
  double Ixx = 0, Iyy = 0, Ixy = 0;
  i = 0;
  for (; i + 4 <= cnt; i += 4) {
    uint32_t p[4];
    double w[4];
#pragma unroll
    for (int k = 0; k < 4; k++) p[k] = reg[i + k];
#pragma unroll
    for (int k = 0; k < 4; k++) w[k] = rc_weight(f, p[k]);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const double ddx = (double)(int)(p[k] & 0xffffu) - x, ddy = (double)(int)(p[k] >> 16) - y;
      Ixx += ddy * ddy * w[k];
      Iyy += ddx * ddx * w[k];
      Ixy += -(ddx * ddy * w[k]);   // Ixy -= v  ==  Ixy += -v
    }
  }
  for (; i < cnt; i++) {
    const uint32_t p = reg[i];
    const double w = rc_weight(f, p);
    const double ddx = (double)(int)(p & 0xffffu) - x, ddy = (double)(int)(p >> 16) - y;
    Ixx += ddy * ddy * w;
    Iyy += ddx * ddx * w;
    Ixy += -(ddx * ddy * w);
  }
  const double theta = lsd_rect_theta(Ixx, Iyy, Ixy, reg_angle, prec);
  const D2 cs = lsd_sincos_inl(theta);
  const double dx = cs.x, dy = cs.y;
  double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
  for (i = 0; i < cnt; i++) {
    const uint32_t p = reg[i];
    const double rdx = (double)(int)(p & 0xffffu) - x, rdy = (double)(int)(p >> 16) - y;
    const double l = rdx * dx + rdy * dy;
    const double w = -rdx * dy + rdy * dx;
    l_max = fmax(l_max, l); l_min = fmin(l_min, l);
    w_max = fmax(w_max, w); w_min = fmin(w_min, w);
  }
  rec[0] = x + l_min * dx; rec[1] = y + l_min * dy;
  rec[2] = x + l_max * dx; rec[3] = y + l_max * dy;
  const double width = w_max - w_min;
  rec[4] = width < 1.0 ? 1.0 : width;
  rec[5] = theta; rec[6] = dx; rec[7] = dy;
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

// rect_improve() after its first rect_nfa() (log_nfa, which was not above LOG_EPS = 0): finer precision, narrower, one side in,
// the other side in, finer precision again (oracle/lsd.cc rect_improve).  Returns whether the rectangle is kept.
__device__ __attribute__((noinline)) bool rc_rect_improve(const RcFrame& f, LsdAdvRect& R, double log_nfa, double logNT) {
  const double delta = 0.5, delta_2 = delta / 2.0, LOG_EPS = 0.0;
  {
    LsdAdvRect r = R;
    for (int n = 0; n < 5; ++n) {          // finer precision
      r.p /= 2;
      r.prec = r.p * kPI;
      const double v = rc_rect_nfa(f, r, logNT);
      if (v > log_nfa) { log_nfa = v; R = r; }
    }
  }
  if (!(log_nfa > LOG_EPS)) {
    LsdAdvRect r = R;
    for (int n = 0; n < 5; ++n)            // reduce width
      if ((r.width - delta) >= 0.5) {
        r.width -= delta;
        const double v = rc_rect_nfa(f, r, logNT);
        if (v > log_nfa) { R = r; log_nfa = v; }
      }
  }
  if (!(log_nfa > LOG_EPS)) {
    LsdAdvRect r = R;
    for (int n = 0; n < 5; ++n)            // reduce one side
      if ((r.width - delta) >= 0.5) {
        r.x1 += -r.dy * delta_2; r.y1 += r.dx * delta_2;
        r.x2 += -r.dy * delta_2; r.y2 += r.dx * delta_2;
        r.width -= delta;
        const double v = rc_rect_nfa(f, r, logNT);
        if (v > log_nfa) { R = r; log_nfa = v; }
      }
  }
  if (!(log_nfa > LOG_EPS)) {
    LsdAdvRect r = R;
    for (int n = 0; n < 5; ++n)            // reduce the other side
      if ((r.width - delta) >= 0.5) {
        r.x1 -= -r.dy * delta_2; r.y1 -= r.dx * delta_2;
        r.x2 -= -r.dy * delta_2; r.y2 -= r.dx * delta_2;
        r.width -= delta;
        const double v = rc_rect_nfa(f, r, logNT);
        if (v > log_nfa) { R = r; log_nfa = v; }
      }
  }
  if (!(log_nfa > LOG_EPS)) {
    LsdAdvRect r = R;
    for (int n = 0; n < 5; ++n)            // finer precision again
      if ((r.width - delta) >= 0.5) {
        r.p /= 2;
        r.prec = r.p * kPI;
        const double v = rc_rect_nfa(f, r, logNT);
        if (v > log_nfa) { R = r; log_nfa = v; }
      }
  }
  return log_nfa > LOG_EPS;
}

__device__ __forceinline__ void rc_store_segment(uint4* slot, const double* rec) {
  float sg[4];
  lsd_segment_of(rec, sg);
  *slot = uint4{__float_as_uint(sg[0]), __float_as_uint(sg[1]), __float_as_uint(sg[2]), __float_as_uint(sg[3])};
}

// One block per frame.  ADV = false: LSD_REFINE_STD, every rectangle is a segment.  ADV = true: the rectangle's first rect_nfa()
// as well; a rectangle that passes is a segment, the others are parked (imp: rectangle, log_nfa, slot; imp's first word counts
// them) for k_lsd_improve.
template <bool ADV>
__device__ __forceinline__ void lsd_rects_frame(const LineDeviceArgs& a) {
  __shared__ uint32_t s_order[RC_CHUNK];
  __shared__ int s_hist[RC_BINS];
  __shared__ int s_nImp;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = min(a.nSegs[b], a.segCap);
  uint4* ent = reinterpret_cast<uint4*>(a.segs + (long long)b * a.arenaStride);
  const uint32_t* log = a.reg + (long long)b * a.arenaStride;
  RcFrame f;
  f.P = a.pix + (long long)b * a.arenaStride; f.A = a.angleTab; f.spitch = a.spitch; f.sw = a.sw; f.sh = a.sh;
  uint32_t* imp = a.scr + (long long)b * a.arenaStride;   // (the scratch area is free by now)
  const int impCap = (int)((a.scaledStride - 2) / RC_IMP_WORDS);
  if (tid == 0) s_nImp = 0;
  for (int c0 = 0; c0 < n; c0 += RC_CHUNK) {
    const int m = min(RC_CHUNK, n - c0);
    // counting sort of the chunk's entries by size class, the largest first (they set the pace of their wavefront)
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
    for (int k = tid; k < m; k += 256) {
      const int slot = (int)s_order[k];
      const uint4 e = ent[slot];   // LsdRegionEntry
      double rec[8];
      rc_region2rect(f, log + e.x, (int)e.y, (double)__uint_as_float(e.z) * kDegToRads, a.prec, rec);
      if constexpr (!ADV) {
        rc_store_segment(&ent[slot], rec);
      } else {
        LsdAdvRect R;
        R.x1 = rec[0]; R.y1 = rec[1]; R.x2 = rec[2]; R.y2 = rec[3]; R.width = rec[4]; R.theta = rec[5]; R.dx = rec[6]; R.dy = rec[7];
        R.prec = a.prec; R.p = a.p;
        const double log_nfa = rc_rect_nfa(f, R, a.logNT);
        if (log_nfa > 0.0) {
          rc_store_segment(&ent[slot], rec);
        } else {
          const int q = atomicAdd(&s_nImp, 1);
          if (q < impCap) {   // improved by k_lsd_improve, densely packed
            double* it = reinterpret_cast<double*>(imp + 2 + (long long)q * RC_IMP_WORDS);
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) it[k2] = rec[k2];
            it[8] = log_nfa;
            imp[2 + (long long)q * RC_IMP_WORDS + 18] = (uint32_t)slot;
          } else {            // (no room to park it: a frame of nothing but minimal rejected regions)
            if (rc_rect_improve(f, R, log_nfa, a.logNT)) {
              rec[0] = R.x1; rec[1] = R.y1; rec[2] = R.x2; rec[3] = R.y2;
              rc_store_segment(&ent[slot], rec);
            } else {
              ent[slot] = uint4{RC_DROPPED, 0u, 0u, 0u};
            }
          }
        }
      }
    }
    __syncthreads();
  }
  if constexpr (ADV) {
    if (tid == 0) imp[0] = (uint32_t)min(s_nImp, impCap);
  }
}
__global__ void __launch_bounds__(256) k_lsd_rects(LineDeviceArgs a) { lsd_rects_frame<false>(a); }
__global__ void __launch_bounds__(256) k_lsd_rects_adv(LineDeviceArgs a) { lsd_rects_frame<true>(a); }

// LSD_REFINE_ADV, second half: rect_improve() of the parked rectangles, one lane each, then a stable compaction of the frame's
// surviving segments.  One block per frame.
__global__ void __launch_bounds__(256) k_lsd_improve(LineDeviceArgs a) {
  __shared__ int s_wave[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = min(a.nSegs[b], a.segCap);
  uint4* ent = reinterpret_cast<uint4*>(a.segs + (long long)b * a.arenaStride);
  RcFrame f;
  f.P = a.pix + (long long)b * a.arenaStride; f.A = a.angleTab; f.spitch = a.spitch; f.sw = a.sw; f.sh = a.sh;
  const uint32_t* imp = a.scr + (long long)b * a.arenaStride;
  const int nImp = n > 0 ? (int)imp[0] : 0;
  for (int q = tid; q < nImp; q += 256) {
    const double* it = reinterpret_cast<const double*>(imp + 2 + (long long)q * RC_IMP_WORDS);
    const int slot = (int)imp[2 + (long long)q * RC_IMP_WORDS + 18];
    LsdAdvRect R;
    R.x1 = it[0]; R.y1 = it[1]; R.x2 = it[2]; R.y2 = it[3]; R.width = it[4]; R.theta = it[5]; R.dx = it[6]; R.dy = it[7];
    R.prec = a.prec; R.p = a.p;
    if (rc_rect_improve(f, R, it[8], a.logNT)) {
      const double rec[4] = {R.x1, R.y1, R.x2, R.y2};
      rc_store_segment(&ent[slot], rec);
    } else {
      ent[slot] = uint4{RC_DROPPED, 0u, 0u, 0u};
    }
  }
  __syncthreads();
  // stable compaction of the surviving segments (a slot moves down or stays: chunks in order never overwrite unread input)
  int outBase = 0;
  const int lane = tid & 63, wv = tid >> 6;
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
