// Line kernels, stage 2c: LSD_REFINE_ADV -- nfa() and rect_improve() on the rectangles of the kept regions
// (cv::LineSegmentDetector created with LSD_REFINE_ADV: what the system opencv_contrib LSDDetector behind
// src/LineExtractor.cpp:39-40 passes as published; oracle/lsd.cc rect_improve / rect_nfa / nfa).
//
// k_lsd_rects_adv (lsd_rects.hip) has left every kept region's rectangle in an LsdAdvRec.  From there:
//   k_adv_scan_first             light   the pixel counts of every rectangle's first rect_nfa(), eight lanes each
//   k_adv_first                  heavy   nfa() of every rectangle; meaningful -> segment, else -> the frame's work list
//   5 x { k_adv_scan(stage)      light   the pixel counts of the stage's five variants of every listed rectangle, eight lanes each
//         k_adv_select(stage) }  heavy   their nfa(), one lane each; the loop's accept rule in order; meaningful -> segment,
//                                        rejected after the last stage -> dropped, else -> the other work list
//   k_adv_compact                light   stable compaction of the surviving segments, nSegs
// One block per frame everywhere; a frame's kernels are ordered by the stream.  Why the five variants of a stage are independent:
// lsd_adv_variant (lsd_rect_dev.h).  Why two kinds of kernels: nfa() needs > 200 registers for its library math, and a kernel
// that also walks the rectangles' pixels with that register budget ran one wavefront per SIMD on dependent gathers (the first two
// versions of this stage: profiles/r04_kernel_stats_adv_v2.csv).
#include "lsd_rect_dev.h"

namespace plh {

struct AdvFrame {
  int n;
  uint4* ent;
  LsdAdvRec* rec;
  uint32_t* park;   // [0], [1]: lengths of the two work lists; list k at park + 2 + k * segCap
};
__device__ __forceinline__ AdvFrame adv_frame(const LineDeviceArgs& a, int b) {
  AdvFrame f;
  f.n = min(a.nSegs[b], a.segCap);
  f.ent = reinterpret_cast<uint4*>(a.segs + (long long)b * a.arenaStride);
  f.rec = a.adv + (long long)b * a.segCap;
  f.park = a.park + (long long)b * a.arenaStride;
  return f;
}

__global__ void __launch_bounds__(256) k_adv_first(LineDeviceArgs a) {
  __shared__ int s_n;
  const int b = blockIdx.x, tid = threadIdx.x;
  const AdvFrame f = adv_frame(a, b);
  uint32_t* list = f.park + 2;
  if (tid == 0) s_n = 0;
  __syncthreads();
  for (int i = tid; i < f.n; i += 256) {
    LsdAdvRec* ar = f.rec + i;
    const double v = lsd_nfa(ar->cnt[0][0], ar->cnt[0][1], a.p, a.logNT);
    ar->log_nfa = v;
    if (v > 0.0) lsd_store_segment(&f.ent[i], ar->r);          // LOG_EPS = 0: meaningful as it is
    else list[atomicAdd(&s_n, 1)] = (uint32_t)i;               // rect_improve()
  }
  __syncthreads();
  if (tid == 0) f.park[0] = (uint32_t)s_n;
}

// k_adv_scan_first: the rectangles as region2rect() left them, all n of the frame; k_adv_scan(stage 0 .. 4): variant m of every
// listed rectangle.
// Eight lanes per rectangle (lsd_rect_counts_g8): 32 rectangles per pass of the block.
// (two kernels: the first scan walks every rectangle of the frame and should not carry the registers of the variants' doubles)
__global__ void __launch_bounds__(256) k_adv_scan_first(LineDeviceArgs a) {
  const int b = blockIdx.x, tid = threadIdx.x, grp = tid >> 3, j = tid & 7;
  const AdvFrame f = adv_frame(a, b);
  RcFrame rf;
  rf.ang = a.advAng + (long long)b * a.scaledStride; rf.spitch = a.spitch; rf.sw = a.sw; rf.sh = a.sh;
  for (int base = 0; base < f.n; base += 32) {   // (uniform over the block: the shuffles below are executed by whole wavefronts)
    const int i = base + grp;
    const bool on = i < f.n;
    LsdAdvRec* ar = f.rec + (on ? i : 0);
    LsdAdvRect r = LsdAdvRect();
    if (on) r = lsd_adv_load(ar->r);
    int total, alg;
    lsd_rect_counts_g8(rf, r, on, j, total, alg);
    alg += __shfl_xor(alg, 1); alg += __shfl_xor(alg, 2); alg += __shfl_xor(alg, 4);
    if (on && j == 0) { ar->cnt[0][0] = total; ar->cnt[0][1] = alg; }
  }
}
__global__ void __launch_bounds__(256) k_adv_scan(LineDeviceArgs a, int stage) {
  const int b = blockIdx.x, tid = threadIdx.x, grp = tid >> 3, j = tid & 7;
  const AdvFrame f = adv_frame(a, b);
  RcFrame rf;
  rf.ang = a.advAng + (long long)b * a.scaledStride; rf.spitch = a.spitch; rf.sw = a.sw; rf.sh = a.sh;
  const uint32_t* list = f.park + 2 + (stage & 1) * a.segCap;
  const int na = f.n > 0 ? (int)f.park[stage & 1] : 0;
  for (int base = 0; base < na * 5; base += 32) {
    const int t = base + grp;
    const bool on = t < na * 5;
    const int q = on ? t / 5 : 0, m = on ? t - 5 * q + 1 : 1;
    LsdAdvRec* ar = f.rec + (on ? list[q] : 0);
    LsdAdvRect r = LsdAdvRect();
    bool ok = false;
    if (on) { r = lsd_adv_load(ar->r); ok = lsd_adv_variant(stage, m, r); }
    int total, alg;
    lsd_rect_counts_g8(rf, r, ok, j, total, alg);
    alg += __shfl_xor(alg, 1); alg += __shfl_xor(alg, 2); alg += __shfl_xor(alg, 4);
    if (on && j == 0) {
      ar->cnt[m - 1][0] = total; ar->cnt[m - 1][1] = alg;
      ar->ok[m - 1] = ok ? 1 : 0;
    }
  }
}

constexpr int ADV_GROUP = 256;   // rectangles selected at a time (5 evaluations each)
__global__ void __launch_bounds__(256) k_adv_select(LineDeviceArgs a, int stage) {
  __shared__ double s_v[ADV_GROUP * 5];
  __shared__ int s_n;
  const int b = blockIdx.x, tid = threadIdx.x;
  const AdvFrame f = adv_frame(a, b);
  const uint32_t* list = f.park + 2 + (stage & 1) * a.segCap;
  uint32_t* next = f.park + 2 + ((stage + 1) & 1) * a.segCap;
  const int na = f.n > 0 ? (int)f.park[stage & 1] : 0;
  if (tid == 0) s_n = 0;
  __syncthreads();
  for (int base = 0; base < na; base += ADV_GROUP) {
    const int ng = min(ADV_GROUP, na - base);
    for (int t = tid; t < ng * 5; t += 256) {   // one evaluation per lane and pass
      const int q = t / 5, m = t - 5 * q + 1;
      const LsdAdvRec* ar = f.rec + list[base + q];
      double v = 0;
      if (ar->ok[m - 1]) {
        double p = ar->r[9];
        if (stage == 0 || stage == 4)
          for (int j = 0; j < m; j++) p /= 2;   // (the variant's p: the two precision stages halve it per iteration)
        v = lsd_nfa(ar->cnt[m - 1][0], ar->cnt[m - 1][1], p, a.logNT);
      }
      s_v[t] = v;
    }
    __syncthreads();
    if (tid < ng) {   // the loop's `if (v > log_nfa)` over the stage's variants, in order
      const int slot = (int)list[base + tid];
      LsdAdvRec* ar = f.rec + slot;
      double log_nfa = ar->log_nfa;
      int best = 0;
      for (int m = 1; m <= 5; m++)
        if (ar->ok[m - 1] && s_v[tid * 5 + m - 1] > log_nfa) { log_nfa = s_v[tid * 5 + m - 1]; best = m; }
      if (best) {
        LsdAdvRect r = lsd_adv_load(ar->r);
        (void)lsd_adv_variant(stage, best, r);
        ar->r[0] = r.x1; ar->r[1] = r.y1; ar->r[2] = r.x2; ar->r[3] = r.y2; ar->r[4] = r.width; ar->r[8] = r.prec; ar->r[9] = r.p;
        ar->log_nfa = log_nfa;
      }
      if (log_nfa > 0.0) lsd_store_segment(&f.ent[slot], ar->r);            // meaningful: the remaining stages are skipped
      else if (stage == 4) f.ent[slot] = uint4{RC_DROPPED, 0u, 0u, 0u};     // log_nfa <= LOG_EPS after all five: no segment
      else next[atomicAdd(&s_n, 1)] = (uint32_t)slot;
    }
    __syncthreads();
  }
  if (tid == 0) f.park[(stage + 1) & 1] = (uint32_t)s_n;
}

// stable compaction of the surviving segments (a slot moves down or stays: chunks in order never overwrite unread input)
__global__ void __launch_bounds__(256) k_adv_compact(LineDeviceArgs a) {
  __shared__ int s_wave[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const AdvFrame f = adv_frame(a, b);
  int outBase = 0;
  for (int c0 = 0; c0 < f.n; c0 += 256) {
    const int i = c0 + tid;
    uint4 v = uint4{RC_DROPPED, 0u, 0u, 0u};
    if (i < f.n) v = f.ent[i];
    const bool valid = v.x != RC_DROPPED;
    const unsigned long long bm = __ballot(valid);
    if (lane == 0) s_wave[wv] = __popcll(bm);
    __syncthreads();
    int off = __popcll(bm & ((1ull << lane) - 1ull));
    for (int w = 0; w < wv; w++) off += s_wave[w];
    const int tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    if (valid) f.ent[outBase + off] = v;
    outBase += tot;
    __syncthreads();
  }
  if (tid == 0) a.nSegs[b] = outBase;
}

void launch_lsd_adv(const LineDeviceArgs& a, hipStream_t s) {
  const dim3 g(a.batch), b(256);
  hipLaunchKernelGGL(k_adv_scan_first, g, b, 0, s, a);
  hipLaunchKernelGGL(k_adv_first, g, b, 0, s, a);
  for (int stage = 0; stage < 5; stage++) {
    hipLaunchKernelGGL(k_adv_scan, g, b, 0, s, a, stage);
    hipLaunchKernelGGL(k_adv_select, g, b, 0, s, a, stage);
  }
  hipLaunchKernelGGL(k_adv_compact, g, b, 0, s, a);
}

}  // namespace plh
