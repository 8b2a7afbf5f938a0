// Line kernels, stage 2c: LSD_REFINE_ADV -- nfa() and rect_improve() on the rectangles of the kept regions
// (cv::LineSegmentDetector created with LSD_REFINE_ADV: what the system opencv_contrib LSDDetector behind
// src/LineExtractor.cpp:39-40 passes as published; oracle/lsd.cc rect_improve / rect_nfa / nfa).
//
// k_lsd_rects_adv (lsd_rects.hip) has left every kept region's rectangle in an LsdAdvRec.  From there (round 5: three launches
// of one-wavefront blocks instead of thirteen of 256-thread blocks):
//   k_adv_first     every rectangle's first rect_nfa(): the pixel counts eight lanes per rectangle, then nfa() one lane per
//                   rectangle; meaningful -> segment, else -> the frame's work list
//   k_adv_improve   rect_improve() of the listed rectangles, eight lanes per rectangle through all five stages: a stage's five
//                   variants follow from its starting rectangle alone (lsd_adv_variant, lsd_rect_dev.h), so their pixel counts
//                   are taken in ONE walk by the eight lanes (lsd_rect_counts_g8_prec5: one rectangle, five tolerances;
//                   lsd_rect_counts_g8_var5: five narrowed rectangles, every pixel loaded and tested once), their nfa() side by
//                   side in five of the eight lanes, and the loop's
//                   `if (v > log_nfa)` is replayed in order; meaningful -> segment, rejected after the last stage -> dropped
//   k_adv_compact   stable compaction of the surviving segments, nSegs
// Rectangles are independent of each other: nothing here is ordered except the compaction.
//
// History (profiles/r04_kernel_stats_adv_*.csv, r05_adv_*): round 4 ran scan and nfa() in separate kernels per stage because nfa()
// with its two log_gamma() evaluations needed 194 registers; thirteen launches of block-per-frame kernels, each of which had to
// find a CU with four free wave slots beside the region-growing wavefronts of the other sub-batches (10 ms alone, 60 - 70 ms on
// the line chain inside the pipeline).  With log_gamma() a table lookup (LineDeviceArgs::lgamma) one kernel holds both halves.
#include "lsd_rect_dev.h"

namespace plh {

struct AdvFrame {
  int n;
  uint4* ent;
  LsdAdvRec* rec;
  uint32_t* count;   // length of the work list (zeroed by k_lsd_rects_sort)
  const uint32_t* order;   // the frame's slots by region size class, largest first (k_lsd_rects_sort)
  uint32_t* list;    // slots of the rectangles that go on to rect_improve(): runs of similar size (appended 64 at a time in `order`)
};
__device__ __forceinline__ AdvFrame adv_frame(const LineDeviceArgs& a, int b) {
  AdvFrame f;
  f.n = min(a.nSegs[b], a.segCap);
  f.ent = reinterpret_cast<uint4*>(a.segs + (long long)b * a.arenaStride);
  f.rec = a.adv + (long long)b * a.segCap;
  uint32_t* park = a.park + (long long)b * a.arenaStride;
  f.count = park;                 // park[0]
  f.order = park + 2;
  f.list = park + 2 + a.segCap;
  return f;
}
__device__ __forceinline__ RcFrame adv_field(const LineDeviceArgs& a, int b) {
  RcFrame rf;
  rf.ang = a.advAng + (long long)b * a.scaledStride; rf.spitch = a.spitch; rf.sw = a.sw; rf.sh = a.sh;
  return rf;
}
// sum over the eight lanes of a rectangle's group
__device__ __forceinline__ int adv_sum8(int v) {
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
  return v;
}

// one-wavefront blocks per frame, by batch size (lsd_blocks_per_frame, lsd_rects.hip): k_adv_first 24 .. 4 (64 rectangles per pass each),
// k_adv_improve 32 .. 6 (8 rectangles per pass each)
constexpr int ADV_FIRST_MAX = 24, ADV_FIRST_MIN = 4, ADV_IMPROVE_MAX = 32, ADV_IMPROVE_MIN = 6;
int lsd_blocks_per_frame(int batch, int lo, int hi);

// lsd_log_gamma(i) for i = 1 .. n - 1 (t[0] is never read)
__global__ void __launch_bounds__(256) k_lsd_lgamma_table(double* t, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) t[i] = i >= 1 ? lsd_log_gamma((double)i) : 0.0;
}
size_t lsd_adv_rec_bytes() { return kLsdAdvRecBytes; }   // (line_host.hip sizes the per-frame record buffer)
void launch_lsd_lgamma_table(double* t, int n, hipStream_t s) {
  hipLaunchKernelGGL(k_lsd_lgamma_table, dim3((n + 255) / 256), dim3(256), 0, s, t, n);
}

// Register budget of rect_improve(): the kernel wants 235 registers (two wavefronts per SIMD); built for three (168 registers, 248 bytes
// of spill per lane) it runs as fast alone and leaves a third wave slot's worth of registers to whatever shares the SIMD inside the
// pipeline: + 2.0 % on the headline in two same-job A/Bs, the same at four / 128 registers (profiles/r06_adv_improve_registers_ab.txt).
// -DPLH_ADV_WAVES=n builds for n (0: no cap).
#ifndef PLH_ADV_WAVES
#define PLH_ADV_WAVES 3
#endif
#if PLH_ADV_WAVES > 0 && !defined(HIPEMU)
#define PLH_ADV_ATTR __attribute__((amdgpu_waves_per_eu(PLH_ADV_WAVES)))
#else
#define PLH_ADV_ATTR
#endif
#if defined(PLH_ADVFIRST_WAVES) && !defined(HIPEMU)
#define PLH_ADVFIRST_ATTR __attribute__((amdgpu_waves_per_eu(PLH_ADVFIRST_WAVES)))
#else
#define PLH_ADVFIRST_ATTR
#endif
__global__ void __launch_bounds__(64) PLH_ADVFIRST_ATTR k_adv_first(LineDeviceArgs a, PlhXcdGrid xg) {
  __shared__ LsdScanGeom s_geom[64];
  __shared__ int s_tot[64], s_alg[64], s_slot[64];
  int blk, b;   // (plh_xcd_decode: the blocks of a frame walk one angle plane -- behind one L2)
  if (!plh_xcd_decode(xg, blk, b)) return;
  const int perFrame = xg.perFrame;
  const int lane = threadIdx.x, grp = lane >> 3, j = lane & 7;
  const AdvFrame f = adv_frame(a, b);
  const RcFrame rf = adv_field(a, b);
  const LsdAlignTol tol0 = lsd_align_tol(0.0, a.prec);   // (every rectangle starts with the launch's tolerance; theta per rectangle)
  // The frame's rectangles in k_lsd_rects' size-class order, cut into chunks of eight: the eight that walk side by side below take
  // about equally long (in slot order the longest of eight set the pace, 2 x the mean).  The chunks go to the frame's blocks
  // round robin, so that every block gets large and small ones; a block takes eight chunks (64 rectangles) per pass.
  const int nChunks = (f.n + 7) >> 3;
  for (int c0 = blk; c0 < nChunks; c0 += 8 * perFrame) {
    // the pass's rectangles: lane (it, pos) -> position pos of chunk c0 + it x blocks.  Their scan geometry, one lane each (the
    // corner sort is as long as a walk: not eight times per rectangle):
    const int k = (c0 + (lane >> 3) * perFrame) * 8 + (lane & 7);
    const int mine = k < f.n ? (int)f.order[k] : -1;
    {
      s_slot[lane] = mine;
      s_geom[lane] = mine >= 0 ? lsd_scan_geom(rf, lsd_adv_load(f.rec[mine].r)) : lsd_scan_none();
    }
    PLH_WAVE_SYNC();
    // their pixel counts, eight rectangles at a time, eight lanes each (lsd_rect_counts_g8)
    for (int it = 0; it < 8; it++) {
      const int i = s_slot[it * 8 + grp];
      LsdAlignTol t = tol0;
      if (i >= 0) { t.theta = f.rec[i].r[5]; t.thDeg = (float)(t.theta * (180.0 / kPI)); }
      int total, alg;
      lsd_rect_counts_g8(rf, s_geom[it * 8 + grp], t, j, total, alg);
      alg = adv_sum8(alg);
      if (j == 0) { s_tot[it * 8 + grp] = total; s_alg[it * 8 + grp] = alg; }
    }
    PLH_WAVE_SYNC();
    // nfa(): one lane per rectangle
    bool again = false;
    if (mine >= 0) {
      LsdAdvRec* ar = f.rec + mine;
      const double v = lsd_nfa(s_tot[lane], s_alg[lane], a.p, a.logNT, a.lgamma);
      if (v > 0.0) {
        lsd_store_segment(&f.ent[mine], ar->r);                    // LOG_EPS = 0: meaningful as it is
      } else {
        ar->log_nfa = v;
        again = true;                                              // rect_improve()
      }
    }
    {   // the wavefront's rectangles that go on, as one run of the work list (one atomic per wavefront)
      const unsigned long long bm = __ballot(again);
      unsigned at = 0;
      if (lane == 0 && bm) at = atomicAdd(f.count, (unsigned)__popcll(bm));
      at = bcast_u32(at, 0);
      if (again) f.list[at + (unsigned)__popcll(bm & ((1ull << lane) - 1ull))] = (uint32_t)mine;
    }
    PLH_WAVE_SYNC();
  }
}

__global__ void __launch_bounds__(64) PLH_ADV_ATTR k_adv_improve(LineDeviceArgs a, PlhXcdGrid xg) {
  __shared__ LsdScanGeom s_geom[8 * 5];   // [rectangle of the pass][variant]
  __shared__ int s_ok[8 * 5];
  int blk, b;
  if (!plh_xcd_decode(xg, blk, b)) return;
  const int perFrame = xg.perFrame;
  const int lane = threadIdx.x, grp = lane >> 3, j = lane & 7, g0 = lane & ~7;
  const AdvFrame f = adv_frame(a, b);
  const RcFrame rf = adv_field(a, b);
  const int na = f.n > 0 ? (int)*f.count : 0;
  for (int base = blk * 8; base < na; base += 8 * perFrame) {   // (uniform: the shuffles below are executed by the whole wavefront)
    const int q = base + grp;
    bool active = q < na;
    const int slot = active ? (int)f.list[q] : 0;
    LsdAdvRect r = LsdAdvRect();
    double log_nfa = 0.0;
    if (active) { r = lsd_adv_load(f.rec[slot].r); log_nfa = f.rec[slot].log_nfa; }
    for (int stage = 0; stage < 5; stage++) {
      if (!__any(active)) break;
      const bool precStage = stage == 0 || stage == 4;   // finer precision: one rectangle, five tolerances
      // lane j < 5 of the group: variant j + 1 of its rectangle -- whether the loop's width gate lets it exist, and its scan geometry
      // (the two precision stages walk the rectangle itself: one geometry, lane 0's)
      PLH_WAVE_SYNC();
      if (j < 5) {
        LsdAdvRect rv = r;
        const bool okv = active && lsd_adv_variant(stage, j + 1, rv);
        s_ok[grp * 5 + j] = okv ? 1 : 0;
        s_geom[grp * 5 + j] = okv ? lsd_scan_geom(rf, precStage ? r : rv) : lsd_scan_none();
      }
      PLH_WAVE_SYNC();
      // (total, aligned) pixel counts of the variants m = 1 .. 5, the same in all eight lanes of the group
      int tot[5], alg[5];
      bool ok[5];
#pragma unroll
      for (int m = 0; m < 5; m++) ok[m] = s_ok[grp * 5 + m] != 0;
      if (precStage) {
        LsdAlignTol5 t5;
        t5.theta = r.theta; t5.thDeg = (float)(r.theta * (180.0 / kPI));
        {
          LsdAdvRect rv = r;
#pragma unroll
          for (int m = 0; m < 5; m++) {
            rv.p /= 2; rv.prec = rv.p * kPI;   // (lsd_adv_variant's iteration)
            t5.prec[m] = rv.prec;
            const float pd = (float)(rv.prec * (180.0 / kPI));
            t5.lo[m] = pd - 1e-3f; t5.hi[m] = pd + 1e-3f;
          }
        }
        int total;
        lsd_rect_counts_g8_prec5(rf, s_geom[grp * 5], t5, j, total, alg);   // (the gate of stage 4 does not move: all five or none)
#pragma unroll
        for (int m = 0; m < 5; m++) { tot[m] = total; alg[m] = adv_sum8(alg[m]); }
      } else {   // a width stage: the five variants in one walk
        const LsdAlignTol t = lsd_align_tol(r.theta, r.prec);
        lsd_rect_counts_g8_var5(rf, s_geom + grp * 5, t, j, g0, tot, alg);   // (tot[m]: lane m's own variant, read below by lane j == m)
#pragma unroll
        for (int m = 0; m < 5; m++) alg[m] = adv_sum8(alg[m]);
      }
      // nfa() of variant j + 1 in lane j < 5 of the group
      double v = 0.0;
      {
        int myTot = 0, myAlg = 0;
        bool myOk = false;
#pragma unroll
        for (int k = 0; k < 5; k++)
          if (k == j) { myTot = tot[k]; myAlg = alg[k]; myOk = ok[k]; }
        if (myOk) {
          double p = r.p;
          if (precStage)
            for (int k = 0; k <= j; k++) p /= 2;   // (the variant's p: the two precision stages halve it per iteration)
          v = lsd_nfa(myTot, myAlg, p, a.logNT, a.lgamma);
        }
      }
      // the loop's `if (v > log_nfa)` over the stage's variants, in order (every lane of the group replays it)
      int best = 0;
#pragma unroll
      for (int m = 0; m < 5; m++) {
        const double vm = __shfl(v, g0 + m);
        if (ok[m] && vm > log_nfa) { log_nfa = vm; best = m + 1; }
      }
      if (best) (void)lsd_adv_variant(stage, best, r);
      if (active) {
        if (log_nfa > 0.0) {   // meaningful: the remaining stages are skipped
          if (j == 0) {
            const double rec[4] = {r.x1, r.y1, r.x2, r.y2};
            lsd_store_segment(&f.ent[slot], rec);
          }
          active = false;
        } else if (stage == 4) {   // log_nfa <= LOG_EPS after all five: no segment
          if (j == 0) f.ent[slot] = uint4{RC_DROPPED, 0u, 0u, 0u};
          active = false;
        }
      }
    }
  }
}

// stable compaction of the surviving segments, one wavefront per frame (a slot moves down or stays: chunks in order never
// overwrite unread input)
__global__ void __launch_bounds__(64) k_adv_compact(LineDeviceArgs a) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const AdvFrame f = adv_frame(a, b);
  int outBase = 0;
  for (int c0 = 0; c0 < f.n; c0 += 64) {
    const int i = c0 + lane;
    uint4 v = uint4{RC_DROPPED, 0u, 0u, 0u};
    if (i < f.n) v = f.ent[i];
    const bool valid = v.x != RC_DROPPED;
    const unsigned long long bm = __ballot(valid);
    PLH_WAVE_SYNC();   // (all loads of the chunk are done before a lane stores into it)
    if (valid) f.ent[outBase + __popcll(bm & ((1ull << lane) - 1ull))] = v;
    outBase += __popcll(bm);
  }
  if (lane == 0) a.nSegs[b] = outBase;
}

void launch_lsd_adv(const LineDeviceArgs& a, hipStream_t s) {
  const int nf = lsd_blocks_per_frame(a.batch, ADV_FIRST_MIN, ADV_FIRST_MAX), ni = lsd_blocks_per_frame(a.batch, ADV_IMPROVE_MIN, ADV_IMPROVE_MAX);
  const PlhXcdGrid xf = plh_xcd_make(nf, a.batch), xi = plh_xcd_make(ni, a.batch);
  hipLaunchKernelGGL(k_adv_first, dim3(plh_xcd_grid(xf)), dim3(64), 0, s, a, xf);
  hipLaunchKernelGGL(k_adv_improve, dim3(plh_xcd_grid(xi)), dim3(64), 0, s, a, xi);
  hipLaunchKernelGGL(k_adv_compact, dim3(a.batch), dim3(64), 0, s, a);
}

}  // namespace plh
