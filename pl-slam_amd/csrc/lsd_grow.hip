// Line kernels, stage 2: LSD region growing / rectangle fit, KeyLine selection, Sobel pack, LBD.
// See line_kernels.hip for the overview and the reference citations.
#include "lsd_rect_dev.h"

namespace plh {

struct GrowCtx {
  uint32_t* P;                 // level-line records of the scaled image (LSD_REC_*: table index | DEF | USED = region growing's mark)
  const LsdAngleEntry* A;      // per-device table of everything a gradient determines (line_plan.h)
  uint32_t* reg;       // region queue (global memory); entries are packed coordinates x | y << 16
  uint32_t* regq;      // one wavefront per frame: gx^2 + gy^2 of the queue's pixels, written beside reg for the region that is kept
                       // (k_lsd_rects takes region2rect()'s weights from it); null in the multi-wavefront kernel (the commit writes it)
  uint32_t* scr;       // scratch of the queue's size (reduce_region_radius compaction)
  uint8_t* M;          // multi-wavefront build only: this wavefront's private `used` marks, one byte per pixel (k_lsd_grow_mw)
  uint16_t* H;         // ... and the frame's claim hints, shared by its wavefronts: tag of the transaction that last marked the pixel
  unsigned hTag;       // this transaction's tag, (sequence number mod 65535) + 1
  unsigned hWin;       // how many sequence numbers below this one may still be uncommitted (0: none -- hints are ignored)
  uint32_t* asmList;   // LDS: pixels this run took for used on the strength of an older transaction's claim (packed coordinates)
  uint32_t* asmCnt;    // LDS word: their count (MW_ASM_CAP = full: no further assumptions)
  uint32_t* ring;      // LDS mirror of the newest LSD_RING queue entries
  double* T;           // LDS [3][64] doubles: lane -> chain transposition buffer of the sequential double sums
  int spitch, sw, sh, lane;
  uint32_t nullIdx;   // record index of pixel (sw - 1, 0): never defined, never marked -- what a lane with nothing to examine loads
  unsigned qThresh;
  double precDef;             // the launch's tolerance and its direction-test margins (host: plh_line_create)
  float cin2Def, cout2Def;
  int fastDef;
  float loDef, hiDef;         // lsd_tol(precDef)'s float band, evaluated once per wavefront: a seed's first region_grow() takes it from here
#if defined(PLH_GROW_PROF)
  unsigned long long* pf;   // per-wave phase counters (debug build only, tools/grow_prof.py)
#endif
};
#if defined(PLH_GROW_PROF)
__device__ unsigned long long g_grow_prof[40];   // [16..31]: k_lsd_grow_mw (transactions, retired unused, reruns, wait / scan / run / commit cycles); [32..]: rare-path lanes
#if defined(HIPEMU) || PLH_GROW_PROF + 0 >= 3
#define PF_NOW() 0ull
#else
#define PF_NOW() __builtin_amdgcn_s_memtime()
#endif
// the per-step clocks of region_grow() (s_memtime is a scalar memory read: it costs a wavefront hundreds of cycles) only with
// -DPLH_GROW_PROF=2; the transaction-level ones of k_lsd_grow_mw with any PLH_GROW_PROF
#if PLH_GROW_PROF + 0 >= 2
#define PF_NOW_FINE() PF_NOW()
#else
#define PF_NOW_FINE() 0ull
#endif
#define PF_ADD(c, k, v) ((c).pf[k] += (unsigned long long)(v))
// -DPLH_GROW_PROF=3: an event trace of k_lsd_grow_mw instead (tools/mw_trace.py): {kind | wavefront << 8 | argument << 16, clock}
#if PLH_GROW_PROF + 0 >= 3 && !defined(HIPEMU)
__device__ unsigned long long g_mw_trace[2 * (1 << 20)];
__device__ unsigned g_mw_trace_n;
#define MW_TRACE(wv, lane, kind, arg)                                                                     \
  do {                                                                                                    \
    if ((lane) == 0) {                                                                                    \
      const unsigned k__ = atomicAdd(&g_mw_trace_n, 1u);                                                  \
      if (k__ < (1u << 20)) {                                                                             \
        g_mw_trace[2 * k__] = (unsigned long long)(kind) | ((unsigned long long)(wv) << 8) | ((unsigned long long)(unsigned)(arg) << 16); \
        g_mw_trace[2 * k__ + 1] = __builtin_amdgcn_s_memtime();                                           \
      }                                                                                                   \
    }                                                                                                     \
  } while (0)
#else
#define MW_TRACE(wv, lane, kind, arg) ((void)0)
#endif
#else
#define PF_NOW() 0ull
#define PF_NOW_FINE() 0ull
#define PF_ADD(c, k, v) ((void)sizeof(v))
#define MW_TRACE(wv, lane, kind, arg) ((void)0)
#endif
constexpr int LSD_GS_D = 9;      // GrowState::d: tolerance of the call | rectangle x1 y1 x2 y2 width | its theta, cos, sin (LSD_REFINE_ADV)
constexpr int LSD_RING = 512;    // 2 KiB; the chain buffer T (1.5 KiB) aliases it (never live at the same time)
constexpr int LSD_PTS = 8;      // queue points examined per step (8 points x 8 neighbours = 64 lanes)
constexpr unsigned LSD_USED = LSD_REC_USED;   // `used` mark: bit 31 of the record word (LsdPix::q holds the record word)
// a pixel region growing may take: defined and not used -- one signed compare on the record word
__device__ __forceinline__ bool rec_is_candidate(unsigned rec) { return (int)rec >= (int)LSD_REC_DEF; }
// Where region growing's marks live.  One wavefront per frame (MW = false): bit 31 of the record itself.  Several
// wavefronts per frame (MW = true, k_lsd_grow_mw): a region in flight is a transaction whose marks must stay invisible to
// the other wavefronts until it commits, so they go to a byte plane private to the wavefront; the record's bit holds the
// committed marks only.  The record word as region growing sees it is the record with the private mark folded into bit 31.
// MW only.  Besides the private marks the wavefronts of a frame share a plane of CLAIM HINTS: whoever marks a pixel also
// writes its transaction's tag there (plain byte stores, last writer wins -- a hint, never a fact).  A claim by an OLDER
// transaction that is still uncommitted predicts that the pixel will be used by the time this one commits, so this run
// takes it for used (and remembers that it did: the commit checks the prediction), instead of growing into a region that
// is being grown elsewhere and finding out at its commit.  `conf` collects "this run is already lost": a pixel this
// wavefront holds a private mark on has a committed mark, or an older transaction's claim, by now.
constexpr int MW_ASM_CAP = 64;
__device__ __forceinline__ bool grow_older_claim(const GrowCtx& c, unsigned h) {
  // how many sequence numbers below this transaction the claimant is (tags are 1 .. 65535, sequence numbers mod 65535)
  const unsigned d = c.hTag >= h ? c.hTag - h : c.hTag + 65535u - h;
  return h != 0u && d != 0u && d <= c.hWin;
}
template <bool MW>
__device__ __forceinline__ unsigned grow_load_rec(const GrowCtx& c, uint32_t idx, unsigned& conf, bool& assumed) {
  if constexpr (MW) {
    const unsigned p = c.P[idx], m = (unsigned)c.M[idx], h = (unsigned)c.H[idx];
    const bool older = grow_older_claim(c, h);
    conf |= ((p >> 31) | (older ? 1u : 0u)) & m;
    assumed = older && m == 0u && rec_is_candidate(p);
    return p | (m << 31) | (assumed ? LSD_USED : 0u);
  } else {
    return c.P[idx];
  }
}
template <bool MW>
__device__ __forceinline__ void grow_mark(const GrowCtx& c, uint32_t idx, unsigned rec) {
  if constexpr (MW) { c.M[idx] = 1; c.H[idx] = (uint16_t)c.hTag; }
  else c.P[idx] = rec | LSD_USED;
}
template <bool MW>
__device__ __forceinline__ void grow_unmark(const GrowCtx& c, uint32_t idx, unsigned rec) {
  if constexpr (MW) { c.M[idx] = 0; c.H[idx] = 0; }
  else c.P[idx] = rec & ~LSD_USED;
}
// store -> load order between the lanes of one wavefront (queue and mark stores read back by other lanes).  A block of the
// one-wavefront kernels is one wavefront, where __syncthreads() is that; the multi-wavefront kernel must not meet a block
// barrier inside a transaction.
template <bool MW>
__device__ __forceinline__ void grow_lane_fence() {
  if constexpr (MW) {
    wave_fence();
    PLH_WAVE_SYNC();
  } else {
    __syncthreads();
  }
}
// the table values of a record (one 16-byte gather; the hot part of the table is L2-resident)
__device__ __forceinline__ LsdPix lsd_fetch_px(const LsdAngleEntry* T, unsigned rec) {
  const uint4 e = *reinterpret_cast<const uint4*>(T + (rec & LSD_REC_IDX));
  LsdPix px;
  px.angf = __uint_as_float(e.x); px.cs = __uint_as_float(e.y); px.sn = __uint_as_float(e.z); px.q = rec;
  return px;
}
__device__ __forceinline__ LsdPix lsd_null_px(unsigned rec) {
  LsdPix px;
  px.angf = 0.f; px.cs = 0.f; px.sn = 0.f; px.q = rec;
  return px;
}

__device__ __forceinline__ int pk_x(uint32_t p) { return (int)(p & 0xffffu); }
__device__ __forceinline__ int pk_y(uint32_t p) { return (int)(p >> 16); }
// rows and pitch are below 2^16 (plh_line_create): one v_mad_u32_u24 (a 32-bit v_mul_lo / 64-bit v_mad are quarter rate)
__device__ __forceinline__ uint32_t pk_lin(const GrowCtx& c, uint32_t p) { return lsd_rec_index((unsigned)pk_x(p), (unsigned)pk_y(p), (unsigned)c.spitch); }   // index of the pixel's record (and of its marks / claim tags)

__device__ __forceinline__ uint32_t reg_get(const GrowCtx& c, int i, int cnt) {
  return (cnt - i <= LSD_RING) ? c.ring[i & (LSD_RING - 1)] : c.reg[i];
}

// (bcast_u32 / bcast_f32 / bcast_f64 -- v_readlane with a uniform lane -- and every other instruction shim: plh_shims.h)

// Per-wavefront state that lives in LDS rather than in registers (uniform values the compiler would keep in VGPRs):
// the parameters of the current region_grow() call, the fitted rectangle, and the first-step prefetch tables.
struct GrowState {
  double* d;          // [0] prec of the call, [1..5] rectangle x1 y1 x2 y2 width, [6..8] its theta, dx, dy (LSD_GS_D doubles)
  uint32_t* u;        // [0] seed (packed), [1] seed q, [2..4] bits of seed angle (degrees), cos, sin
  const LsdPix* fstPx;      // [64] neighbour records of the batch's seeds (lane group t = seed t)
  const uint32_t* fstIdx;   // [64] linear index, 0xffffffff = out of bounds / no seed
  const uint32_t* fstPk;    // [64] packed coordinates
};

// A seed as flsd() hands it to its first region_grow(): wave-uniform values (scalar registers).  refine()'s second call takes its
// parameters from GrowState instead (lane 0 writes them).
struct LsdSeed {
  uint32_t pk;          // packed coordinates
  unsigned q;           // its record (not marked)
  float ang, cx, sy;    // its level-line angle in degrees, (float)cos / (float)sin of the double angle
};

// (isAligned() of cv::LineSegmentDetector for a defined pixel -- lsd_aligned -- is in lsd_rect_dev.h)

// The same test decided in float degrees whenever the angles are clearly inside / outside the tolerance; only the
// pixels within 2e-3 degrees of a decision boundary (tolerance or the 270-degree fold) take the exact double path.
// (theta = thF * DEG_TO_RADS and a = aF * DEG_TO_RADS carry ~1e-15 rad of rounding, the float difference ~3e-5 degrees.)
struct LsdTol {
  double prec;
  float lo, hi;
  float cin2, cout2;   // cos^2(prec -+ LSD_ALIGN_MARGIN): the direction test of lsd_classify
  float posT;          // 0; -inf (with cin2 = +inf, cout2 = -1) when prec + margin >= 89 degrees and the test does not apply
};
// The margin of the direction test, in degrees.  It has to cover |fastAtan2 - atan2| (0.0096 degrees for this polynomial,
// scanned over 4 M ratios, plus the float rounding of the octant folding, < 1e-4), the direction error of the stored float
// cos / sin of a pixel angle (< 1e-5) and the float rounding of the test itself (< 2e-4).
constexpr double LSD_ALIGN_MARGIN_DEG = 0.05;
// cos^2 of the tolerance -+ margin for a tolerance that is not the launch's default (refine()'s tau; rare): out of line, the
// library cosine must not cost the hot loop registers
struct LsdMargins { float cin2, cout2; int fast; };
__device__ __attribute__((noinline)) LsdMargins lsd_tol_margins(double prec) {
  const double m = LSD_ALIGN_MARGIN_DEG * kDegToRads;
  LsdMargins r;
  r.fast = prec + m < 89.0 * kDegToRads;
  const double ci = prec > m ? cos(prec - m) : 2.0, co = cos(prec + m);
  r.cin2 = (float)(ci * ci);
  r.cout2 = (float)(co * co);
  return r;
}
__device__ __forceinline__ LsdTol lsd_tol(double prec) {
  LsdTol t;
  t.prec = prec;
  const float pd = (float)(prec * (180.0 / kPI));
  const float m = 2e-3f + pd * 1e-6f;
  t.lo = pd - m;
  t.hi = pd + m;
  t.cin2 = __builtin_inff(); t.cout2 = -1.f; t.posT = -__builtin_inff();
  return t;
}
__device__ __forceinline__ bool lsd_aligned_f(float thF, float aF, const LsdTol& t) {
  float d = fabsf(thF - aF);
  const bool nearFold = fabsf(d - 270.f) < 1e-2f;
  if (d > 270.f) d = fabsf(d - 360.f);
  bool r = d < t.lo;
  if (nearFold || (d >= t.lo && d <= t.hi)) r = lsd_aligned((double)thF * kDegToRads, (double)aF * kDegToRads, t.prec);
  return r;
}

#define LSD_INV_BALLOT(c, m) PLH_INV_BALLOT(m)   // a wave mask as a per-lane predicate (plh_shims.h)

// lsd_aligned_f for a whole step, as wave masks: every compare result is used as the 64-bit mask it already is and the
// set logic runs on the scalar unit.  `act` = lanes whose answer matters; only they can trigger the exact double path.
__device__ __forceinline__ unsigned long long lsd_aligned_mask(const GrowCtx& c, float thF, float aF, const LsdTol& t,
                                                              unsigned long long act) {
  float d = fabsf(thF - aF);
  const unsigned long long nearFold = wballot(fabsf(d - 270.f) < 1e-2f);
  if (d > 270.f) d = fabsf(d - 360.f);
  unsigned long long r = wballot(d < t.lo);
  const unsigned long long ex = ((~r & wballot(d <= t.hi)) | nearFold) & act;   // d >= lo  <=>  !(d < lo): angles are finite
  PF_ADD(c, 32, __popcll(act)); PF_ADD(c, 33, __popcll(ex));   // lanes decided by the reference's fastAtan2 arithmetic / by its double form
  if (ex) {
    bool e = false;
    if (LSD_INV_BALLOT(c, ex)) e = lsd_aligned((double)thF * kDegToRads, (double)aF * kDegToRads, t.prec);
    r = (r & ~ex) | (wballot(e) & ex);
  }
  return r & act;
}

// fast_atan2_deg (plh_common.h) for the running sums of region_grow(), with the IEEE division written out as the
// reciprocal / residual sequence the compiler emits between v_div_scale and v_div_fixup -- without those three.  They
// only act on operands whose quotient or reciprocal leaves the normal range, and here it cannot: the divisor is
// max(|x|, |y|) + 2.2e-16 with |x|, |y| <= 2^18 (a region has at most 2^18 pixels, each adds a cosine and a sine), and the
// dividend min(|x|, |y|) is either 0 or at least 2^-48 (every term is a float of magnitude >= 4e-8 -- the cosine at the
// float next to pi/2 -- so a sum is a multiple of 2^-48).  Same operations, same roundings, same result.
__device__ __forceinline__ float lsd_atan2_deg(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
  const float ax = fabsf(x), ay = fabsf(y);
  const bool swap = !(ax >= ay);
  const float num = swap ? ax : ay, den = (swap ? ay : ax) + 2.2204460492503131e-16f;
  const float c = div_normal(num, den);   // (plh_shims.h: the division without v_div_scale / v_div_fixup; plh_selftest compares it with `/`)
  const float c2 = c * c;
  float a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  if (swap) a = 90.f - a;
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// The walk of lsd_resolve: the predicted-accepted lanes of mask P in lane order; lane k adds its (cos, sin) to the running
// sums of every lane behind it and cancels its later duplicates (same pixel examined from another queue point).  On return
// acc = the lanes walked, canc = the duplicates dropped (possibly with lanes outside the candidate set: only ever used
// masked).  Both forms -- the hand-scheduled gfx950 loop and the plain one that says what it computes -- are in plh_shims.h.
__device__ __forceinline__ void lsd_walk(const GrowCtx&, unsigned long long P, bool mayDup, uint32_t nidx, float cs, float sn,
                                         float& preX, float& preY, unsigned long long& accOut, unsigned long long& cancOut) {
  plh::lsd_walk(P, mayDup, nidx, cs, sn, preX, preY, accOut, cancOut);
}

// Which way does a candidate's decision go, judged by directions instead of fastAtan2 values?  The reference compares the
// pixel angle a with reg_angle = fastAtan2(sumdy, sumdx) (or the seed's own angle before the first accept), folded to the
// circular distance, against prec.  fastAtan2 is within 0.01 degrees of the true direction of (x, y) = the running sums, so
// with D = the true angle between (x, y) and (cos a, sin a):  D <= prec - margin  =>  aligned,  D >= prec + margin  =>  not.
// cos D = (cs x + sn y) / |(x, y)|, compared in squares (needs prec + margin < 90 degrees).  Only the lanes in between
// -- a 0.1-degree band around the tolerance -- need the reference's arithmetic.  Ten VALU instructions; the exact test costs
// a fastAtan2 (30) per state plus the compare (9).  `in` / `unc` = certainly aligned / undecided lanes of `act`.
__device__ __forceinline__ void lsd_classify(float x, float y, float cs, float sn, const LsdTol& t, unsigned long long act,
                                             unsigned long long& in, unsigned long long& unc) {
  // (a tolerance the test does not apply to carries posT = -inf, cin2 = +inf, cout2 = -1: every lane comes out undecided)
  const float dot = __builtin_fmaf(sn, y, cs * x), n2 = __builtin_fmaf(y, y, x * x), dd = dot * dot;
  const unsigned long long pos = wballot(dot > t.posT);
  const unsigned long long ge = wballot(dd >= t.cin2 * n2), le = wballot(dd <= t.cout2 * n2);
  in = pos & ge & act;
  unc = act & ~(in | ~pos | le);
}

// remember the pixels taken for used (rare: a region growing next to an older one in flight); when the list is full the
// run stops assuming (hWin = 0 would need a non-const context: the caller sees the count and gives the run up)
template <bool MW>
__device__ __forceinline__ void grow_note_assumed(const GrowCtx& c, bool assumed, uint32_t npk) {
  if constexpr (MW) {
    const unsigned long long am = wballot(assumed);
    if (am) {
      const unsigned n0 = bcast_u32(*c.asmCnt, 0);
      if (LSD_INV_BALLOT(c, am)) {
        const unsigned k = n0 + (unsigned)mbcnt64(am);
        if (k < (unsigned)MW_ASM_CAP) c.asmList[k] = npk;
      }
      PLH_WAVE_SYNC();
      if (c.lane == 0) *c.asmCnt = n0 + (unsigned)__popcll(am);   // may exceed the capacity: the caller checks
      PLH_WAVE_SYNC();
    }
  }
}
// One step's candidates: lane order == the reference's examination order (queue point, then yy, then xx).
struct LsdCand {
  LsdPix px;
  uint32_t nidx, npk;
  bool inb;
};

// Resolve the candidates of one step with the sequential semantics of region_grow(): every accepted pixel
// moves the running region angle, a pixel accepted for an earlier point cancels its later duplicates.
//
// The sequential chain is cut down to one packed float add per accepted pixel.  Each pass
//   1. predicts every remaining candidate against the current (exact) running sums,
//   2. walks the predicted-accepted lanes in order, adding their (cos, sin) to the running sums of all LATER
//      lanes -- so lane j holds exactly the sums the reference has when it examines candidate j, built by the
//      same sequence of float additions,
//   3. lets every lane test its alignment against the state in front of it (its own pre-sums) -- in parallel,
//   4. commits everything up to the first lane whose real decision differs from the prediction (decisions in
//      front of it were taken on exact states, so they are the reference's), and repeats from there.
// Both tests (1, 3) are decided by lsd_classify (directions, ten instructions) for every lane that is not within 0.05 degrees of
// the tolerance; only for those does a pass evaluate what the reference evaluates -- reg_angle = fastAtan2 of the sums in front
// of the lane, then the folded angle difference (lsd_aligned_mask).  The region angle as a number is therefore not kept up
// to date: `angValid` says whether regAngF still is the reference's reg_angle for (sumdx, sumdy); it is recomputed when a lane
// needs it and once at the end of region_grow().  Mispredictions only happen for pixels within the step's angle drift of the
// tolerance boundary.  Returns the mask of the lanes whose pixel was accepted.
template <bool MW>
__device__ __forceinline__ unsigned long long lsd_resolve(const GrowCtx& c, unsigned long long rem, const LsdCand& cd,
                                                          bool mayDup, const LsdTol& tol, float& sumdx, float& sumdy,
                                                          float& regAngF, bool& angValid, int& cnt) {
  // rem = mask of the candidate lanes; all bookkeeping below is on 64-bit masks (scalar unit)
  unsigned long long accAll = 0;
  PF_ADD(c, 10, __popcll(rem));
  if (!rem) return 0;
  while (rem) {
    unsigned long long P, unc;
    lsd_classify(sumdx, sumdy, cd.px.cs, cd.px.sn, tol, rem, P, unc);
    if (unc) {
      if (!angValid) { regAngF = lsd_atan2_deg(sumdy, sumdx); angValid = true; }
      P |= lsd_aligned_mask(c, regAngF, cd.px.angf, tol, unc);
    }
    if (!P) break;   // the state cannot change any more: every remaining candidate is rejected on the exact state
    PF_ADD(c, 12, 1);
    float preX = sumdx, preY = sumdy;
    unsigned long long acc, canc;
    lsd_walk(c, P, mayDup, cd.nidx, cd.px.cs, cd.px.sn, preX, preY, acc, canc);
    const unsigned long long live = rem & ~canc;   // (canc may hold lanes outside rem: harmless here)
    unsigned long long D;
    lsd_classify(preX, preY, cd.px.cs, cd.px.sn, tol, live, D, unc);
    if (unc) {
      // the reference's reg_angle in front of a lane: fastAtan2 of its pre-sums once something was accepted below it in this
      // pass, the region angle as it stood before
      const unsigned long long moved = ~1ull << (__ffsll((long long)acc) - 1);   // lanes behind the first walked lane
      if ((unc & ~moved) && !angValid) { regAngF = lsd_atan2_deg(sumdy, sumdx); angValid = true; }
      float angPrev = lsd_atan2_deg(preY, preX);
      if (!LSD_INV_BALLOT(c, moved)) angPrev = regAngF;
      D |= lsd_aligned_mask(c, angPrev, cd.px.angf, tol, unc);
    }
    const unsigned long long mism = (D ^ acc) & live;
    unsigned long long A = acc;
    int f = 64;
    if (mism) {
      f = __ffsll((long long)mism) - 1;
      A &= (1ull << f) - 1ull;
      if (!((acc >> f) & 1ull)) A |= 1ull << f;   // predicted rejected, really accepted
    }
    if (A) {
      if (LSD_INV_BALLOT(c, A)) {
        const unsigned slot = (unsigned)cnt + (unsigned)mbcnt64(A);   // accepted lanes below this one: v_mbcnt on the scalar mask
        c.reg[slot] = cd.npk;
        c.ring[slot & (LSD_RING - 1)] = cd.npk;
        grow_mark<MW>(c, cd.nidx, cd.px.q);
      }
      accAll |= A;
      const int last = 63 - __clzll((long long)A);
      sumdx = bcast_f32(preX + cd.px.cs, last);
      sumdy = bcast_f32(preY + cd.px.sn, last);
      angValid = false;
      cnt += __popcll(A);
    }
    if (f >= 64) break;
    PF_ADD(c, 13, 1);
    rem &= ~((2ull << f) - 1ull);
    if (mayDup && rem) {   // drop the duplicates of what has just been committed
      unsigned long long am = A;
      while (am) {
        const int k = __ffsll((long long)am) - 1;
        am &= am - 1;
        rem &= ~wballot(cd.nidx == bcast_u32(cd.nidx, k));
      }
    }
  }
  return accAll;
}

// Addressing of one neighbour lane: lane 8g+n looks at neighbour n (yy outer, xx inner, centre skipped) of queue
// point q = base+g.  Returns whether the lane has an in-bounds pixel to examine.
__device__ __forceinline__ bool lsd_addr(const GrowCtx& c, int i, int cnt, uint32_t& nidx, uint32_t& npk) {
  const int n = c.lane & 7, q = i + (c.lane >> 3);
  const int nb = n < 4 ? n : n + 1;
  const int dy = nb / 3 - 1, dx = nb - (nb / 3) * 3 - 1;
  uint32_t p = c.ring[q & (LSD_RING - 1)];
  if (cnt - i > LSD_RING) {   // the queue ran ahead of the LDS mirror (rare; uniform test, kept a real branch)
    if (q < cnt && cnt - q > LSD_RING) p = *(volatile const uint32_t*)&c.reg[q];
  }
  // Queue points are defined pixels, and k_lsd_grad never defines the last column / row: x + dx <= sw - 1 and
  // y + dy <= sh - 1 always hold, only the -1 side can leave the image.
  const int xx = pk_x(p) + dx, yy = pk_y(p) + dy;
  const bool inb = q < cnt && (xx | yy) >= 0;
  nidx = inb ? lsd_rec_index((unsigned)xx, (unsigned)yy, (unsigned)c.spitch) : c.nullIdx;   // (sw-1, 0): never defined, never marked
  npk = (uint32_t)xx | ((uint32_t)yy << 16);
  return inb;
}

// region_grow(): BFS over reg[] used as a queue, 8 queued points (64 neighbour lanes) per step; each lane fetches
// its neighbour's 4-byte level-line record (table index | DEF | used mark), issued right after the previous step's
// marks, and -- if the pixel is a candidate -- its (angle, cos, sin) from the gradient table.  (A frontier wider than 8 points is rare -- 7 % of the steps -- so prefetching across steps
// costs more instructions than it hides; latency is hidden by the other resident frames.)  `first` holds the seed's
// 8 neighbours prefetched in lane group `firstGrp` (firstGrp < 0: not prefetched).
// Returns the region size; regAngF = final reg_angle in degrees (reg_angle = regAngF * DEG_TO_RADS, exactly the
// reference's float fastAtan2 result; evaluated only if the region has at least minCnt pixels).  All lanes hold identical
// (uniform) state.
// `fstQ` (dirtyFst only): lane group firstGrp's records of the seed's neighbours as they are now.
// `firstCall` (a literal at both call sites): a seed's first region_grow() -- the seed comes in `sd` and the tolerance is the launch's, with
// everything lsd_tol() would derive from it in the context; gs is not read.  (Round 4 handed every call its parameters through LDS:
// lane 0 stores, everybody loads, v_readlane -- five dependent LDS round trips, an f64 multiply and, in the 72-register build, two
// reloads from scratch for each of a frame's 8 k calls, three in four of which end after one step.)  refine()'s call: parameters in gs.
template <bool MW>
__device__ __forceinline__ int lsd_region_grow(const GrowCtx& c, const GrowState& gs, const LsdSeed& sd, int firstGrp, bool dirtyFst, unsigned fstQ,
                                               int minCnt, float* regAngOut, bool* conflictOut, const bool firstCall) {
  const int lane = c.lane, g = lane >> 3;
  uint32_t seedPk;
  unsigned seedQ;
  float regAngF, sumdx, sumdy;
  LsdTol tol;
  if (firstCall) {
    seedPk = sd.pk; seedQ = sd.q; regAngF = sd.ang; sumdx = sd.cx; sumdy = sd.sy;
    tol.prec = c.precDef; tol.lo = c.loDef; tol.hi = c.hiDef;
    tol.cin2 = __builtin_inff(); tol.cout2 = -1.f; tol.posT = -__builtin_inff();
    if (c.fastDef) { tol.cin2 = c.cin2Def; tol.cout2 = c.cout2Def; tol.posT = 0.f; }
  } else {
    // uniform values from LDS: nothing of the caller's state has to stay in registers across the loop
    seedPk = bcast_u32(gs.u[0], 0);
    seedQ = bcast_u32(gs.u[1], 0);
    regAngF = bcast_f32(__uint_as_float(gs.u[2]), 0); sumdx = bcast_f32(__uint_as_float(gs.u[3]), 0);
    sumdy = bcast_f32(__uint_as_float(gs.u[4]), 0);
    tol = lsd_tol(bcast_f64(gs.d[0], 0));
    if (tol.prec == c.precDef) {   // the launch's tolerance: margins from the host
      if (c.fastDef) { tol.cin2 = c.cin2Def; tol.cout2 = c.cout2Def; tol.posT = 0.f; }
    } else {
      const LsdMargins mg = lsd_tol_margins(tol.prec);   // results of a call come back in vector registers: make them scalar again
      if (bcast_u32((unsigned)mg.fast, 0) != 0u) { tol.cin2 = bcast_f32(mg.cin2, 0); tol.cout2 = bcast_f32(mg.cout2, 0); tol.posT = 0.f; }
    }
  }
  bool angValid = true;   // before the first accept reg_angle is the seed's own angle
  const uint32_t seed = pk_lin(c, seedPk);
  PLH_WAVE_SYNC();
  if (lane == 0) {
    c.reg[0] = seedPk;
    c.ring[0] = seedPk;
    grow_mark<MW>(c, seed, seedQ);
  }
  PF_ADD(c, 11, 1);
  int cnt = 1, i = 0;
  PLH_WAVE_SYNC();
  if (firstGrp >= 0) {
    const unsigned long long pt1 = PF_NOW_FINE();
    // the seed's 8 neighbours were prefetched into LDS by lane group firstGrp; their marks may have changed since
    // (table addresses from an opaque copy of the lane number: as loop invariants of the whole kernel they end up in scratch)
    LsdCand first;
    const int l1 = (int)plh_opaque_u32((unsigned)lane);
    first.nidx = gs.fstIdx[l1];
    first.inb = g == firstGrp && first.nidx != 0xffffffffu;
    if (!first.inb) first.nidx = 0;
    first.npk = gs.fstPk[l1];
    first.px = gs.fstPx[l1];
    // (marks may have changed since the prefetch: the caller re-read the records of lane group firstGrp -- fstQ -- in the same round
    // trip as the seed's own)
    if (dirtyFst && first.inb) first.px.q = fstQ;
    // one signed compare on the record word covers "not marked and above the gradient threshold" (the table values were
    // fetched with the neighbourhood for every DEFINED pixel, so a pixel that refine() un-marked in between has them too)
    const unsigned long long candM = wballot(first.inb) & wballot(rec_is_candidate(first.px.q));
    lsd_resolve<MW>(c, candM, first, false, tol, sumdx, sumdy, regAngF, angValid, cnt);
    PF_ADD(c, 4, PF_NOW_FINE() - pt1); PF_ADD(c, 8, 1);
    i = 1;
  }
  if (i >= cnt) {
    PF_ADD(c, 9, cnt);
    *regAngOut = regAngF;   // nothing accepted: the seed's angle (and the caller drops the region anyway)
    return cnt;
  }
  PLH_WAVE_SYNC();
  LsdCand cur;
  unsigned conf = 0;
  bool assumed = false;
  cur.inb = lsd_addr(c, i, cnt, cur.nidx, cur.npk);
  {
    const unsigned rec = grow_load_rec<MW>(c, cur.nidx, conf, assumed);
    cur.px = rec_is_candidate(rec) ? lsd_fetch_px(c.A, rec) : lsd_null_px(rec);
    grow_note_assumed<MW>(c, assumed, cur.npk);
  }
  while (i < cnt) {
    const unsigned long long pt0 = PF_NOW_FINE();
    const int m = min(LSD_PTS, cnt - i);
    // lanes with nothing to examine loaded the NOTDEF pixel (sw-1, 0): one signed compare on the record word decides
    const unsigned long long candM = wballot(rec_is_candidate(cur.px.q));
    const unsigned long long pt1 = PF_NOW_FINE();
    PF_ADD(c, 3, pt1 - pt0); PF_ADD(c, 8, 1);
    lsd_resolve<MW>(c, candM, cur, true, tol, sumdx, sumdy, regAngF, angValid, cnt);
    PF_ADD(c, 4, PF_NOW_FINE() - pt1);
    i += m;
    // the next step's records, requested after this step's marks were stored (a wavefront observes its own stores);
    // every lane loads (record 0 when it has nothing to examine) so the carried registers are simply overwritten
    PLH_WAVE_SYNC();
    cur.inb = lsd_addr(c, i, cnt, cur.nidx, cur.npk);
    {   // the 4-byte record, then -- for candidates only -- its table values
      const unsigned rec = grow_load_rec<MW>(c, cur.nidx, conf, assumed);
      cur.px = rec_is_candidate(rec) ? lsd_fetch_px(c.A, rec) : lsd_null_px(rec);
      grow_note_assumed<MW>(c, assumed, cur.npk);
    }
    if constexpr (MW) {
      if (wballot(conf != 0u)) {   // overtaken by an older transaction's commit: the caller takes the marks back and starts over
        *conflictOut = true;
        *regAngOut = regAngF;
        return cnt;
      }
    }
  }
  PF_ADD(c, 9, cnt);
  if (!angValid && cnt >= minCnt) regAngF = lsd_atan2_deg(sumdy, sumdx);   // the caller only looks at regions it keeps
  *regAngOut = regAngF;
  return cnt;
}

// Sequential double sums of up to three series at once.  Every lane has stored its three terms for element
// base+lane into c.T ([3][64], zero beyond the end); lane ch < 3 then adds series ch in element order from LDS,
// so the three dependent v_add_f64 chains of the reference run side by side in three lanes, one add per element
// (adding the +0.0 padding is exact: the accumulators are never -0.0).
__device__ __forceinline__ double lsd_chain_add(const GrowCtx& c, double acc, int n) {
  const int ch = min(c.lane, 2);
  const D2* row = reinterpret_cast<const D2*>(c.T + ch * 64);
  const int n2 = ((n + 7) >> 3) << 2;
  for (int l = 0; l < n2; l += 4) {
    const D2 v0 = row[l], v1 = row[l + 1], v2 = row[l + 2], v3 = row[l + 3];
    acc += v0.x; acc += v0.y; acc += v1.x; acc += v1.y;
    acc += v2.x; acc += v2.y; acc += v3.x; acc += v3.y;
  }
  return acc;
}

// region2rect() + get_theta().  The weighted sums are accumulated in region order (lsd_chain_add) so the doubles
// are bit-identical to the sequential reference; the extents are min/max (exact).
// cos / sin of the rectangle angle (theta in [0, 3 pi)): head + tail evaluation (plh_common.h), out of line (inlined twice it
// costs the kernel registers it does not have): 60 instructions instead of the library routine's 153, 2 300 times per frame,
// and as close to the host libm as the library routine is (either differs from glibc in 3.1 % of the values, by one unit in the
// last place; line extractor 141.1 -> 139.3 ms per 6144 frames).
__device__ __attribute__((noinline)) D2 lsd_sincos(double t) { return lsd_sincos_inl(t); }   // x = cos, y = sin

__device__ void lsd_region2rect(const GrowCtx& c, int cnt, double reg_angle, double prec, double* rec) {
  const int lane = c.lane;
  double acc = 0;
  for (int base = 0; base < cnt; base += 64) {
    const int i = base + lane;
    double w = 0, wx = 0, wy = 0;
    if (i < cnt) {
      const uint32_t p = c.reg[i];
      w = c.A[c.P[pk_lin(c, p)] & LSD_REC_IDX].modgrad;
      wx = (double)pk_x(p) * w;
      wy = (double)pk_y(p) * w;
    }
    PLH_WAVE_SYNC();
    c.T[lane] = wx; c.T[64 + lane] = wy; c.T[128 + lane] = w;
    PLH_WAVE_SYNC();
    acc = lsd_chain_add(c, acc, min(64, cnt - base));
  }
  double x = bcast_f64(acc, 0), y = bcast_f64(acc, 1);
  const double sum = bcast_f64(acc, 2);
  x /= sum;
  y /= sum;
  acc = 0;
  for (int base = 0; base < cnt; base += 64) {
    const int i = base + lane;
    double a = 0, b = 0, cc = 0;
    if (i < cnt) {
      const uint32_t p = c.reg[i];
      const double w = c.A[c.P[pk_lin(c, p)] & LSD_REC_IDX].modgrad;
      const double ddx = (double)pk_x(p) - x, ddy = (double)pk_y(p) - y;
      a = ddy * ddy * w;
      b = ddx * ddx * w;
      cc = -(ddx * ddy * w);   // Ixy -= v  ==  Ixy += -v
    }
    PLH_WAVE_SYNC();
    c.T[lane] = a; c.T[64 + lane] = b; c.T[128 + lane] = cc;
    PLH_WAVE_SYNC();
    acc = lsd_chain_add(c, acc, min(64, cnt - base));
  }
  const double Ixx = bcast_f64(acc, 0), Iyy = bcast_f64(acc, 1), Ixy = bcast_f64(acc, 2);
  const double theta = lsd_rect_theta(Ixx, Iyy, Ixy, reg_angle, prec);
  const D2 cs = lsd_sincos(theta);
  const double dx = cs.x, dy = cs.y;
  double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
  for (int i = lane; i < cnt; i += 64) {
    const uint32_t p = c.reg[i];
    const double rdx = (double)pk_x(p) - x, rdy = (double)pk_y(p) - y;
    const double l = rdx * dx + rdy * dy;
    const double w = -rdx * dy + rdy * dx;
    l_max = fmax(l_max, l); l_min = fmin(l_min, l);
    w_max = fmax(w_max, w); w_min = fmin(w_min, w);
  }
  // Wave-wide extremes of the four series through LDS (exact in any order): the minima as maxima of the negated values,
  // series s in lanes 16 s .. 16 s + 15, three levels of four-to-one (two 16-byte reads + three v_max_f64 each).  The
  // shuffle butterfly this replaces took 48 ds_bpermute and 70 VALU instructions per call, 2 300 calls per frame.
  {
    PLH_WAVE_SYNC();
    c.T[lane] = l_max; c.T[64 + lane] = -l_min; c.T[128 + lane] = w_max; c.T[192 + lane] = -w_min;
    PLH_WAVE_SYNC();
    const int sr = lane >> 4, j = lane & 15;
    const D2* rp = reinterpret_cast<const D2*>(c.T + sr * 64 + 4 * j);
    D2 u = rp[0], v = rp[1];
    double m = fmax(fmax(u.x, u.y), fmax(v.x, v.y));
    PLH_WAVE_SYNC();
    c.T[lane] = m;   // series sr: 16 partial maxima at [16 sr, 16 sr + 16)
    PLH_WAVE_SYNC();
    rp = reinterpret_cast<const D2*>(c.T + sr * 16 + 4 * (j & 3));
    u = rp[0]; v = rp[1];
    m = fmax(fmax(u.x, u.y), fmax(v.x, v.y));
    PLH_WAVE_SYNC();
    c.T[lane] = m;   // lanes j < 4 of a series: its four partial maxima at [16 sr, 16 sr + 4)
    PLH_WAVE_SYNC();
    rp = reinterpret_cast<const D2*>(c.T + sr * 16);
    u = rp[0]; v = rp[1];
    m = fmax(fmax(u.x, u.y), fmax(v.x, v.y));
    l_max = bcast_f64(m, 0); l_min = -bcast_f64(m, 16);
    w_max = bcast_f64(m, 32); w_min = -bcast_f64(m, 48);
  }
  PLH_WAVE_SYNC();
  if (lane == 0) {   // rec[] is in LDS: x1 y1 x2 y2 width
    rec[0] = x + l_min * dx; rec[1] = y + l_min * dy;
    rec[2] = x + l_max * dx; rec[3] = y + l_max * dy;
    const double width = w_max - w_min;
    rec[4] = width < 1.0 ? 1.0 : width;
    rec[5] = theta; rec[6] = dx; rec[7] = dy;   // (read by rect_nfa only)
  }
  PLH_WAVE_SYNC();
}

__device__ __forceinline__ double rect_density(int cnt, const double* r) {
  return (double)cnt / (sqrt(dist_sq(r[0], r[1], r[2], r[3])) * r[4]);
}

// ---------------------------------------------------------------------------------------------
// The density screen.  What region growing needs from region2rect() is, nearly always, one bit: is the region's density
// cnt / (length x width) of its rectangle at least densityTh?  (refine() and reduce_region_radius() only run when it is not, and
// the rectangle of a region that is kept changes no mark.)  The exact rectangle costs a wavefront ~800 instructions -- two
// passes of sequential double sums on 3 of 64 lanes, two divisions, sqrt, sincos -- for a decision that is rarely close.  So
// the wavefront first brackets the exact density with float arithmetic in any order (~200 instructions, one gather):
//   * weights sqrt(gx^2 + gy^2) from the record's table index (the common factor 1/2 of modgrad cancels in the centroid and
//     in the direction), raw moments about the seed in float, summed through LDS;
//   * the direction from the same fastAtan2 formula on the float inertia terms, cos / sin by v_cos / v_sin;
//   * extents L~, W~ of the region about the float centroid along that direction.
// Error budget (R = region radius <= L~, all in pixels / radians):
//   - the exact theta is fastAtan2 of float casts of double inertia terms; ours is fastAtan2 of float terms whose relative
//     error is <= 2e-6 x (raw trace / eigenvalue gap) -- the screen gives up unless that is <= 1e-4 -- so the two arguments
//     agree to 1e-4 and the results to 1e-4 rad, EXCEPT across the polynomial's octant seam (|x| = |y|: a jump of 0.019
//     degrees) and the |Ixx| > |Iyy| choice of formula (both formulas give the same direction to twice the polynomial's
//     error, 0.019 degrees, or its opposite -- a rotation by pi leaves length and width alone).  delta = 5e-4 rad covers all
//     of it (3.4e-4 + 1e-4 + v_sin / v_cos 1e-5) without looking at which case applies;
//   - a rotation of the axes by delta moves an extent by at most delta x (the other extent); float rounding of the
//     coordinates, the centroid and the products moves it by < 1e-5 x (L~ + W~ + 1).
// Hence  L* in [L~ - a, L~ + a],  W* in [W~ - b, W~ + b]  with a = delta W~ + eps, b = delta L~ + eps, and the exact density
// lies between cnt / ((L~ + a) max(W~ + b, 1)) and cnt / ((L~ - a) max(W~ - b, 1)).  Verdict +1 / -1 only when the whole
// bracket lies on one side of densityTh (with 2e-5 of slack for the float products); 0 = undecided: the caller evaluates the
// exact rectangle, as it always did.  Either way the decisions taken are the reference's, so the segments are; the exact
// rectangle of a kept region is evaluated later, one lane per region (k_lsd_rects).  A -DPLH_GROW_PROF build checks every
// verdict against the exact density (counter 39 = contradictions: must stay 0; tools/grow_prof.py, tests/test_soak_gpu.py).
// ---------------------------------------------------------------------------------------------
constexpr float LSD_SCREEN_DELTA = 5e-4f;
constexpr int LSD_SCREEN_MAX = 4096;   // beyond this the float sums' own error grows with the count: not screened (rare)

__device__ __forceinline__ float screen_f(unsigned u) { return __uint_as_float(u); }
__device__ __forceinline__ float screen_sum4(const uint4 v) { return (screen_f(v.x) + screen_f(v.y)) + (screen_f(v.z) + screen_f(v.w)); }
__device__ __forceinline__ float screen_max4(const uint4 v) { return fmaxf(fmaxf(screen_f(v.x), screen_f(v.y)), fmaxf(screen_f(v.z), screen_f(v.w))); }

// `fromRing`: the queue is as region_grow() left it (its newest LSD_RING entries mirrored in LDS); false after
// reduce_region_radius() has permuted it in global memory.
// (one wavefront per frame: regq of a region the screen did not walk -- screen off, or more than LSD_SCREEN_MAX pixels)
__device__ __forceinline__ void lsd_fill_regq(const GrowCtx& c, int cnt) {
  for (int i = c.lane; i < cnt; i += 64) c.regq[i] = lsd_rec_q(c.P[pk_lin(c, c.reg[i])]);
}
template <bool MW>
__device__ __forceinline__ int lsd_density_screen(const GrowCtx& c, int cnt, bool fromRing, float thLo, float thHi) {
  if (cnt > LSD_SCREEN_MAX) return 0;
  const int lane = (int)plh_opaque_u32((unsigned)c.lane);   // (per-lane queue addresses formed here, per call: hoisted out of the
                                                            // transaction loop they are 64-bit pairs the 72-register build keeps in scratch)
  const bool inRing = fromRing && cnt <= LSD_RING;   // the whole queue is still in its LDS mirror
  if (!inRing || cnt > 64) grow_lane_fence<MW>();   // the queue in global memory is read below: stores of all lanes visible
  PLH_WAVE_SYNC();
  const uint32_t seedPk = bcast_u32(inRing ? c.ring[0] : c.reg[0], 0);
  const int x0 = pk_x(seedPk), y0 = pk_y(seedPk);
  float S[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // sums of w, w x, w y, w x x, w y y, w x y (x, y relative to the seed)
  uint32_t p0 = seedPk;
  for (int base = 0; base < cnt; base += 64) {
    const int i = base + lane;
    const bool on = i < cnt;
    uint32_t p = seedPk;
    if (on) p = inRing ? c.ring[i] : c.reg[i];
    if (base == 0) p0 = p;
    const unsigned rec = c.P[pk_lin(c, p)];   // (lanes beyond the end read the seed's record; their weight is zero)
    const unsigned q = lsd_rec_q(rec);
    if constexpr (!MW) {
      if (on) c.regq[i] = q;   // beside the queue: if this region is kept, k_lsd_rects needs no record of it again
    }
    const float w = on ? plh_sqrt_approx((float)q) : 0.f;
    const float dx = (float)(pk_x(p) - x0), dy = (float)(pk_y(p) - y0);
    const float wx = w * dx, wy = w * dy;
    S[0] += w; S[1] += wx; S[2] += wy;
    S[3] = __builtin_fmaf(wx, dx, S[3]); S[4] = __builtin_fmaf(wy, dy, S[4]); S[5] = __builtin_fmaf(wx, dy, S[5]);
  }
  float* F = reinterpret_cast<float*>(c.T);   // 512 floats: the ring, dead from here on (region_grow() restarts it)
  {
    PLH_WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < 6; k++) F[64 * k + lane] = S[k];
    PLH_WAVE_SYNC();
    const int s = lane >> 3, j = lane & 7;
    float t = 0.f;
    if (s < 6) {
      const uint4* q = reinterpret_cast<const uint4*>(F + 64 * s + 8 * j);
      t = screen_sum4(q[0]) + screen_sum4(q[1]);
    }
    PLH_WAVE_SYNC();
    F[lane] = t;   // series s: eight partial sums at [8 s, 8 s + 8)
    PLH_WAVE_SYNC();
    float u = 0.f;
    if (lane < 6) {
      const uint4* q = reinterpret_cast<const uint4*>(F + 8 * lane);
      u = screen_sum4(q[0]) + screen_sum4(q[1]);
    }
#pragma unroll
    for (int k = 0; k < 6; k++) S[k] = bcast_f32(u, k);
  }
  const float inv = 1.0f / S[0];
  const float mx = S[1] * inv, my = S[2] * inv;
  const float Ixx = S[4] - S[2] * my, Iyy = S[3] - S[1] * mx, Ixy = S[1] * my - S[5];
  const float d = Ixx - Iyy, g = sqrtf(d * d + 4.f * (Ixy * Ixy));
  // conditioning of the direction: float error of the inertia terms (2e-6 of the raw trace) against the eigenvalue gap
  if (!(2e-6f * (S[3] + S[4]) <= 1e-4f * g)) return 0;   // (also catches NaN)
  const float thDeg = fabsf(Ixx) > fabsf(Iyy) ? fast_atan2_deg(-0.5f * (d + g), Ixy) : fast_atan2_deg(Ixy, -0.5f * (g - d));
  const float turns = thDeg * (1.0f / 360.0f);
  const float cs = cos_turns(turns), sn = sin_turns(turns);
  float E[4] = {0.f, 0.f, 0.f, 0.f};   // max l, max -l, max w, max -w (the reference's extremes start at 0)
  for (int base = 0; base < cnt; base += 64) {
    const int i = base + lane;
    uint32_t p = p0;
    if (base > 0 && i < cnt) p = c.reg[i];
    if (i < cnt) {
      const float rx = (float)(pk_x(p) - x0) - mx, ry = (float)(pk_y(p) - y0) - my;
      const float l = __builtin_fmaf(ry, sn, rx * cs), w = __builtin_fmaf(ry, cs, -(rx * sn));
      E[0] = fmaxf(E[0], l); E[1] = fmaxf(E[1], -l); E[2] = fmaxf(E[2], w); E[3] = fmaxf(E[3], -w);
    }
  }
  {
    PLH_WAVE_SYNC();
#pragma unroll
    for (int k = 0; k < 4; k++) F[64 * k + lane] = E[k];
    PLH_WAVE_SYNC();
    const float t = screen_max4(*reinterpret_cast<const uint4*>(F + 4 * lane));   // lane (s, j): series s = lane >> 4
    PLH_WAVE_SYNC();
    F[lane] = t;   // series s: sixteen partial maxima at [16 s, 16 s + 16)
    PLH_WAVE_SYNC();
    float u = 0.f;
    if (lane < 4) {
      const uint4* q = reinterpret_cast<const uint4*>(F + 16 * lane);
      u = fmaxf(fmaxf(screen_max4(q[0]), screen_max4(q[1])), fmaxf(screen_max4(q[2]), screen_max4(q[3])));
    }
#pragma unroll
    for (int k = 0; k < 4; k++) E[k] = bcast_f32(u, k);
    PLH_WAVE_SYNC();
  }
  const float L = E[0] + E[1], W = E[2] + E[3];
  const float eps = 1e-5f * (L + W + 1.0f);
  const float a = LSD_SCREEN_DELTA * W + eps, b = LSD_SCREEN_DELTA * L + eps;
  const float n = (float)cnt;
  if (n >= thHi * ((L + a) * fmaxf(W + b, 1.0f))) return 1;
  if (n < thLo * (fmaxf(L - a, 0.f) * fmaxf(W - b, 1.0f))) return -1;
  return 0;
}

// One iteration of reduce_region_radius(): drop every point farther than sqrt(radSq) from reg[0].
// The reference removes with "swap with the last element, pop, re-check": near points of the final prefix
// [0, K) stay in place and the holes (far points with index < K, ascending) are filled with the near points
// of the tail [K, cnt) in DESCENDING index order.  That permutation is reproduced here with parallel passes.
template <bool MW>
__device__ int lsd_reduce_radius_step(const GrowCtx& c, int cnt, double xc, double yc, double radSq) {
  const int lane = c.lane;
  int K = 0;
  for (int base = 0; base < cnt; base += 64) {
    const int i = base + lane;
    bool near = false;
    if (i < cnt) {
      const uint32_t p = c.reg[i];
      near = !(dist_sq(xc, yc, (double)pk_x(p), (double)pk_y(p)) > radSq);
      if (!near) {
        const uint32_t li = pk_lin(c, p);
        if constexpr (MW) { c.M[li] = 0; c.H[li] = 0; }
        else c.P[li] &= ~LSD_USED;
      }
    }
    K += __popcll(__ballot(near));
  }
  int H = 0;
  for (int base = 0; base < K; base += 64) {
    const int i = base + lane;
    bool hole = false;
    if (i < K) {
      const uint32_t p = c.reg[i];
      hole = dist_sq(xc, yc, (double)pk_x(p), (double)pk_y(p)) > radSq;
    }
    const unsigned long long m = __ballot(hole);
    if (hole) c.scr[H + __popcll(m & lanemask_lt())] = (uint32_t)i;
    H += __popcll(m);
  }
  int F = 0;
  for (int top = cnt; top > K; top -= 64) {
    const int i = top - 1 - lane;
    bool fil = false;
    uint32_t p = 0;
    if (i >= K) {
      p = c.reg[i];
      fil = !(dist_sq(xc, yc, (double)pk_x(p), (double)pk_y(p)) > radSq);
    }
    const unsigned long long m = __ballot(fil);
    if (fil) c.scr[H + F + __popcll(m & lanemask_lt())] = p;
    F += __popcll(m);
  }
  grow_lane_fence<MW>();
  for (int j = lane; j < H; j += 64) c.reg[c.scr[j]] = c.scr[H + j];
  grow_lane_fence<MW>();
  return K;
}

// What a transaction -- the body of flsd()'s loop for one seed that is free at its turn (oracle/lsd.cc run(): region_grow ->
// region2rect -> refine [-> region_grow -> region2rect -> reduce_region_radius]) -- leaves behind.  The rectangle itself is
// not part of it: a region that is kept goes to the frame's log and k_lsd_rects evaluates its rectangle (lsd_rects.hip).
struct LsdTxn {
  int logLen;            // several wavefronts per frame: pixels ever accepted, base[0 .. logLen)
  int finBase, finCnt;   // the pixels still marked at the end (the final region): base[finBase .. finBase + finCnt)
  float ang;             // reg_angle of the final region's region_grow() in float degrees: what its region2rect() takes
  bool keep;             // the final region is a line-support region
  bool conflict;         // several wavefronts per frame: the run met a committed mark on a pixel it holds -- not the reference's
                         // course, to be run again
};

// One transaction for seed `sd` (its first region_grow() runs at the launch's tolerance; gs.u / gs.d[0] are refine()'s way of
// handing over the second call's).  The caller has put c.reg at the start of free queue space.
// One wavefront per frame (MW = false): marks in the records, every phase reuses the queue.  Several (MW = true): private marks,
// the queues of the phases laid end to end so that the log keeps every pixel ever accepted, reduce_region_radius() on a copy.
// Where flsd() looks at the density of a rectangle the wavefront asks lsd_density_screen() first and evaluates the exact
// rectangle only if the bracket straddles the threshold -- or if the rectangle itself is needed next: refine() takes its
// width, reduce_region_radius() its end points.  Inside reduce_region_radius()'s loop only the decision is.
template <bool MW>
__device__ __forceinline__ LsdTxn lsd_txn(GrowCtx& c, const GrowState& gs, const LineDeviceArgs& a, const LsdSeed& sd, int firstGrp, bool dirtyFst, unsigned fstQ) {
  const int lane = c.lane;
  double* rec = gs.d + 1;
  uint32_t* const base = c.reg;
  LsdTxn t;
  t.keep = false; t.conflict = false; t.finBase = 0; t.ang = 0.f;
  float regAngF;
  const unsigned long long pg0 = PF_NOW();
  int cnt = lsd_region_grow<MW>(c, gs, sd, firstGrp, dirtyFst, fstQ, a.minRegSize, &regAngF, &t.conflict, true);
  PF_ADD(c, 2, PF_NOW() - pg0);
  t.logLen = cnt; t.finCnt = cnt;
  if constexpr (MW) {
    if (bcast_u32(*c.asmCnt, 0) > (unsigned)MW_ASM_CAP) t.conflict = true;   // more predictions than the list holds: not this time
  }
  if (t.conflict || cnt < a.minRegSize) return t;   // (a region below the minimum is dropped, its pixels stay used)
  int phase = 0, cnt1 = 0;
  bool fromRing = true;
  float angS = regAngF;
  for (;;) {
    const unsigned long long pg1 = PF_NOW();
    const int v = a.screen ? lsd_density_screen<MW>(c, cnt, fromRing, a.screenLo, a.screenHi) : 0;
    bool dense = v > 0;
    bool exact = v == 0 || (v < 0 && phase < 2);
#if defined(PLH_GROW_PROF)
    exact = true;   // every verdict is checked against the exact density
    PF_ADD(c, 37, v > 0); PF_ADD(c, 38, v < 0);
#endif
    if (exact) {
      grow_lane_fence<MW>();   // queue stores visible to every lane
      PF_ADD(c, 14, 1); PF_ADD(c, 1, cnt);
      lsd_region2rect(c, cnt, (double)angS * kDegToRads, a.prec, rec);
      const bool d2 = !(rect_density(cnt, rec) < a.densityTh);
      PF_ADD(c, 39, v != 0 && d2 != dense);
      dense = d2;
    }
    PF_ADD(c, 5, PF_NOW() - pg1);
    if (dense) {
      t.keep = true;
      if constexpr (!MW) {
        if (!a.screen || cnt > LSD_SCREEN_MAX) { grow_lane_fence<MW>(); lsd_fill_regq(c, cnt); }
      }
      break;
    }
    const unsigned long long pg2 = PF_NOW();
    if (phase == 0) {
      // refine(): tolerance from the angle spread near the seed, everything un-marked, grown again
      PF_ADD(c, 15, 1);
      const uint32_t cPk = c.reg[0];
      const double xc = (double)pk_x(cPk), yc = (double)pk_y(cPk);
      const uint32_t cLin = pk_lin(c, cPk);
      const unsigned rec0 = c.P[cLin];
      const LsdAngleEntry* e0 = c.A + (rec0 & LSD_REC_IDX);
      const float ang0 = e0->angf, sx0 = e0->seedx, sy0 = e0->seedy;
      const double ang_c = (double)ang0 * kDegToRads, width = rec[4];
      double acc = 0;
      int n = 0;
      bool overtaken = false;
      for (int b0 = 0; b0 < cnt; b0 += 64) {
        const int i = b0 + lane;
        bool flag = false;
        double ang_d = 0;
        if (i < cnt) {
          const uint32_t p = c.reg[i];
          const uint32_t li = pk_lin(c, p);
          const unsigned rp = c.P[li];
          if constexpr (MW) {
            overtaken = overtaken || (rp & LSD_USED) != 0u;
            c.M[li] = 0; c.H[li] = 0;
          } else {
            c.P[li] = rp & ~LSD_USED;
          }
          if (sqrt(dist_sq(xc, yc, (double)pk_x(p), (double)pk_y(p))) < width) {
            flag = true;
            ang_d = angle_diff_signed((double)c.A[rp & LSD_REC_IDX].angf * kDegToRads, ang_c);
          }
        }
        n += __popcll(__ballot(flag));
        PLH_WAVE_SYNC();
        c.T[lane] = ang_d; c.T[64 + lane] = ang_d * ang_d; c.T[128 + lane] = 0.0;
        PLH_WAVE_SYNC();
        acc = lsd_chain_add(c, acc, min(64, cnt - b0));
      }
      if constexpr (MW) {
        if (__ballot(overtaken) != 0ull) { t.conflict = true; PF_ADD(c, 6, PF_NOW() - pg2); return t; }
      }
      const double sum = bcast_f64(acc, 0), s_sum = bcast_f64(acc, 1);
      const double mean_angle = sum / (double)n;
      PLH_WAVE_SYNC();
      if (lane == 0) {   // parameters of the second region_grow()
        gs.d[0] = 2.0 * sqrt((s_sum - 2.0 * mean_angle * sum) / (double)n + mean_angle * mean_angle);
        gs.u[0] = cPk; gs.u[1] = rec0 & ~LSD_USED;
        gs.u[2] = __float_as_uint(ang0); gs.u[3] = __float_as_uint(sx0); gs.u[4] = __float_as_uint(sy0);
      }
      grow_lane_fence<MW>();
      cnt1 = cnt;
      if constexpr (MW) c.reg = base + cnt1;   // the first region stays in the log
      cnt = lsd_region_grow<MW>(c, gs, sd, -1, false, 0u, 2, &regAngF, &t.conflict, false);
      t.finCnt = cnt;
      if constexpr (MW) {
        t.logLen = cnt1 + cnt; t.finBase = cnt1;
        if (bcast_u32(*c.asmCnt, 0) > (unsigned)MW_ASM_CAP) t.conflict = true;
      }
      PF_ADD(c, 6, PF_NOW() - pg2);
      if (t.conflict || cnt < 2) return t;
      angS = regAngF;
      phase = 1; fromRing = true;
      continue;
    }
    if (phase == 1) {
      // reduce_region_radius() starts: the radius from the end points of the exact rectangle (evaluated above: a verdict
      // other than "dense" takes the exact path in this phase), carried in LDS (gs.d[0] is free now)
      const uint32_t cPk = c.reg[0];
      const double xc = (double)pk_x(cPk), yc = (double)pk_y(cPk);
      if constexpr (MW) {   // it permutes and drops queue entries: it works on a copy, the log keeps the grown region
        uint32_t* cp = c.reg + cnt;
        for (int i = lane; i < cnt; i += 64) cp[i] = c.reg[i];
        grow_lane_fence<true>();
        c.reg = cp;
        t.finBase = cnt1 + cnt;
      }
      const double r1 = dist_sq(xc, yc, rec[0], rec[1]), r2 = dist_sq(xc, yc, rec[2], rec[3]);
      PLH_WAVE_SYNC();
      if (lane == 0) gs.d[0] = r1 > r2 ? r1 : r2;
      PLH_WAVE_SYNC();
      phase = 2;
    }
    {   // one step of reduce_region_radius()
      const uint32_t oPk = c.reg[0];
      const double radSq = gs.d[0] * (0.75 * 0.75);
      PLH_WAVE_SYNC();
      if (lane == 0) gs.d[0] = radSq;
      cnt = lsd_reduce_radius_step<MW>(c, cnt, (double)pk_x(oPk), (double)pk_y(oPk), radSq);
      t.finCnt = cnt;
      PF_ADD(c, 7, 1);
      PF_ADD(c, 6, PF_NOW() - pg2);
      if (cnt < 2) break;
      fromRing = false;
    }
  }
  t.finCnt = cnt; t.ang = angS;
  return t;
}

// flsd(): one wavefront per frame, seeds in pseudo-order, sequential semantics.
// VGPR budget.  Rounds 2-3 built this kernel for 64 registers = 8 wavefronts per SIMD (it needed 74; six values spilled around
// region_grow() calls; the two extra wave slots were worth 3 - 4 % on the whole front end, profiles/r02_waves_per_simd.txt).  With
// round 4's density screen inlined between the region_grow() calls the 64-register build spills 144 bytes, part of it on paths
// every region takes, and the 72-register build (7 wavefronts per SIMD: 7168 frames resident, more than the bench's 6144) is
// 3.7 % faster on the whole front end, 7 % on the kernel alone (profiles/r04_grow_waves_8_vs_7.txt).
// -DPLH_GROW_WAVES=0 builds without a cap.
#ifndef PLH_GROW_WAVES
#define PLH_GROW_WAVES 7
#endif
#if PLH_GROW_WAVES > 0
#define PLH_GROW_ATTR __attribute__((amdgpu_waves_per_eu(PLH_GROW_WAVES)))
#else
#define PLH_GROW_ATTR
#endif
__device__ __forceinline__ void lsd_grow_frame(const LineDeviceArgs& a, unsigned char* smem) {
  const int b = blockIdx.x, lane = threadIdx.x;
  GrowCtx c;
  // LDS: the 2 KiB ring only.  T aliases it: the ring is only live inside lsd_region_grow (which restarts it), T only
  // inside region2rect / refine, which read the queue from global memory.  The `used` map of the reference lives in
  // bit 31 of the level-line records themselves (k_lsd_grad rewrites them every frame): the test comes for free with
  // the record load, and with no per-frame bitmap in LDS the number of resident frames per CU is bounded by registers
  // only.  Marks are plain stores: a frame is owned by one wavefront, whose later loads observe its earlier stores.
  c.ring = (uint32_t*)smem;
  c.T = (double*)smem;
  c.P = a.pix + (long long)b * a.arenaStride;
  c.A = a.angleTab;
  c.reg = a.reg + (long long)b * a.arenaStride;
  c.regq = a.regq + (long long)b * a.arenaStride;
  c.scr = c.regq;   // reduce_region_radius()'s scratch (at most one word per queue entry) is the queue's own part of regq: whatever
                    // a reduce step leaves there is rewritten by the next density decision on the permuted queue (or at keep time)
  c.spitch = a.spitch; c.sw = a.sw; c.sh = a.sh; c.lane = lane; c.qThresh = a.qThresh;
  c.nullIdx = lsd_rec_index((unsigned)(a.sw - 1), 0u, (unsigned)a.spitch);
  c.precDef = a.prec; c.cin2Def = a.alignCin2; c.cout2Def = a.alignCout2; c.fastDef = a.alignFast;
  { const LsdTol t0 = lsd_tol(a.prec); c.loDef = bcast_f32(t0.lo, 0); c.hiDef = bcast_f32(t0.hi, 0); }
  const uint32_t* ord = a.ordered + (long long)b * a.arenaStride;   // packed coordinates x | y << 16
  float* segs = a.segs + (long long)b * a.arenaStride;
#if defined(PLH_GROW_PROF)
  unsigned long long pfv[40];
  for (int i = 0; i < 40; i++) pfv[i] = 0;
  c.pf = pfv;
  const unsigned long long pfStart = PF_NOW();
#endif
  const int nOrd = a.nOrdered[b];
  if (a.batch <= 8) {
    // latency mode (a handful of frames, one lone wavefront each): every step of the walk below waits for one dependent
    // record fetch, so pull the frame's records through this XCD's L2 once, with all loads in flight, before it starts
    const uint4* P4 = reinterpret_cast<const uint4*>(c.P);
    const int n16 = (a.spitch * lsd_rec_rows(a.sh)) >> 2;   // the pitch is a multiple of 64
    unsigned acc = 0;
    for (int i = lane; i < n16; i += 64 * 8) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int j = i + 64 * k;
        if (j < n16) {
          const uint4 r4 = P4[j];
          acc |= r4.w & 0x20000000u;   // bit 29 is never set (index < 2^20, DEF = bit 30, USED = bit 31)
          // ... and the table entries the frame's defined pixels point to (about 19 k distinct ones)
          const unsigned rr[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
          for (int q = 0; q < 4; q++)
            if (rr[q] & LSD_REC_DEF) acc |= __float_as_uint(c.A[rr[q] & LSD_REC_IDX].pad);   // pad is 0
        }
      }
    }
    if (acc) atomicOr(a.status, 8);   // keeps the loads alive; never taken
  }
  const int grp = lane >> 3, nbr = lane & 7;
  const int nbq = nbr < 4 ? nbr : nbr + 1;
  const int ndy = nbq / 3 - 1, ndx = nbq - (nbq / 3) * 3 - 1;
  int nseg = 0, logOff = 0;   // segment slots handed out; words of the frame's log (a.reg) taken by the kept regions
  // LDS tables that keep the seed scan and the prefetched neighbourhoods out of the registers (the wavefront's VGPR
  // count decides how many frames are resident): the 64 seeds of a scan, and per lane the first-step record
  uint4* tabX = (uint4*)(smem + LSD_RING * 4);         // [64] per seed: packed coordinates, record, bits of the angle and of the seed cos: one
                                                       // 16-byte LDS read per seed
  float* tabS = (float*)(tabX + 64);                   // [64] ... and the seed sin
  LsdPix* fstPx = (LsdPix*)(tabS + 64);                // [64] neighbour records of the batch's seeds (lane group t = seed t)
  uint32_t* fstIdx = (uint32_t*)(fstPx + 64);          // [64] linear index, 0xffffffff = out of bounds / no seed
  uint32_t* fstPk = fstIdx + 64;
  int* batchSk = (int*)(fstPk + 64);                   // [8] (unused since round 5: the batch's scan lanes travel in a scalar register pair)
  GrowState gs;
  gs.d = (double*)(batchSk + 8);
  gs.u = (uint32_t*)(gs.d + LSD_GS_D);
  gs.fstPx = fstPx; gs.fstIdx = fstIdx; gs.fstPk = fstPk;
  for (int sbase = 0; sbase < nOrd; sbase += 64) {
    // 64 seeds per scan: one coalesced load + one parallel `used` test; the survivors' own records (angle, seed
    // cos/sin) are fetched by their scan lanes, all at once
    unsigned long long fm;
    {
      const int si = sbase + lane;
      const uint32_t seedP = si < nOrd ? ord[si] : 0u;
      const uint32_t seedL = pk_lin(c, seedP);
      unsigned sRec = LSD_USED;
      if (si < nOrd) sRec = c.P[seedL];
      fm = __ballot(!(sRec & LSD_USED));
      float sAng = 0.f, sCx = 0.f, sSy = 0.f;   // seed angle, (float)cos / (float)sin of its double angle
      if (!(sRec & LSD_USED)) {
        const LsdAngleEntry* e = c.A + (sRec & LSD_REC_IDX);
        const uint4 e0 = *reinterpret_cast<const uint4*>(e);
        sAng = __uint_as_float(e0.x); sCx = __uint_as_float(e0.w); sSy = e->seedy;
      }
      PLH_WAVE_SYNC();
      tabX[lane] = uint4{seedP, sRec, __float_as_uint(sAng), __float_as_uint(sCx)}; tabS[lane] = sSy;
      PLH_WAVE_SYNC();
    }
    bool dirtySeed = false, dirtyFst = false;   // a region has been grown since this scan / since fst was fetched
    while (fm) {
      // up to 8 surviving seeds at a time: lane group t prefetches the 8 neighbours of the t-th of them
      int nb = 0;
      unsigned long long skPack = 0;   // scan lane of the batch's t-th seed in byte t (a scalar register pair, not an LDS table)
      {
        int mySk = 0;
        for (; nb < 8 && fm; nb++) {
          const int k = __ffsll((long long)fm) - 1;
          fm &= fm - 1;
          if (grp == nb) mySk = k;
          skPack |= (unsigned long long)k << (8 * nb);
        }
        PLH_WAVE_SYNC();
        const uint32_t sp = tabX[mySk].x;
        const int xx = pk_x(sp) + ndx, yy = pk_y(sp) + ndy;
        LsdPix px = lsd_null_px(0u);
        uint32_t nidx = 0xffffffffu;
        if (grp < nb && xx >= 0 && xx < c.sw && yy >= 0 && yy < c.sh) {
          nidx = lsd_rec_index((unsigned)xx, (unsigned)yy, (unsigned)c.spitch);
          const unsigned rec = c.P[nidx];
          px = (rec & LSD_REC_DEF) ? lsd_fetch_px(c.A, rec) : lsd_null_px(rec);   // every defined pixel: a mark may be taken back by refine()
        }
        fstPx[lane] = px; fstIdx[lane] = nidx; fstPk[lane] = (uint32_t)xx | ((uint32_t)yy << 16);
        PLH_WAVE_SYNC();
      }
      dirtyFst = false;
      for (int t = 0; t < nb; t++) {
        const int sk = (int)((skPack >> (8 * t)) & 63ull);
        const uint4 sx = tabX[sk];
        const float ss = tabS[sk];
        uint32_t seedPk = bcast_u32(sx.x, 0);
        uint32_t seed = pk_lin(c, seedPk);
        unsigned seedQ = bcast_u32(sx.y, 0);
        LsdSeed sd;   // (the scan's table values with the same LDS round trip, whether or not the seed survives the test below)
        sd.pk = seedPk; sd.ang = __uint_as_float(bcast_u32(sx.z, 0)); sd.cx = __uint_as_float(bcast_u32(sx.w, 0)); sd.sy = bcast_f32(ss, 0);
        PLH_WAVE_SYNC();
        // marks may have changed since the scan / the neighbourhood prefetch: re-read the seed's record and, in lane group t, its
        // neighbours' -- ONE round trip for both (round 4 made two: the neighbours' only after the seed had passed)
        unsigned fstQ = 0u;
        if (dirtyFst) {
          const uint32_t fi = fstIdx[lane];
          if (grp == t && fi != 0xffffffffu) fstQ = c.P[fi];
        }
        if (dirtySeed) seedQ = c.P[seed];
        if (seedQ & LSD_USED) continue;   // swallowed by a region grown since the scan
        sd.q = bcast_u32(seedQ, 0);
        // region_grow -> [density] -> refine: tighter tolerance, re-grow -> [density] -> reduce_region_radius (lsd_txn)
        PLH_WAVE_SYNC();
        const LsdTxn tx = lsd_txn<false>(c, gs, a, sd, t, dirtyFst, fstQ);
        dirtySeed = true; dirtyFst = true;
        if (!tx.keep) continue;
        // a line-support region: its pixels stay where they are -- the queue moves on behind them -- and its segment slot
        // says where (LsdRegionEntry); k_lsd_rects puts the segment there.  The kept regions are disjoint sets of marked
        // pixels and a queue only ever holds marked pixels outside them, so log + queue fit the frame's pixel count.
        if (lane == 0 && nseg < a.segCap)
          reinterpret_cast<uint4*>(segs)[nseg] = uint4{(unsigned)logOff, (unsigned)tx.finCnt, __float_as_uint(tx.ang), 0u};
        logOff += tx.finCnt;
        c.reg += tx.finCnt;
        c.regq += tx.finCnt;
        c.scr = c.regq;
        nseg++;
      }
    }
  }
  if (lane == 0) {
    if (nseg > a.segCap) { atomicOr(a.status, 4); nseg = a.segCap; }
    a.nSegs[b] = nseg;
  }
#if defined(PLH_GROW_PROF)
  pfv[0] = PF_NOW() - pfStart;
  if (lane == 0)
    for (int i = 0; i < 40; i++) atomicAdd(&g_grow_prof[i], pfv[i]);
#endif
}

// ---------------------------------------------------------------------------------------------
// Multi-wavefront region growing (k_lsd_grow_mw): W wavefronts of ONE workgroup share a frame.
//
// flsd() is a sequence of transactions, one per seed that is still free at its turn: region_grow -> region2rect ->
// [refine -> region_grow -> region2rect -> reduce_region_radius].  A transaction reads the `used` map and marks pixels in
// it; nothing else couples two of them.  Across transactions the map only ever changes from free to used (refine() and
// reduce_region_radius() take back marks of their OWN region only), so a transaction that ran against a stale map -- one
// that lacks the marks of older transactions still in flight -- took exactly the reference's course unless a pixel it
// ACCEPTED at any point is used in the true map: a pixel it rejected for its angle is rejected either way, a pixel it saw
// used is used.  Hence optimistic execution with in-order commit:
//   * seeds are handed out in the reference's order (a FIFO in LDS, refilled 64 seeds at a time by whichever wavefront
//     finds it low); the FIFO index is the transaction's sequence number;
//   * a wavefront runs its transaction with its marks in a private byte plane (grow_mark<true>), reading the committed
//     marks from the records, and keeps a log of every pixel it ever accepted (the region queues of its phases, laid
//     end to end).  When the transaction is through it takes its private marks back, POSTS the result (log position, final
//     region, segment) in a ring in LDS and goes on to the next seed -- it does not wait for its turn;
//   * posted transactions are COMMITTED strictly in sequence order by whichever wavefront holds the drain lock: it re-reads
//     the records of the log; if none is used it sets the USED bits of the final region and appends the segment; otherwise
//     it runs the transaction again itself -- it is the oldest one now, so this run is the reference's -- and commits that.
//     A transaction that notices a committed mark on a pixel it holds (while growing, or when it is through) starts over
//     at once.  A seed that is already used when it is handed out retires without effect.
// The result is the one-wavefront kernel's, segment for segment; only the schedule differs.  All wavefronts of a frame sit
// on one CU: they share its L1 (coherent for their own stores, workgroup scope) and talk through LDS.
// Used for small batches, where one wavefront per frame leaves the GPU empty and a frame takes 47 ms (Frame.cc:224-227
// calls the extractor once per frame).
// ---------------------------------------------------------------------------------------------
constexpr int MW_N = 512;    // ring of posted transactions (power of two): bounds how far the transactions may run ahead of the commits
constexpr int MW_F = 128;    // seed FIFO (power of two)
constexpr int MW_LOW = 24;   // a wavefront that finds fewer seeds queued refills the FIFO
constexpr int MW_MAX_WAVES = 16;
enum { MWC_CURSOR = 0, MWC_LOCK, MWC_PUSH, MWC_POP, MWC_HEAD, MWC_DONE, MWC_NSEG, MWC_ABORT, MWC_DLOCK, MWC_LOGOFF, MWC_RET = 16,
       MWC_WORDS = MWC_RET + MW_MAX_WAVES };
constexpr int MW_PEND_WORDS = 16;   // a posted transaction (MwPost)
// A wait that lasts this many polls (some seconds) cannot be a wait for work: the kernel gives up instead of hanging the GPU
// -- every wavefront leaves at its next wait, status bit 4 (16) reports it and the frame's segments are void.
constexpr unsigned MW_SPIN_LIMIT = 1u << 26;
constexpr int MW_WAVE_LDS = LSD_RING * 4 + (LSD_GS_D + 1) * 8 + 8 * 4 + MW_ASM_CAP * 4 + 16;   // per wavefront: ring (aliased by T) + GrowState + assumed-used list + its count

// -DPLH_MW_PARANOID (a checking build for tools, never the product): the commit re-runs EVERY posted transaction against the committed
// map -- the run the sequential algorithm makes at that point -- publishes that run, and compares it with what was posted whenever the
// post passed the commit's validation: [0] validated inline posts compared ([10]: general posts -- kept or refined regions), [1] of them different (the protocol's claim is that this is 0),
// [2..9] the first offender: frame, sequence number, seed, flags, posted final count, exact final count, posted keep, exact keep.
#if defined(PLH_MW_PARANOID)
__device__ unsigned g_mw_paranoid[16];
#endif

// the control words in LDS, the pause of a polling loop and the two fences: plh_shims.h (lds_*, spin_pause, wg_release, wg_acquire)
__device__ __forceinline__ int mw_ld(const int* p) { return lds_load(p); }
__device__ __forceinline__ void mw_st(int* p, int v) { lds_store(p, v); }
__device__ __forceinline__ int mw_cas(int* p, int cmp, int v) { return lds_cas(p, cmp, v); }
__device__ __forceinline__ void mw_pause() { spin_pause(); }
// publish: this wavefront's global stores are complete (L1 / L2 of its CU) before the LDS word that hands them over
__device__ __forceinline__ void mw_release() { wg_release(); }
__device__ __forceinline__ void mw_acquire() { wg_acquire(); }

// -DPLH_MW_JITTER (a stress build for tests / tools, never the product): pseudo-random pauses at the hand-over points of the protocol, so
// that interleavings a timing-stable build never produces are exercised (round 6: the counter build of k_lsd_grow_mw16 -- much slower per
// transaction -- disagreed with the product on a handful of the soak's frames, different ones from run to run).
#if defined(PLH_MW_JITTER) && !defined(HIPEMU)
__device__ __forceinline__ void mw_jitter(unsigned salt) {
  const unsigned t = (unsigned)__builtin_amdgcn_s_memtime() * 2654435761u + salt * 40503u;
  const unsigned k = (t >> 13) & 15u;
  if (k < 6u) {
    for (unsigned i = 0; i <= ((t >> 20) & 31u); i++) __builtin_amdgcn_s_sleep(64);
  }
}
#else
__device__ __forceinline__ void mw_jitter(unsigned) {}
#endif

__device__ __forceinline__ bool mw_give_up(int* ctl, unsigned& polls, int* status) {
  if (++polls > MW_SPIN_LIMIT) {
    mw_st(&ctl[MWC_ABORT], 1);
    atomicOr(status, 16);
    return true;
  }
  return mw_ld(&ctl[MWC_ABORT]) != 0;
}

// a control word as ONE value for the whole wavefront (lane 0's read): the words change under the reader's feet, and every
// decision taken on them has to be the same in all lanes
__device__ __forceinline__ int mw_ld_u(const int* p) { return (int)bcast_u32((unsigned)mw_ld(p), 0); }
__device__ __forceinline__ int mw_try_lock(int* lock, int lane) {
  int got = 0;
  if (lane == 0) got = mw_cas(lock, 0, 1) == 0;
  return (int)bcast_u32((unsigned)got, 0);
}

// Refill the seed FIFO (caller holds MWC_LOCK): 64 seeds per pass -- one coalesced load of their coordinates, their records,
// and for those not used yet the seed terms from the gradient table -- pushed in order.  A seed that is used by the time it
// is popped costs its wavefront one record load.
__device__ void mw_scan(const GrowCtx& c, int* ctl, uint4* fifo, const uint32_t* ord, int nOrd) {
  const int lane = c.lane;
  int cursor = mw_ld_u(&ctl[MWC_CURSOR]), push = mw_ld_u(&ctl[MWC_PUSH]);   // only the lock holder moves these
  for (;;) {
    if (cursor >= nOrd) {
      if (lane == 0) mw_st(&ctl[MWC_DONE], 1);
      break;
    }
    const int pop = mw_ld_u(&ctl[MWC_POP]);
    if (push - pop >= MW_LOW || push + 64 - pop > MW_F) break;
    const int si = cursor + lane;
    bool fr = false;
    uint32_t seedP = 0;
    uint4 ent = uint4{0u, 0u, 0u, 0u};
    if (si < nOrd) {
      seedP = ord[si];
      const unsigned sRec = c.P[pk_lin(c, seedP)];
      fr = !(sRec & LSD_USED);
      if (fr) {
        const LsdAngleEntry* e = c.A + (sRec & LSD_REC_IDX);
        const uint4 e0 = *reinterpret_cast<const uint4*>(e);
        ent.x = seedP; ent.y = e0.x; ent.z = e0.w; ent.w = __float_as_uint(e->seedy);   // angle (degrees), seed cos, seed sin
      }
    }
    const unsigned long long m = __ballot(fr);
    if (fr) fifo[(push + __popcll(m & lanemask_lt())) & (MW_F - 1)] = ent;
    push += __popcll(m);
    cursor += 64;
    PLH_WAVE_SYNC();
    if (lane == 0) mw_st(&ctl[MWC_PUSH], push);   // after the entries (LDS operations of a wavefront execute in order)
  }
  PLH_WAVE_SYNC();
  if (lane == 0) mw_st(&ctl[MWC_CURSOR], cursor);
}

// One transaction of the multi-wavefront kernel: lsd_txn<true> for the seed, its log at regBase.
__device__ LsdTxn lsd_txn_mw(GrowCtx& c, const GrowState& gs, const LineDeviceArgs& a, uint32_t* regBase, uint32_t seedPk,
                             unsigned seedRec, unsigned sAngBits, unsigned sCxBits, unsigned sSyBits) {
  c.reg = regBase;
  PLH_WAVE_SYNC();
  if (c.lane == 0) *c.asmCnt = 0;
  PLH_WAVE_SYNC();
  LsdSeed sd;   // (lane 0's values, as when they went through LDS)
  sd.pk = bcast_u32(seedPk, 0); sd.q = bcast_u32(seedRec, 0);
  sd.ang = __uint_as_float(bcast_u32(sAngBits, 0)); sd.cx = __uint_as_float(bcast_u32(sCxBits, 0)); sd.sy = __uint_as_float(bcast_u32(sSyBits, 0));
  return lsd_txn<true>(c, gs, a, sd, -1, false, 0u);
}

// A posted transaction: 16 words in LDS, ring slot = sequence number mod MW_N.
//   w[0]  flags: bit 0 has a log (0: the seed was used already, or is predicted to be), bit 1 ran against a possibly stale map,
//         bit 2 the final region is kept (a segment), bit 3 INLINE; bits 8-15 wavefront; bits 16-23 pixels taken for used on an older claim;
//         bits 24-31 (inline form) pixels accepted
//   w[1..4]  seed (packed coordinates, angle, seed cos / sin bits): to run it again
//   inline form (a region that was dropped for its size, never refined -- four transactions in five): w[5..12] = the accepted
//         pixels, then the assumed ones (packed coordinates; at most 8 together): nothing of it lies in global memory
//   general form: w[5] log offset in the wavefront's arena, w[6] log length, w[7] / w[8] first / count of the final region,
//         w[9] region angle of a kept region (bits of the float; the commit copies the region to the frame's log and
//         writes its LsdRegionEntry), w[13..15] up to 3 assumed pixels (more: in the arena behind the log)
struct MwPost {
  unsigned w[MW_PEND_WORDS];
};
enum { MWP_FLAGS = 0, MWP_SEED = 1, MWP_PIX = 5, MWP_OFF = 5, MWP_LOGLEN = 6, MWP_FINBASE = 7, MWP_FINCNT = 8, MWP_SEG = 9, MWP_ASM = 13 };
constexpr int MW_INLINE_MAX = 8;

struct MwShared {
  int* ctl;        // MWC_*
  int* state;      // [MW_N] sequence tag of the posted transaction in the slot
  uint4* fifo;     // [MW_F] seeds
  MwPost* pend;    // [MW_N]
};

// Commit the posted transactions at the head of the sequence, in order, for as long as there are any (caller holds
// MWC_DLOCK).  `ch` is the drain context: a log arena and a mark plane of its own (slot W of the frame) for the transactions
// that have to be run again.
// The commits are one dependency chain per frame (every one reads the marks the previous one set), so the chain is made
// short: up to eight consecutive INLINE posts are taken together, lane group g = post h + g, lane j of it = its pixel j:
// one LDS round trip for the posts, one load of all their records; a pixel an older post of the batch publishes counts as
// used for the younger ones (compared lane against lane); everything in front of the first post that fails is committed
// with one store per pixel.
// A kept region goes to the frame's log (frameLog: the one-wavefront kernel's queue area, free here) and its LsdRegionEntry into
// the next segment slot, exactly as the one-wavefront kernel leaves them; k_lsd_rects evaluates the rectangles.
__device__ void mw_drain(GrowCtx& ch, const GrowState& gs, const LineDeviceArgs& a, const MwShared& sh, uint32_t* frameReg,
                         uint32_t* drainReg, float* segs, uint32_t* frameLog, uint32_t* frameLogQ) {
  const int lane = ch.lane, g = lane >> 3, j = lane & 7;
  int h = mw_ld_u(&sh.ctl[MWC_HEAD]);
  int nseg = mw_ld_u(&sh.ctl[MWC_NSEG]);
  int logOff = mw_ld_u(&sh.ctl[MWC_LOGOFF]);
  const int h0 = h;
  for (;;) {
    // ---- the inline posts among the next eight sequence numbers
    const int slot = (h + g) & (MW_N - 1);
    const bool ready = mw_ld(&sh.state[slot]) == h + g + 1;
    const unsigned long long readyM = wballot(ready && j == 0);   // (one lane per group decides: the words change while they are read)
    if (!(readyM & 1ull)) break;   // the head is not posted yet
    const bool rdy = ((readyM >> (8 * g)) & 1ull) != 0ull;
    const unsigned flags = rdy ? sh.pend[slot].w[MWP_FLAGS] : 0u;
    const unsigned long long inlM = wballot(rdy && (flags & 8u) != 0u && j == 0);
    const unsigned long long gap = ~inlM & 0x0101010101010101ull;
#if defined(PLH_MW_PARANOID) && PLH_MW_PARANOID + 0 == 1
    const int nb = 0;   // every post goes through the general path below, one by one
    (void)gap;
#else
    const int nb = gap ? (__ffsll((long long)gap) - 1) >> 3 : 8;   // leading inline posts
#endif
    if (nb > 0) {
      const int nAcc = (int)(flags >> 24), nAsm = (int)((flags >> 16) & 255u);
      const bool has = g < nb && j < nAcc + nAsm, isAcc = j < nAcc, spec = (flags & 2u) != 0u;
      const uint32_t pk = has ? sh.pend[slot].w[MWP_PIX + j] : 0u;
      const uint32_t idx = has ? pk_lin(ch, pk) : ch.nullIdx;   // (sw - 1, 0): never defined, never marked
      const unsigned p = ch.P[idx];
      const bool usedNow = (p & LSD_USED) != 0u;
      // pixels published by an older post of the batch
      bool earlier = false;
      unsigned long long pm = wballot(has && isAcc);
      while (pm) {
        const int k = __ffsll((long long)pm) - 1;
        pm &= pm - 1;
        const uint32_t ik = bcast_u32(idx, k);
        earlier = earlier || (idx == ik && g > (k >> 3));
      }
      const bool bad = has && spec && (isAcc ? (usedNow || earlier) : !(usedNow || earlier));
      const unsigned long long badM = wballot(bad);
      const int gb = badM ? min((__ffsll((long long)badM) - 1) >> 3, nb) : nb;   // posts in front of the first failure
      if (has && isAcc && g < gb) ch.P[idx] = p | LSD_USED;
      h += gb;
      PF_ADD(ch, 34, gb);
      if (gb == nb) continue;
      // post h failed: run it again below (general path handles both forms)
    }
    // ---- one post, general path
    mw_jitter((unsigned)h + 101u);
    const MwPost* e = &sh.pend[h & (MW_N - 1)];
    const uint4* e4 = reinterpret_cast<const uint4*>(e);
    const uint4 q0 = e4[0], q1 = e4[1], q2 = e4[2], q3 = e4[3];
    const unsigned fl = bcast_u32(q0.x, 0);
    const int asmLen = (int)((fl >> 16) & 255u);
    const bool inl = (fl & 8u) != 0u;
    const int wv = (int)((fl >> 8) & 255u);
    bool bad = inl;   // an inline post gets here only when the batch found it invalid
#if defined(PLH_MW_PARANOID)
    // the exact run of post h against the committed map, compared with what was posted (before anything of it is published)
    auto paranoid = [&](bool validated) {
      int pFin = 0;
      bool pKeep = false;
      if (inl) {
        // validity of an inline post as the batch path would judge it (its pixels travel in the post)
        const int nAcc = (int)(fl >> 24), nAsm = (int)((fl >> 16) & 255u);
        const unsigned w5[8] = {q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x};
        bool ib = false;
        for (int k = 0; k < nAcc + nAsm && k < 8; k++) {
          const bool u = (ch.P[pk_lin(ch, w5[k])] & LSD_USED) != 0u;
          if ((fl & 2u) && (k < nAcc ? u : !u)) ib = true;
        }
        validated = !ib;
        pFin = nAcc;
      } else if ((fl & 1u) || asmLen) {
        pFin = (fl & 1u) ? (int)bcast_u32(q2.x, 0) : 0;
        pKeep = (fl & 4u) != 0u;
      } else {
        validated = false;   // a retired seed: nothing was run
      }
      const uint32_t seedPk2 = bcast_u32(q0.y, 0);
      const unsigned seedRec2 = bcast_u32(ch.P[pk_lin(ch, seedPk2)], 0);
      if (validated && !(seedRec2 & LSD_USED) && (inl ? (fl >> 24) != 0u : (fl & 1u) != 0u)) {
        ch.hTag = (unsigned)h % 65535u + 1u;
        ch.hWin = 0;
        const LsdTxn t2 = lsd_txn_mw(ch, gs, a, drainReg, seedPk2, seedRec2, bcast_u32(q0.z, 0), bcast_u32(q0.w, 0), bcast_u32(q1.x, 0));
        grow_lane_fence<true>();
        bool diff = t2.finCnt != pFin || t2.keep != pKeep;
        if (!diff) {
          if (inl) {
            const unsigned w5[8] = {q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x};
            for (int k = 0; k < pFin && k < 8; k++) diff = diff || drainReg[t2.finBase + k] != w5[k];
          } else {
            const uint32_t* plog = frameReg + (long long)wv * a.mwRegStride + bcast_u32(q1.y, 0) + (int)bcast_u32(q1.w, 0);
            bool d = false;
            for (int i = lane; i < pFin; i += 64) d = d || plog[i] != drainReg[t2.finBase + i];
            diff = __ballot(d) != 0ull || (pKeep && q2.y != __float_as_uint(t2.ang));
          }
        }
        for (int i = lane; i < t2.finCnt; i += 64) ch.M[pk_lin(ch, drainReg[t2.finBase + i])] = 0;   // (a re-run below marks them again)
        grow_lane_fence<true>();
        if (lane == 0) {
          atomicAdd(&g_mw_paranoid[inl ? 0 : 10], 1u);
          if (diff && atomicAdd(&g_mw_paranoid[1], 1u) == 0u) {
            g_mw_paranoid[2] = blockIdx.x; g_mw_paranoid[3] = (unsigned)h; g_mw_paranoid[4] = seedPk2; g_mw_paranoid[5] = fl;
            g_mw_paranoid[6] = (unsigned)pFin; g_mw_paranoid[7] = (unsigned)t2.finCnt; g_mw_paranoid[8] = pKeep; g_mw_paranoid[9] = t2.keep;
          }
        }
      }
    };
#endif
    if (!inl && ((fl & 1u) || asmLen)) {
      const uint32_t* log = frameReg + (long long)wv * a.mwRegStride + bcast_u32(q1.y, 0);
      const int logLen = (int)bcast_u32(q1.z, 0), finBase = (int)bcast_u32(q1.w, 0), finCnt = (int)bcast_u32(q2.x, 0);
      // a log of up to 64 pixels whose final region lies inside it (nearly all): every lane keeps its pixel's record from the
      // validation, so that publishing is a store and not another two dependent fetches
      const bool oneGo = logLen <= 64 && finBase + finCnt <= logLen;
      uint32_t myIdx = 0, myPk = 0;
      unsigned myRec = 0;
      if (oneGo && lane < logLen && (fl & 3u)) {
        myPk = log[lane];
        myIdx = pk_lin(ch, myPk);
        myRec = ch.P[myIdx];
      }
      if (fl & 2u) {
        if (oneGo) {
          bad = lane < logLen && (myRec & LSD_USED) != 0u;
        } else {
          for (int base = 0; base < logLen; base += 64) {
            const int i = base + lane;
            if (i < logLen) bad = bad || (ch.P[pk_lin(ch, log[i])] & LSD_USED) != 0u;
          }
        }
#if defined(PLH_GROW_PROF)
        if (__ballot(bad) != 0ull) PF_ADD(ch, 29, 1);
#endif
        // the predictions: every pixel taken for used on an older transaction's claim is used now
        bool badA = false;
        if (asmLen <= 3) {
          const unsigned ap = lane == 0 ? q3.y : (lane == 1 ? q3.z : q3.w);
          if (lane < asmLen) badA = !(ch.P[pk_lin(ch, ap)] & LSD_USED);
        } else {
          const uint32_t* al = log + max(logLen, finBase + finCnt);
          if (lane < asmLen) badA = !(ch.P[pk_lin(ch, al[lane])] & LSD_USED);
        }
#if defined(PLH_GROW_PROF)
        if (__ballot(badA) != 0ull) PF_ADD(ch, 30, 1);
#endif
        bad = __ballot(bad || badA) != 0ull;
      }
#if defined(PLH_MW_PARANOID)
      paranoid(!bad);
#if PLH_MW_PARANOID + 0 == 1
      bad = true;
#endif
#endif
      if (!bad) {
        if (fl & 1u) {
          if (oneGo) {
            if (lane >= finBase && lane < finBase + finCnt) ch.P[myIdx] = myRec | LSD_USED;
          } else {
            for (int i = lane; i < finCnt; i += 64) ch.P[pk_lin(ch, log[finBase + i])] |= LSD_USED;
          }
        }
        if (fl & 4u) {
          if (oneGo) {
            if (lane >= finBase && lane < finBase + finCnt) {
              frameLog[logOff + lane - finBase] = myPk;
              frameLogQ[logOff + lane - finBase] = lsd_rec_q(myRec);
            }
          } else {
            for (int i = lane; i < finCnt; i += 64) {
              const uint32_t pk = log[finBase + i];
              frameLog[logOff + i] = pk;
              frameLogQ[logOff + i] = lsd_rec_q(ch.P[pk_lin(ch, pk)]);
            }
          }
          if (lane == 0 && nseg < a.segCap)   // (every lane read the same post: q2.y is the region angle in all of them)
            reinterpret_cast<uint4*>(segs)[nseg] = uint4{(unsigned)logOff, (unsigned)finCnt, q2.y, 0u};
          nseg++;
          logOff += finCnt;
        }
      }
    }
#if defined(PLH_MW_PARANOID)
    if (inl || !((fl & 1u) || asmLen)) paranoid(!bad);
#if PLH_MW_PARANOID + 0 == 1
    bad = true;   // publish the exact run in every case (-DPLH_MW_PARANOID=2: compare only, the post is published the usual way)
#endif
#endif
    if (bad) {
      // an older transaction took a pixel this one accepted (or left one this one counted on): run it again here --
      // everything older is committed, so this is the reference's run
      PF_ADD(ch, 24, 1);
      MW_TRACE(threadIdx.x >> 6, lane, 8, h);   // re-run of post h begins
      const unsigned long long pr0 = PF_NOW();
      const uint32_t seedPk = bcast_u32(q0.y, 0);
      const unsigned seedRec = bcast_u32(ch.P[pk_lin(ch, seedPk)], 0);
      ch.hTag = (unsigned)h % 65535u + 1u;
      ch.hWin = 0;   // nothing older is in flight: no claim is believed
      if (!(seedRec & LSD_USED)) {
        const LsdTxn t = lsd_txn_mw(ch, gs, a, drainReg, seedPk, seedRec, bcast_u32(q0.z, 0), bcast_u32(q0.w, 0), bcast_u32(q1.x, 0));
        grow_lane_fence<true>();
        for (int i = lane; i < t.finCnt; i += 64) {
          const uint32_t li = pk_lin(ch, drainReg[t.finBase + i]);
          ch.P[li] |= LSD_USED;
          ch.M[li] = 0;
        }
        if (t.keep) {
          for (int i = lane; i < t.finCnt; i += 64) {
            const uint32_t pk = drainReg[t.finBase + i];
            frameLog[logOff + i] = pk;
            frameLogQ[logOff + i] = lsd_rec_q(ch.P[pk_lin(ch, pk)]);
          }
          if (lane == 0 && nseg < a.segCap)
            reinterpret_cast<uint4*>(segs)[nseg] = uint4{(unsigned)logOff, (unsigned)t.finCnt, __float_as_uint(t.ang), 0u};
          nseg++;
          logOff += t.finCnt;
        }
      }
      PF_ADD(ch, 10, PF_NOW() - pr0);
      MW_TRACE(threadIdx.x >> 6, lane, 9, h);   // ... ends
    }
    PF_ADD(ch, 35, 1);
    PLH_WAVE_SYNC();
    if (!inl && ((fl & 1u) || asmLen > 3))
      if (lane == 0) sh.ctl[MWC_RET + wv] += 1;   // the owner may reuse the log's space (only the lock holder writes these)
    h++;
  }
  if (h != h0) {
    mw_release();   // the USED bits are in place before `head` says so: a transaction that starts as the oldest trusts them
    if (lane == 0) {
      mw_st(&sh.ctl[MWC_NSEG], nseg);
      mw_st(&sh.ctl[MWC_LOGOFF], logOff);
      mw_st(&sh.ctl[MWC_HEAD], h);
    }
  }
  PLH_WAVE_SYNC();
}

// take the drain lock if there is something to commit and nobody is at it; returns whether anything was done
__device__ bool mw_try_drain(GrowCtx& ch, const GrowState& gs, const LineDeviceArgs& a, const MwShared& sh, uint32_t* frameReg,
                             uint32_t* drainReg, float* segs, uint32_t* frameLog, uint32_t* frameLogQ) {
  bool did = false;
  const int wvTrace = (int)(threadIdx.x >> 6);
  (void)wvTrace;
  for (;;) {
    const int h = mw_ld_u(&sh.ctl[MWC_HEAD]);
    if (mw_ld_u(&sh.state[h & (MW_N - 1)]) != h + 1) break;
    if (!mw_try_lock(&sh.ctl[MWC_DLOCK], ch.lane)) break;
    mw_acquire();
    const unsigned long long pd0 = PF_NOW();
    MW_TRACE(wvTrace, ch.lane, 5, h);   // drain session begins at head h
    mw_drain(ch, gs, a, sh, frameReg, drainReg, segs, frameLog, frameLogQ);
    MW_TRACE(wvTrace, ch.lane, 6, mw_ld(&sh.ctl[MWC_HEAD]));   // ... ends
    PF_ADD(ch, 23, PF_NOW() - pd0); PF_ADD(ch, 25, 1);
    if (ch.lane == 0) mw_st(&sh.ctl[MWC_DLOCK], 0);
    PLH_WAVE_SYNC();
    did = true;   // and look again: a transaction posted while the lock was held found it taken
  }
  return did;
}

__device__ __forceinline__ void lsd_grow_frame_mw(const LineDeviceArgs& a, unsigned char* smem) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, W = (int)(blockDim.x >> 6);
  MwShared sh;
  sh.ctl = (int*)smem;
  sh.state = sh.ctl + MWC_WORDS;
  sh.fifo = (uint4*)(sh.state + MW_N);
  sh.pend = (MwPost*)(sh.fifo + MW_F);
  int* const ctl = sh.ctl;
  unsigned char* wsm = (unsigned char*)(sh.pend + MW_N) + wv * MW_WAVE_LDS;
  const long long S = a.mwRegStride / 5;   // words: an arena holds posted logs below S, a running transaction (3 S) and scratch (S)
  uint32_t* const frameReg = a.mwReg + (long long)b * (W + 1) * a.mwRegStride;
  uint8_t* const frameMark = a.mwMark + (long long)b * (W + 1) * a.mwMarkStride;
  GrowCtx c;
  c.ring = (uint32_t*)wsm;
  c.T = (double*)wsm;
  c.P = a.pix + (long long)b * a.arenaStride;
  c.A = a.angleTab;
  uint32_t* const regBase = frameReg + (long long)wv * a.mwRegStride;
  c.reg = regBase;
  c.regq = nullptr;
  c.scr = regBase + 4 * S;
  c.M = frameMark + (long long)wv * a.mwMarkStride;
  c.H = a.mwHint + (long long)b * a.mwMarkStride;   // (mwMarkStride 16-bit tags)
  c.hTag = 1; c.hWin = 0;
  c.asmList = (uint32_t*)(wsm + LSD_RING * 4 + (LSD_GS_D + 1) * 8 + 8 * 4);
  c.asmCnt = c.asmList + MW_ASM_CAP;
  c.spitch = a.spitch; c.sw = a.sw; c.sh = a.sh; c.lane = lane; c.qThresh = a.qThresh;
  c.nullIdx = lsd_rec_index((unsigned)(a.sw - 1), 0u, (unsigned)a.spitch);
  c.precDef = a.prec; c.cin2Def = a.alignCin2; c.cout2Def = a.alignCout2; c.fastDef = a.alignFast;
  { const LsdTol t0 = lsd_tol(a.prec); c.loDef = bcast_f32(t0.lo, 0); c.hiDef = bcast_f32(t0.hi, 0); }
  GrowState gs;
  gs.d = (double*)(wsm + LSD_RING * 4);
  gs.u = (uint32_t*)(gs.d + LSD_GS_D + 1);
  gs.fstPx = nullptr; gs.fstIdx = nullptr; gs.fstPk = nullptr;
  const uint32_t* ord = a.ordered + (long long)b * a.arenaStride;
  float* segs = a.segs + (long long)b * a.arenaStride;
  uint32_t* const frameLog = a.reg + (long long)b * a.arenaStride;   // the kept regions, in commit order (k_lsd_rects reads them)
  uint32_t* const frameLogQ = a.regq + (long long)b * a.arenaStride;  // ... and gx^2 + gy^2 of their pixels
  const int nOrd = a.nOrdered[b];
#if defined(PLH_GROW_PROF)
  unsigned long long pfv[40];
  for (int i = 0; i < 40; i++) pfv[i] = 0;
  c.pf = pfv;
  const unsigned long long pfStart = PF_NOW();
#endif
  GrowCtx ch = c;   // the drain context: slot W of the frame
  uint32_t* const drainReg = frameReg + (long long)W * a.mwRegStride;
  ch.reg = drainReg;
  ch.scr = drainReg + 4 * S;
  ch.M = frameMark + (long long)W * a.mwMarkStride;
  for (int i = tid; i < MWC_WORDS + MW_N; i += (int)blockDim.x) ctl[i] = 0;   // control words and sequence tags
  {   // no claims yet (the plane holds the previous launch's)
    uint4* H4 = reinterpret_cast<uint4*>(c.H);
    const int n16 = (a.spitch * lsd_rec_rows(a.sh)) >> 3;   // 16-byte stores of eight tags; the pitch is a multiple of 64
    for (int i = tid; i < n16; i += (int)blockDim.x) H4[i] = uint4{0u, 0u, 0u, 0u};
    mw_release();
  }
  if (a.batch <= 8) {
    // a handful of frames: pull the frame's records and the table entries they point to through this XCD's L2, all loads
    // in flight, before the dependent fetches of region growing start (as k_lsd_grow_lone does)
    const uint4* P4 = reinterpret_cast<const uint4*>(c.P);
    const int n16 = (a.spitch * lsd_rec_rows(a.sh)) >> 2;
    unsigned acc = 0;
    for (int i = tid; i < n16; i += (int)blockDim.x * 4) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int j = i + (int)blockDim.x * k;
        if (j < n16) {
          const uint4 r4 = P4[j];
          acc |= r4.w & 0x20000000u;
          const unsigned rr[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
          for (int q = 0; q < 4; q++)
            if (rr[q] & LSD_REC_DEF) acc |= __float_as_uint(c.A[rr[q] & LSD_REC_IDX].pad);
        }
      }
    }
    if (acc) atomicOr(a.status, 8);   // keeps the loads alive; never taken
  }
  __syncthreads();
  int off = 0, posted = 0;   // this wavefront's arena: posted logs lie in [0, off); `posted` of them so far
  unsigned polls = 0;        // consecutive fruitless waits (watchdog)
  for (;;) {
    // ---- may this wavefront start another transaction?  Not with its arena full of posted logs, and not further than
    // mwLag sequence numbers ahead of the commits (a stale map makes wasted runs).  While it cannot, it helps committing.
    if (off > 0 && mw_ld_u(&ctl[MWC_RET + wv]) == posted) off = 0;
    const int done = mw_ld_u(&ctl[MWC_DONE]), pop = mw_ld_u(&ctl[MWC_POP]), push = mw_ld_u(&ctl[MWC_PUSH]);
    const int head = mw_ld_u(&ctl[MWC_HEAD]);
    int s = -1;
    uint4 ent = uint4{0u, 0u, 0u, 0u};
    if (!done && push - pop < MW_LOW && mw_try_lock(&ctl[MWC_LOCK], lane)) {
      const unsigned long long ps0 = PF_NOW();
      mw_scan(c, ctl, sh.fifo, ord, nOrd);
      PLH_WAVE_SYNC();
      if (lane == 0) mw_st(&ctl[MWC_LOCK], 0);
      PF_ADD(c, 20, PF_NOW() - ps0);
      polls = 0;
      continue;
    }
    if (pop < push && off <= S && pop - head < a.mwLag) {
      ent = sh.fifo[pop & (MW_F - 1)];   // read before the pop: the slot may be refilled right after it
      ent.x = bcast_u32(ent.x, 0); ent.y = bcast_u32(ent.y, 0); ent.z = bcast_u32(ent.z, 0); ent.w = bcast_u32(ent.w, 0);
      int ok = 0;
      if (lane == 0) ok = mw_cas(&ctl[MWC_POP], pop, pop + 1) == pop;
      ok = (int)bcast_u32((unsigned)ok, 0);
      if (!ok) continue;
      s = pop;
    }
    if (s < 0) {
      if (done && pop >= push && head >= push) break;   // every seed handed out and committed
      const unsigned long long pw0 = PF_NOW();
      if (mw_try_drain(ch, gs, a, sh, frameReg, drainReg, segs, frameLog, frameLogQ)) { polls = 0; continue; }
      int stop = 0;
      MW_TRACE(wv, lane, 7, (pop < push ? 1 : 0) | (off > S ? 2 : 0) | (pop - head >= a.mwLag ? 4 : 0) | (done ? 8 : 0));   // nothing to do: why
      if (lane == 0) { stop = mw_give_up(ctl, polls, a.status); mw_pause(); }
      if (bcast_u32((unsigned)stop, 0)) break;
      PF_ADD(c, 19, PF_NOW() - pw0);
      continue;
    }
    polls = 0;
    mw_jitter((unsigned)s);
    MW_TRACE(wv, lane, 1, s);   // popped
    // ---- the transaction
    const uint32_t seedPk = ent.x;
    const uint32_t seedLin = pk_lin(c, seedPk);
    const bool spec = head != s;   // older transactions are still in flight: what this one reads may be stale
    PF_ADD(c, 16, 1);
    // the post, one word per lane (MwPost): what the transaction leaves behind for its commit
    unsigned pFlags = 8u;   // the seed was used already: nothing to check, nothing to do
    int pMode = 0;          // 1: predicted unused, 2: inline, 3: general
    unsigned pGen[4] = {0u, 0u, 0u, 0u};   // general form: off, logLen, finBase, finCnt
    unsigned pAng = 0u;     // general form, kept region: its region angle (float bits)
    int pAcc = 0, pAsm = 0;
    c.hTag = (unsigned)s % 65535u + 1u;
    for (;;) {
      mw_acquire();
      const unsigned seedRecL = c.P[seedLin], seedHintL = (unsigned)c.H[seedLin];   // one round trip for both
      // ... and for the committed records of the seed's eight neighbours (lanes 0 - 7), see below
      unsigned nbRec = 0u;
      {
        const int k = (lane & 7) + ((lane & 7) >= 4 ? 1 : 0);   // 0 .. 8 without the centre
        const int nx = pk_x(seedPk) + k % 3 - 1, ny = pk_y(seedPk) + k / 3 - 1;
        if (lane < 8 && nx >= 0 && ny >= 0 && nx < c.sw && ny < c.sh) nbRec = c.P[lsd_rec_index((unsigned)nx, (unsigned)ny, (unsigned)c.spitch)];
      }
      const unsigned seedRec = bcast_u32(seedRecL, 0);
      if (seedRec & LSD_USED) {   // swallowed by a committed region: certain, marks are never taken back once committed
        PF_ADD(c, 17, 1);
        break;
      }
      c.hWin = (unsigned)max(s - mw_ld_u(&ctl[MWC_HEAD]), 0);
      if (grow_older_claim(c, bcast_u32(seedHintL, 0))) {
        // an older transaction in flight holds the seed: predicted to be swallowed -- nothing to run, the commit checks
        PF_ADD(c, 28, 1);
        pFlags = 2u | 8u | ((unsigned)wv << 8) | (1u << 16);
        pMode = 1;
        break;
      }
      if (wballot(rec_is_candidate(nbRec)) == 0ull) {
        // No neighbour of the seed is both defined and free in the COMMITTED map (2 200 of a frame's 8 200 regions): region_grow()
        // can accept nothing, whatever the angles, and nothing can change that -- committed marks stay, undefined pixels stay
        // undefined.  The region is the seed alone and is dropped for its size: post exactly that, without running it.  (The commit
        // still checks that the seed itself has not been taken by an older transaction, like for any other accepted pixel.)
        PF_ADD(c, 36, 1);
        if (lane == 0) { c.ring[0] = seedPk; c.H[seedLin] = (uint16_t)c.hTag; }
        PLH_WAVE_SYNC();
        pAcc = 1; pAsm = 0;
        pFlags = 1u | (spec ? 2u : 0u) | 8u | ((unsigned)wv << 8) | (1u << 24);
        pMode = 2;
        break;
      }
      const unsigned long long pt0 = PF_NOW();
      MW_TRACE(wv, lane, 2, s);   // run begins
      const LsdTxn t = lsd_txn_mw(c, gs, a, regBase + off, seedPk, seedRec, ent.y, ent.z, ent.w);
      grow_lane_fence<true>();
      mw_jitter((unsigned)s + 211u);
      MW_TRACE(wv, lane, 3, t.logLen);   // run ends
      // through: take the private marks back (the plane is clean for the next transaction) and look once more whether an
      // older transaction has committed a pixel of the log meanwhile
      const uint32_t* log = regBase + off;
      const unsigned long long pv0 = PF_NOW();
      const int asmLen = (int)bcast_u32(*c.asmCnt, 0);
      // a region that was dropped for its size without a refine() is all in the LDS mirror of its queue: its post goes inline
      const bool inl = !t.keep && t.finBase == 0 && t.finCnt == t.logLen && t.logLen + asmLen <= MW_INLINE_MAX;
      bool bad = t.conflict;
      if (!bad) {
        for (int base = 0; base < t.logLen; base += 64) {
          const int i = base + lane;
          if (i < t.logLen) {
            const uint32_t li = pk_lin(c, inl ? c.ring[i] : log[i]);
            bad = bad || (c.P[li] & LSD_USED) != 0u;
            // ... or an older transaction still in flight has claimed a pixel of the final region since: it will commit first
            if (i >= t.finBase) bad = bad || grow_older_claim(c, (unsigned)c.H[li]);
          }
        }
        bad = spec && __ballot(bad) != 0ull;
      }
      if (bad) {
        for (int i = lane; i < t.logLen; i += 64) {
          const uint32_t li = pk_lin(c, log[i]);
          c.M[li] = 0; c.H[li] = 0;
        }
        grow_lane_fence<true>();
        PF_ADD(c, 18, 1); PF_ADD(c, 22, PF_NOW() - pt0);
        continue;   // again, against a fresher map
      }
      pAcc = t.logLen; pAsm = asmLen;
      if (inl) {
        if (lane < t.logLen) c.M[pk_lin(c, c.ring[lane])] = 0;
        pFlags = 1u | (spec ? 2u : 0u) | 8u | ((unsigned)wv << 8) | ((unsigned)asmLen << 16) | ((unsigned)t.logLen << 24);
        pMode = 2;
      } else {
        for (int i = lane; i < t.finCnt; i += 64) c.M[pk_lin(c, log[t.finBase + i])] = 0;
        const int used = max(t.logLen, t.finBase + t.finCnt);
        pFlags = 1u | (spec ? 2u : 0u) | (t.keep ? 4u : 0u) | ((unsigned)wv << 8) | ((unsigned)asmLen << 16);
        pMode = 3;
        pGen[0] = (unsigned)off; pGen[1] = (unsigned)t.logLen; pGen[2] = (unsigned)t.finBase; pGen[3] = (unsigned)t.finCnt;
        pAng = __float_as_uint(t.ang);
        if (asmLen <= 3) {
          off += used;
        } else {
          if (lane < asmLen) regBase[off + used + lane] = c.asmList[lane];
          off += used + asmLen;
        }
        posted++;
      }
      PF_ADD(c, 22, PF_NOW() - pt0); PF_ADD(c, 26, PF_NOW() - pv0);
      break;
    }
    const unsigned long long pp0 = PF_NOW();
    {
      unsigned pw = 0u;   // word `lane` of the post
      if (lane == MWP_FLAGS) pw = pFlags;
      else if (lane < MWP_PIX) pw = lane == 1 ? seedPk : (lane == 2 ? ent.y : (lane == 3 ? ent.z : ent.w));
      else if (pMode == 1) pw = seedPk;   // (only word MWP_PIX is read)
      else if (pMode == 2) {
        const int k = lane - MWP_PIX;
        if (k < pAcc) pw = c.ring[k];
        else if (k < pAcc + pAsm) pw = c.asmList[k - pAcc];
      } else if (pMode == 3) {
        if (lane < MWP_SEG) pw = lane == MWP_OFF ? pGen[0] : (lane == MWP_LOGLEN ? pGen[1] : (lane == MWP_FINBASE ? pGen[2] : pGen[3]));
        else if (lane < MWP_ASM) pw = lane == MWP_SEG ? pAng : 0u;
        else if (lane - MWP_ASM < min(pAsm, 3)) pw = c.asmList[lane - MWP_ASM];
      }
      mw_jitter((unsigned)s + 17u);
      MW_TRACE(wv, lane, 4, pMode);   // posting
      if (!(pFlags & 8u)) mw_release();   // a log in global memory is complete before its post is visible
      if (lane < MW_PEND_WORDS) sh.pend[s & (MW_N - 1)].w[lane] = pw;
      PLH_WAVE_SYNC();
      if (lane == 0) mw_st(&sh.state[s & (MW_N - 1)], s + 1);   // after the words (LDS operations of a wavefront execute in order)
    }
    PLH_WAVE_SYNC();
    PF_ADD(c, 27, PF_NOW() - pp0);
    // Commit now?  A drain session costs its set-up however few posts it finds: worth it when this was the oldest transaction
    // (nobody else can move the head past it) or a batch's worth has piled up; otherwise the next one to come by does it
    // (and a wavefront with nothing else to do always does).
    {
      const int hd = mw_ld_u(&ctl[MWC_HEAD]);
      if (s == hd || s - hd >= a.mwDrainGap) mw_try_drain(ch, gs, a, sh, frameReg, drainReg, segs, frameLog, frameLogQ);
    }
  }
  __syncthreads();
  if (tid == 0) {
    int nseg = ctl[MWC_NSEG];
    if (nseg > a.segCap) { atomicOr(a.status, 4); nseg = a.segCap; }
    a.nSegs[b] = nseg;
  }
#if defined(PLH_GROW_PROF)
  pfv[0] = PF_NOW() - pfStart;
  if (lane == 0)
    for (int i = 0; i < 40; i++) atomicAdd(&g_grow_prof[i], pfv[i]);
#endif
}

// ---------------------------------------------------------------------------------------------
// KeyLine construction + LINEextractor's selection.  One block per frame.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void clamp_extremes(float e[4], int w, int h) {   // checkLineExtremes()
  if (e[0] < 0) e[0] = 0;
  if (e[0] >= w) e[0] = (float)w - 1.0f;
  if (e[2] < 0) e[2] = 0;
  if (e[2] >= w) e[2] = (float)w - 1.0f;
  if (e[1] < 0) e[1] = 0;
  if (e[1] >= h) e[1] = (float)h - 1.0f;
  if (e[3] < 0) e[3] = 0;
  if (e[3] >= h) e[3] = (float)h - 1.0f;
}
__device__ __forceinline__ float seg_length(const float e[4]) {
  const double dxx = (double)(e[0] - e[2]), dyy = (double)(e[1] - e[3]);
  return (float)sqrt(dxx * dxx + dyy * dyy);
}

// Two builds of the same body: k_lsd_grow with the 64-register budget for batches (residency is what counts), and
// k_lsd_grow_lone without the cap (74 registers, no spills) for a handful of frames, where a lone wavefront per frame runs at
// the latency of its own instruction stream.
__global__ void __launch_bounds__(64) PLH_GROW_ATTR k_lsd_grow(LineDeviceArgs a) {
  HIP_DYNAMIC_SHARED(unsigned char, smem)
  lsd_grow_frame(a, smem);
}
__global__ void __launch_bounds__(64) k_lsd_grow_lone(LineDeviceArgs a) {
  HIP_DYNAMIC_SHARED(unsigned char, smem)
  lsd_grow_frame(a, smem);
}
// (LSD_REFINE_ADV needs no build of its own any more: rect_improve() reads the immutable level-line field only and decides
// nothing that region growing depends on, so it runs with the rectangles in k_lsd_rects)

// up to 8 wavefronts per frame: 256 registers to spare; 9 .. 16: half of that (the workgroup is 1024 threads)
__global__ void __launch_bounds__(512) k_lsd_grow_mw(LineDeviceArgs a) {
  HIP_DYNAMIC_SHARED(unsigned char, smem)
  lsd_grow_frame_mw(a, smem);
}
__global__ void __launch_bounds__(1024) k_lsd_grow_mw16(LineDeviceArgs a) {
  HIP_DYNAMIC_SHARED(unsigned char, smem)
  lsd_grow_frame_mw(a, smem);
}

// One item of the frame's detection list: octave 0's segments first, then octave 1's (LSDDetector_custom.cpp:161-199 walks the
// octaves in this order; with one octave n1 = 0).  Loads the segment, clamps it to its octave's image (checkLineExtremes) and says
// which octave it came from.
struct KlFrame {
  const float *segs0, *segs1;
  int n0, n1, w0, h0, w1, h1;
  float scale1;
};
__device__ __forceinline__ int kl_item(const KlFrame& f, int i, float e[4], int& w, int& h, float& scale) {
  const bool o1 = i >= f.n0;
  const float* sg = o1 ? f.segs1 + (i - f.n0) * 4 : f.segs0 + i * 4;
  e[0] = sg[0]; e[1] = sg[1]; e[2] = sg[2]; e[3] = sg[3];
  w = o1 ? f.w1 : f.w0; h = o1 ? f.h1 : f.h0; scale = o1 ? f.scale1 : 1.0f;
  clamp_extremes(e, w, h);
  return o1 ? 1 : 0;
}

// KeyLine selection, ONE WAVEFRONT per frame (round 5).  Rounds 1-4 ran a 256-thread block per frame that found the outCap best
// keys by outCap passes over ALL n keys: 0.6 ms alone, but such a block needs a free wave slot and 96 registers on all four SIMDs of
// one CU at the same moment, which the region-growing wavefronts of the other sub-batches rarely leave (17 ms inside the pipeline,
// on the line chain's critical path).  Now: the keys' responses go into a 256-bin histogram, the bins from the top that hold the
// first outCap keys give a candidate set (a few hundred of the ~1500 keys), and every candidate's rank among the candidates -- the
// number of larger keys; keys are unique -- is its place in the selection.  Same order as the repeated arg-max: (response
// descending, detection index ascending).  More candidates than KL_CAND (a frame whose lines all have one length): the old loop.
constexpr int KL_THREADS = 64, KL_BINS = 256, KL_CAND = 1024;
__device__ __forceinline__ int kl_bin(unsigned long long key) {   // monotone in the key's response (a length over the image's larger side: < 1.5)
  return min(KL_BINS - 1, (int)(__uint_as_float((unsigned)(key >> 32)) * 170.0f));
}
__global__ void __launch_bounds__(KL_THREADS) k_keylines(LineDeviceArgs a, plh_keyline* outKl, double* outFn, int* nOut) {
  HIP_DYNAMIC_SHARED(unsigned char, smem)
  unsigned long long* sel = (unsigned long long*)smem;   // [outCap]
  __shared__ unsigned long long s_cand[KL_CAND];
  __shared__ int s_hist[KL_BINS];
  __shared__ int s_valid, s_keep;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid;
  KlFrame f;
  f.segs0 = a.segs + (long long)b * a.arenaStride;
  f.n0 = min(a.nSegs[b], a.segCap);
  f.w0 = a.w; f.h0 = a.h;
  f.segs1 = a.segs1 ? a.segs1 + (long long)b * a.arena1Stride : nullptr;
  f.n1 = a.segs1 ? min(a.nSegs1[b], a.segCap1) : 0;
  f.w1 = a.w1; f.h1 = a.h1; f.scale1 = a.octScale1;
  const int n = f.n0 + f.n1;
  unsigned long long* keys = (unsigned long long*)(a.reg + (long long)b * a.arenaStride);   // scratch (free after k_lsd_rects)
  for (int i = tid; i < KL_BINS; i += KL_THREADS) s_hist[i] = 0;
  __syncthreads();
  int myValid = 0;
  for (int i = tid; i < n; i += KL_THREADS) {
    float e[4], sc;
    int w, h;
    kl_item(f, i, e, w, h, sc);
    bool valid = true;
    if (a.mask) {   // drop only if BOTH endpoints lie on mask == 0 (looked up at the full-resolution end points)
      const float sx = e[0] * sc, sy = e[1] * sc, ex = e[2] * sc, ey = e[3] * sc;
      if (a.mask[(long long)(int)sy * a.w + (int)sx] == 0 && a.mask[(long long)(int)ey * a.w + (int)ex] == 0) valid = false;
    }
    const float response = seg_length(e) / (float)max(w, h);
    const unsigned long long key = valid ? (((unsigned long long)__float_as_uint(response) << 32) | (unsigned long long)(0xffffffffu - (unsigned)i)) : 0ull;
    keys[i] = key;
    if (valid) atomicAdd(&s_hist[KL_BINS - 1 - kl_bin(key)], 1);   // (stored from the top: bin 0 holds the largest responses)
    myValid += valid;
  }
  int nValid = myValid;
  for (int m = 32; m >= 1; m >>= 1) nValid += __shfl_xor(nValid, m);
  __syncthreads();
  // the first (reversed) bin at which the count from the top reaches outCap: every key above it is selected for sure, the keys
  // of that bin compete for the remaining places
  int tRev = KL_BINS - 1;
  {
    const int h0 = s_hist[4 * lane], h1 = s_hist[4 * lane + 1], h2 = s_hist[4 * lane + 2], h3 = s_hist[4 * lane + 3];
    int incl = h0 + h1 + h2 + h3;
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    const int before = incl - (h0 + h1 + h2 + h3);
    int mine = 4 * KL_BINS;   // (no bin of this lane reaches outCap)
    if (before < a.outCap && incl >= a.outCap) {
      int c = before + h0;
      mine = 4 * lane;
      if (c < a.outCap) { c += h1; mine = 4 * lane + 1; }
      if (c < a.outCap) { c += h2; mine = 4 * lane + 2; }
      if (c < a.outCap) mine = 4 * lane + 3;
    }
    for (int m = 32; m >= 1; m >>= 1) mine = min(mine, __shfl_xor(mine, m));
    if (mine < KL_BINS) tRev = mine;   // (fewer valid keys than outCap: every one is a candidate)
  }
  // the candidates, in no particular order
  int C = 0;
  for (int i0 = 0; i0 < n; i0 += KL_THREADS) {
    const int i = i0 + lane;
    const unsigned long long key = i < n ? keys[i] : 0ull;
    const bool cand = key != 0ull && KL_BINS - 1 - kl_bin(key) <= tRev;
    const unsigned long long bm = __ballot(cand);
    const int at = C + __popcll(bm & ((1ull << lane) - 1ull));
    if (cand && at < KL_CAND) s_cand[at] = key;
    C += __popcll(bm);
  }
  __syncthreads();
  int K = min(nValid, a.outCap);
  if (C <= KL_CAND) {
    // rank = number of larger candidates
    for (int m = lane; m < C; m += KL_THREADS) {
      const unsigned long long key = s_cand[m];
      int rank = 0;
      for (int x = 0; x < C; x++) rank += s_cand[x] > key ? 1 : 0;
      if (rank < a.outCap) sel[rank] = key;
    }
  } else {
    // top-(outCap) by (response desc, detection index asc): repeated arg-max below the previous pick, over all keys
    unsigned long long prev = ~0ull;
    K = 0;
    for (int k = 0; k < a.outCap; k++) {
      unsigned long long best = 0;
      for (int i = tid; i < n; i += KL_THREADS) {
        const unsigned long long v = keys[i];
        if (v < prev && v > best) best = v;
      }
      for (int m = 32; m >= 1; m >>= 1) {
        const unsigned long long o = __shfl_xor(best, m);
        best = o > best ? o : best;
      }
      if (best == 0) break;
      if (tid == 0) sel[k] = best;
      prev = best;
      K++;
    }
  }
  if (tid == 0) s_valid = nValid;
  __syncthreads();
  if (tid == 0) {   // LineExtractor.cpp:44-64
    const int nv = s_valid;
    int total, index;
    if (nv > a.nFeature) { total = a.nFeature; index = a.nFeature; }
    else { total = nv; index = nv; }
    auto lenOf = [&](int k) -> float {
      const int i = (int)(0xffffffffu - (unsigned)(sel[k] & 0xffffffffull));
      float e[4], sc;
      int w, h;
      kl_item(f, i, e, w, h, sc);
      return seg_length(e);
    };
    if (total >= 1 && (double)lenOf(total - 1) < a.minLineLength) {
      for (int i = 0; i < total - 1; i++) {
        if ((double)lenOf(i) >= a.minLineLength && (double)lenOf(i + 1) < a.minLineLength) { index = i; break; }
      }
    }
    int keep = min(index + 1, nv);
    keep = min(keep, K);
    s_keep = keep;
    nOut[b] = keep;
  }
  __syncthreads();
  const int keep = s_keep;
  for (int k = tid; k < keep; k += KL_THREADS) {
    const int i = (int)(0xffffffffu - (unsigned)(sel[k] & 0xffffffffull));
    float e[4], sc;
    int w, h;
    const int oct = kl_item(f, i, e, w, h, sc);
    plh_keyline kl;
    kl.startPointX = e[0] * sc; kl.startPointY = e[1] * sc; kl.endPointX = e[2] * sc; kl.endPointY = e[3] * sc;
    kl.sPointInOctaveX = e[0]; kl.sPointInOctaveY = e[1]; kl.ePointInOctaveX = e[2]; kl.ePointInOctaveY = e[3];
    kl.lineLength = seg_length(e);
    const int x1 = cv_round(e[0]), y1 = cv_round(e[1]), x2 = cv_round(e[2]), y2 = cv_round(e[3]);
    kl.numOfPixels = max(abs(x2 - x1), abs(y2 - y1)) + 1;
    kl.angle = (float)atan2((double)(kl.endPointY - kl.startPointY), (double)(kl.endPointX - kl.startPointX));
    kl.class_id = k;
    kl.octave = oct;
    kl.size = (kl.endPointX - kl.startPointX) * (kl.endPointY - kl.startPointY);
    kl.response = kl.lineLength / (float)max(w, h);
    kl.pt_x = (kl.endPointX + kl.startPointX) / 2;
    kl.pt_y = (kl.endPointY + kl.startPointY) / 2;
    outKl[(long long)b * a.outCap + k] = kl;
    const double sx = kl.startPointX, sy = kl.startPointY, ex = kl.endPointX, ey = kl.endPointY;
    const double l0 = sy * 1.0 - 1.0 * ey, l1 = 1.0 * ex - sx * 1.0, l2 = sx * ey - sy * ex;
    const double nrm = sqrt(l0 * l0 + l1 * l1);
    double* fn = outFn + ((long long)b * a.outCap + k) * 3;
    fn[0] = l0 / nrm; fn[1] = l1 / nrm; fn[2] = l2 / nrm;
  }
}

// ---------------------------------------------------------------------------------------------
// Sobel 3x3 (dx, dy) of the 5x5-blurred frame, packed int16 x 2, REFLECT_101.
// ---------------------------------------------------------------------------------------------
// 4 bytes at p (any alignment) from aligned dword loads
__device__ __forceinline__ unsigned ld4_any(const uint8_t* p) {
  const int m = (int)((size_t)p & 3);
  const unsigned* ap = reinterpret_cast<const unsigned*>(p - m);
  const unsigned lo = ap[0];
  return m ? (unsigned)(((((unsigned long long)ap[1]) << 32) | lo) >> (8 * m)) : lo;
}

// packed 16-bit lanes (v_pk_add_u16 / v_pk_sub_i16 / v_pk_mad_u16: pk_add16, pk_sub16, pk_twice_plus16), v_perm_b32, v_alignbyte_b32: plh_shims.h
__device__ __forceinline__ unsigned perm_b32(unsigned hi, unsigned lo, unsigned sel) { return plh_perm(hi, lo, sel); }
__device__ __forceinline__ unsigned align_b32(unsigned hi, unsigned lo, unsigned sh) { return plh_alignbyte(hi, lo, sh); }

// One thread per 4 adjacent pixels.  Columns x-1 .. x+4 of the three rows come from three aligned dwords per row and one
// v_alignbyte pair; the sums are taken on pairs of columns in packed 16-bit lanes: the vertical smoothing r0 + 2 r1 + r2
// and the vertical difference r2 - r0 of six columns (three pairs each), then dx = smooth[c + 2] - smooth[c] and
// dy = diff[c] + 2 diff[c + 1] + diff[c + 2] for the four outputs (two pairs each), re-paired into (dx, dy) dwords with
// v_perm: 36 VALU instructions instead of 18 byte extractions and 4 x 11 scalar-lane additions.
__global__ void __launch_bounds__(256) k_sobel_pack(LineDeviceArgs a, PlhXcdGrid xg) {
  int bx, y, b;   // (plh_xcd_decode_tiles: a block is a row and reads its two neighbours -- behind one L2 a frame is fetched once, not thrice)
  if (!plh_xcd_decode_tiles(xg, bx, y, b)) return;
  const int x = (bx * 256 + threadIdx.x) * 4;
  if (x >= a.w) return;
  const uint8_t* S = a.tmpA + (long long)b * a.fullStride;
  const int ym = refl101(y - 1, a.h), yp = refl101(y + 1, a.h);
  const uint8_t *r0 = S + __mul24(ym, a.w), *r1 = S + __mul24(y, a.w), *r2 = S + __mul24(yp, a.w);
  uint32_t out[4];
  if (x >= 4 && x + 12 <= a.w) {   // the three aligned dwords around x-1 .. x+4 stay inside the row
    unsigned A[3], B[3], C[3];     // column pairs (x-1, x), (x+1, x+2), (x+3, x+4) of the three rows as 16-bit lanes
    const uint8_t* rows[3] = {r0, r1, r2};
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const uint8_t* p = rows[r] + x - 1;
      const unsigned sh = (unsigned)((size_t)p & 3);
      const unsigned* q = reinterpret_cast<const unsigned*>(p - sh);
      const unsigned q0 = q[0], q1 = q[1], q2 = q[2];
      const unsigned w0 = align_b32(q1, q0, sh), w1 = align_b32(q2, q1, sh);   // columns x-1 .. x+2, x+3 .. x+6
      A[r] = perm_b32(0u, w0, 0x0c010c00u);
      B[r] = perm_b32(0u, w0, 0x0c030c02u);
      C[r] = perm_b32(0u, w1, 0x0c010c00u);
    }
    const unsigned sA = pk_twice_plus16(A[1], pk_add16(A[0], A[2])), sB = pk_twice_plus16(B[1], pk_add16(B[0], B[2])),
                   sC = pk_twice_plus16(C[1], pk_add16(C[0], C[2]));
    const unsigned dA = pk_sub16(A[2], A[0]), dB = pk_sub16(B[2], B[0]), dC = pk_sub16(C[2], C[0]);
    const unsigned gx01 = pk_sub16(sB, sA), gx23 = pk_sub16(sC, sB);
    const unsigned gy01 = pk_add16(pk_twice_plus16(align_b32(dB, dA, 2u), dA), dB);
    const unsigned gy23 = pk_add16(pk_twice_plus16(align_b32(dC, dB, 2u), dB), dC);
    out[0] = perm_b32(gy01, gx01, 0x05040100u);
    out[1] = perm_b32(gy01, gx01, 0x07060302u);
    out[2] = perm_b32(gy23, gx23, 0x05040100u);
    out[3] = perm_b32(gy23, gx23, 0x07060302u);
  } else {
    int p0[6], p1[6], p2[6];   // columns x-1 .. x+4 of the three rows
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const int xx = refl101(min(x - 1 + k, a.w), a.w);   // columns beyond the row only feed discarded outputs
      p0[k] = r0[xx]; p1[k] = r1[xx]; p2[k] = r2[xx];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int gx = (p0[k + 2] - p0[k]) + 2 * (p1[k + 2] - p1[k]) + (p2[k + 2] - p2[k]);
      const int gy = (p2[k] - p0[k]) + 2 * (p2[k + 1] - p0[k + 1]) + (p2[k + 2] - p0[k + 2]);
      out[k] = pack_g(gx, gy);
    }
  }
  uint32_t* o = a.dxdy + (long long)b * a.fullStride + (__mul24(y, a.w) + x);
  if (x + 4 <= a.w && (((size_t)o) & 15) == 0) {
    *reinterpret_cast<uint4*>(o) = uint4{out[0], out[1], out[2], out[3]};
  } else {
    for (int k = 0; k < 4 && x + k < a.w; k++) o[k] = out[k];
  }
}

// ---------------------------------------------------------------------------------------------
// LBD.  One wavefront per line: lane = row of the 63-row line support region, walking the line's
// numOfPixels columns with the reference's sequential float accumulation; bands are then accumulated
// in row order by 9 band lanes, and the 72-float vector is normalised / binarised.
// coef = { gaussCoefL_[21], gaussCoefG_[63] } as float (host: exp() in double, cast at use).
// ---------------------------------------------------------------------------------------------
__device__ const unsigned char c_lbd_comb[64] = {0, 1, 0, 2, 0, 3, 0, 4, 0, 5, 0, 6, 1, 2, 1, 3, 1, 4, 1, 5, 1, 6, 2, 3,
                                                 2, 4, 2, 5, 2, 6, 2, 7, 2, 8, 3, 4, 3, 5, 3, 6, 3, 7, 3, 8, 4, 5, 4, 6,
                                                 4, 7, 4, 8, 5, 6, 5, 7, 5, 8, 6, 7, 6, 8, 7, 8};

// clamp((int)(short)roundf(v), 0, hi) of the walk (binary_descriptor_custom.cpp:1117-1124) for |v| < 32767, where the cast
// to short is the identity: everything below zero clamps to 0, and for c >= 0 roundf(c) = trunc(c) + (fract(c) >= 0.5).
__device__ __forceinline__ int lbd_coord(float v, int hi) {
  const float c = fmaxf(v, 0.f);
  const int i = (int)c + (plh_fract(c) >= 0.5f ? 1 : 0);
  return min(i, hi);
}
__device__ __forceinline__ int lbd_coord_wide(float v, int hi) {   // images from 16384 pixels a side: the literal form
  const int tc = (int)(short)roundf(v);
  return tc < 0 ? 0 : (tc > hi ? hi : tc);
}

constexpr int LBD_TILE_PITCH = 20;   // dwords per row of the offset / value tile: 16 samples + pad (rows stay 16-byte aligned)
__global__ void __launch_bounds__(64) k_lbd(LineDeviceArgs a, const plh_keyline* kls, const int* nOut, const float* coef,
                                            uint8_t* desc, PlhXcdGrid xg) {
  __shared__ float rows[4][64];                  // per support-region row: pL nL pO nO (after the global Gaussian)
  __shared__ float des[LBD_NUM_BANDS * 8];
  __shared__ float sq[LBD_NUM_BANDS * 8];
  __shared__ float scl[2];
  int li, b;   // (plh_xcd_decode: the bands of a frame's lines overlap -- one gradient plane behind one L2)
  if (!plh_xcd_decode(xg, li, b)) return;
  const int lane = threadIdx.x;
  if (li >= nOut[b]) return;
  const plh_keyline L = kls[(long long)b * a.outCap + li];
  // the gradient images of the line's octave (binary_descriptor_custom.cpp:1080-1104)
  const bool oct1 = L.octave != 0;
  const uint32_t* D = oct1 ? a.dxdy1 + (long long)b * a.full1Stride : a.dxdy + (long long)b * a.fullStride;
  const int imgW = oct1 ? a.w1 : a.w, imgH = oct1 ? a.h1 : a.h;
  const short lengthOfLSP = (short)L.numOfPixels;
  const short halfWidth = (lengthOfLSP - 1) / 2;
  const short halfHeight = (LBD_ROWS - 1) / 2;
  const int imageWidth = imgW - 1, imageHeight = imgH - 1;
  const bool wide = imgW >= 16384 || imgH >= 16384;   // walk coordinates can leave the range of a short
  const float midX = (float)(0.5 * (L.sPointInOctaveX + L.ePointInOctaveX));
  const float midY = (float)(0.5 * (L.sPointInOctaveY + L.ePointInOctaveY));
  // (float)cos / (float)sin of the double angle: the short evaluation on |angle| in [0, pi] (cos is even, sin is odd, and so
  // is rounding), the library's only when a result sits next to a float rounding boundary (plh_common.h) -- the two
  // library calls were 600 of this kernel's VALU instructions per line
  float dL0, dL1;
  {
    const double ad = fabs((double)L.angle);
    double sd, cd;
    sincos_0_2pi(ad, sd, cd);
    if (!(ad <= 6.2831853071795862 && float_round_is_safe(cd) && float_round_is_safe(sd))) { cd = cos(ad); sd = sin(ad); }
    dL0 = (float)cd;
    dL1 = __builtin_signbit(L.angle) ? -(float)sd : (float)sd;   // sin(-0.0) = -0.0
  }
  const float dO0 = -dL1, dO1 = dL0;
  float pL = 0, nL = 0, pO = 0, nO = 0;
  {
    float sCorX0 = -dL0 * halfWidth + dL1 * halfHeight + midX;
    float sCorY0 = -dL1 * halfWidth - dL0 * halfHeight + midY;
    // row r starts r sequential float steps from row 0: step r is taken by the lanes above r (the mask is scalar)
    // (the mask is shifted from a run-time ballot: with compile-time masks, ROCm 7.2's lowering of inverse_ballot to
    // "s_mov_b64 sN, <32-bit literal>" lost the upper 32 lanes of the masks 0xffffffffffffff80 .. 0xffffffff80000000 on
    // gfx950 -- measured, rows 32..62 started 25 steps short)
    unsigned long long stepMask = wballot(true);
#pragma unroll
    for (int r = 0; r < LBD_ROWS - 1; r++) {
      stepMask <<= 1;
      if (PLH_INV_BALLOT(stepMask)) { sCorX0 -= dL1; sCorY0 += dL0; }
    }
    float sCorX = sCorX0, sCorY = sCorY0;
    // A sum only ever grows by positive terms, so "if (g > 0) p += g; else n -= g;" is p += max(g, 0); n += max(-g, 0):
    // adding +0 changes nothing (the sums start at +0 and stay non-negative).
    auto accumulate = [&](uint32_t gv) {
      const float dx = (float)g_x(gv), dy = (float)g_y(gv);
      const float gDL = dx * dL0 + dy * dL1;
      const float gDO = dx * dO0 + dy * dO1;
      pL += fmaxf(gDL, 0.f);
      nL += fmaxf(-gDL, 0.f);
      pO += fmaxf(gDO, 0.f);
      nO += fmaxf(-gDO, 0.f);
    };
    auto next_offset = [&]() -> uint32_t {   // the current sample's linear offset; advances the walk by one float step
      const int xCor = wide ? lbd_coord_wide(sCorX, imageWidth) : lbd_coord(sCorX, imageWidth);
      const int yCor = wide ? lbd_coord_wide(sCorY, imageHeight) : lbd_coord(sCorY, imageHeight);
      sCorX += dL0;
      sCorY += dL1;
      return (uint32_t)(__mul24(yCor, imgW) + xCor);   // 32-bit offset: the 64-bit multiply-add of a long long index is a quarter-rate instruction
    };
    if (fabsf(dL0) <= fabsf(dL1)) {
      // Steep line: the 63 rows of one walk step are neighbours along an image row, a gather touches a few cache lines.  The
      // walk's addresses do not depend on the data: 8 gathers are issued together, then accumulated in walk order (coordinates
      // and sums advance by the same float additions, in the same order, as the one-pixel-at-a-time loop).
      for (int w0 = 0; w0 < lengthOfLSP; w0 += 8) {
        uint32_t g[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
          g[k] = 0;
          if (w0 + k < lengthOfLSP) g[k] = D[next_offset()];
        }
#pragma unroll
        for (int k = 0; k < 8; k++)
          if (w0 + k < lengthOfLSP) accumulate(g[k]);
      }
    } else {
      // Flat line: the 63 rows of one walk step lie in 63 different image rows -- 63 cache lines per gather, and the kernel ran
      // at the rate the texture-address unit walks them (2.8 ms per 1536 frames for 0.5 M VALU instructions per frame).  So
      // the gathers run ALONG the line instead: every row lane writes the offsets of its next 16 samples to an LDS tile, the
      // wavefront re-reads the tile as 4 rows x 16 consecutive samples per instruction (consecutive samples of a flat line are
      // neighbours in an image row), gathers, writes the values back in place, and every row lane then accumulates its 16
      // values in walk order -- same coordinates, same float sums, a tenth of the cache lines per gather.
      __shared__ __attribute__((aligned(16))) uint32_t tile[64 * LBD_TILE_PITCH];
      const int gr = lane >> 4, gj = lane & 15;
      for (int w0 = 0; w0 < lengthOfLSP; w0 += 16) {
        uint32_t o[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
          o[k] = 0;   // beyond the line: any valid address, the value is not accumulated
          if (w0 + k < lengthOfLSP) o[k] = next_offset();
        }
        uint4* trow = reinterpret_cast<uint4*>(tile + lane * LBD_TILE_PITCH);
#pragma unroll
        for (int q = 0; q < 4; q++) { uint4 v; v.x = o[4 * q]; v.y = o[4 * q + 1]; v.z = o[4 * q + 2]; v.w = o[4 * q + 3]; trow[q] = v; }
        __syncthreads();
        uint32_t gv[16];
#pragma unroll
        for (int rb = 0; rb < 16; rb++) gv[rb] = D[tile[(4 * rb + gr) * LBD_TILE_PITCH + gj]];
#pragma unroll
        for (int rb = 0; rb < 16; rb++) tile[(4 * rb + gr) * LBD_TILE_PITCH + gj] = gv[rb];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint4 v = trow[q];
          if (w0 + 4 * q < lengthOfLSP) accumulate(v.x);
          if (w0 + 4 * q + 1 < lengthOfLSP) accumulate(v.y);
          if (w0 + 4 * q + 2 < lengthOfLSP) accumulate(v.z);
          if (w0 + 4 * q + 3 < lengthOfLSP) accumulate(v.w);
        }
        __syncthreads();
      }
    }
    const float cg = coef[21 + min(lane, LBD_ROWS - 1)];
    rows[0][lane] = cg * pL; rows[1][lane] = cg * nL; rows[2][lane] = cg * pO; rows[3][lane] = cg * nO;   // lane 63: an unused 64th row
  }
  __syncthreads();
  // band accumulation in row order: band lane bb takes rows 7 (bb - 1) .. 7 (bb + 2) - 1.  Row hID = 7 (bb - 1) + r sits in
  // the band above (r < 7), the band itself or the band below; its local Gaussian weight gaussCoefL_[hID % 7 + 7 (r / 7)]
  // is coef[r] -- the same for every band.
  const int bb = lane;
  if (bb < LBD_NUM_BANDS) {
    float spL = 0, snL = 0, spL2 = 0, snL2 = 0, spO = 0, snO = 0, spO2 = 0, snO2 = 0;
    const int row0 = LBD_BAND_WIDTH * (bb - 1);
#pragma unroll
    for (int r = 0; r < 3 * LBD_BAND_WIDTH; r++) {
      const bool valid = (r >= LBD_BAND_WIDTH || bb > 0) && (r < 2 * LBD_BAND_WIDTH || bb < LBD_NUM_BANDS - 1);
      if (valid) {
        const float rpL = rows[0][row0 + r], rnL = rows[1][row0 + r], rpO = rows[2][row0 + r], rnO = rows[3][row0 + r];
        const float cL = coef[r];
        spL += cL * rpL;
        snL += cL * rnL;
        spL2 += cL * cL * (rpL * rpL);
        snL2 += cL * cL * (rnL * rnL);
        spO += cL * rpO;
        snO += cL * rnO;
        spO2 += cL * cL * (rpO * rpO);
        snO2 += cL * cL * (rnO * rnO);
      }
    }
    const float invN2 = (float)(1.0 / (LBD_BAND_WIDTH * 2.0)), invN3 = (float)(1.0 / (LBD_BAND_WIDTH * 3.0));
    const float invN = (bb == 0 || bb == LBD_NUM_BANDS - 1) ? invN2 : invN3;
    float* d = &des[bb * 8];
    float temp = spL * invN;
    d[0] = temp; d[4] = sqrtf(spL2 * invN - temp * temp);
    temp = snL * invN;
    d[1] = temp; d[5] = sqrtf(snL2 * invN - temp * temp);
    temp = spO * invN;
    d[2] = temp; d[6] = sqrtf(spO2 * invN - temp * temp);
    temp = snO * invN;
    d[3] = temp; d[7] = sqrtf(snO2 * invN - temp * temp);
  }
  __syncthreads();
  // normalise means and deviations separately, clamp at 0.4, renormalise (binary_descriptor_custom.cpp:1322-1366).  The
  // squares are taken by all lanes; the sums, whose order the float result depends on, by one lane per sum.
  if (lane < 36) {   // chain position i = 4 q + k: des[8 q + k] (means), des[8 q + 4 + k] (deviations)
    const int e = 8 * (lane >> 2) + (lane & 3);
    const float m = des[e], sd = des[e + 4];
    sq[lane] = m * m;
    sq[36 + lane] = sd * sd;
  }
  __syncthreads();
  if (lane < 2) {
    const float* c = sq + 36 * lane;
    float t = 0;
#pragma unroll
    for (int i = 0; i < 36; i++) t += c[i];
    scl[lane] = 1 / sqrtf(t);
  }
  __syncthreads();
#pragma unroll
  for (int e0 = 0; e0 < LBD_NUM_BANDS * 8; e0 += 64) {
    const int e = e0 + lane;
    if (e < LBD_NUM_BANDS * 8) {
      float v = des[e] * scl[(e >> 2) & 1];
      if (v >= 0.4f) v = 0.4f;   // (double)v > 0.4  <=>  v >= 0.4f (the float next to 0.4 from above); NaN stays
      des[e] = v;
      sq[e] = v * v;
    }
  }
  __syncthreads();
  if (lane == 0) {
    float t = 0;
#pragma unroll
    for (int i = 0; i < LBD_NUM_BANDS * 8; i++) t += sq[i];
    scl[0] = 1 / sqrtf(t);
  }
  __syncthreads();
  if (lane < 32) {   // des[i] * temp on both sides of each comparison
    const float t = scl[0];
    const float* f1 = &des[8 * c_lbd_comb[lane * 2]];
    const float* f2 = &des[8 * c_lbd_comb[lane * 2 + 1]];
    int result = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
      if (f1[i] * t > f2[i] * t) result += 1 << i;
    desc[((long long)b * a.outCap + li) * 32 + lane] = (uint8_t)result;
  }
}

// ---------------------------------------------------------------------------------------------
size_t lsd_grow_lds_bytes(int spitch, int sh);
void launch_lsd_grow(const LineDeviceArgs& a, hipStream_t s) {
  if (a.mwWaves > 0) {
    const size_t ldsMw = (size_t)MWC_WORDS * 4 + (size_t)MW_N * (4 + MW_PEND_WORDS * 4) + (size_t)MW_F * 16 + (size_t)a.mwWaves * MW_WAVE_LDS;
    // (beyond 64 KiB the dynamic LDS has to be requested per kernel and device: plh_line_create does, lsd_grow_request_lds)
    // (the roomy build holds three wavefronts per SIMD: beyond two per SIMD over the whole GPU take the 128-register one)
    const bool roomy = a.mwWaves <= 8 && (long long)a.batch * a.mwWaves <= 2048;
    const dim3 g(a.batch), b(64 * a.mwWaves);
    if (roomy) hipLaunchKernelGGL(k_lsd_grow_mw, g, b, ldsMw, s, a);
    else hipLaunchKernelGGL(k_lsd_grow_mw16, g, b, ldsMw, s, a);
    return;
  }
  const size_t lds = lsd_grow_lds_bytes(a.spitch, a.sh);
  if (a.batch <= 8) hipLaunchKernelGGL(k_lsd_grow_lone, dim3(a.batch), dim3(64), lds, s, a);
  else hipLaunchKernelGGL(k_lsd_grow, dim3(a.batch), dim3(64), lds, s, a);
}
// The multi-wavefront kernels with more than ~11 wavefronts per frame need more than 64 KiB of dynamic LDS, which has to be
// asked for per kernel on the CURRENT device: once per handle at create time (not behind a process-wide flag -- a second GPU in
// the same process would never get it).
plh_status lsd_grow_request_lds() {
  plh_status st = lds_request(k_lsd_grow_mw, 160u * 1024u, "k_lsd_grow_mw");
  if (st == PLH_OK) st = lds_request(k_lsd_grow_mw16, 160u * 1024u, "k_lsd_grow_mw16");
  return st;
}
void launch_keylines(const LineDeviceArgs& a, plh_keyline* kl, double* fn, int* n, hipStream_t s) {
  hipLaunchKernelGGL(k_keylines, dim3(a.batch), dim3(KL_THREADS), (size_t)a.outCap * 8 + 64, s, a, kl, fn, n);
}
void launch_sobel(const LineDeviceArgs& a, hipStream_t s) {
  const PlhXcdGrid xg = plh_xcd_make((a.w + 1023) / 1024, a.h, a.batch);
  hipLaunchKernelGGL(k_sobel_pack, dim3(plh_xcd_grid(xg)), dim3(256), 0, s, a, xg);
}
void launch_lbd(const LineDeviceArgs& a, const plh_keyline* kl, const int* n, const float* coef, uint8_t* desc, hipStream_t s) {
  const PlhXcdGrid xg = plh_xcd_make(a.outCap, a.batch);
  hipLaunchKernelGGL(k_lbd, dim3(plh_xcd_grid(xg)), dim3(64), 0, s, a, kl, n, coef, desc, xg);
}
#if defined(PLH_GROW_PROF)
#if PLH_GROW_PROF + 0 >= 3 && !defined(HIPEMU)
extern "C" __attribute__((visibility("default"))) int plh_debug_mw_trace(unsigned long long* out, unsigned cap, int reset) {
  unsigned n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_mw_trace_n), 4) != hipSuccess) return -1;
  n = n < cap ? n : cap;
  if (n > (1u << 20)) n = 1u << 20;
  if (out && n && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mw_trace), (size_t)n * 16) != hipSuccess) return -1;
  if (reset) {
    const unsigned z = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_mw_trace_n), &z, 4) != hipSuccess) return -1;
  }
  return (int)n;
}
#endif
#endif
#if defined(PLH_MW_PARANOID) && !defined(HIPEMU)
extern "C" __attribute__((visibility("default"))) int plh_debug_mw_paranoid(unsigned* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_mw_paranoid), 64) != hipSuccess) return 1;
  if (reset) {
    unsigned z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_mw_paranoid), z, 64) != hipSuccess) return 1;
  }
  return 0;
}
#endif
#if defined(PLH_GROW_PROF)
extern "C" __attribute__((visibility("default"))) int plh_debug_grow_prof(unsigned long long* out40, int reset) {
#if defined(HIPEMU)
  memcpy(out40, g_grow_prof, sizeof(g_grow_prof));
  if (reset) memset(g_grow_prof, 0, sizeof(g_grow_prof));
#else
  if (hipMemcpyFromSymbol(out40, HIP_SYMBOL(g_grow_prof), sizeof(unsigned long long) * 40) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[40] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_grow_prof), z, sizeof(z)) != hipSuccess) return 1;
  }
#endif
  return 0;
}
#endif
size_t lsd_grow_lds_bytes(int spitch, int sh) {   // ring + scan table + first-step records + batch index
  (void)spitch; (void)sh;
  return (size_t)LSD_RING * 4 + 5 * 64 * 4 + 64 * (16 + 4 + 4) + 8 * 4 + LSD_GS_D * 8 + 8 * 4;
}

}  // namespace plh
