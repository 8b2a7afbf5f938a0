// The rectangle arithmetic of cv::LineSegmentDetector (region2rect / get_theta, and LSD_REFINE_ADV's nfa) as per-value device
// functions shared by the two kernels that evaluate it: k_lsd_grow* (lsd_grow.hip: one wavefront per frame, only where a
// decision of region growing needs the exact rectangle) and k_lsd_rects (lsd_rects.hip: one lane per kept region, the
// rectangles that become segments).  Same expressions, same order of operations, no contraction (-ffp-contract=off): the two
// kernels produce the same doubles, which are the oracle's (oracle/lsd.cc region2rect, get_theta, nfa).
#pragma once
#include "line_dev.h"
#include "plh_sincos_cr.h"

namespace plh {

struct alignas(16) D2 {
  double x, y;
};

// angle_diff_signed() of lsd.cpp: `a -= b; while (a <= -pi) a += 2 pi; while (a > pi) a -= 2 pi;`.  Every caller in the kernels
// hands in two angles of [0, 3 pi] (level-line angles, region angles, theta + pi), so |a - b| <= 3 pi and each loop body runs at
// most twice: the loops are written as that many conditional steps -- the same additions in the same order, selects instead of
// EXEC-masked loops.  (Round 6: a data-dependent loop here put a divergent join block right in front of the out-of-line sincos call of
// region2rect(), and ROCm 7.2's register allocator placed the copies that save caller-saved registers around that call IN FRONT of the
// join block's EXEC restore -- profiles/r06_prof_build_mw16_fault_root_cause.txt; tools/isa_exec_split_check.py watches for the shape.)
__device__ __forceinline__ double angle_diff_signed(double a, double b) {
  double diff = a - b;
  diff = (diff <= -kPI) ? diff + k2PI : diff;
  diff = (diff <= -kPI) ? diff + k2PI : diff;
  diff = (diff > kPI) ? diff - k2PI : diff;
  diff = (diff > kPI) ? diff - k2PI : diff;
  return diff;
}

__device__ __forceinline__ double dist_sq(double x1, double y1, double x2, double y2) {
  return (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1);
}

// isAligned() of cv::LineSegmentDetector for a defined pixel: |theta - a| folded at 3pi/2, compared with prec.
__device__ __forceinline__ bool lsd_aligned(double theta, double a, double prec) {
  double n_theta = theta - a;
  if (n_theta < 0) n_theta = -n_theta;
  if (n_theta > k3_2PI) {
    n_theta -= k2PI;
    if (n_theta < 0) n_theta = -n_theta;
  }
  return n_theta <= prec;
}

// get_theta(): the rectangle angle from the inertia sums (oracle/lsd.cc get_theta)
__device__ __forceinline__ double lsd_rect_theta(double Ixx, double Iyy, double Ixy, double reg_angle, double prec) {
  const double lambda = 0.5 * (Ixx + Iyy - sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
  double theta = (fabs(Ixx) > fabs(Iyy)) ? (double)fast_atan2_deg((float)(lambda - Ixx), (float)Ixy)
                                         : (double)fast_atan2_deg((float)Ixy, (float)(lambda - Iyy));
  theta *= kDegToRads;
  if (fabs(angle_diff_signed(theta, reg_angle)) > prec) theta += kPI;
  return theta;
}

// cos / sin of the rectangle angle (theta in [0, 3 pi)), x = cos, y = sin: correctly rounded (plh_sincos_cr.h)
__device__ __forceinline__ D2 lsd_sincos_inl(double t) {
  D2 r;
  sincos_cr(t, r.y, r.x);
  return r;
}

// ---------------------------------------------------------------------------------------------
// LSD_REFINE_ADV (cv::LineSegmentDetector created with LSD_REFINE_ADV, what the system opencv_contrib LSDDetector behind
// src/LineExtractor.cpp:39-40 passes as published; oracle/lsd.cc restates it with the published code's quirks): a rectangle
// is kept only if its number of false alarms says it is meaningful, after up to five kinds of adjustment.
//   nfa()          -log10(NT x binomial tail), Lanczos / Windschitl log-gamma
// ---------------------------------------------------------------------------------------------
struct LsdAdvRect {
  double x1, y1, x2, y2, width, theta, dx, dy, prec, p;
};

__device__ __attribute__((noinline)) static double lsd_log_gamma(double x) {
  if (x > 15.0) return 0.918938533204673 + (x - 0.5) * log(x) - x + 0.5 * x * log(x * sinh(1 / x) + 1 / (810.0 * pow(x, 6.0)));
  const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511};
  double a = (x + 0.5) * log(x + 5.5) - (x + 5.5);
  double b = 0;
  for (int n = 0; n < 7; ++n) {
    a -= log(x + double(n));
    b += q[n] * pow(x, double(n));
  }
  return a + log(b);
}

// nfa()'s log_gamma() is only ever asked for integers -- k + 1 and n - k + 1 with n, k pixel counts of a rectangle, at most the
// scaled image's pixel count -- so the handle keeps lsd_log_gamma(i) for every i it can be asked for (LineDeviceArgs::lgamma,
// filled once by k_lsd_lgamma_table with the function above: the same doubles).  Round 4 evaluated the two calls per nfa():
// Lanczos below 15 (seven log + pow pairs: every well-aligned rectangle, where n - k is small) or Windschitl (log, sinh, pow) --
// most of a 194-register kernel.  What is left is log(p), log(1 - p), exp, the tail loop and log10.
__device__ __forceinline__ double lsd_nfa(int n, int k, double p, double logNT, const double* lgamma) {
  if (n == 0 || k == 0) return -logNT;
  if (n == k) return -logNT - double(n) * log10(p);
  const double p_term = p / (1 - p);
  // (n + 1) where the original algorithm has log_gamma(n + 1): as published (oracle/lsd.cc)
  const double log1term = (double(n) + 1) - lgamma[k + 1] - lgamma[n - k + 1] + double(k) * log(p) +
                          (double(n - k)) * log(1.0 - p);
  double term = exp(log1term);
  {
    // double_equal(term, 0)
    const double aa = fabs(term);
    const double abs_max = aa < 2.2250738585072014e-308 ? 2.2250738585072014e-308 : aa;
    if (term == 0.0 || (aa / abs_max) <= (100.0 * 2.2204460492503131e-16)) {
      if (k > n * p) return -log1term / 2.30258509299404568402 - logNT;
      return -logNT;
    }
  }
  double bin_tail = term;
  const double tolerance = 0.1;
  for (int i = k + 1; i <= n; ++i) {
    const double bin_term = double(n - i + 1) / double(i);
    const double mult_term = bin_term * p_term;
    term *= mult_term;
    bin_tail += term;
    if (bin_term < 1) {
      const double err = term * ((1 - pow(mult_term, double(n - i + 1))) / (1 - mult_term) - 1);
      if (err < tolerance * fabs(-log10(bin_tail) - logNT) * bin_tail) break;
    }
  }
  return -log10(bin_tail) - logNT;
}

// flsd(): + 0.5, / SCALE
__device__ __forceinline__ void lsd_segment_of(const double* rec, float seg[4]) {
  seg[0] = (float)((rec[0] + 0.5) / 0.8); seg[1] = (float)((rec[1] + 0.5) / 0.8);
  seg[2] = (float)((rec[2] + 0.5) / 0.8); seg[3] = (float)((rec[3] + 0.5) / 0.8);
}

// What k_lsd_grow* leaves behind for every region it keeps, in the region's segment slot (16 bytes, like the segment that
// k_lsd_rects puts in its place): where the region's pixels lie in the frame's log (a.reg: packed coordinates, region order),
// how many, and the region angle of its last region_grow() (float degrees, as fastAtan2 returned it).
struct LsdRegionEntry {
  uint32_t logOff, cnt, angBits, pad;
};

// ---------------------------------------------------------------------------------------------
// LSD_REFINE_ADV on the kept regions' rectangles (lsd_rects.hip computes them, lsd_adv.hip improves them).  rect_improve()
// reads the immutable level-line field only and decides only whether the segment is kept, so it runs per rectangle, not per
// frame.  Its two halves: rect_nfa()'s scan of the rectangle's pixels (eight lanes per rectangle along the scan lines, one float
// per pixel) and nfa() (one lane per evaluation; with log_gamma() from the handle's table it is log, exp and a short tail loop).
// Between the kernels a rectangle lives in an LsdAdvRec (a lazily allocated buffer of segCap records per frame).
// ---------------------------------------------------------------------------------------------
struct alignas(16) LsdAdvRec {
  double r[10];          // x1 y1 x2 y2 width theta dx dy prec p: the rectangle as region2rect() left it (k_lsd_rects_adv)
  double log_nfa;        // its log_nfa (k_adv_first), for the rectangles that go on to rect_improve() (k_adv_improve)
};
constexpr size_t kLsdAdvRecBytes = 96;   // line_host.hip sizes the per-frame buffer with this
static_assert(sizeof(LsdAdvRec) == kLsdAdvRecBytes, "LsdAdvRec: the host's allocation and the kernels' indexing must agree");

constexpr uint32_t RC_DROPPED = 0xffffffffu;   // first word of a slot whose rectangle LSD_REFINE_ADV rejected (a NaN: never a coordinate)

struct RcFrame {
  const float* ang;   // level-line angle per pixel, float degrees, -1024 = NOTDEF (LineDeviceArgs::advAng)
  int spitch, sw, sh;
};
// rc_row(): where row y of the angle plane starts; rc_at(): pixel x of that row.  The plane is row-major.  A rectangle is scanned row
// by row, a few pixels per row unless it is flat, so every scan line of a steep rectangle is a sector of its own for 3 - 5 floats; with
// the plane in the record plane's 4 x 4-pixel blocks (lsd_rec_index, line_plan.h; -DPLH_ANG_TILED) four consecutive scan lines share
// theirs: measured (profiles/r06b_angle_plane_tiled_ab.txt) k_adv_first 3.2 -> 1.9 MB and k_adv_improve 2.0 -> 1.7 MB of PMC traffic per
// frame, but the three more address instructions per load make the walks dearer than the misses they save: the headline - 0.65 % in
// two same-job A/Bs.  Not adopted; the switch stays for the next look at these kernels.
#if defined(PLH_ANG_TILED)
__device__ __forceinline__ const float* rc_row(const RcFrame& f, int y) {
  return f.ang + (__umul24((unsigned)y >> 2, (unsigned)f.spitch << 2) + (((unsigned)y & 3u) << 2));
}
__device__ __forceinline__ float rc_at(const float* row, int x) { return row[(((unsigned)x >> 2) << 4) + ((unsigned)x & 3u)]; }
#else
__device__ __forceinline__ const float* rc_row(const RcFrame& f, int y) { return f.ang + __umul24((unsigned)y, (unsigned)f.spitch); }
__device__ __forceinline__ float rc_at(const float* row, int x) { return row[x]; }
#endif

__device__ __forceinline__ LsdAdvRect lsd_adv_load(const double* q) {
  LsdAdvRect r;
  r.x1 = q[0]; r.y1 = q[1]; r.x2 = q[2]; r.y2 = q[3]; r.width = q[4]; r.theta = q[5]; r.dx = q[6]; r.dy = q[7]; r.prec = q[8]; r.p = q[9];
  return r;
}

// The pixel counts of rect_nfa() (oracle/lsd.cc rect_nfa, with the published code's quirks: integer scan-line steps, the tail
// point's x where a y is meant).  The scan-line bounds advance by integer steps from an integer start; rows outside the image
// are skipped before the step update (`continue`), so they do not count (LsdScanWalk below).  One float per pixel from the angle
// plane k_lsd_grad writes for this level (no record -> table chain).
struct LsdScanGeom {   // what the scan of a rectangle's rows needs: the top corner's x, the rows of the left / right corners, the steps
  int mx, ly, ry, fl, sl, fr, sr, yA, yB;
};
__device__ __forceinline__ LsdScanGeom lsd_scan_geom(const RcFrame& f, const LsdAdvRect& r) {
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
  // (int arithmetic: corners lie within a rectangle's width of the image, |coordinates| < 2^15, so steps x rows stay below 2^31)
  LsdScanGeom g;
  g.mx = mx; g.ly = ly; g.ry = ry;
  g.fl = (my != ly) ? (mx - lx) / (my - ly) : 0; g.sl = (ly != tx) ? (lx - tx) / (ly - tx) : 0;
  g.fr = (my != ry) ? (mx - rx) / (my - ry) : 0; g.sr = (ry != tx) ? (rx - tx) / (ry - tx) : 0;
  g.yA = max(my, 0); g.yB = min(oy[iMax], f.sh - 1);   // the scan lines inside the image
  return g;
}
// (an empty geometry: no scan line at all -- what a lane group without a rectangle walks)
__device__ __forceinline__ LsdScanGeom lsd_scan_none() {
  LsdScanGeom g;
  g.mx = 0; g.ly = 0; g.ry = 0; g.fl = 0; g.sl = 0; g.fr = 0; g.sr = 0; g.yA = 1; g.yB = 0;
  return g;
}
__device__ __forceinline__ bool lsd_aligned_deg(double theta, float angDeg, double prec) {
  return angDeg >= 0.f && lsd_aligned(theta, (double)angDeg * kDegToRads, prec);
}

// isAligned() of a pixel against a rectangle, decided in float degrees wherever that is safe.  The reference folds |theta - a| at
// 3 pi / 2 and compares with prec in double.  The float difference of the two angles in degrees is off by < 2e-4 degrees (rounding
// of theta to a float below 540, of the subtraction), so more than 1e-3 degrees away from the tolerance the float verdict is the
// double one; only the lanes inside that band take lsd_aligned() itself (rare).  Near the
// fold (a difference of 270 degrees) either side is "not aligned" for every tolerance below 89 degrees (rect_improve's are at most
// 22.5), and NOTDEF (-1024) is more than 664 degrees from any theta: no special cases.
struct LsdAlignTol {
  double theta, prec;
  float thDeg, lo, hi;
};
__device__ __forceinline__ LsdAlignTol lsd_align_tol(double theta, double prec) {
  LsdAlignTol t;
  t.theta = theta; t.prec = prec;
  t.thDeg = (float)(theta * (180.0 / kPI));
  const float pd = (float)(prec * (180.0 / kPI));
  t.lo = pd - 1e-3f; t.hi = pd + 1e-3f;
  return t;
}
__device__ __forceinline__ float lsd_fold_deg(float thDeg, float angDeg) {   // the folded difference in float degrees
  const float d = fabsf(thDeg - angDeg);
  return d > 270.f ? fabsf(d - 360.f) : d;
}

// The walk over a rectangle's scan lines, EIGHT LANES per rectangle: lane j of the group takes pixel xa + j (+ 8, + 16, ...) of
// every scan line, so that a group reads runs of consecutive floats (one cache line per scan line and group where a lane per
// rectangle touched 64 lines per load), two scan lines per trip.  The published loop advances its bounds AFTER a scan line -- by
// the first kind of step while the line lies above the left (right) corner, by the second kind from that corner's line on -- and
// only on lines inside the image (`continue` in front of the update): the walk starts at the first line inside the image and does
// exactly that.  The geometry is uniform in the group (computed once per rectangle by ONE lane and handed over through LDS by
// value: the corner sort is ~500 instructions).  Loads are unconditional (clamped addresses, the value replaced by NOTDEF where the
// lane has no pixel) and the alignment test has no branch except the one around the rare exact path: round 5's first version of
// this loop compiled to ~200 instructions per pair of scan lines, this one to ~60.
// totalOut is the rectangle's pixel count (the same in all eight lanes), algOut THIS LANE's share of the aligned pixels: the
// caller adds the eight up.  The counts are sums over pixels, so how the pixels are dealt to lanes does not matter.
__device__ __forceinline__ void lsd_rect_counts_g8(const RcFrame& f, const LsdScanGeom g, const LsdAlignTol& t, int j, int& totalOut, int& algOut) {
  const int ly = g.ly, ry = g.ry, fl = g.fl, sl = g.sl, fr = g.fr, sr = g.sr, yB = g.yB, xMax = f.sw - 1;
  int left = g.mx, right = g.mx, total = 0, alg = 0;
#pragma clang loop unroll(disable) vectorize(disable)
  for (int y = g.yA; y <= yB; y += 2) {
    const int xa0 = max(left, 0), n0 = max(min(right, xMax) - xa0 + 1, 0);
    left += y < ly ? fl : sl; right += y < ry ? fr : sr;
    const bool two = y + 1 <= yB;
    const int xa1 = max(left, 0), n1 = two ? max(min(right, xMax) - xa1 + 1, 0) : 0;
    left += two ? (y + 1 < ly ? fl : sl) : 0; right += two ? (y + 1 < ry ? fr : sr) : 0;
    total += n0 + n1;
    const float* row0 = rc_row(f, y);
    const float* row1 = rc_row(f, two ? y + 1 : y);
    const int nmax = max(n0, n1);
#pragma clang loop unroll(disable) vectorize(disable)
    for (int k = j; k < nmax; k += 8) {
      const float v0 = rc_at(row0, min(xa0 + k, xMax)), v1 = rc_at(row1, min(xa1 + k, xMax));
      const float d0 = lsd_fold_deg(t.thDeg, k < n0 ? v0 : -1024.f), d1 = lsd_fold_deg(t.thDeg, k < n1 ? v1 : -1024.f);
      bool r0 = d0 < t.lo, r1 = d1 < t.lo;
      const bool m0 = !r0 && d0 <= t.hi, m1 = !r1 && d1 <= t.hi;
      if (m0 || m1) {   // within 1e-3 degrees of the tolerance (rare): the reference's own arithmetic
        if (m0) r0 = lsd_aligned_deg(t.theta, v0, t.prec);   // (no wave vote here: the lane groups' loops are not in step)
        if (m1) r1 = lsd_aligned_deg(t.theta, v1, t.prec);
      }
      alg += (r0 ? 1 : 0) + (r1 ? 1 : 0);
    }
  }
  totalOut = total; algOut = alg;
}

// ONE walk for the five variants of a width stage of rect_improve() (reduce width / one side / the other side): the variants are
// the stage's rectangle narrowed by 0.5 .. 2.5 pixels, so their scan lines and spans nearly coincide -- but each has its own corner
// sort, its own integer steps and its own first and last scan line, so each keeps its own walk (left / right bounds advanced on its
// own lines only); what is shared is the pixel: a scan line's union span is loaded once, its pixels are tested for alignment once
// (same theta, same tolerance in these stages) and counted for every variant whose span holds them.  Same counts as five separate
// walks, a third of the loads -- and the walks, not nfa(), are what k_adv_improve's duration is made of
// (profiles/r05_rects_improve_alone_variants.txt).  gs: the five geometries in LDS (lsd_scan_none() for a variant the width gate
// stopped).
struct LsdVar5Line {   // one scan line of the five variants: their spans (empty: xa = 0xffff, xb = 0) and the union of the non-empty ones
  int xa[5], xb[5], ua, ub;
};
__device__ __forceinline__ void lsd_var5_count(const LsdAlignTol& t, const LsdVar5Line& L, int x, float v, int alg[5]) {
  const float d = lsd_fold_deg(t.thDeg, v);
  bool r = d < t.lo;
  if (!r && d <= t.hi) r = lsd_aligned_deg(t.theta, v, t.prec);   // within 1e-3 degrees of the tolerance (rare): the reference's own arithmetic
#pragma unroll
  for (int m = 0; m < 5; m++) alg[m] += (r && x >= L.xa[m] && x <= L.xb[m]) ? 1 : 0;
}
// Round 6, second half: the BOUNDS of the five variants are walked by five lanes, one variant each -- lane m < 5 of the rectangle's
// eight advances variant m's left / right bounds and counts its pixels (tot), and a scan line's span travels to the other lanes as one
// packed word (xa | xb << 16) through a shuffle.  Before, every lane advanced all five variants itself: 70 of the ~ 95 instructions a
// scan line of a steep rectangle cost were that bookkeeping, eight times over.  g0 = first lane of the rectangle's group.  The scan-line
// loop is uniform over the wavefront (a group whose walk is over takes part with empty spans), so the shuffles are executed by all lanes;
// a lane's result: tot[m] = ITS OWN variant's pixel count in every m (the caller reads tot[j] in lane j), alg[m] = its share of the
// aligned pixels of variant m (the caller adds the eight up).  Image widths < 65535 (the spans are packed in 16 bits).
__device__ __forceinline__ void lsd_rect_counts_g8_var5(const RcFrame& f, const LsdScanGeom* gs, const LsdAlignTol& t, int j, int g0, int tot[5], int alg[5]) {
  const int xMax = f.sw - 1;
  int yLo = 1 << 30, yHi = -1;
#pragma unroll
  for (int m = 0; m < 5; m++) {
    const int yA = gs[m].yA, yB = gs[m].yB;
    alg[m] = 0;
    if (yA <= yB) { yLo = min(yLo, yA); yHi = max(yHi, yB); }
  }
  const LsdScanGeom g = gs[min(j, 4)];
  const int gyA = j < 5 ? g.yA : 1, gyB = j < 5 ? g.yB : 0;   // (lanes 5 .. 7: no variant, never on a scan line)
  int left = g.mx, right = g.mx, mytot = 0;
  auto span = [&](int y) -> unsigned {   // this lane's variant on scan line y
    const bool on = y >= gyA && y <= gyB;
    const int a0 = max(left, 0), b0 = min(right, xMax);
    const bool some = on && b0 >= a0;
    mytot += some ? b0 - a0 + 1 : 0;
    left += on ? (y < g.ly ? g.fl : g.sl) : 0;
    right += on ? (y < g.ry ? g.fr : g.sr) : 0;
    return some ? ((unsigned)a0 | ((unsigned)b0 << 16)) : 0xffffu;
  };
  auto spread = [&](unsigned s, LsdVar5Line& L) {
    L.ua = 0xffff; L.ub = 0;
#pragma unroll
    for (int m = 0; m < 5; m++) {
      const unsigned w = (unsigned)__shfl((int)s, g0 + m);
      L.xa[m] = (int)(w & 0xffffu); L.xb[m] = (int)(w >> 16);
      L.ua = min(L.ua, L.xa[m]); L.ub = max(L.ub, L.xb[m]);
    }
  };
#pragma clang loop unroll(disable) vectorize(disable)
  for (int y = yLo; __any(y <= yHi); y += 2) {   // two scan lines per trip to memory
    const bool live = y <= yHi;
    unsigned s0 = 0xffffu, s1 = 0xffffu;
    if (live) {
      s0 = span(y);
      if (y + 1 <= yHi) s1 = span(y + 1);
    }
    LsdVar5Line L0, L1;
    spread(s0, L0);
    spread(s1, L1);
    const float* row0 = rc_row(f, live ? y : 0);
    const float* row1 = rc_row(f, live ? y + 1 : 0);   // (read only when the line y + 1 has a span: it exists then)
    const int x0 = L0.ua + j, x1 = L1.ua + j;
    float v0 = -1024.f, v1 = -1024.f;   // NOTDEF
    if (x0 <= L0.ub) v0 = rc_at(row0, x0);
    if (x1 <= L1.ub) v1 = rc_at(row1, x1);
    lsd_var5_count(t, L0, x0, v0, alg);   // (a lane without a pixel: NOTDEF is aligned with nothing)
    lsd_var5_count(t, L1, x1, v1, alg);
#pragma clang loop unroll(disable) vectorize(disable)
    for (int x = x0 + 8; x <= L0.ub; x += 8) lsd_var5_count(t, L0, x, rc_at(row0, x), alg);
#pragma clang loop unroll(disable) vectorize(disable)
    for (int x = x1 + 8; x <= L1.ub; x += 8) lsd_var5_count(t, L1, x, rc_at(row1, x), alg);
  }
#pragma unroll
  for (int m = 0; m < 5; m++) tot[m] = mytot;
}

// The same walk for the two "finer precision" stages of rect_improve(): their five variants share the rectangle and differ in
// the tolerance only (p / 2^m, prec = p pi, descending), so ONE walk counts all five: the folded difference of a pixel is formed
// once and compared with the five tolerances; a pixel within 1e-3 degrees of any of them takes lsd_aligned()'s own expressions.
struct LsdAlignTol5 {
  double theta, prec[5];
  float thDeg, lo[5], hi[5];
};
__device__ __forceinline__ int lsd_aligned5(const LsdAlignTol5& t, float v, int alg[5]) {   // returns 1 when the pixel is in a margin
  const float d = lsd_fold_deg(t.thDeg, v);
  int inLo = 0, inHi = 0;
#pragma unroll
  for (int m = 0; m < 5; m++) {
    const int a = d < t.lo[m] ? 1 : 0;
    alg[m] += a; inLo += a; inHi += d <= t.hi[m] ? 1 : 0;
  }
  return inLo != inHi ? 1 : 0;
}
__device__ __forceinline__ void lsd_aligned5_exact(const LsdAlignTol5& t, float v, int alg[5]) {   // undo the float verdicts, add the exact ones
  const float d = lsd_fold_deg(t.thDeg, v);
  double n_theta = t.theta - (double)v * kDegToRads;   // lsd_aligned()'s n_theta (v is a defined angle here: |d| < 23)
  if (n_theta < 0) n_theta = -n_theta;
  if (n_theta > k3_2PI) {
    n_theta -= k2PI;
    if (n_theta < 0) n_theta = -n_theta;
  }
#pragma unroll
  for (int m = 0; m < 5; m++) alg[m] += (n_theta <= t.prec[m] ? 1 : 0) - (d < t.lo[m] ? 1 : 0);
}
__device__ __forceinline__ void lsd_rect_counts_g8_prec5(const RcFrame& f, const LsdScanGeom g, const LsdAlignTol5& t, int j, int& totalOut, int alg[5]) {
  const int ly = g.ly, ry = g.ry, fl = g.fl, sl = g.sl, fr = g.fr, sr = g.sr, yB = g.yB, xMax = f.sw - 1;
  int left = g.mx, right = g.mx, total = 0;
#pragma unroll
  for (int m = 0; m < 5; m++) alg[m] = 0;
#pragma clang loop unroll(disable) vectorize(disable)
  for (int y = g.yA; y <= yB; y += 2) {
    const int xa0 = max(left, 0), n0 = max(min(right, xMax) - xa0 + 1, 0);
    left += y < ly ? fl : sl; right += y < ry ? fr : sr;
    const bool two = y + 1 <= yB;
    const int xa1 = max(left, 0), n1 = two ? max(min(right, xMax) - xa1 + 1, 0) : 0;
    left += two ? (y + 1 < ly ? fl : sl) : 0; right += two ? (y + 1 < ry ? fr : sr) : 0;
    total += n0 + n1;
    const float* row0 = rc_row(f, y);
    const float* row1 = rc_row(f, two ? y + 1 : y);
    const int nmax = max(n0, n1);
#pragma clang loop unroll(disable) vectorize(disable)
    for (int k = j; k < nmax; k += 8) {
      float v0 = rc_at(row0, min(xa0 + k, xMax)), v1 = rc_at(row1, min(xa1 + k, xMax));
      v0 = k < n0 ? v0 : -1024.f; v1 = k < n1 ? v1 : -1024.f;
      const int m0 = lsd_aligned5(t, v0, alg), m1 = lsd_aligned5(t, v1, alg);
      if (m0 | m1) {
        if (m0) lsd_aligned5_exact(t, v0, alg);
        if (m1) lsd_aligned5_exact(t, v1, alg);
      }
    }
  }
  totalOut = total;
}

// One variant of rect_improve(): the rectangle R after `m` iterations (1 .. 5) of stage `stage`'s loop body.  The loops modify
// their rectangle whether or not the variant is accepted, so the five variants of a stage follow from the stage's starting
// rectangle alone and can be evaluated side by side.  Returns false when the loop's width gate stops before iteration m.
__device__ __forceinline__ bool lsd_adv_variant(int stage, int m, LsdAdvRect& r) {
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

__device__ __forceinline__ void lsd_store_segment(uint4* slot, const double* rec) {
  float sg[4];
  lsd_segment_of(rec, sg);
  *slot = uint4{__float_as_uint(sg[0]), __float_as_uint(sg[1]), __float_as_uint(sg[2]), __float_as_uint(sg[3])};
}

}  // namespace plh
