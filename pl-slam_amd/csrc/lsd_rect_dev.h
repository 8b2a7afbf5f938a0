// The rectangle arithmetic of cv::LineSegmentDetector (region2rect / get_theta, and LSD_REFINE_ADV's nfa) as per-value device
// functions shared by the two kernels that evaluate it: k_lsd_grow* (lsd_grow.hip: one wavefront per frame, only where a
// decision of region growing needs the exact rectangle) and k_lsd_rects (lsd_rects.hip: one lane per kept region, the
// rectangles that become segments).  Same expressions, same order of operations, no contraction (-ffp-contract=off): the two
// kernels produce the same doubles, which are the oracle's (oracle/lsd.cc region2rect, get_theta, nfa).
#pragma once
#include "line_dev.h"

namespace plh {

struct alignas(16) D2 {
  double x, y;
};

__device__ __forceinline__ double angle_diff_signed(double a, double b) {
  double diff = a - b;
  while (diff <= -kPI) diff += k2PI;
  while (diff > kPI) diff -= k2PI;
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

// cos / sin of the rectangle angle (theta in [0, 3 pi)), x = cos, y = sin
__device__ __forceinline__ D2 lsd_sincos_inl(double t) {
  D2 r;
#if defined(PLH_LIB_SINCOS)
  sincos(t, &r.y, &r.x);
#else
  sincos_head_tail(t, r.y, r.x);
#endif
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

__device__ __attribute__((noinline)) static double lsd_nfa(int n, int k, double p, double logNT) {
  if (n == 0 || k == 0) return -logNT;
  if (n == k) return -logNT - double(n) * log10(p);
  const double p_term = p / (1 - p);
  // (n + 1) where the original algorithm has log_gamma(n + 1): as published (oracle/lsd.cc)
  const double log1term = (double(n) + 1) - lsd_log_gamma(double(k) + 1) - lsd_log_gamma(double(n - k) + 1) + double(k) * log(p) +
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

}  // namespace plh
