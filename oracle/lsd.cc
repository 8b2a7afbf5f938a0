// ORACLE (test infrastructure) -- CPU restatement of cv::LineSegmentDetector (LSD_REFINE_STD, default
// parameters; LSD_REFINE_ADV -- rect_improve / rect_nfa / nfa on top of it -- behind the `refine` argument of the _ex entry
// points) as called by cv::line_descriptor::LSDDetector::detect, which LINEextractor::operator() uses
// (reference src/LineExtractor.cpp:39-40; in-tree twin of the wrapper:
// Thirdparty/line_descriptor/src/LSDDetector_custom.cpp:105-215, LSD created at :149).
// The detector itself lives in OpenCV imgproc (lsd.cpp), NOT under /root/reference and absent from this image:
// restated from the published algorithm as pinned in SURVEY.md Appendix B.7.  PARITY UNPINNED (oracle/plo.h).
//
// Pinned definitions:
//   * seed order: bins descending, RASTER order inside a bin (upstream std::sort is unstable there);
//   * internal 8-bit blur = classic Q8 separable path, internal 0.8x resize = INTER_LINEAR fixed point with
//     scale_x = 1/0.8 exactly (cv::resize called with dsize = Size(), fx = fy = 0.8);
//   * cos/sin of float arguments inside region growing = correctly rounded float ((float)cos((double)a));
//   * no FMA contraction.
// Which refine level the reference actually runs: the twin in the tree (LSDDetector_custom.cpp:149) creates the detector with
// the default (LSD_REFINE_STD); src/LineExtractor.cpp:39-40 however calls the SYSTEM opencv_contrib line_descriptor, whose
// LSDDetector::detectImpl -- as published for 3.x -- passes cv::LSD_REFINE_ADV.  Neither binary is in this image: both levels
// are restated, STD is the default (what the in-tree source says), ADV is selectable end to end (plh_line_set_refine).
// The ADV functions are restated WITH the published code's quirks, which decide which rectangles survive: the integer
// divisions of the scan-line steps and the `tailp->p.x` read where a y is meant (rect_nfa), and the first term of nfa()'s
// log1term being (n + 1) rather than log_gamma(n + 1).
#include <quadmath.h>

#include <cfloat>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "plo.h"

extern "C" void plo_resize_linear_u8_scale(const uint8_t* src, int sw, int sh, size_t sstep, uint8_t* dst, int dw, int dh,
                                           size_t dstep, double inv_scale_x, double inv_scale_y);

namespace {

const double NOTDEF = -1024.0;
const double M_3_2_PI = 3 * M_PI / 2, M_2__PI = 2 * M_PI;
const double DEG_TO_RADS = M_PI / 180;
const double kScale = 0.8, kSigmaScale = 0.6, kQuant = 2.0, kAngTh = 22.5, kDensityTh = 0.7;
const int kNBins = 1024;

struct RegionPoint {
  int x, y;
  double angle, modgrad;
};

struct Rect {
  double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p;
};

inline double distSq(double x1, double y1, double x2, double y2) { return (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1); }
inline double dist(double x1, double y1, double x2, double y2) { return std::sqrt(distSq(x1, y1, x2, y2)); }
inline double angle_diff_signed(double a, double b) {
  double diff = a - b;
  while (diff <= -M_PI) diff += M_2__PI;
  while (diff > M_PI) diff -= M_2__PI;
  return diff;
}
inline double angle_diff(double a, double b) { return std::fabs(angle_diff_signed(a, b)); }

// counters for tools/ (how much sequential work a frame carries); not part of any result
static thread_local long g_stats[8];

struct Lsd {
  int w = 0, h = 0;
  std::vector<uint8_t> img;
  std::vector<double> angles, modgrad;
  std::vector<uint8_t> used;
  std::vector<int> ordered;   // pixel index y*w+x, seed order

  // ll_angle
  void ll_angle(double threshold) {
    angles.assign((size_t)w * h, NOTDEF);
    modgrad.assign((size_t)w * h, 0.0);
    double max_grad = -1;
    for (int y = 0; y < h - 1; ++y)
      for (int x = 0; x < w - 1; ++x) {
        const uint8_t* r0 = &img[(size_t)y * w];
        const uint8_t* r1 = &img[(size_t)(y + 1) * w];
        int DA = r1[x + 1] - r0[x];
        int BC = r0[x + 1] - r1[x];
        int gx = DA + BC, gy = DA - BC;
        double norm = std::sqrt((gx * gx + gy * gy) / 4.0);
        modgrad[(size_t)y * w + x] = norm;
        if (norm <= threshold) {
          angles[(size_t)y * w + x] = NOTDEF;
        } else {
          angles[(size_t)y * w + x] = plo_fast_atan2((float)gx, (float)-gy) * DEG_TO_RADS;
          if (norm > max_grad) max_grad = norm;
        }
      }
    const double bin_coef = (max_grad > 0) ? double(kNBins - 1) / max_grad : 0;
    // counting sort: bins descending, raster order inside a bin (PINNED)
    std::vector<int> count(kNBins + 1, 0);
    const size_t npts = (size_t)(w - 1) * (h - 1);
    std::vector<int> bin(npts);
    size_t k = 0;
    for (int y = 0; y < h - 1; ++y)
      for (int x = 0; x < w - 1; ++x) {
        int i = int(modgrad[(size_t)y * w + x] * bin_coef);
        bin[k++] = i;
        count[i]++;
      }
    std::vector<int> start(kNBins + 1, 0);
    int acc = 0;
    for (int b = kNBins - 1; b >= 0; --b) { start[b] = acc; acc += count[b]; }
    ordered.assign(npts, 0);
    k = 0;
    for (int y = 0; y < h - 1; ++y)
      for (int x = 0; x < w - 1; ++x) ordered[start[bin[k++]]++] = y * w + x;
  }

  bool isAligned(int x, int y, double theta, double prec) const {
    if (x < 0 || y < 0 || x >= w || y >= h) return false;
    const double a = angles[(size_t)y * w + x];
    if (a == NOTDEF) return false;
    double n_theta = theta - a;
    if (n_theta < 0) n_theta = -n_theta;
    if (n_theta > M_3_2_PI) {
      n_theta -= M_2__PI;
      if (n_theta < 0) n_theta = -n_theta;
    }
    return n_theta <= prec;
  }

  void region_grow(int sx, int sy, std::vector<RegionPoint>& reg, double& reg_angle, double prec) {
    reg.clear();
    g_stats[0]++;
    RegionPoint seed;
    seed.x = sx; seed.y = sy;
    reg_angle = angles[(size_t)sy * w + sx];
    seed.angle = reg_angle;
    seed.modgrad = modgrad[(size_t)sy * w + sx];
    reg.push_back(seed);
    float sumdx = float(std::cos(reg_angle));
    float sumdy = float(std::sin(reg_angle));
    used[(size_t)sy * w + sx] = 1;
    for (size_t i = 0; i < reg.size(); i++) {
      const int px = reg[i].x, py = reg[i].y;
      int xx_min = std::max(px - 1, 0), xx_max = std::min(px + 1, w - 1);
      int yy_min = std::max(py - 1, 0), yy_max = std::min(py + 1, h - 1);
      for (int yy = yy_min; yy <= yy_max; ++yy)
        for (int xx = xx_min; xx <= xx_max; ++xx) {
          uint8_t& is_used = used[(size_t)yy * w + xx];
          if (is_used != 1 && isAligned(xx, yy, reg_angle, prec)) {
            const double angle = angles[(size_t)yy * w + xx];
            is_used = 1;
            RegionPoint rp;
            rp.x = xx; rp.y = yy;
            rp.modgrad = modgrad[(size_t)yy * w + xx];
            rp.angle = angle;
            reg.push_back(rp);
            sumdx += (float)std::cos((double)(float)angle);   // cos(float(angle)) -> float (PINNED: correctly rounded)
            sumdy += (float)std::sin((double)(float)angle);
            reg_angle = plo_fast_atan2(sumdy, sumdx) * DEG_TO_RADS;
          }
        }
    }
    g_stats[1] += (long)reg.size();
  }

  double get_theta(const std::vector<RegionPoint>& reg, double x, double y, double reg_angle, double prec) const {
    double Ixx = 0.0, Iyy = 0.0, Ixy = 0.0;
    for (size_t i = 0; i < reg.size(); ++i) {
      const double regx = reg[i].x, regy = reg[i].y, weight = reg[i].modgrad;
      double dx = regx - x, dy = regy - y;
      Ixx += dy * dy * weight;
      Iyy += dx * dx * weight;
      Ixy -= dx * dy * weight;
    }
    double lambda = 0.5 * (Ixx + Iyy - std::sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
    double theta = (std::fabs(Ixx) > std::fabs(Iyy)) ? double(plo_fast_atan2(float(lambda - Ixx), float(Ixy)))
                                                      : double(plo_fast_atan2(float(Ixy), float(lambda - Iyy)));
    theta *= DEG_TO_RADS;
    if (angle_diff(theta, reg_angle) > prec) theta += M_PI;
    return theta;
  }

  void region2rect(const std::vector<RegionPoint>& reg, double reg_angle, double prec, double p, Rect& rec) const {
    g_stats[2]++; g_stats[3] += (long)reg.size();
    double x = 0, y = 0, sum = 0;
    for (size_t i = 0; i < reg.size(); ++i) {
      const double weight = reg[i].modgrad;
      x += double(reg[i].x) * weight;
      y += double(reg[i].y) * weight;
      sum += weight;
    }
    x /= sum;
    y /= sum;
    double theta = get_theta(reg, x, y, reg_angle, prec);
    // PINNED (extended in round 4): cos / sin of the rectangle angle are the CORRECTLY ROUNDED doubles -- what the C
    // library of the reference's build returns differs from build to build in the last bit (glibc >= 2.28 is not correctly
    // rounded).  113-bit libquadmath, rounded to double (tools/sincos_cr_check.c counts the arguments whose 113-bit value sits
    // close enough to a tie for that second rounding to matter: none).
    __float128 qs, qc;
    sincosq((__float128)theta, &qs, &qc);
    const double dx = (double)qc, dy = (double)qs;
    double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
    for (size_t i = 0; i < reg.size(); ++i) {
      double regdx = double(reg[i].x) - x, regdy = double(reg[i].y) - y;
      double l = regdx * dx + regdy * dy;
      double wv = -regdx * dy + regdy * dx;
      if (l > l_max) l_max = l; else if (l < l_min) l_min = l;
      if (wv > w_max) w_max = wv; else if (wv < w_min) w_min = wv;
    }
    rec.x1 = x + l_min * dx; rec.y1 = y + l_min * dy;
    rec.x2 = x + l_max * dx; rec.y2 = y + l_max * dy;
    rec.width = w_max - w_min;
    rec.x = x; rec.y = y; rec.theta = theta; rec.dx = dx; rec.dy = dy; rec.prec = prec; rec.p = p;
    if (rec.width < 1.0) rec.width = 1.0;
  }

  bool reduce_region_radius(std::vector<RegionPoint>& reg, double reg_angle, double prec, double p, Rect& rec,
                            double density, double density_th) {
    double xc = double(reg[0].x), yc = double(reg[0].y);
    double radSq1 = distSq(xc, yc, rec.x1, rec.y1), radSq2 = distSq(xc, yc, rec.x2, rec.y2);
    double radSq = radSq1 > radSq2 ? radSq1 : radSq2;
    while (density < density_th) {
      g_stats[5]++; g_stats[6] += (long)reg.size();
      radSq *= 0.75 * 0.75;
      for (size_t i = 0; i < reg.size(); ++i) {
        if (distSq(xc, yc, double(reg[i].x), double(reg[i].y)) > radSq) {
          used[(size_t)reg[i].y * w + reg[i].x] = 0;
          std::swap(reg[i], reg[reg.size() - 1]);
          reg.pop_back();
          --i;
        }
      }
      if (reg.size() < 2) return false;
      region2rect(reg, reg_angle, prec, p, rec);
      density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    }
    return true;
  }

  bool refine(std::vector<RegionPoint>& reg, double reg_angle, double prec, double p, Rect& rec, double density_th) {
    double density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    if (density >= density_th) return true;
    g_stats[4]++;
    double xc = double(reg[0].x), yc = double(reg[0].y);
    const double ang_c = reg[0].angle;
    double sum = 0, s_sum = 0;
    int n = 0;
    for (size_t i = 0; i < reg.size(); ++i) {
      used[(size_t)reg[i].y * w + reg[i].x] = 0;
      if (dist(xc, yc, reg[i].x, reg[i].y) < rec.width) {
        const double angle = reg[i].angle;
        double ang_d = angle_diff_signed(angle, ang_c);
        sum += ang_d;
        s_sum += ang_d * ang_d;
        ++n;
      }
    }
    double mean_angle = sum / double(n);
    double tau = 2.0 * std::sqrt((s_sum - 2.0 * mean_angle * sum) / double(n) + mean_angle * mean_angle);
    region_grow(reg[0].x, reg[0].y, reg, reg_angle, tau);
    if (reg.size() < 2) return false;
    region2rect(reg, reg_angle, prec, p, rec);
    density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    if (density < density_th) return reduce_region_radius(reg, reg_angle, prec, p, rec, density, density_th);
    return true;
  }

  // ---- LSD_REFINE_ADV -------------------------------------------------------------------------------------------
  double LOG_NT = 0;

  static double log_gamma_windschitl(double x) {
    return 0.918938533204673 + (x - 0.5) * std::log(x) - x + 0.5 * x * std::log(x * std::sinh(1 / x) + 1 / (810.0 * std::pow(x, 6.0)));
  }
  static double log_gamma_lanczos(double x) {
    static const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511};
    double a = (x + 0.5) * std::log(x + 5.5) - (x + 5.5);
    double b = 0;
    for (int n = 0; n < 7; ++n) {
      a -= std::log(x + double(n));
      b += q[n] * std::pow(x, double(n));
    }
    return a + std::log(b);
  }
  static double log_gamma(double x) { return x > 15.0 ? log_gamma_windschitl(x) : log_gamma_lanczos(x); }
  static bool double_equal(double a, double b) {
    if (a == b) return true;
    const double abs_diff = std::fabs(a - b), aa = std::fabs(a), bb = std::fabs(b);
    double abs_max = aa > bb ? aa : bb;
    if (abs_max < DBL_MIN) abs_max = DBL_MIN;
    return (abs_diff / abs_max) <= (100.0 * DBL_EPSILON);   // RELATIVE_ERROR_FACTOR
  }

  // nfa(n, k, p) = -log10(NT * binomial tail B(n, k, p))
  double nfa(int n, int k, double p) const {
    if (n == 0 || k == 0) return -LOG_NT;
    if (n == k) return -LOG_NT - double(n) * std::log10(p);
    const double p_term = p / (1 - p);
    const double log1term = (double(n) + 1) - log_gamma(double(k) + 1) - log_gamma(double(n - k) + 1) + double(k) * std::log(p) +
                            (double(n - k)) * std::log(1.0 - p);
    double term = std::exp(log1term);
    if (double_equal(term, 0)) {
      if (k > n * p) return -log1term / M_LN10 - LOG_NT;
      return -LOG_NT;
    }
    double bin_tail = term;
    const double tolerance = 0.1;
    for (int i = k + 1; i <= n; ++i) {
      const double bin_term = double(n - i + 1) / double(i);
      const double mult_term = bin_term * p_term;
      term *= mult_term;
      bin_tail += term;
      if (bin_term < 1) {
        const double err = term * ((1 - std::pow(mult_term, double(n - i + 1))) / (1 - mult_term) - 1);
        if (err < tolerance * std::fabs(-std::log10(bin_tail) - LOG_NT) * bin_tail) break;
      }
    }
    return -std::log10(bin_tail) - LOG_NT;
  }

  struct Edge { int x, y; bool taken; };

  // points of the rectangle, scan line by scan line, and how many of them are aligned with it
  double rect_nfa(const Rect& rec) const {
    int total_pts = 0, alg_pts = 0;
    const double half_width = rec.width / 2.0;
    const double dyhw = rec.dy * half_width, dxhw = rec.dx * half_width;
    Edge o[4];
    o[0] = {int(rec.x1 - dyhw), int(rec.y1 + dxhw), false};
    o[1] = {int(rec.x2 - dyhw), int(rec.y2 + dxhw), false};
    o[2] = {int(rec.x2 + dyhw), int(rec.y2 - dxhw), false};
    o[3] = {int(rec.x1 + dyhw), int(rec.y1 - dxhw), false};
    std::sort(o, o + 4, [](const Edge& a, const Edge& b) { return a.x == b.x ? a.y < b.y : a.x < b.x; });   // AsmallerB_XoverY
    Edge *min_y = &o[0], *max_y = &o[0];
    for (unsigned i = 1; i < 4; ++i) {
      if (min_y->y > o[i].y) min_y = &o[i];
      if (max_y->y < o[i].y) max_y = &o[i];
    }
    min_y->taken = true;
    Edge* leftmost = nullptr;
    for (unsigned i = 0; i < 4; ++i)
      if (!o[i].taken) {
        if (!leftmost) leftmost = &o[i];
        else if (leftmost->x > o[i].x) leftmost = &o[i];
      }
    leftmost->taken = true;
    Edge* rightmost = nullptr;
    for (unsigned i = 0; i < 4; ++i)
      if (!o[i].taken) {
        if (!rightmost) rightmost = &o[i];
        else if (rightmost->x < o[i].x) rightmost = &o[i];
      }
    rightmost->taken = true;
    Edge* tailp = nullptr;
    for (unsigned i = 0; i < 4; ++i)
      if (!o[i].taken) {
        if (!tailp) tailp = &o[i];
        else if (tailp->x > o[i].x) tailp = &o[i];
      }
    tailp->taken = true;
    // (integer divisions, and tailp->x where the published code means a y: as published)
    const double flstep = (min_y->y != leftmost->y) ? (min_y->x - leftmost->x) / (min_y->y - leftmost->y) : 0;
    const double slstep = (leftmost->y != tailp->x) ? (leftmost->x - tailp->x) / (leftmost->y - tailp->x) : 0;
    const double frstep = (min_y->y != rightmost->y) ? (min_y->x - rightmost->x) / (min_y->y - rightmost->y) : 0;
    const double srstep = (rightmost->y != tailp->x) ? (rightmost->x - tailp->x) / (rightmost->y - tailp->x) : 0;
    double lstep = flstep, rstep = frstep;
    double left_x = min_y->x, right_x = min_y->x;
    const int min_iter = min_y->y, max_iter = max_y->y;
    for (int y = min_iter; y <= max_iter; ++y) {
      // as published (OpenCV 3.x lsd.cpp rect_nfa): a scan line outside the image is skipped BEFORE the step update, so the
      // spans do not advance on it (rounds 2-3 restated this with the update applied to skipped rows as well)
      if (y < 0 || y >= h) continue;
      for (int x = int(left_x); x <= int(right_x); ++x) {
        if (x < 0 || x >= w) continue;
        ++total_pts;
        if (isAligned(x, y, rec.theta, rec.prec)) ++alg_pts;
      }
      if (y >= leftmost->y) lstep = slstep;
      if (y >= rightmost->y) rstep = srstep;
      left_x += lstep;
      right_x += rstep;
    }
    g_stats[7]++;
    return nfa(total_pts, alg_pts, rec.p);
  }

  double rect_improve(Rect& rec, double LOG_EPS) const {
    const double delta = 0.5, delta_2 = delta / 2.0;
    double log_nfa = rect_nfa(rec);
    if (log_nfa > LOG_EPS) return log_nfa;
    Rect r = rec;                                   // finer precision
    for (int n = 0; n < 5; ++n) {
      r.p /= 2;
      r.prec = r.p * M_PI;
      const double v = rect_nfa(r);
      if (v > log_nfa) { log_nfa = v; rec = r; }
    }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = rec;                                        // reduce width
    for (unsigned n = 0; n < 5; ++n)
      if ((r.width - delta) >= 0.5) {
        r.width -= delta;
        const double v = rect_nfa(r);
        if (v > log_nfa) { rec = r; log_nfa = v; }
      }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = rec;                                        // reduce one side
    for (unsigned n = 0; n < 5; ++n)
      if ((r.width - delta) >= 0.5) {
        r.x1 += -r.dy * delta_2; r.y1 += r.dx * delta_2;
        r.x2 += -r.dy * delta_2; r.y2 += r.dx * delta_2;
        r.width -= delta;
        const double v = rect_nfa(r);
        if (v > log_nfa) { rec = r; log_nfa = v; }
      }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = rec;                                        // reduce the other side
    for (unsigned n = 0; n < 5; ++n)
      if ((r.width - delta) >= 0.5) {
        r.x1 -= -r.dy * delta_2; r.y1 -= r.dx * delta_2;
        r.x2 -= -r.dy * delta_2; r.y2 -= r.dx * delta_2;
        r.width -= delta;
        const double v = rect_nfa(r);
        if (v > log_nfa) { rec = r; log_nfa = v; }
      }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = rec;                                        // finer precision again
    for (unsigned n = 0; n < 5; ++n)
      if ((r.width - delta) >= 0.5) {
        r.p /= 2;
        r.prec = r.p * M_PI;
        const double v = rect_nfa(r);
        if (v > log_nfa) { rec = r; log_nfa = v; }
      }
    return log_nfa;
  }

  // flsd
  int run(const uint8_t* src, int sw, int sh, size_t sstep, float* segs, int cap, int refine_level = 0) {
    const double prec = M_PI * kAngTh / 180;
    const double p = kAngTh / 180;
    const double rho = kQuant / std::sin(prec);
    const double sigma = kSigmaScale / kScale;
    const double sprec = 3;
    const unsigned hh = (unsigned)(std::ceil(sigma * std::sqrt(2 * sprec * std::log(10.0))));
    const int ksize = 1 + 2 * (int)hh;
    std::vector<uint8_t> g((size_t)sw * sh);
    plo_gaussian_blur_u8(src, sw, sh, sstep, g.data(), sw, ksize, sigma);
    w = (int)lrint(sw * kScale);
    h = (int)lrint(sh * kScale);
    img.assign((size_t)w * h, 0);
    plo_resize_linear_u8_scale(g.data(), sw, sh, sw, img.data(), w, h, w, kScale, kScale);
    ll_angle(rho);
    LOG_NT = 5 * (std::log10(double(w)) + std::log10(double(h))) / 2 + std::log10(11.0);
    const size_t min_reg_size = size_t(-LOG_NT / std::log10(p));
    const double LOG_EPS = 0;
    used.assign((size_t)w * h, 0);
    std::vector<RegionPoint> reg;
    int n = 0;
    for (size_t i = 0; i < ordered.size(); ++i) {
      const int px = ordered[i] % w, py = ordered[i] / w;
      if (used[ordered[i]] == 0 && angles[ordered[i]] != NOTDEF) {
        double reg_angle;
        region_grow(px, py, reg, reg_angle, prec);
        if (reg.size() < min_reg_size) continue;
        Rect rec;
        region2rect(reg, reg_angle, prec, p, rec);
        if (!refine(reg, reg_angle, prec, p, rec, kDensityTh)) continue;
        if (refine_level >= 1) {   // LSD_REFINE_ADV: the rectangle has to be meaningful (NFA), after up to five kinds of adjustment
          const double log_nfa = rect_improve(rec, LOG_EPS);
          if (log_nfa <= LOG_EPS) continue;
        }
        rec.x1 += 0.5; rec.y1 += 0.5; rec.x2 += 0.5; rec.y2 += 0.5;
        rec.x1 /= kScale; rec.y1 /= kScale; rec.x2 /= kScale; rec.y2 /= kScale; rec.width /= kScale;
        if (n < cap) {
          segs[n * 4 + 0] = float(rec.x1); segs[n * 4 + 1] = float(rec.y1);
          segs[n * 4 + 2] = float(rec.x2); segs[n * 4 + 3] = float(rec.y2);
        }
        n++;
      }
    }
    return n;
  }
};

}  // namespace

extern "C" {

void plo_lsd_stats(long* out8, int reset) {
  for (int i = 0; i < 8; i++) { out8[i] = g_stats[i]; if (reset) g_stats[i] = 0; }
}

int plo_lsd_detect_ex(const uint8_t* img, int w, int h, size_t step, float* segs_xyxy, int cap, int refine) {
  if (w < 8 || h < 8) return 0;
  Lsd lsd;
  return lsd.run(img, w, h, step, segs_xyxy, cap, refine);
}
int plo_lsd_detect(const uint8_t* img, int w, int h, size_t step, float* segs_xyxy, int cap) {
  return plo_lsd_detect_ex(img, w, h, step, segs_xyxy, cap, 0);
}

// known-answer taps of the ADV functions: nfa(n, k, p) for an image of w x h (LOG_NT), log_gamma
double plo_lsd_nfa(int w, int h, int n, int k, double p) {
  Lsd lsd;
  lsd.LOG_NT = 5 * (std::log10(double(w)) + std::log10(double(h))) / 2 + std::log10(11.0);
  return lsd.nfa(n, k, p);
}
double plo_lsd_log_gamma(double x) { return Lsd::log_gamma(x); }

// taps for stage-wise parity: the scaled 8-bit image, angle (double, NOTDEF = -1024) and seed order
int plo_lsd_stage_taps(const uint8_t* img, int w, int h, size_t step, uint8_t* scaled, double* angles, double* modgrad,
                       int32_t* ordered, int* sw_out, int* sh_out) {
  Lsd lsd;
  std::vector<float> segs(4);
  const double prec = M_PI * kAngTh / 180;
  const double rho = kQuant / std::sin(prec);
  const double sigma = kSigmaScale / kScale;
  std::vector<uint8_t> g((size_t)w * h);
  plo_gaussian_blur_u8(img, w, h, step, g.data(), w, 7, sigma);
  lsd.w = (int)lrint(w * kScale);
  lsd.h = (int)lrint(h * kScale);
  lsd.img.assign((size_t)lsd.w * lsd.h, 0);
  plo_resize_linear_u8_scale(g.data(), w, h, w, lsd.img.data(), lsd.w, lsd.h, lsd.w, kScale, kScale);
  lsd.ll_angle(rho);
  *sw_out = lsd.w; *sh_out = lsd.h;
  if (scaled) memcpy(scaled, lsd.img.data(), lsd.img.size());
  if (angles) memcpy(angles, lsd.angles.data(), lsd.angles.size() * 8);
  if (modgrad) memcpy(modgrad, lsd.modgrad.data(), lsd.modgrad.size() * 8);
  if (ordered) memcpy(ordered, lsd.ordered.data(), lsd.ordered.size() * 4);
  return (int)lsd.ordered.size();
}

}  // extern "C"
