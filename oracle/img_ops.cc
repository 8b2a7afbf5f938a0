// ORACLE (test infrastructure) -- restatement of the OpenCV 3.2-3.4.0 primitives the PL-SLAM
// front end calls.  OpenCV is NOT vendored under /root/reference and is absent from this image;
// these follow the published algorithms (modules/imgproc resize/smooth/filter/deriv/imgwarp/undistort,
// modules/features2d fast.cpp/fast_score.cpp, modules/core mathfuncs_core) as pinned in
// SURVEY.md Appendix B.  Call sites in the reference are cited per routine.
// PARITY UNPINNED (no upstream fixtures, reference unbuildable here) -- see oracle/plo.h.
#include "plo.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

inline int cv_round_f(float v) { return (int)lrintf(v); }   // FE_TONEAREST: round-half-even
inline int cv_round_d(double v) { return (int)lrint(v); }
inline int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }

inline int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) {
    if (p < 0) p = -p;
    else p = 2 * n - 2 - p;
  }
  return p;
}

inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

}  // namespace

extern "C" {

int plo_cv_round_f(float v) { return cv_round_f(v); }

// cv::fastAtan2(y, x): used by IC_Angle (ORBextractor.cc:103) and inside cv::LineSegmentDetector.
float plo_fast_atan2(float y, float x) {
  static const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
  static const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
  static const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
  static const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
  float ax = std::fabs(x), ay = std::fabs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) for CV_8UC1 -- ORBextractor.cc:1120 (pyramid) and
// LSD's internal 0.8x rescale.  Fixed point: 11-bit coefficients, >>4 / >>16 / (+2)>>2 vertical pass.
// inv_scale_* > 0: the fx/fy form (dsize = Size(), scale = 1/fx exactly, as LSD calls it); <= 0: derive from dsize.
void plo_resize_linear_u8_scale(const uint8_t* src, int sw, int sh, size_t sstep, uint8_t* dst, int dw, int dh,
                                size_t dstep, double inv_scale_x, double inv_scale_y) {
  if (inv_scale_x <= 0) inv_scale_x = (double)dw / sw;
  if (inv_scale_y <= 0) inv_scale_y = (double)dh / sh;
  const double scale_x = 1.0 / inv_scale_x;
  const double scale_y = 1.0 / inv_scale_y;
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> ialpha(dw * 2), ibeta(dh * 2);
  int xmax = dw;
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cv_floor_f(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx + 1 >= sw) {
      xmax = std::min(xmax, dx);
      if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    }
    xofs[dx] = sx;
    ialpha[dx * 2] = (short)cv_round_f((1.f - fx) * 2048.f);
    ialpha[dx * 2 + 1] = (short)cv_round_f(fx * 2048.f);
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cv_floor_f(fy);
    fy -= sy;
    yofs[dy] = sy;
    ibeta[dy * 2] = (short)cv_round_f((1.f - fy) * 2048.f);
    ibeta[dy * 2 + 1] = (short)cv_round_f(fy * 2048.f);
  }
  std::vector<int> r0(dw), r1(dw);
  auto hrow = [&](int sy, std::vector<int>& out) {
    sy = std::min(std::max(sy, 0), sh - 1);
    const uint8_t* S = src + (size_t)sy * sstep;
    int dx = 0;
    for (; dx < xmax; dx++) {
      int sx = xofs[dx];
      out[dx] = S[sx] * ialpha[dx * 2] + S[sx + 1] * ialpha[dx * 2 + 1];
    }
    for (; dx < dw; dx++) out[dx] = S[xofs[dx]] * 2048;
  };
  for (int dy = 0; dy < dh; dy++) {
    hrow(yofs[dy], r0);
    hrow(yofs[dy] + 1, r1);
    const int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
    uint8_t* D = dst + (size_t)dy * dstep;
    for (int x = 0; x < dw; x++)
      D[x] = (uint8_t)((((b0 * (r0[x] >> 4)) >> 16) + ((b1 * (r1[x] >> 4)) >> 16) + 2) >> 2);
  }
}

void plo_resize_linear_u8(const uint8_t* src, int sw, int sh, size_t sstep, uint8_t* dst, int dw, int dh,
                          size_t dstep) {
  plo_resize_linear_u8_scale(src, sw, sh, sstep, dst, dw, dh, dstep, 0, 0);
}

// getGaussianKernel(ksize, sigma, CV_32F) then convertTo(CV_32S, 256): the 8-bit "classic" separable path
// (OpenCV <= 3.4.0, createSeparableLinearFilter with bits = 8 per pass).
void plo_gaussian_kernel_q8(int n, double sigma, int32_t* out) {
  std::vector<float> cf(n);
  double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
  double scale2X = -0.5 / (sigmaX * sigmaX);
  double sum = 0;
  for (int i = 0; i < n; i++) {
    double x = i - (n - 1) * 0.5;
    double t = std::exp(scale2X * x * x);
    cf[i] = (float)t;
    sum += cf[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < n; i++) {
    cf[i] = (float)(cf[i] * sum);
    out[i] = cv_round_f(cf[i] * 256.f);
  }
}

// cv::GaussianBlur on CV_8UC1, BORDER_REFLECT_101: ORBextractor.cc:1086 (7x7, sigma 2),
// BinaryDescriptor::computeGaussianPyramid (5x5, sigma 1), LSD (7x7, sigma 0.75).
void plo_gaussian_blur_u8(const uint8_t* src, int w, int h, size_t sstep, uint8_t* dst, size_t dstep, int ksize,
                          double sigma) {
  std::vector<int32_t> k(ksize);
  plo_gaussian_kernel_q8(ksize, sigma, k.data());
  const int r = ksize / 2;
  std::vector<int32_t> rows((size_t)w * h);
  for (int y = 0; y < h; y++) {
    const uint8_t* S = src + (size_t)y * sstep;
    int32_t* R = rows.data() + (size_t)y * w;
    for (int x = 0; x < w; x++) {
      int32_t s = 0;
      for (int t = -r; t <= r; t++) s += k[t + r] * S[reflect101(x + t, w)];
      R[x] = s;
    }
  }
  for (int y = 0; y < h; y++) {
    uint8_t* D = dst + (size_t)y * dstep;
    for (int x = 0; x < w; x++) {
      int32_t s = 0;
      for (int t = -r; t <= r; t++) s += k[t + r] * rows[(size_t)reflect101(y + t, h) * w + x];
      D[x] = sat_u8((s + (1 << 15)) >> 16);
    }
  }
}

static const int kFastRing[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                     {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

// cornerScore<16>(ptr, pixel, threshold) of fast_score.cpp, written out literally.
int plo_fast_score(const uint8_t* ptr, size_t step, int threshold) {
  const int K = 8, N = K * 3 + 1;
  int v = ptr[0];
  short d[N];
  for (int k = 0; k < N; k++) {
    const int* o = kFastRing[k & 15];
    d[k] = (short)(v - ptr[(ptrdiff_t)o[1] * (ptrdiff_t)step + o[0]]);
  }
  int a0 = threshold;
  for (int k = 0; k < 16; k += 2) {
    int a = std::min((int)d[k + 1], (int)d[k + 2]);
    a = std::min(a, (int)d[k + 3]);
    if (a <= a0) continue;
    a = std::min(a, (int)d[k + 4]);
    a = std::min(a, (int)d[k + 5]);
    a = std::min(a, (int)d[k + 6]);
    a = std::min(a, (int)d[k + 7]);
    a = std::min(a, (int)d[k + 8]);
    a0 = std::max(a0, std::min(a, (int)d[k]));
    a0 = std::max(a0, std::min(a, (int)d[k + 9]));
  }
  int b0 = -a0;
  for (int k = 0; k < 16; k += 2) {
    int b = std::max((int)d[k + 1], (int)d[k + 2]);
    b = std::max(b, (int)d[k + 3]);
    b = std::max(b, (int)d[k + 4]);
    b = std::max(b, (int)d[k + 5]);
    if (b >= b0) continue;
    b = std::max(b, (int)d[k + 6]);
    b = std::max(b, (int)d[k + 7]);
    b = std::max(b, (int)d[k + 8]);
    b0 = std::min(b0, std::max(b, (int)d[k]));
    b0 = std::min(b0, std::max(b, (int)d[k + 9]));
  }
  return -b0 - 1;
}

// cv::FAST(img, kps, threshold, nonmax, TYPE_9_16) -- called per cell at ORBextractor.cc:809,814.
// Literal restatement of FAST_t<16> (rolling 3-row score buffer, strict-greater 3x3 NMS).
int plo_fast9_16(const uint8_t* img, int w, int h, size_t step, int threshold, int nonmax, plo_keypoint* out,
                 int cap) {
  const int K = 8, N = 25;
  int pixel[25];
  for (int k = 0; k < 16; k++) pixel[k] = kFastRing[k][1] * (int)step + kFastRing[k][0];
  for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
  threshold = std::min(std::max(threshold, 0), 255);
  uint8_t tab[512];
  for (int i = -255; i <= 255; i++) tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
  std::vector<uint8_t> bufmem((size_t)w * 3, 0);
  uint8_t* buf[3] = {bufmem.data(), bufmem.data() + w, bufmem.data() + 2 * w};
  std::vector<int> cpmem((size_t)(w + 1) * 3, 0);
  int* cpbuf[3] = {cpmem.data() + 1, cpmem.data() + (w + 1) + 1, cpmem.data() + 2 * (w + 1) + 1};
  int n = 0;
  for (int i = 3; i < h - 2; i++) {
    const uint8_t* ptr = img + (size_t)i * step + 3;
    uint8_t* curr = buf[(i - 3) % 3];
    int* cornerpos = cpbuf[(i - 3) % 3];
    memset(curr, 0, w);
    int ncorners = 0;
    if (i < h - 3) {
      for (int j = 3; j < w - 3; j++, ptr++) {
        int v = ptr[0];
        const uint8_t* t = &tab[0] - v + 255;
        int d = t[ptr[pixel[0]]] | t[ptr[pixel[8]]];
        if (d == 0) continue;
        d &= t[ptr[pixel[2]]] | t[ptr[pixel[10]]];
        d &= t[ptr[pixel[4]]] | t[ptr[pixel[12]]];
        d &= t[ptr[pixel[6]]] | t[ptr[pixel[14]]];
        if (d == 0) continue;
        d &= t[ptr[pixel[1]]] | t[ptr[pixel[9]]];
        d &= t[ptr[pixel[3]]] | t[ptr[pixel[11]]];
        d &= t[ptr[pixel[5]]] | t[ptr[pixel[13]]];
        d &= t[ptr[pixel[7]]] | t[ptr[pixel[15]]];
        if (d & 1) {
          int vt = v - threshold, count = 0;
          for (int k = 0; k < N; k++) {
            int x = ptr[pixel[k]];
            if (x < vt) {
              if (++count > K) {
                cornerpos[ncorners++] = j;
                if (nonmax) curr[j] = (uint8_t)plo_fast_score(ptr, step, threshold);
                break;
              }
            } else
              count = 0;
          }
        }
        if (d & 2) {
          int vt = v + threshold, count = 0;
          for (int k = 0; k < N; k++) {
            int x = ptr[pixel[k]];
            if (x > vt) {
              if (++count > K) {
                cornerpos[ncorners++] = j;
                if (nonmax) curr[j] = (uint8_t)plo_fast_score(ptr, step, threshold);
                break;
              }
            } else
              count = 0;
          }
        }
      }
    }
    cornerpos[-1] = ncorners;
    if (i == 3) continue;
    const uint8_t* prev = buf[(i - 4 + 3) % 3];
    const uint8_t* pprev = buf[(i - 5 + 3) % 3];
    cornerpos = cpbuf[(i - 4 + 3) % 3];
    ncorners = cornerpos[-1];
    for (int k = 0; k < ncorners; k++) {
      int j = cornerpos[k];
      int score = prev[j];
      if (!nonmax || (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
                      score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1])) {
        if (n < cap) {
          plo_keypoint kp;
          kp.x = (float)j;
          kp.y = (float)(i - 1);
          kp.size = 7.f;
          kp.angle = -1.f;
          kp.response = (float)score;
          kp.octave = 0;
          kp.class_id = -1;
          out[n] = kp;
        }
        n++;
      }
    }
  }
  return n;
}

// cv::pyrDown(src8u, dst, Size(dw, dh)) (imgproc pyramids.cpp, pyrDown_<FixPtCast<uchar, 8>>; BORDER_DEFAULT = REFLECT_101): the
// 5 x 5 kernel [1 4 6 4 1] x [1 4 6 4 1] centred on source pixel (2x, 2y), integer sums, (v + 128) >> 8.  Called by
// LSDDetector::computeGaussianPyramid (LSDDetector_custom.cpp:56-73) and BinaryDescriptor::computeGaussianPyramid
// (binary_descriptor_custom.cpp:350-370) with dst = Size(cols / ratio, rows / ratio); OpenCV asserts |2 dw - w| <= 2 and
// |2 dh - h| <= 2 (returns -1 here).  OpenCV source absent from /root/reference: PARITY UNPINNED (oracle/plo.h).
int plo_pyr_down_u8(const uint8_t* src, int w, int h, size_t sstep, uint8_t* dst, int dw, int dh, size_t dstep) {
  if (std::abs(dw * 2 - w) > 2 || std::abs(dh * 2 - h) > 2 || dw <= 0 || dh <= 0) return -1;
  static const int k5[5] = {1, 4, 6, 4, 1};
  for (int y = 0; y < dh; y++)
    for (int x = 0; x < dw; x++) {
      int v = 0;
      for (int ky = 0; ky < 5; ky++) {
        const uint8_t* row = src + (size_t)reflect101(2 * y + ky - 2, h) * sstep;
        int r = 0;
        for (int kx = 0; kx < 5; kx++) r += k5[kx] * row[reflect101(2 * x + kx - 2, w)];
        v += k5[ky] * r;
      }
      dst[(size_t)y * dstep + x] = (uint8_t)((v + 128) >> 8);
    }
  return 0;
}

// cv::Sobel(src8u, dst, CV_16S, 1,0,3) and (0,1,3), BORDER_REFLECT_101 (binary_descriptor_custom.cpp:395-396).
void plo_sobel3_s16(const uint8_t* src, int w, int h, size_t sstep, int16_t* dx, int16_t* dy) {
  auto P = [&](int y, int x) -> int { return src[(size_t)reflect101(y, h) * sstep + reflect101(x, w)]; };
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int gx = (P(y - 1, x + 1) - P(y - 1, x - 1)) + 2 * (P(y, x + 1) - P(y, x - 1)) + (P(y + 1, x + 1) - P(y + 1, x - 1));
      int gy = (P(y + 1, x - 1) - P(y - 1, x - 1)) + 2 * (P(y + 1, x) - P(y - 1, x)) + (P(y + 1, x + 1) - P(y - 1, x + 1));
      dx[(size_t)y * w + x] = (int16_t)gx;
      dy[(size_t)y * w + x] = (int16_t)gy;
    }
}

// cv::initUndistortRectifyMap(K, D, I, K, size, CV_32F) (Frame.cc:221).  Pinned: evaluated directly per
// pixel in double (OpenCV accumulates along the row; differs in the last ulp at most).
void plo_undistort_maps(const float K[4], const float D[5], int w, int h, float* mapx, float* mapy) {
  const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
  const double k1 = D[0], k2 = D[1], p1 = D[2], p2 = D[3], k3 = D[4];
  for (int v = 0; v < h; v++)
    for (int u = 0; u < w; u++) {
      double x = (u - cx) / fx, y = (v - cy) / fy;
      double x2 = x * x, y2 = y * y, r2 = x2 + y2, _2xy = 2 * x * y;
      double kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2;
      double xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2);
      double yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy;
      mapx[(size_t)v * w + u] = (float)(fx * xd + cx);
      mapy[(size_t)v * w + u] = (float)(fy * yd + cy);
    }
}

// cv::remap(src, dst, mapx, mapy, INTER_LINEAR) with BORDER_CONSTANT(0) (Frame.cc:222):
// 5-bit fractional coordinates, 15-bit weights.
void plo_remap_linear_u8(const uint8_t* src, int w, int h, size_t sstep, const float* mapx, const float* mapy,
                         uint8_t* dst, size_t dstep) {
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int sx = cv_round_f(mapx[(size_t)y * w + x] * 32.f);
      int sy = cv_round_f(mapy[(size_t)y * w + x] * 32.f);
      int ix = sx >> 5, iy = sy >> 5, ax = sx & 31, ay = sy & 31;
      auto P = [&](int yy, int xx) -> int {
        return (xx >= 0 && xx < w && yy >= 0 && yy < h) ? src[(size_t)yy * sstep + xx] : 0;
      };
      int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32, w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
      int s = w00 * P(iy, ix) + w01 * P(iy, ix + 1) + w10 * P(iy + 1, ix) + w11 * P(iy + 1, ix + 1);
      dst[(size_t)y * dstep + x] = sat_u8((s + (1 << 14)) >> 15);
    }
}

}  // extern "C"
