// Device helpers shared by the line kernels (LSD level-line field encoding, constants).
#pragma once
#include "line_plan.h"
#include "plh_common.h"

namespace plh {

constexpr double kPI = 3.14159265358979323846;
constexpr double kDegToRads = kPI / 180;
constexpr double k3_2PI = 3 * kPI / 2, k2PI = 2 * kPI;

struct Taps7 {
  int k[7];
};
// The Q8 taps of k_blur7_u8 as the dot-product operands its two passes use, built once on the host (launch_blur7):
// h[k][part] = byte weights of bytes 4 part .. 4 part + 3 of a 12-byte row window for output k (tap j of output k sits at
// byte 4 + k + j); v[odd][i] = the (t, t+1) tap pair that multiplies dword i of a column of 16-bit row sums for an even /
// odd output row (odd rows: the pairs slid by one tap).
struct BlurWeights {
  unsigned h[4][3];
  unsigned v[2][4];
};

__device__ __forceinline__ int refl101(int p, int n) {
  if (p < 0) p = -p;
  if (p >= n) p = 2 * n - 2 - p;
  return p;
}

// What region growing keeps in registers for a candidate pixel: its table values (line_plan.h, LsdAngleEntry) and its
// record word `q` (LSD_REC_*: table index, DEF, USED).
//   angf : fastAtan2(gx, -gy) in degrees (float); the reference's double angle is angf * DEG_TO_RADS
//   cs,sn: (float)cos / sin of float(angle) -- the increments region_grow() adds to sumdx / sumdy
// All are exactly the values cv::LineSegmentDetector would compute on the fly.
struct LsdPix {
  float angf, cs, sn;
  unsigned q;
};
__device__ __forceinline__ double pix_angle(const LsdPix& p) { return (double)p.angf * kDegToRads; }
__device__ __forceinline__ double q_modgrad(unsigned q) { return sqrt((double)(int)q / 4.0); }
// gx^2 + gy^2 of a record
__device__ __forceinline__ unsigned lsd_rec_q(uint32_t rec) {
  const int gx = (int)(rec & 1023u) - LSD_GRAD_MAX, gy = (int)((rec >> LSD_ANGLE_PITCH_LOG2) & 1023u) - LSD_GRAD_MAX;
  return (unsigned)(__mul24(gx, gx) + __mul24(gy, gy));
}

// packed int16 pair (used for the Sobel dx,dy image of LBD)
__device__ __forceinline__ uint32_t pack_g(int gx, int gy) { return ((uint32_t)gx & 0xffffu) | ((uint32_t)gy << 16); }
__device__ __forceinline__ int g_x(uint32_t g) { return (int)(short)(g & 0xffffu); }
__device__ __forceinline__ int g_y(uint32_t g) { return (int)(short)(g >> 16); }
__device__ __forceinline__ unsigned g_q(uint32_t g) { const int x = g_x(g), y = g_y(g); return (unsigned)(x * x + y * y); }
__device__ __forceinline__ double g_angle(uint32_t g) {
  return (double)fast_atan2_deg((float)g_x(g), (float)(-g_y(g))) * kDegToRads;
}
__device__ __forceinline__ double g_modgrad(uint32_t g) { return sqrt((double)(int)g_q(g) / 4.0); }

}  // namespace plh
