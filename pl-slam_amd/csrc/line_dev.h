// Device helpers shared by the line kernels (LSD level-line field encoding, constants).
#pragma once
#include "line_plan.h"
#include "plh_common.h"

namespace plh {

#if defined(HIPEMU)
#define PLH_WAVE_SYNC() hipemu::wave_barrier()
#else
#define PLH_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#endif

constexpr double kPI = 3.14159265358979323846;
constexpr double kDegToRads = kPI / 180;
constexpr double k3_2PI = 3 * kPI / 2, k2PI = 2 * kPI;

struct Taps7 {
  int k[7];
};

__device__ __forceinline__ int refl101(int p, int n) {
  if (p < 0) p = -p;
  if (p >= n) p = 2 * n - 2 - p;
  return p;
}

// LSD level-line field.  gx = DA+BC, gy = DA-BC are stored packed (int16 x 2); the angle
// fastAtan2(gx,-gy)*DEG_TO_RADS and the gradient norm sqrt((gx^2+gy^2)/4.0) are recomputed from the pair
// wherever needed -- exactly the doubles cv::LineSegmentDetector keeps in its angles/modgrad Mats.
__device__ __forceinline__ uint32_t pack_g(int gx, int gy) { return ((uint32_t)gx & 0xffffu) | ((uint32_t)gy << 16); }
__device__ __forceinline__ int g_x(uint32_t g) { return (int)(short)(g & 0xffffu); }
__device__ __forceinline__ int g_y(uint32_t g) { return (int)(short)(g >> 16); }
__device__ __forceinline__ unsigned g_q(uint32_t g) { const int x = g_x(g), y = g_y(g); return (unsigned)(x * x + y * y); }
__device__ __forceinline__ double g_angle(uint32_t g) {
  return (double)fast_atan2_deg((float)g_x(g), (float)(-g_y(g))) * kDegToRads;
}
__device__ __forceinline__ double g_modgrad(uint32_t g) { return sqrt((double)(int)g_q(g) / 4.0); }

}  // namespace plh
