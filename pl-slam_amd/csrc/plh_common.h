// Shared host/device helpers of libplslam_hip (MI355X / gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/plslam_hip.h"
#include "plh_shims.h"
#include "plh_xcd.h"

#define PLH_WAVE 64

namespace plh {

char* tls_error();
void set_error(const char* fmt, ...);
// First thing every C-ABI entry point does: make sure the HIP runtime is initialised in this process
// (the library may have been dlopen'ed before the caller's own HIP user, e.g. PyTorch) and drop any stale
// sticky error so that the launch checks below report only our own failures.
plh_status ensure_runtime();

// Dynamic LDS above 64 KiB has to be requested per kernel; gfx950 has 160 KiB per workgroup.  Called by the launchers
// whose LDS footprint scales with a caller-supplied capacity.
template <typename K>
inline plh_status lds_request(K kernel, size_t bytes, const char* who) {
  if (bytes > 160u * 1024u) {
    set_error("%s: %zu bytes of LDS needed (capacity too large for one workgroup)", who, bytes);
    return PLH_ERR_INVALID;
  }
  if (bytes > 64u * 1024u &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
    (void)hipGetLastError();
    set_error("%s: cannot reserve %zu bytes of LDS", who, bytes);
    return PLH_ERR_INVALID;
  }
  return PLH_OK;
}

#define PLH_HIP(call)                                                                         \
  do {                                                                                        \
    hipError_t e__ = (call);                                                                  \
    if (e__ != hipSuccess) {                                                                  \
      plh::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__));    \
      return PLH_ERR_HIP;                                                                     \
    }                                                                                         \
  } while (0)

#define PLH_LAUNCH_CHECK()                                                                    \
  do {                                                                                        \
    hipError_t e__ = hipGetLastError();                                                       \
    if (e__ != hipSuccess) {                                                                  \
      plh::set_error("%s:%d kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return PLH_ERR_HIP;                                                                     \
    }                                                                                         \
  } while (0)

template <typename T>
static inline T align_up(T v, T a) { return (v + a - 1) / a * a; }

// ---- device helpers ----  (lane_id, wballot, PLH_INV_BALLOT, the broadcasts and every other instruction shim: plh_shims.h)
// cvRound: round-half-to-even (v_cvt_i32_f32 with the default RNE mode / v_rndne)
__device__ __forceinline__ int cv_round(float v) { return __float2int_rn(v); }

// number of set bits of a wave mask below this lane (v_mbcnt_lo + v_mbcnt_hi)
__device__ __forceinline__ int mbcnt64(unsigned long long m) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

__device__ __forceinline__ unsigned long long lanemask_lt() {
  return (1ull << lane_id()) - 1ull;
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// cv::fastAtan2 (degrees), float32 Horner evaluation, no FMA (translation unit is built with
// -ffp-contract=off); used by IC_Angle (reference src/ORBextractor.cc:103) and by LSD.
__host__ __device__ __forceinline__ float fast_atan2_deg(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
  // both branches of the reference (ax >= ay ? ay / (ax + eps) : ax / (ay + eps)) are min / (max + eps): one division and
  // one polynomial for every lane, bit-identical to the branched form
  const float ax = fabsf(x), ay = fabsf(y);
  const bool swap = !(ax >= ay);
  const float c = (swap ? ax : ay) / ((swap ? ay : ax) + 2.2204460492503131e-16f);
  const float c2 = c * c;
  float a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  if (swap) a = 90.f - a;
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// sin / cos of a double in [0, 2 pi] for results that are going to be rounded to float.  Three-part pi/2 reduction (the
// first two products are exact, Cody-Waite) and the classic degree-13 / degree-14 minimax kernels on [-pi/4, pi/4]: the
// values carry a relative error below 2^-48, the library's below 2^-52.  `float_round_is_safe` says whether a double is
// further than 2^-40 (relative) from the nearest float rounding boundary, in which case both round to the same float;
// the caller falls back to the library sincos for the one pixel in ~10^4 where that is not so, which keeps the stored
// floats bit-identical to a plain (float)sincos(ad).
__device__ __forceinline__ void sincos_0_2pi(double ad, double& s, double& c) {
  const double kd = (double)(int)(ad * 0.63661977236758138 + 0.5);
  const int k = (int)kd;
  double x = __builtin_fma(-kd, 1.57079632673412561417e+00, ad);   // 33-bit head of pi/2: exact
  x = __builtin_fma(-kd, 6.07710050630396597660e-11, x);           // next 33 bits
  x = __builtin_fma(-kd, 2.02226624879595063154e-21, x);           // tail
  const double z = x * x;
  double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
  ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
  ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
  ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
  const double sp = __builtin_fma(x * z, ps, x);
  double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
  pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
  pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
  pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
  const double cp = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
  const bool swap = (k & 1) != 0;
  const double s0 = swap ? cp : sp, c0 = swap ? sp : cp;
  s = (k & 2) ? -s0 : s0;
  c = ((k + 1) & 2) ? -c0 : c0;
}
// sin / cos of a double in [0, 3 pi] for results that stay doubles (region2rect's rectangle direction): reduction to head + tail
// (x + y, the two-step form of the classic pi/2 reduction) and the classic kernels *with* the tail.  Against the host libm
// (glibc) on 2^24 arguments of the form float-degrees x pi/180 (+ pi): 3.1 % of the values differ, by one unit in the last
// place -- exactly the share by which the device library's sincos differs (tools/ubench/sincos_ulp.hip,
// profiles/r03_sincos_ulp.txt); 60 instructions against the library's 153 + call.
__device__ __forceinline__ void sincos_head_tail(double ad, double& s, double& c) {
  const double kd = (double)(int)(ad * 0.63661977236758138 + 0.5);
  const int k = (int)kd;
  const double t = __builtin_fma(-kd, 1.57079632673412561417e+00, ad);   // 33-bit head of pi/2: exact
  double w = kd * 6.07710050630396597660e-11;                            // next 33 bits
  const double r = t - w;
  w = __builtin_fma(kd, 2.02226624879595063154e-21, -((t - r) - w));     // the rest, and what the subtraction above lost
  const double x = r - w;
  const double y = (r - x) - w;
  const double z = x * x, v = z * x;
  const double rs = 8.33333333332248946124e-03 +
                    z * (-1.98412698298579493134e-04 +
                         z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
  const double sp = x - ((z * (0.5 * y - v * rs) - y) - v * -1.66666666666666324348e-01);
  const double rc =
      z * (4.16666666666666019037e-02 +
           z * (-1.38888888888741095749e-03 +
                z * (2.48015872894767294178e-05 +
                     z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
  const double hz = 0.5 * z, ww = 1.0 - hz;
  const double cp = ww + (((1.0 - ww) - hz) + (z * rc - x * y));
  const bool swap = (k & 1) != 0;
  const double s0 = swap ? cp : sp, c0 = swap ? sp : cp;
  s = (k & 2) ? -s0 : s0;
  c = ((k + 1) & 2) ? -c0 : c0;
}
__device__ __forceinline__ bool float_round_is_safe(double v) {
  unsigned long long u;
  __builtin_memcpy(&u, &v, 8);
  const unsigned t = (unsigned)u & 0x1fffffffu;            // the 29 mantissa bits a float drops; the boundary is 2^28
  return (unsigned)(t - (0x10000000u - 4096u)) >= 8192u;
}

// 256-bit Hamming distance of two 32-byte rows held as 4 x u64.
__device__ __forceinline__ int hamming256(const unsigned long long a[4], const unsigned long long b[4]) {
  return __popcll(a[0] ^ b[0]) + __popcll(a[1] ^ b[1]) + __popcll(a[2] ^ b[2]) + __popcll(a[3] ^ b[3]);
}

// XCD-aware block order: plh_xcd.h (PlhXcdGrid, built by the launcher).  q / d for a launch constant d with m = floor(2^32 / d): the
// estimate (q m) >> 32 is the quotient or one below it for every q < 2^32.
__device__ __forceinline__ unsigned plh_udiv_magic(unsigned q, unsigned d, unsigned m, unsigned& rem) {
  unsigned b = (unsigned)(((unsigned long long)q * m) >> 32);
  unsigned x = q - b * d;
  if (x >= d) { b++; x -= d; }
  rem = x;
  return b;
}
__device__ __forceinline__ bool plh_xcd_decode(const PlhXcdGrid& g, int& x, int& b) {
  const unsigned L = blockIdx.x;
  unsigned r;
  if (g.batch < PLH_XCD_MIN_BATCH) {
    b = (int)plh_udiv_magic(L, (unsigned)g.perFrame, g.mPer, r);
    x = (int)r;
    return b < g.batch;
  }
  b = (int)(plh_udiv_magic(L >> 3, (unsigned)g.perFrame, g.mPer, r) * 8u + (L & 7u));
  x = (int)r;
  return b < g.batch;
}
// the same for a (tiles in x, tiles in y) x frames grid: neighbouring tiles of a stencil share their halo sectors behind one L2
__device__ __forceinline__ bool plh_xcd_decode_tiles(const PlhXcdGrid& g, int& bx, int& by, int& b) {
  int t;
  if (!plh_xcd_decode(g, t, b)) return false;
  unsigned r;
  by = (int)plh_udiv_magic((unsigned)t, (unsigned)g.nx, g.mNx, r);
  bx = (int)r;
  return true;
}

}  // namespace plh
