// plh_selftest: every instruction shim of plh_shims.h, gfx950 form against its portable description, on the device.
// One wavefront; per shim a few thousand operand sets drawn from a counter-based generator plus the corner cases that bit
// before (masks with bits in both halves, every lane as broadcast source, byte selectors of all kinds).  failures[k] counts the
// mismatching lanes of shim k; the host call returns their number (0 = the instructions do what the twins say).
#include "plh_common.h"

namespace plh {

enum { ST_BALLOT = 0, ST_INV_BALLOT, ST_BCAST32, ST_BCAST64, ST_WAVE_MIN, ST_PERM, ST_ALIGNBYTE, ST_UDOT4, ST_UDOT2, ST_PK_MIN, ST_PK_ADD,
       ST_PK_SUB, ST_PK_MAD, ST_SAT255, ST_SBFE1, ST_FRACT, ST_SQRT, ST_DIV, ST_WALK, ST_WALK_DUP, ST_TURNS, ST_COUNT };

__device__ __forceinline__ unsigned long long st_mix(unsigned long long z) {   // splitmix64
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

#if !defined(HIPEMU)
__global__ void __launch_bounds__(64) k_shim_selftest(int* failures, int rounds) {
  const int lane = (int)threadIdx.x;
  int bad[ST_COUNT];
  for (int k = 0; k < ST_COUNT; k++) bad[k] = 0;
  for (int it = 0; it < rounds; it++) {
    const unsigned long long r0 = st_mix((unsigned long long)it * 64 + lane), r1 = st_mix(r0), r2 = st_mix(r1);
    const unsigned a = (unsigned)r0, b = (unsigned)(r0 >> 32), c = (unsigned)r1;
    // uniform values of this round
    const unsigned long long u0 = st_mix(0xABCDull + it), u1 = st_mix(u0);
    // masks: random, and the shapes that lost their upper half as compile-time constants (here they are run-time values)
    unsigned long long mask = u0;
    if ((it & 7) == 1) mask = ~0ull << (it % 64);
    if ((it & 7) == 2) mask = 0xffffffff80000000ull >> (it % 31);
    if ((it & 7) == 3) mask = 0x8000000000000001ull;
    const int src = (int)(u1 % 64);
    // votes and broadcasts
    const bool pred = ((r2 >> 7) & 1ull) != 0ull;
    bad[ST_BALLOT] += hw::wballot(pred) != ref::wballot(pred);
    bad[ST_INV_BALLOT] += hw::inv_ballot(mask) != ref::inv_ballot(mask);
    bad[ST_BCAST32] += hw::bcast_u32(a, src) != ref::bcast_u32(a, src);
    bad[ST_BCAST32] += __float_as_uint(hw::bcast_f32(__uint_as_float(b & 0x7f7fffffu), src)) != __float_as_uint(ref::bcast_f32(__uint_as_float(b & 0x7f7fffffu), src));
    {
      const double d = __hiloint2double((int)(a & 0x7fefffffu), (int)b);
      bad[ST_BCAST64] += __double2hiint(hw::bcast_f64(d, src)) != __double2hiint(ref::bcast_f64(d, src)) ||
                         __double2loint(hw::bcast_f64(d, src)) != __double2loint(ref::bcast_f64(d, src));
    }
    bad[ST_WAVE_MIN] += hw::wave_min_i32((int)a) != ref::wave_min_i32((int)a);
    // byte and half-word arithmetic
    unsigned sel = c;
    if (it & 1) {   // selectors in the documented range: bytes 0..7 and the constant-zero code 0x0c
      sel = 0;
      for (int i = 0; i < 4; i++) { const unsigned q = (c >> (8 * i)) & 15u; sel |= (q < 8u ? q : 0x0cu) << (8 * i); }
    } else {
      sel &= 0x07070707u;
    }
    bad[ST_PERM] += hw::perm(a, b, sel) != ref::perm(a, b, sel);
    bad[ST_ALIGNBYTE] += hw::alignbyte(a, b, c & 3u) != ref::alignbyte(a, b, c & 3u);
    bad[ST_UDOT4] += hw::udot4(a, b, c) != ref::udot4(a, b, c);
    bad[ST_UDOT2] += hw::udot2(a, b, c) != ref::udot2(a, b, c);
    bad[ST_PK_MIN] += hw::pk_min_u16(a, b) != ref::pk_min_u16(a, b);
    bad[ST_PK_ADD] += hw::pk_add16(a, b) != ref::pk_add16(a, b);
    bad[ST_PK_SUB] += hw::pk_sub16(a, b) != ref::pk_sub16(a, b);
    bad[ST_PK_MAD] += hw::pk_twice_plus16(a, b) != ref::pk_twice_plus16(a, b);
    bad[ST_SAT255] += hw::hi_halves_sat255(a, b) != ref::hi_halves_sat255(a, b);
    bad[ST_SBFE1] += hw::sbfe1(a, (int)(b & 31u)) != ref::sbfe1(a, (int)(b & 31u));
    {
      const float x = (float)(a >> 8) * (1.0f / 4096.0f);   // >= 0, as the callers guarantee
      bad[ST_FRACT] += __float_as_uint(hw::fract(x)) != __float_as_uint(ref::fract(x));
      const float y = (float)(b >> 6);
      const float e = hw::sqrt_approx(y), t = ref::sqrt_approx(y);
      bad[ST_SQRT] += !(fabsf(e - t) <= 1.2e-7f * t);   // 1 ulp: the callers correct the estimate with exact compares
    }
    {
      // lsd_atan2_deg's operands: divisor = max(|x|, |y|) + 2.2e-16 with |x|, |y| <= 2^18, dividend = min(|x|, |y|), zero or >= 2^-48
      const float big = ldexpf((float)((a >> 9) | 1u) * (1.0f / 8388608.0f), (int)(b % 67u) - 48);   // [2^-48, 2^18]
      const float fr = (it & 3) == 0 ? 0.f : (float)(c >> 8) * (1.0f / 16777216.0f);
      const float num = big * fr, den = big + 2.2204460492503131e-16f;
      bad[ST_DIV] += __float_as_uint(hw::div_normal(num, den)) != __float_as_uint(ref::div_normal(num, den));
    }
    {
      // sin / cos of an angle in turns: estimates, the caller (lsd_density_screen) budgets 1e-5 of absolute error
      const float t = (float)(a >> 8) * (1.0f / 16777216.0f) * ((it & 1) ? 1.f : 1.5f);
      bad[ST_TURNS] += !(fabsf(hw::sin_turns(t) - ref::sin_turns(t)) <= 4e-6f) || !(fabsf(hw::cos_turns(t) - ref::cos_turns(t)) <= 4e-6f);
    }
    // the walk of lsd_resolve, with and without duplicate pixels
    {
      const unsigned long long P = mask & u1;
      const float cs = __uint_as_float(0x3f000000u | (a & 0x007fffffu)) - 0.75f, sn = __uint_as_float(0x3f000000u | (b & 0x007fffffu)) - 0.75f;
      const uint32_t nidx = (it & 4) ? (uint32_t)(c % 23u) : (uint32_t)(lane + 64 * it);   // few distinct pixels: many duplicates
      for (int dup = 0; dup < 2; dup++) {
        float hx = 1.25f, hy = -0.5f, gx = 1.25f, gy = -0.5f;
        unsigned long long ha, hc, ga, gc;
        hw::lsd_walk(P, dup != 0, nidx, cs, sn, hx, hy, ha, hc);
        ref::lsd_walk(P, dup != 0, nidx, cs, sn, gx, gy, ga, gc);
        // canc may hold lanes outside P in either form (documented): compare what the caller uses
        const bool diff = ha != ga || ((hc ^ gc) & P) != 0ull || __float_as_uint(hx) != __float_as_uint(gx) || __float_as_uint(hy) != __float_as_uint(gy);
        bad[dup ? ST_WALK_DUP : ST_WALK] += diff;
      }
    }
  }
  for (int k = 0; k < ST_COUNT; k++)
    if (bad[k]) atomicAdd(&failures[k], bad[k]);
}
#endif

// A fixed amount of VALU work (independent 32-bit multiply-add chains, no memory): 4096 blocks x 256 threads x `iters` x 64
// operations.  Its duration depends on the box's CU count, clocks and power state only -- what bench.py reports beside a measured
// rate so that numbers from different boxes can be compared (plh_box_probe).
__global__ void __launch_bounds__(256) k_box_probe(unsigned* sink, int iters) {
  unsigned a[8];
#pragma unroll
  for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 2654435761u + blockIdx.x + (unsigned)i;
  const unsigned m = (unsigned)iters | 3u;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int i = 0; i < 8; i++) a[i] = a[i] * m + (unsigned)r;
    }
  }
  unsigned x = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) x ^= a[i];
  if (x == 0x12345678u) *sink = x;
}

}  // namespace plh

using namespace plh;

extern "C" {

plh_status plh_selftest(int device, int* failing_checks, int32_t* per_shim, int per_shim_cap) {
  if (!failing_checks) return PLH_ERR_INVALID;
  *failing_checks = 0;
  for (int k = 0; per_shim && k < per_shim_cap; k++) per_shim[k] = 0;
#if defined(HIPEMU)
  (void)device;   // the emulator build has only the portable forms: nothing to compare
  return PLH_OK;
#else
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  PLH_HIP(hipSetDevice(device));
  int* d = nullptr;
  PLH_HIP(hipMalloc((void**)&d, ST_COUNT * sizeof(int)));
  PLH_HIP(hipMemset(d, 0, ST_COUNT * sizeof(int)));
  hipLaunchKernelGGL(k_shim_selftest, dim3(1), dim3(64), 0, nullptr, d, 4096);
  PLH_LAUNCH_CHECK();
  int h[ST_COUNT];
  PLH_HIP(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  PLH_HIP(hipFree(d));
  static const char* names[ST_COUNT] = {"wballot", "inv_ballot", "bcast_u32/f32", "bcast_f64", "wave_min_i32", "perm", "alignbyte", "udot4",
                                        "udot2", "pk_min_u16", "pk_add16", "pk_sub16", "pk_twice_plus16", "hi_halves_sat255", "sbfe1", "fract",
                                        "sqrt_approx", "div_normal", "lsd_walk", "lsd_walk (duplicates)", "sin_turns / cos_turns"};
  int total = 0;
  for (int k = 0; k < ST_COUNT; k++) {
    if (per_shim && k < per_shim_cap) per_shim[k] = h[k];
    if (h[k]) {
      if (!total) set_error("plh_selftest: shim %s differs from its description on this device (%d lanes)", names[k], h[k]);
      total += h[k];
    }
  }
  *failing_checks = total;
  return PLH_OK;
#endif
}

int plh_selftest_shims(void) { return ST_COUNT; }

// bench.py's box normaliser: the duration of a fixed VALU-only launch (k_box_probe), twice -- the first run finds the GPU in
// whatever power state it is in, the second one at its running clocks.
plh_status plh_box_probe(int device, int iters, float ms[2]) {
  if (!ms || iters <= 0) return PLH_ERR_INVALID;
  if (ensure_runtime() != PLH_OK) return PLH_ERR_NO_DEVICE;
  PLH_HIP(hipSetDevice(device));
  // (a private non-blocking stream: the legacy null stream would serialise with every blocking stream of the process; everything
  // acquired here is released on every path -- ADVICE r5)
  unsigned* d = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipStream_t st = nullptr;
  hipError_t e = hipMalloc((void**)&d, 4);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  for (int k = 0; k < 2 && e == hipSuccess; k++) {
    e = hipEventRecord(e0, st);
    if (e != hipSuccess) break;
    hipLaunchKernelGGL(k_box_probe, dim3(4096), dim3(256), 0, st, d, iters);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipEventRecord(e1, st);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms[k], e0, e1);
  }
  if (st) (void)hipStreamDestroy(st);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (d) (void)hipFree(d);
  if (e != hipSuccess) {
    set_error("plh_box_probe: %s", hipGetErrorString(e));
    return PLH_ERR_HIP;
  }
  return PLH_OK;
}

}  // extern "C"
