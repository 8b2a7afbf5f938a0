// Instruction shims: every place where the kernels use a gfx950 instruction (or builtin) that plain C++ cannot say.
//
// ONE definition each, here: namespace plh::hw holds the gfx950 form, namespace plh::ref the portable twin that says what the
// instruction computes.  The product build binds the names to hw, the CPU emulator build (tests/hipemu, -DHIPEMU: test
// infrastructure, never the product) to ref -- the only `#if defined(HIPEMU)` a kernel source needs is this one switch.
// Because the twins are compiled into the product build as well, k_shim_selftest (selftest.hip, plh_selftest) runs both forms
// of every shim on the GPU over all 64 lanes -- masks with bits in both halves, every lane as broadcast source -- and
// compares them: a mismatch between an instruction and its description (the constant-mask inverse_ballot of round 2 that lost
// lanes 32..63 and that the emulator could not see) shows up in seconds, not in a parity test.
#pragma once
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdint>

namespace plh {

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// ---------------------------------------------------------------------------------------------------------------------
namespace ref {

// wave vote as the mask of lanes whose predicate holds
__device__ __forceinline__ unsigned long long wballot(bool p) { return __ballot(p); }
// a wave mask used as a per-lane predicate
__device__ __forceinline__ bool inv_ballot(unsigned long long m) { return ((m >> lane_id()) & 1ull) != 0ull; }
// value of lane l in every lane (l uniform)
__device__ __forceinline__ unsigned bcast_u32(unsigned v, int l) { return __shfl(v, l); }
__device__ __forceinline__ float bcast_f32(float v, int l) { return __shfl(v, l); }
__device__ __forceinline__ double bcast_f64(double v, int l) { return __shfl(v, l); }
// minimum over the wavefront, uniform result
__device__ __forceinline__ int wave_min_i32(int v) {
  for (int s = 32; s >= 1; s >>= 1) v = min(v, __shfl_xor(v, s));
  return v;
}
// byte i of the result = byte sel[i] of {hi:lo}; selector 0x0c -> 0                                   (v_perm_b32)
__device__ __forceinline__ unsigned perm(unsigned hi, unsigned lo, unsigned sel) {
  const unsigned long long v = ((unsigned long long)hi << 32) | lo;
  unsigned r = 0;
  for (int i = 0; i < 4; i++) {
    const unsigned c = (sel >> (8 * i)) & 255u;
    r |= (c <= 7u ? (unsigned)((v >> (8 * c)) & 255u) : 0u) << (8 * i);
  }
  return r;
}
// bytes sh .. sh + 3 of {hi:lo}, sh in 0..3                                                           (v_alignbyte_b32)
__device__ __forceinline__ unsigned alignbyte(unsigned hi, unsigned lo, unsigned sh) {
  return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (8 * sh));
}
// c + sum of the products of the four byte pairs / the two half-word pairs                            (v_dot4_u32_u8, v_dot2_u32_u16)
__device__ __forceinline__ unsigned udot4(unsigned a, unsigned b, unsigned c) {
  for (int i = 0; i < 4; i++) c += ((a >> (8 * i)) & 255u) * ((b >> (8 * i)) & 255u);
  return c;
}
__device__ __forceinline__ unsigned udot2(unsigned a, unsigned b, unsigned c) { return (a & 0xffffu) * (b & 0xffffu) + (a >> 16) * (b >> 16) + c; }
// packed 16-bit lanes                                                                                 (v_pk_min_u16, v_pk_add_u16, v_pk_sub_u16, v_pk_mad_u16)
__device__ __forceinline__ unsigned pk_min_u16(unsigned a, unsigned b) { return min(a & 0xffffu, b & 0xffffu) | (min(a >> 16, b >> 16) << 16); }
__device__ __forceinline__ unsigned pk_add16(unsigned a, unsigned b) { return ((a + b) & 0xffffu) | (((a >> 16) + (b >> 16)) << 16); }
__device__ __forceinline__ unsigned pk_sub16(unsigned a, unsigned b) { return ((a - b) & 0xffffu) | (((a >> 16) - (b >> 16)) << 16); }
__device__ __forceinline__ unsigned pk_twice_plus16(unsigned a, unsigned b) { return pk_add16(pk_add16(a, a), b); }
// bytes 2, 3 of lo and of hi as two halves (a >> 16 of two accumulators), both saturated at 255
__device__ __forceinline__ unsigned hi_halves_sat255(unsigned hi, unsigned lo) { return min(lo >> 16, 255u) | (min(hi >> 16, 255u) << 16); }
// bit k of v as 0 / -1                                                                                (v_bfe_i32, width 1)
__device__ __forceinline__ int sbfe1(unsigned v, int k) { return -(int)((v >> k) & 1u); }
// x - floor(x), exact for x >= 0                                                                      (v_fract_f32)
__device__ __forceinline__ float fract(float x) { return x - floorf(x); }
// square root where an estimate is all that is needed (the caller corrects it with exact compares)  (v_sqrt_f32, 1 ulp)
__device__ __forceinline__ float sqrt_approx(float x) { return sqrtf(x); }
// IEEE float division for operands whose quotient and reciprocal stay in the normal range (lsd_atan2_deg, lsd_grow.hip)
__device__ __forceinline__ float div_normal(float num, float den) { return num / den; }
// sin / cos of an angle given in turns (1.0 = 360 degrees), |turns| <= 256: estimates (absolute error of a few 1e-6) for
// callers that carry their own error bound (lsd_density_screen, lsd_grow.hip)                          (v_sin_f32, v_cos_f32)
__device__ __forceinline__ float sin_turns(float t) { return sinf(t * 6.28318530717958647692f); }
__device__ __forceinline__ float cos_turns(float t) { return cosf(t * 6.28318530717958647692f); }
// words in LDS shared by the wavefronts of a workgroup (k_lsd_grow_mw): relaxed, workgroup scope
__device__ __forceinline__ int lds_load(const int* p) { return *(volatile const int*)p; }
__device__ __forceinline__ void lds_store(int* p, int v) { *(volatile int*)p = v; }
__device__ __forceinline__ int lds_cas(int* p, int cmp, int v) { const int o = *p; if (o == cmp) *p = v; return o; }

// The walk of lsd_resolve (lsd_grow.hip): the lanes of mask P in lane order; lane k adds its (cs, sn) to the running sums of
// every lane behind it and, when mayDup, cancels the later lanes that examine the same pixel (nidx).  acc = lanes walked,
// canc = lanes cancelled.
__device__ __forceinline__ void lsd_walk(unsigned long long P, bool mayDup, uint32_t nidx, float cs, float sn, float& preX, float& preY,
                                         unsigned long long& accOut, unsigned long long& cancOut) {
  unsigned long long m = P, canc = 0;
  const int lane = lane_id();
  while (m) {
    const int k = __ffsll((long long)m) - 1;
    m &= ~(1ull << k);
    const unsigned long long above = ~1ull << k;
    if (mayDup) {
      const unsigned long long dup = __ballot(nidx == bcast_u32(nidx, k)) & above;
      m &= ~dup;
      canc |= dup;
    }
    const float ck = bcast_f32(cs, k), sk = bcast_f32(sn, k);
    if ((above >> lane) & 1ull) { preX += ck; preY += sk; }
  }
  accOut = P & ~canc;
  cancOut = canc;
}

}  // namespace ref

// ---------------------------------------------------------------------------------------------------------------------
#if !defined(HIPEMU)
namespace hw {

// HIP's __ballot materialises the predicate as an int first (v_cndmask + v_cmp, 8 issue cycles per vote on gfx950); this is
// the compare mask itself: feed it direct comparisons and combine the masks with scalar logic
__device__ __forceinline__ unsigned long long wballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// the SGPR pair feeds exec / v_cndmask directly (no v_cmp).  Never hand it a compile-time constant with bits in both halves:
// ROCm 7.2 lowers that to `s_mov_b64 sN, <32-bit literal>` and loses lanes 32..63 (measured; plh_selftest checks the form used)
__device__ __forceinline__ bool inv_ballot(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
__device__ __forceinline__ unsigned bcast_u32(unsigned v, int l) { return (unsigned)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ float bcast_f32(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ double bcast_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
// row_shr 1/2/4/8 + row_bcast 15/31 DPP steps (register to register, no LDS crossbar round trips) and one v_readlane
__device__ __forceinline__ int wave_min_i32(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x111, 0xf, 0xf, false));   // row_shr:1
  v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x112, 0xf, 0xf, false));   // row_shr:2
  v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x114, 0xf, 0xf, false));   // row_shr:4
  v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x118, 0xf, 0xf, false));   // row_shr:8: lane 15 of a row = row min
  v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 into rows 1, 3
  v = min(v, __builtin_amdgcn_update_dpp(INT_MAX, v, 0x143, 0xc, 0xf, false));   // row_bcast:31 into rows 2, 3
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ unsigned perm(unsigned hi, unsigned lo, unsigned sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
__device__ __forceinline__ unsigned alignbyte(unsigned hi, unsigned lo, unsigned sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 pk_of(unsigned a) { u16x2 v; __builtin_memcpy(&v, &a, 4); return v; }
__device__ __forceinline__ unsigned pk_to(u16x2 v) { unsigned a; __builtin_memcpy(&a, &v, 4); return a; }
__device__ __forceinline__ unsigned udot4(unsigned a, unsigned b, unsigned c) { return __builtin_amdgcn_udot4(a, b, c, false); }
__device__ __forceinline__ unsigned udot2(unsigned a, unsigned b, unsigned c) { return __builtin_amdgcn_udot2(pk_of(a), pk_of(b), c, false); }
__device__ __forceinline__ unsigned pk_min_u16(unsigned a, unsigned b) { return pk_to(__builtin_elementwise_min(pk_of(a), pk_of(b))); }
__device__ __forceinline__ unsigned pk_add16(unsigned a, unsigned b) { return pk_to(pk_of(a) + pk_of(b)); }
__device__ __forceinline__ unsigned pk_sub16(unsigned a, unsigned b) { return pk_to(pk_of(a) - pk_of(b)); }
__device__ __forceinline__ unsigned pk_twice_plus16(unsigned a, unsigned b) { const u16x2 two = {2, 2}; return pk_to(pk_of(a) * two + pk_of(b)); }
__device__ __forceinline__ unsigned hi_halves_sat255(unsigned hi, unsigned lo) { return pk_min_u16(__builtin_amdgcn_perm(hi, lo, 0x07060302u), 0x00ff00ffu); }
__device__ __forceinline__ int sbfe1(unsigned v, int k) { return __builtin_amdgcn_sbfe((int)v, (unsigned)k, 1u); }
__device__ __forceinline__ float fract(float x) { return __builtin_amdgcn_fractf(x); }
__device__ __forceinline__ float sqrt_approx(float x) { return __builtin_amdgcn_sqrtf(x); }
// the reciprocal / residual sequence the compiler emits between v_div_scale and v_div_fixup -- without those two, which only
// act on operands whose quotient or reciprocal leaves the normal range: same operations, same roundings, same result
__device__ __forceinline__ float div_normal(float num, float den) {
  float r = __builtin_amdgcn_rcpf(den);
  r = __builtin_fmaf(__builtin_fmaf(-den, r, 1.0f), r, r);
  float q = num * r;
  q = __builtin_fmaf(__builtin_fmaf(-den, q, num), r, q);
  return __builtin_fmaf(__builtin_fmaf(-den, q, num), r, q);
}
__device__ __forceinline__ float sin_turns(float t) { return __builtin_amdgcn_sinf(t); }
__device__ __forceinline__ float cos_turns(float t) { return __builtin_amdgcn_cosf(t); }
// The handshakes of k_lsd_grow_mw publish plain LDS stores (a post, FIFO entries) by a later store to a control word and read
// them after loading that word.  The hardware keeps a wavefront's LDS operations in order; what has to be pinned is the
// COMPILER's order of the plain accesses around the relaxed atomic -- a wavefront-scope fence does exactly that and emits no
// instruction (a workgroup-scope release would also drain the wavefront's global stores: s_waitcnt vmcnt(0) on every control
// word; the places that do hand over global memory say so with wg_release / wg_acquire).
__device__ __forceinline__ int lds_load(const int* p) {
  const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  return v;
}
__device__ __forceinline__ void lds_store(int* p, int v) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int lds_cas(int* p, int cmp, int v) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __hip_atomic_compare_exchange_strong(p, &cmp, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  return cmp;
}

// Hand-scheduled: it runs once per accepted pixel (130 k times per frame) and the per-lane updates are cheapest under
// EXEC = "lanes behind k", which the scalar unit produces in one instruction (s_lshl_b64 exec, -2, k): the compare then needs no
// mask and the adds no selects -- 6 VALU + 6 SALU instructions per pixel where the compiled form had 8 + 9 (the set of walked
// lanes is P & ~canc afterwards; the loop branches on the SCC of its last mask update).  Wait states (gfx940 family: an SGPR
// written by v_readlane may be read by a VALU instruction no sooner than the third instruction after it) are kept by the order
// of the instructions.
__device__ __forceinline__ void lsd_walk(unsigned long long P, bool mayDup, uint32_t nidx, float cs, float sn, float& preX, float& preY,
                                         unsigned long long& accOut, unsigned long long& cancOut) {
  unsigned long long m = P, canc = 0;
  unsigned long long saved;
  int k;
  unsigned t0;
  float t1, t2;
  if (mayDup) {
    asm volatile(
        "s_mov_b64 %[sv], exec\n"
        "lsdwalk%=:\n\t"
        "s_ff1_i32_b64 %[k], %[m]\n\t"
        "s_bitset0_b64 %[m], %[k]\n\t"
        "v_readlane_b32 %[t0], %[nidx], %[k]\n\t"
        "v_readlane_b32 %[t1], %[cs], %[k]\n\t"
        "v_readlane_b32 %[t2], %[sn], %[k]\n\t"
        "s_lshl_b64 exec, -2, %[k]\n\t"
        "v_cmp_eq_u32_e32 vcc, %[t0], %[nidx]\n\t"
        "v_add_f32_e32 %[px], %[t1], %[px]\n\t"
        "v_add_f32_e32 %[py], %[t2], %[py]\n\t"
        "s_or_b64 %[canc], %[canc], vcc\n\t"
        "s_andn2_b64 %[m], %[m], vcc\n\t"      // SCC = lanes left to walk
        "s_cbranch_scc1 lsdwalk%=\n\t"
        "s_mov_b64 exec, %[sv]"
        : [m] "+s"(m), [canc] "+s"(canc), [px] "+v"(preX), [py] "+v"(preY), [k] "=&s"(k),
          [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2), [sv] "=&s"(saved)
        : [nidx] "v"(nidx), [cs] "v"(cs), [sn] "v"(sn)
        : "vcc", "scc");
  } else {
    asm volatile(
        "s_mov_b64 %[sv], exec\n"
        "lsdwalk%=:\n\t"
        "s_ff1_i32_b64 %[k], %[m]\n\t"
        "s_bitset0_b64 %[m], %[k]\n\t"
        "v_readlane_b32 %[t1], %[cs], %[k]\n\t"
        "v_readlane_b32 %[t2], %[sn], %[k]\n\t"
        "s_lshl_b64 exec, -2, %[k]\n\t"
        "s_cmp_lg_u64 %[m], 0\n\t"
        "v_add_f32_e32 %[px], %[t1], %[px]\n\t"
        "v_add_f32_e32 %[py], %[t2], %[py]\n\t"
        "s_cbranch_scc1 lsdwalk%=\n\t"
        "s_mov_b64 exec, %[sv]"
        : [m] "+s"(m), [px] "+v"(preX), [py] "+v"(preY), [k] "=&s"(k), [t1] "=&s"(t1),
          [t2] "=&s"(t2), [sv] "=&s"(saved)
        : [cs] "v"(cs), [sn] "v"(sn)
        : "scc");
  }
  accOut = P & ~canc;   // the lanes walked: predicted, and not cancelled by an earlier walked lane
  cancOut = canc;
}

}  // namespace hw
namespace shim = hw;
#define PLH_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
// what is not an instruction with a portable twin but a scheduling / ordering primitive of the machine
__device__ __forceinline__ void spin_pause() { __builtin_amdgcn_s_sleep(8); }
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
__device__ __forceinline__ void wg_release() {   // this wavefront's global stores are complete (L1 / L2 of its CU)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void wg_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
// the identity, with the value's origin hidden from the optimiser (no instruction): keeps a load from being merged with another one
// through a selected pointer, or a loop-invariant derived from the value from being hoisted into a register the build has to spill
__device__ __forceinline__ unsigned plh_opaque_u32(unsigned v) { asm volatile("" : "+v"(v)); return v; }
#else
namespace shim = ref;
#define PLH_WAVE_SYNC() hipemu::wave_barrier()
__device__ __forceinline__ void spin_pause() { hipemu::spin_yield(); }   // the fibers of the other wavefronts get a turn
__device__ __forceinline__ void wave_fence() {}
__device__ __forceinline__ void wg_release() {}
__device__ __forceinline__ void wg_acquire() {}
__device__ __forceinline__ unsigned plh_opaque_u32(unsigned v) { return v; }
#endif

using shim::wballot;
using shim::bcast_u32;
using shim::bcast_f32;
using shim::bcast_f64;
using shim::wave_min_i32;
using shim::lsd_walk;
__device__ __forceinline__ unsigned plh_perm(unsigned hi, unsigned lo, unsigned sel) { return shim::perm(hi, lo, sel); }
__device__ __forceinline__ unsigned plh_alignbyte(unsigned hi, unsigned lo, unsigned sh) { return shim::alignbyte(hi, lo, sh); }
__device__ __forceinline__ unsigned plh_udot4(unsigned a, unsigned b, unsigned c) { return shim::udot4(a, b, c); }
__device__ __forceinline__ unsigned plh_udot2(unsigned a, unsigned b, unsigned c) { return shim::udot2(a, b, c); }
__device__ __forceinline__ unsigned plh_pk_min_u16(unsigned a, unsigned b) { return shim::pk_min_u16(a, b); }
using shim::pk_add16;
using shim::pk_sub16;
using shim::pk_twice_plus16;
using shim::hi_halves_sat255;
__device__ __forceinline__ int plh_sbfe1(unsigned v, int k) { return shim::sbfe1(v, k); }
__device__ __forceinline__ float plh_fract(float x) { return shim::fract(x); }
__device__ __forceinline__ float plh_sqrt_approx(float x) { return shim::sqrt_approx(x); }
using shim::div_normal;
using shim::sin_turns;
using shim::cos_turns;
using shim::lds_load;
using shim::lds_store;
using shim::lds_cas;
// a wave mask as a per-lane predicate (a macro for historical call sites; the mask must not be a compile-time constant)
#define PLH_INV_BALLOT(m) plh::shim::inv_ballot(m)

}  // namespace plh
