// hipemu -- a minimal single-process CPU emulation of the HIP subset used by pl-slam_amd/csrc.
// TEST INFRASTRUCTURE ONLY (tests/hipemu): lets `pytest -m "not gpu"` execute the *same kernel
// sources* on the CPU to debug indexing / ordering logic without a GPU.  It is never built into,
// linked with or loaded by the product library (libplslam_hip.so is always compiled by hipcc for
// gfx950); results from the emulator are not parity evidence.
//
// Model: blocks run one after another; the threads of a block are ucontext fibers scheduled
// round-robin on one OS thread.  __syncthreads() and the wave intrinsics (__ballot/__shfl*)
// are rendezvous points among the block's / wave's fibers (wave = 64 consecutive threads).
// `__shared__` becomes `static` (valid because blocks are sequential).
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__
#define HIPEMU 1

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3e { unsigned x, y, z; };
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
struct float2 { float x, y; };

typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost, hipMemcpyDefault };

namespace hipemu {
extern uint3e threadIdx_, blockIdx_;
extern dim3 blockDim_, gridDim_;
extern unsigned char* dyn_smem;
void block_barrier();
void wave_barrier();
void spin_yield();
unsigned long long wave_ballot(int pred);
unsigned long long wave_exchange(unsigned long long v, int src_lane, int width);  // value of lane src
void run_grid(dim3 grid, dim3 block, size_t shmem, void (*tramp)(void*), void* args);
int lane_id();
}  // namespace hipemu

#define threadIdx hipemu::threadIdx_
#define blockIdx hipemu::blockIdx_
#define blockDim hipemu::blockDim_
#define gridDim hipemu::gridDim_
#define warpSize 64
#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)hipemu::dyn_smem;

inline void __syncthreads() { hipemu::block_barrier(); }
inline void __threadfence() {}
inline void __threadfence_block() {}
inline unsigned long long __ballot(int pred) { return hipemu::wave_ballot(pred); }
inline int __any(int pred) { return hipemu::wave_ballot(pred) != 0; }
inline int __all(int pred) { return hipemu::wave_ballot(!pred) == 0; }

template <typename T>
inline T hipemu_shfl_raw(T v, int src, int width) {
  static_assert(sizeof(T) <= 8, "shfl type");
  unsigned long long raw = 0;
  memcpy(&raw, &v, sizeof(T));
  raw = hipemu::wave_exchange(raw, src, width);
  T out;
  memcpy(&out, &raw, sizeof(T));
  return out;
}
template <typename T> inline T __shfl(T v, int src, int width = 64) {
  int lane = hipemu::lane_id();
  int base = lane & ~(width - 1);
  return hipemu_shfl_raw(v, base + (src & (width - 1)), width);
}
template <typename T> inline T __shfl_xor(T v, int mask, int width = 64) {
  int lane = hipemu::lane_id();
  int src = lane ^ mask;
  if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
  return hipemu_shfl_raw(v, src, width);
}
template <typename T> inline T __shfl_down(T v, unsigned d, int width = 64) {
  int lane = hipemu::lane_id();
  int src = lane + (int)d;
  if ((src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
  return hipemu_shfl_raw(v, src, width);
}
template <typename T> inline T __shfl_up(T v, unsigned d, int width = 64) {
  int lane = hipemu::lane_id();
  int src = lane - (int)d;
  if (src < 0 || (src & ~(width - 1)) != (lane & ~(width - 1))) src = lane;
  return hipemu_shfl_raw(v, src, width);
}

inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __mul24(int a, int b) { return a * b; }
inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
inline unsigned __float_as_uint(float v) { unsigned u; memcpy(&u, &v, 4); return u; }
inline float __uint_as_float(unsigned u) { float v; memcpy(&v, &u, 4); return v; }
template <typename T> inline T atomicAnd(T* p, T v) { T o = *p; *p = (T)(o & v); return o; }
inline int __float2int_rn(float v) { return (int)lrintf(v); }
inline int __double2int_rn(double v) { return (int)lrint(v); }
inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned add) {
  int lane = hipemu::lane_id();
  unsigned m = lane >= 32 ? 0xffffffffu : ((1u << lane) - 1u);
  return add + __builtin_popcount(mask & m);
}
inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned add) {
  int lane = hipemu::lane_id();
  unsigned m = lane < 32 ? 0u : ((1u << (lane - 32)) - 1u);
  return add + __builtin_popcount(mask & m);
}
template <typename T> inline T __builtin_amdgcn_readfirstlane_emu(T v) { return hipemu_shfl_raw(v, hipemu::lane_id() & ~63, 64); }

template <typename T> inline T atomicAdd(T* p, T v) { T o = *p; *p = (T)(o + v); return o; }
template <typename T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> inline T atomicOr(T* p, T v) { T o = *p; *p = (T)(o | v); return o; }
template <typename T> inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <typename T> inline T atomicCAS(T* p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }

using std::max;
using std::min;

// ---- host runtime subset ----
inline hipError_t hipMalloc(void** p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <typename T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMallocAsync(void** p, size_t n, hipStream_t) { return hipMalloc(p, n); }
inline hipError_t hipFreeAsync(void* p, hipStream_t) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
template <typename T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipMalloc((void**)p, n); }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = 0) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t = 0) {
  for (size_t y = 0; y < h; y++) memcpy((char*)d + y * dp, (const char*)s + y * sp, w);
  return hipSuccess;
}
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = 0) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = nullptr; return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }   // the emulator's LDS is the heap
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)1; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
#define hipEventDisableTiming 2
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = 0) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0; return hipSuccess; }
#define hipStreamNonBlocking 1
#define hipHostMallocDefault 0

namespace hipemu {
template <typename... A> struct ArgPack;
template <typename F, typename Tuple, size_t... I>
inline void call_with(F f, Tuple& t, std::index_sequence<I...>) { f(std::get<I>(t)...); }
}  // namespace hipemu
#include <tuple>
template <typename... KA, typename... A>
inline void hipLaunchKernelGGL(void (*kernel)(KA...), dim3 grid, dim3 block, size_t shmem, hipStream_t, A... args) {
  std::tuple<KA...> packed(static_cast<KA>(args)...);
  struct Ctx { void (*k)(KA...); std::tuple<KA...>* t; } ctx{kernel, &packed};
  hipemu::run_grid(grid, block, shmem,
                   [](void* p) {
                     Ctx* c = (Ctx*)p;
                     hipemu::call_with(c->k, *c->t, std::index_sequence_for<KA...>{});
                   },
                   &ctx);
}
