// What does one 64-lane dword gather cost the texture addresser / vector L1 of a CU, as a function of WHERE the lanes point?
//
// k_lsd_grow's step is a gather of 64 four-byte records: lane 8g+n reads neighbour n of queue point g (rows y-1, y, y+1 of a
// row-major plane, pitch 512).  At full residency the TA of a CU is busy 54 % of the time (profiles/r04_grow_memory_path_counters.txt)
// at 23 busy cycles per vector-memory instruction and 11.7 "cache accesses" each: is an instruction charged per distinct cache
// line, per group of four lanes, or both?  This program times independent gathers (many in flight, 7 wavefronts per SIMD, every
// CU) for lane -> address maps that differ in exactly those quantities:
//   0 coalesced        64 consecutive dwords                                      2 lines, 16 quads of one line each
//   1 grow             8 adjacent queue points x 8 neighbours, row-major plane    what k_lsd_grow issues
//   2 scattered        64 random lines                                            64 lines
//   3 quads            16 random lines, 4 consecutive dwords in each              16 lines, 16 quads of one line each
//   4 grow, tiled      the points of (1) in a plane of 8 x 4-pixel tiles (128 B)  fewer lines, same quads
//   5 grow, n-major    the points of (1), lane n*8+g                              same lines as (1), other quads
//   6 grow, far        8 queue points far from each other, map of (1)             24 lines
//   7 one line         all 64 lanes in one 128-byte line, random order            1 line
// mode 0: the centre moves by a pixel or two per gather (what a growing region does: L1 hits); mode 1: it jumps (L1 misses).
//   hipcc --offload-arch=gfx950 -O3 -o gather_cost gather_cost.hip && ./gather_cost
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { if ((x) != hipSuccess) { std::fprintf(stderr, "%s failed\n", #x); return 1; } } while (0)

constexpr int PITCH = 512, ROWS = 384, WORDS = PITCH * ROWS;

__device__ __forceinline__ unsigned tiled(unsigned x, unsigned y) { return ((y >> 2) * (PITCH >> 3) + (x >> 3)) * 32u + (y & 3u) * 8u + (x & 7u); }

__device__ __forceinline__ unsigned hashu(unsigned s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
__device__ __forceinline__ unsigned word_of(unsigned s) { return (s & 0x1ffffu) + ((s >> 15) & 0xffffu); }   // < WORDS, no division

// (address arithmetic kept to a few instructions per gather: the centre is wave-uniform -- scalar unit -- and everything that only
// depends on the lane is computed once, so that the vector ALU is not what the loop measures)
template <int PAT>
__global__ void __launch_bounds__(64) k_gather(const unsigned* p, int iters, int jump, unsigned* sink) {
  const unsigned* w = p + (size_t)blockIdx.x * WORDS;
  const unsigned lane = threadIdx.x;
  const unsigned g = PAT == 5 ? (lane & 7u) : (lane >> 3), n = PAT == 5 ? (lane >> 3) : (lane & 7u);
  const unsigned nb = n < 4 ? n : n + 1;
  const int dy = (int)(nb / 3) - 1 + (int)(g / 3u), dx = (int)(nb % 3) - 1 + (int)(g % 3u);   // neighbour n of queue point g (a 3 x 3 cluster of points)
  const int offRow = dy * PITCH + dx;
  const unsigned laneKey = (PAT == 3 ? (lane >> 2) : PAT == 6 ? g : lane) * 0x9E3779B9u;
  unsigned s = 0x9E3779B9u * (blockIdx.x + 1u), acc = 0;
  unsigned cx = 8 + (s & 255u), cy = 8 + ((s >> 10) & 255u);
  for (int k = 0; k < iters; k += 4) {
    unsigned a[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      s = hashu(s);
      if (jump) { cx = 4 + ((s >> 3) & 0xffu) + ((s >> 12) & 0x7fu) + ((s >> 20) & 0x3fu); cy = 4 + ((s >> 5) & 0xffu) + ((s >> 14) & 0x3fu) + ((s >> 22) & 0x1fu); }
      else { cx = 8 + ((cx - 8 + (s & 3u)) & 255u); cy = 8 + ((cy - 8 + ((s >> 2) & 1u)) & 255u); }
      const unsigned c = cy * PITCH + cx;
      if (PAT == 0) a[u] = (c & ~63u) + lane;
      else if (PAT == 2) a[u] = word_of(hashu(s ^ laneKey));
      else if (PAT == 3) a[u] = (word_of(hashu(s ^ laneKey)) & ~3u) + (lane & 3u);
      else if (PAT == 7) a[u] = (c & ~31u) + ((lane * 13u + s) & 31u);
      else if (PAT == 6) { const unsigned hh = hashu(s ^ laneKey); const unsigned h = 2 * PITCH + (hh & 0x1ffffu) + ((hh >> 17) & 0x7fffu); a[u] = h + (unsigned)((int)(nb / 3) - 1) * PITCH + (unsigned)((int)(nb % 3) - 1); }
      else if (PAT == 4) a[u] = tiled(cx + (unsigned)dx, cy + (unsigned)dy);
      else a[u] = c + (unsigned)offRow;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) acc ^= w[a[u]];
  }
  if (acc == 0x12345678u) *sink = acc;
}

template <int PAT>
static double run(const unsigned* d, int waves, int iters, int jump, unsigned* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_gather<PAT>, dim3(waves), dim3(64), 0, 0, d, iters / 4, jump, sink);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k_gather<PAT>, dim3(waves), dim3(64), 0, 0, d, iters, jump, sink);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main(int argc, char** argv) {
  const int waves = argc > 1 ? std::atoi(argv[1]) : 256 * 28;   // 7 per SIMD
  const int iters = argc > 2 ? std::atoi(argv[2]) : 4000;
  unsigned* d = nullptr;
  unsigned* sink = nullptr;
  CHECK(hipMalloc((void**)&d, (size_t)waves * WORDS * 4));
  CHECK(hipMalloc((void**)&sink, 4));
  CHECK(hipMemset(d, 1, (size_t)waves * WORDS * 4));
  CHECK(hipDeviceSynchronize());
  const char* names[8] = {"coalesced", "grow (row-major)", "scattered 64 lines", "16 quads", "grow, 8x4 tiles", "grow, n-major lanes", "grow, far points", "one line"};
  std::printf("%d wavefronts (%.1f per SIMD), %d gathers each; cycles at 2.4 GHz per gather and CU\n", waves, waves / 1024.0, iters);
  for (int jump = 0; jump < 2; jump++) {
    std::printf("-- centre %s\n", jump ? "jumps (L1 misses)" : "walks (L1 hits)");
    double ms[8];
    ms[0] = run<0>(d, waves, iters, jump, sink); ms[1] = run<1>(d, waves, iters, jump, sink); ms[2] = run<2>(d, waves, iters, jump, sink);
    ms[3] = run<3>(d, waves, iters, jump, sink); ms[4] = run<4>(d, waves, iters, jump, sink); ms[5] = run<5>(d, waves, iters, jump, sink);
    ms[6] = run<6>(d, waves, iters, jump, sink); ms[7] = run<7>(d, waves, iters, jump, sink);
    for (int i = 0; i < 8; i++) {
      const double perCu = (double)waves / 256.0 * iters;   // gathers per CU
      std::printf("%-22s %9.3f ms  %7.1f cycles / gather / CU\n", names[i], ms[i], ms[i] * 1e-3 * 2.4e9 / perCu);
    }
  }
  return 0;
}
