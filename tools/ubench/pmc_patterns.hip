// Calibration of the L2 memory-side counters (FETCH_SIZE / WRITE_SIZE) for the ACCESS PATTERNS of this code base.
//
// MI355X_MICROARCH.md derives its x2 FETCH_SIZE correction on gfx950 from wide coalesced 16-byte-per-lane streaming reads and says
// other widths are uncalibrated; round 3 applied x2 to every kernel, which made k_remap_u8 / k_blur7_u8 "fetch" 2.04 x their
// 307 200-byte input.  This program reads a buffer FAR larger than the 256 MiB Infinity Cache exactly once, in the
// four ways the front end's kernels read memory, so that every kernel below has a known number of bytes to fetch:
//   k_stream16   16 bytes per lane, consecutive lanes                    (the guide's reference pattern)
//   k_stream4    aligned dwords, consecutive lanes                       (dword-staged rows: k_fast_strips, k_blur7_u8, k_pyr_down)
//   k_stream1    single bytes, consecutive lanes                         (k_remap_u8's taps, byte loads)
//   k_gather4    random 4-byte gathers inside a 0.8 MB window per wavefront, `frames` windows: k_lsd_grow's record loads
//                (every 64-byte sector of a window is hit about `touch` times; bytes "needed" = sectors touched x 64)
// Run under rocprofv3 --pmc FETCH_SIZE (kernel trace); tools/pmc_calibrate.sh divides the known bytes by the counter.
//   hipcc --offload-arch=gfx950 -O3 -o pmc_patterns pmc_patterns.hip && ./pmc_patterns [MiB = 2048]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { if ((x) != hipSuccess) { std::fprintf(stderr, "%s failed\n", #x); return 1; } } while (0)

__global__ void k_stream16(const uint4* p, size_t n, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = p[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_stream4(const unsigned* p, size_t n, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_stream1(const unsigned char* p, size_t n, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 0x12345678u) *sink = acc;
}
// one wavefront per window of `words` dwords; `loads` dependent-free random gathers per lane (xorshift), all lanes distinct
__global__ void __launch_bounds__(64) k_gather4(const unsigned* p, unsigned words, int loads, unsigned* sink) {
  const unsigned* w = p + (size_t)blockIdx.x * words;
  unsigned s = 0x9E3779B9u * (blockIdx.x * 64u + threadIdx.x + 1u), acc = 0;
  for (int k = 0; k < loads; k++) {
    s ^= s << 13; s ^= s >> 17; s ^= s << 5;
    acc ^= w[s % words];
  }
  if (acc == 0x12345678u) *sink = acc;
}

int main(int argc, char** argv) {
  const size_t mib = argc > 1 ? (size_t)std::atoll(argv[1]) : 2048;
  const size_t bytes = mib << 20;
  unsigned char* d = nullptr;
  unsigned* sink = nullptr;
  CHECK(hipMalloc((void**)&d, bytes));
  CHECK(hipMalloc((void**)&sink, 4));
  CHECK(hipMemset(d, 1, bytes));
  CHECK(hipDeviceSynchronize());
  const int grid = 256 * 32, block = 256;
  // every kernel is preceded by a pass over a second buffer of the same size, so that nothing of `d` is left in the caches
  unsigned char* flush = nullptr;
  CHECK(hipMalloc((void**)&flush, bytes));
  CHECK(hipMemset(flush, 2, bytes));
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k_stream16, dim3(grid), dim3(block), 0, nullptr, (const uint4*)flush, bytes / 16, sink);
    hipLaunchKernelGGL(k_stream16, dim3(grid), dim3(block), 0, nullptr, (const uint4*)d, bytes / 16, sink);
    hipLaunchKernelGGL(k_stream4, dim3(grid), dim3(block), 0, nullptr, (const unsigned*)flush, bytes / 4, sink);
    hipLaunchKernelGGL(k_stream4, dim3(grid), dim3(block), 0, nullptr, (const unsigned*)d, bytes / 4, sink);
    hipLaunchKernelGGL(k_stream1, dim3(grid), dim3(block), 0, nullptr, (const unsigned char*)flush, bytes, sink);
    hipLaunchKernelGGL(k_stream1, dim3(grid), dim3(block), 0, nullptr, (const unsigned char*)d, bytes, sink);
    // gathers: windows of 196 608 dwords (a 512 x 384 record plane), as many as fit; 3 x 196 608 / 64 loads per lane = every
    // sector of the window is expected ~48 x: all 12 288 sectors of a window are touched (P(miss) = e^-48)
    const unsigned words = 196608;
    const unsigned frames = (unsigned)(bytes / 4 / words);
    hipLaunchKernelGGL(k_stream16, dim3(grid), dim3(block), 0, nullptr, (const uint4*)flush, bytes / 16, sink);
    hipLaunchKernelGGL(k_gather4, dim3(frames), dim3(64), 0, nullptr, (const unsigned*)d, words, 3 * (int)words / 64, sink);
    CHECK(hipDeviceSynchronize());
    if (rep == 0)
      std::printf("known bytes: k_stream16 %zu (x2 launches: flush + measured), k_stream4 %zu, k_stream1 %zu, k_gather4 %zu (%u windows x %u bytes, "
                  "%u lane-loads of 4 bytes = %zu requested bytes)\n",
                  bytes, bytes, bytes, (size_t)frames * words * 4, frames, words * 4, frames * 3 * words, (size_t)frames * 3 * words * 4);
  }
  CHECK(hipFree(d));
  CHECK(hipFree(flush));
  CHECK(hipFree(sink));
  return 0;
}
