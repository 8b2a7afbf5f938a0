// Device run of plh::sincos_cr (pl-slam_amd/csrc/plh_sincos_cr.h) over EVERY argument region2rect() can produce -- theta = (double)f x
// (pi / 180) for every float f in [0, 360] and theta + pi -- against the host's evaluation of the same source, which
// tools/sincos_cr_check.c has proven equal to the correctly rounded value on all of them (profiles/r04_sincos_cr.txt).  Host and device
// results are compared through per-block 64-bit sums of the result bits (s + 3 c), 1 M arguments per block: any differing value shows.
// The host half is compiled by g++ (tools/ubench/sincos_cr_host.cc): hipcc's host pass cannot read the header's __device__ tables.
//   g++ -O2 -march=x86-64-v3 -ffp-contract=off -std=c++17 -c -o /tmp/sincos_cr_host.o tools/ubench/sincos_cr_host.cc
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -c -o /tmp/sincos_cr_dev.o tools/ubench/sincos_cr_dev.hip
//   hipcc -o sincos_cr_dev /tmp/sincos_cr_dev.o /tmp/sincos_cr_host.o -lpthread
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../../pl-slam_amd/csrc/plh_sincos_cr.h"

static constexpr double kPI = 3.14159265358979323846;
static constexpr double kDegToRads = kPI / 180;
static constexpr uint32_t kLast = 0x43B40000u;   // 360.0f
static constexpr uint32_t kBlock = 1u << 20;

extern "C" void sincos_cr_host_sums(unsigned long long* out, uint32_t nblk, uint32_t block, uint32_t last);

__device__ inline unsigned long long one(uint32_t bits) {
  float f;
  memcpy(&f, &bits, 4);
  const double t1 = (double)f * kDegToRads;
  unsigned long long acc = 0;
  for (int flip = 0; flip < 2; flip++) {
    double s, c;
    plh::sincos_cr(flip ? t1 + kPI : t1, s, c);
    unsigned long long sb, cb;
    memcpy(&sb, &s, 8);
    memcpy(&cb, &c, 8);
    acc += sb + 3ull * cb;
  }
  return acc;
}

__global__ void k_sums(unsigned long long* out) {
  const uint32_t base = blockIdx.x * kBlock;
  unsigned long long acc = 0;
  for (uint32_t i = threadIdx.x; i < kBlock; i += blockDim.x) {
    const uint32_t bits = base + i;
    if (bits <= kLast) acc += one(bits);
  }
  atomicAdd(&out[blockIdx.x], acc);
}

int main() {
  const uint32_t nblk = kLast / kBlock + 1;
  unsigned long long* d = nullptr;
  if (hipMalloc((void**)&d, nblk * 8) != hipSuccess || hipMemset(d, 0, nblk * 8) != hipSuccess) return 2;
  hipLaunchKernelGGL(k_sums, dim3(nblk), dim3(256), 0, nullptr, d);
  std::vector<unsigned long long> dev(nblk), host(nblk, 0);
  sincos_cr_host_sums(host.data(), nblk, kBlock, kLast);
  if (hipMemcpy(dev.data(), d, nblk * 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  unsigned bad = 0;
  for (uint32_t b = 0; b < nblk; b++) bad += dev[b] != host[b];
  std::printf("sincos_cr on the device against the host's evaluation of the same source: %u blocks of %u float arguments (x 2: theta, theta + pi), "
              "%u blocks with a differing sum of result bits\n", nblk, kBlock, bad);
  return bad ? 1 : 0;
}
