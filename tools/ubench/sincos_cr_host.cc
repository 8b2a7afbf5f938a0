// Host half of tools/ubench/sincos_cr_dev.hip, compiled by g++ (hipcc's host pass cannot read the __device__ tables of
// plh_sincos_cr.h): per-block sums of the result bits of plh::sincos_cr over float-degree arguments, on all hardware threads.
//   g++ -O2 -march=x86-64-v3 -ffp-contract=off -std=c++17 -c tools/ubench/sincos_cr_host.cc
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../pl-slam_amd/csrc/plh_sincos_cr.h"

static const double kPI = 3.14159265358979323846;
static const double kDegToRads = kPI / 180;

extern "C" void sincos_cr_host_sums(unsigned long long* out, uint32_t nblk, uint32_t block, uint32_t last) {
  const unsigned nt = std::max(1u, std::thread::hardware_concurrency());
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; t++)
    th.emplace_back([=] {
      for (uint32_t b = t; b < nblk; b += nt) {
        unsigned long long acc = 0;
        for (uint32_t i = 0; i < block; i++) {
          const uint32_t bits = b * block + i;
          if (bits > last) break;
          float f;
          memcpy(&f, &bits, 4);
          const double t1 = (double)f * kDegToRads;
          for (int flip = 0; flip < 2; flip++) {
            double s, c;
            plh::sincos_cr(flip ? t1 + kPI : t1, s, c);
            unsigned long long sb, cb;
            memcpy(&sb, &s, 8);
            memcpy(&cb, &c, 8);
            acc += sb + 3ull * cb;
          }
        }
        out[b] = acc;
      }
    });
  for (auto& x : th) x.join();
}
