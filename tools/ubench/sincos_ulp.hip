// How far are the device sincos forms from the host libm (glibc) on the arguments region2rect() produces?
// theta = float degrees * pi/180 (+ pi): the library routine (ocml) and the head + tail evaluation of plh_common.h, each against
// std::sin / std::cos, in units in the last place.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I pl-slam_amd/csrc -I include
// -o sincos_ulp tools/ubench/sincos_ulp.hip && ./sincos_ulp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "plh_common.h"

__global__ void k(const double* th, int n, double* lib, double* sh) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s, c;
  sincos(th[i], &s, &c);
  lib[2 * i] = s; lib[2 * i + 1] = c;
  plh::sincos_head_tail(th[i], s, c);
  sh[2 * i] = s; sh[2 * i + 1] = c;
}
static long long ulpdiff(double a, double b) { int64_t x, y; memcpy(&x, &a, 8); memcpy(&y, &b, 8); return llabs(x - y); }
int main() {
  const int n = 1 << 24;
  std::mt19937_64 rng(1);
  const double PI = 3.14159265358979323846, D2R = PI / 180;
  std::vector<double> th(n), lib(2 * n), sh(2 * n);
  for (int i = 0; i < n; i++) {
    const float deg = (float)((rng() >> 11) * (360.0 / 9007199254740992.0));
    th[i] = (double)deg * D2R + ((rng() & 1) ? PI : 0.0);
  }
  double *dt, *dl, *ds;
  hipMalloc(&dt, n * 8); hipMalloc(&dl, n * 16); hipMalloc(&ds, n * 16);
  hipMemcpy(dt, th.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dt, n, dl, ds);
  hipMemcpy(lib.data(), dl, n * 16, hipMemcpyDeviceToHost);
  hipMemcpy(sh.data(), ds, n * 16, hipMemcpyDeviceToHost);
  long long ml = 0, ms = 0, xl = 0, xs = 0, mls = 0;
  for (int i = 0; i < n; i++) {
    const double g[2] = {std::sin(th[i]), std::cos(th[i])};
    for (int j = 0; j < 2; j++) {
      const long long a = ulpdiff(lib[2 * i + j], g[j]), b = ulpdiff(sh[2 * i + j], g[j]);
      ml += a != 0; ms += b != 0; if (a > xl) xl = a; if (b > xs) xs = b;
      mls += lib[2 * i + j] != sh[2 * i + j];
    }
  }
  printf("%d arguments (sin and cos each): library sincos differs from glibc in %lld values (max %lld ulp); head + tail evaluation in %lld (max %lld ulp); "
         "library vs head + tail: %lld\n", n, ml, xl, ms, xs, mls);
  return 0;
}
