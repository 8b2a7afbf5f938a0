// Exhaustive proof that plh::sincos_cr (pl-slam_amd/csrc/plh_sincos_cr.h) returns the correctly rounded cos / sin on EVERY
// argument region2rect() can hand it: theta = (double)f * (pi / 180) for every float f in [0, 360] (fastAtan2's range) and
// theta + pi (get_theta()'s flip) -- 2 x 1 135 869 953 values -- against libquadmath's sinq / cosq (113-bit), rounded to double.
// The same source runs on the device (IEEE operations and explicit FMAs only).
//   g++ -O2 -march=x86-64-v3 -ffp-contract=off -fopenmp -x c++ -o sincos_cr_check tools/sincos_cr_check.c -lquadmath
//   ./sincos_cr_check [first_bits last_bits]      (defaults: the whole range; prints a summary for profiles/r04_sincos_cr.txt)
#include <quadmath.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../pl-slam_amd/csrc/plh_sincos_cr.h"

static const double kPI = 3.14159265358979323846;
static const double kDegToRads = kPI / 180;

int main(int argc, char** argv) {
  uint32_t first = 0, last = 0x43B40000u;   // 0.0f .. 360.0f
  if (argc >= 3) { first = (uint32_t)strtoul(argv[1], 0, 0); last = (uint32_t)strtoul(argv[2], 0, 0); }
  unsigned long long bad = 0, n = 0, tie_risk = 0;
  double worst_arg = 0;
#pragma omp parallel for schedule(dynamic, 1 << 16) reduction(+ : bad, n, tie_risk)
  for (uint32_t bits = first; bits <= last; bits++) {
    float f;
    memcpy(&f, &bits, 4);
    const double t1 = (double)f * kDegToRads;
    for (int flip = 0; flip < 2; flip++) {
      const double t = flip ? t1 + kPI : t1;
      double s, c;
      plh::sincos_cr(t, s, c);
      __float128 qs, qc;
      sincosq((__float128)t, &qs, &qc);
      const double rs = (double)qs, rc = (double)qc;
      // a 113-bit value within 2^-105 (relative) of a rounding boundary of the double format could itself be misrounded
      const __float128 es = fabsq(qs - (__float128)rs), ec = fabsq(qc - (__float128)rc);
      const __float128 us = (__float128)ldexp(1.0, ilogb(rs == 0 ? 1e-300 : rs) - 53), uc = (__float128)ldexp(1.0, ilogb(rc == 0 ? 1e-300 : rc) - 53);
      if (fabsq(es - us) < us * 1e-28Q || fabsq(ec - uc) < uc * 1e-28Q) tie_risk++;
      if (s != rs || c != rc) {
        bad++;
#pragma omp critical
        { if (bad < 10) fprintf(stderr, "mismatch at f bits 0x%08x flip %d: sin %a vs %a, cos %a vs %a\n", bits, flip, s, rs, c, rc); worst_arg = t; }
      }
      n++;
    }
  }
  printf("sincos_cr against libquadmath (113-bit, rounded to double): %llu arguments (float degrees 0x%08x .. 0x%08x x pi/180, and + pi), "
         "%llu differing values, %llu arguments where the 113-bit value itself sits within 1e-28 ulp of a tie\n",
         n, first, last, bad, tie_risk);
  (void)worst_arg;
  return bad ? 1 : 0;
}
