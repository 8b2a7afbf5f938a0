// Correctly rounded cos / sin of a double in [0, 3 pi]: the direction of region2rect()'s rectangle (cv::LineSegmentDetector
// evaluates cos(theta), sin(theta) with the C library; which last bit that gives is a property of the libm the reference was
// linked with -- glibc >= 2.28 is not correctly rounded, older ones were).  The pinned definition is the mathematically defined
// one: the double nearest to the true value.  Rounds 1-3 evaluated a head + tail form that differed from the host libm in 3 % of
// the values by one unit in the last place, as the device library's routine does (profiles/r03_sincos_ulp.txt): equal segments
// then rested on the float rounding of the end points.  This is exact by construction:
//   1. reduction by pi/2 in three parts (33 + 33 + 53 bits; the first two products are exact for k <= 6): r = x + y;
//   2. quick phase: x = j / 128 + d, |d| <= 2^-8; sin / cos of j / 128 from a table of double-double values, the angle-sum
//      formulas with the two leading terms kept exactly (the table's head and an exact product) and the rest (< 2^-16 of the
//      result) in double: total error below 2^-63 of the result;
//   3. Ziv's test: if the quick result rounds the same way with the error bound added and subtracted, it is the correctly
//      rounded value; otherwise (about one argument in 2^9) --
//   4. accurate phase: the Taylor series of sin / cos on r in double-double arithmetic (terms to r^29 / 29!: < 2^-100).
// Host and device run the same source (only IEEE operations and explicit FMAs), so tools/sincos_cr_check.c can prove it on the
// host: EVERY argument the kernels can produce -- theta = (double)f x (pi / 180) and that + pi for every float f in [0, 360],
// 2.27e9 values -- against libquadmath's 113-bit sinq / cosq (profiles/r04_sincos_cr.txt: 0 differences); the GPU runs the same
// arguments against the host's results (tools/ubench/sincos_ulp.hip).
#pragma once

#if defined(__HIPCC__) || defined(__HIP__)
#define PLH_HD __host__ __device__ __forceinline__
#define PLH_DTAB __device__ const
#else
#define PLH_HD static inline
#define PLH_DTAB static const
#endif

#include "sincos_cr_table.inc"

namespace plh {

PLH_DTAB double k_sincos_tab[102][4] = {PLH_SINCOS_TABLE};
PLH_DTAB double k_inv_fact[30][2] = {PLH_INV_FACT_TABLE};

struct DD {
  double h, l;
};
PLH_HD DD dd_fast2sum(double a, double b) {   // |a| >= |b| (or a == 0)
  DD r;
  r.h = a + b;
  r.l = b - (r.h - a);
  return r;
}
PLH_HD DD dd_2sum(double a, double b) {
  DD r;
  r.h = a + b;
  const double bb = r.h - a;
  r.l = (a - (r.h - bb)) + (b - bb);
  return r;
}
PLH_HD DD dd_2prod(double a, double b) {
  DD r;
  r.h = a * b;
  r.l = __builtin_fma(a, b, -r.h);
  return r;
}
PLH_HD DD dd_add(DD a, DD b) {
  DD s = dd_2sum(a.h, b.h);
  s.l += a.l + b.l;
  return dd_fast2sum(s.h, s.l);
}
PLH_HD DD dd_mul(DD a, DD b) {
  DD p = dd_2prod(a.h, b.h);
  p.l += __builtin_fma(a.h, b.l, a.l * b.h);
  return dd_fast2sum(p.h, p.l);
}

// accurate phase: sin / cos of r = x + y, |r| <= pi/4 + 2^-7, to 2^-100
PLH_HD void sincos_dd(double x, double y, DD& s, DD& c) {
  const DD r = dd_fast2sum(x, y);
  const DD z = dd_mul(r, r);
  DD ps, pc;   // Horner in z: sin = r (1 - z/3! + z^2/5! - ...), cos = 1 - z/2! + z^2/4! - ...
  ps.h = k_inv_fact[29][0]; ps.l = k_inv_fact[29][1];
  pc.h = k_inv_fact[28][0]; pc.l = k_inv_fact[28][1];
  for (int n = 27; n >= 1; n -= 2) {
    DD cs, cc;
    cs.h = k_inv_fact[n][0]; cs.l = k_inv_fact[n][1];
    cc.h = k_inv_fact[n - 1][0]; cc.l = k_inv_fact[n - 1][1];
    DD t = dd_mul(ps, z);
    t.h = -t.h; t.l = -t.l;
    ps = dd_add(cs, t);
    t = dd_mul(pc, z);
    t.h = -t.h; t.l = -t.l;
    pc = dd_add(cc, t);
  }
  s = dd_mul(ps, r);
  c = pc;
}

// Is h -- with l, the rest of an unevaluated sum h + l that is within err of the true value, |l| <= ulp(h) / 2 -- the double
// nearest to the true value whatever the error is?
PLH_HD bool dd_round_safe(double h, double l, double err) { return h + (l + err) == h && h + (l - err) == h; }

PLH_HD void sincos_cr(double ad, double& s, double& c) {
  const double kd = (double)(int)(ad * 0.63661977236758138 + 0.5);
  const int k = (int)kd;
  const double t = __builtin_fma(-kd, 1.57079632673412561417e+00, ad);   // 33-bit head of pi/2: exact
  double w = kd * 6.07710050630396597660e-11;                            // next 33 bits: exact
  const double r = t - w;
  w = __builtin_fma(kd, 2.02226624879595063154e-21, -((t - r) - w));     // the rest, and what the subtraction above lost
  const double x = r - w;
  const double y = (r - x) - w;                                          // r = x + y to ~2^-110
  // ---- quick phase
  const double ax = x < 0 ? -x : x;
  int j = (int)(ax * 128.0 + 0.5);
  j = j < 0 ? 0 : (j > 101 ? 101 : j);   // (a NaN argument -- a lane without a region -- must not index outside the table)
  const double xk = (double)j * 0.0078125;
  const double sg = x < 0 ? -1.0 : 1.0;           // sin is odd, cos even: work on |r| = sg (x + y)
  const double d0 = ax - xk;                      // exact
  const DD d = dd_2sum(d0, sg * y);
  const double u = d.h, u2 = u * u;
  // sin(u) - u and cos(u) - 1 for |u| <= 2^-8 (+ the next term than needed for 2^-68)
  const double sm = u * u2 * (-1.66666666666666666667e-01 + u2 * (8.33333333333333333333e-03 + u2 * -1.98412698412698412698e-04));
  const double cm1 = u2 * (-0.5 + u2 * (4.16666666666666666667e-02 + u2 * (-1.38888888888888888889e-03 + u2 * 2.48015873015873015873e-05)));
  const double Sh = k_sincos_tab[j][0], Sl = k_sincos_tab[j][1], Ch = k_sincos_tab[j][2], Cl = k_sincos_tab[j][3];
  double sh, sl, ch, cl;
  bool okS, okC;
  {   // sin(xk + d) = S cos d + C sin d = Sh + Ch u + { Sl + Ch dl + Cl u + Sh (cos d - 1) + Ch (sin u - u) - Sh u dl }
    const DD p = dd_2prod(Ch, u);
    const double small = Sl + ((Ch * d.l + Cl * u) + (Sh * cm1 + (Ch * sm - Sh * (u * d.l))));
    const DD a = dd_2sum(Sh, p.h);
    const DD q = dd_fast2sum(a.h, a.l + (p.l + small));
    sh = q.h; sl = q.l;
    const double mag = sh < 0 ? -sh : sh;
    okS = dd_round_safe(sh, sl, mag * 1.0842021724855044e-19 + 1e-300);   // 2^-63 of the result
  }
  {   // cos(xk + d) = C cos d - S sin d = Ch - Sh u + { Cl - Sh dl - Sl u + Ch (cos d - 1) - Sh (sin u - u) - Ch u dl }
    const DD p = dd_2prod(-Sh, u);
    const double small = Cl + ((-(Sh * d.l) - Sl * u) + (Ch * cm1 - (Sh * sm + Ch * (u * d.l))));
    const DD a = dd_2sum(Ch, p.h);
    const DD q = dd_fast2sum(a.h, a.l + (p.l + small));
    ch = q.h; cl = q.l;
    const double mag = ch < 0 ? -ch : ch;
    okC = dd_round_safe(ch, cl, mag * 1.0842021724855044e-19 + 1e-300);
  }
  if (!(okS && okC)) {   // ---- accurate phase (rare)
    DD S, C;
    sincos_dd(ax, sg * y, S, C);
    sh = S.h; ch = C.h;
  }
  sh *= sg;
  (void)sl; (void)cl;
  const bool swap = (k & 1) != 0;
  const double s0 = swap ? ch : sh, c0 = swap ? sh : ch;
  s = (k & 2) ? -s0 : s0;
  c = ((k + 1) & 2) ? -c0 : c0;
}

}  // namespace plh
