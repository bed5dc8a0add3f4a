// cr_cos.h -- a CORRECTLY ROUNDED f64 cos for the f-vectors of the Tucker objective (host + device).
//
// The reference evaluates f = float32(a*cos(b*w + c) + d) (TD_Tester.py:25-28,37) with numpy, i.e. the host libm's cos.  The
// reference-order objective reproduces the reference bit for bit EXCEPT for that cos: the device math library's cos may differ from
// libm's in the last place, and where a*cos+d nearly cancels such a difference flips the f32 rounding of f about once per 2e6
// values (measured over 9e6: 5 flips) -- a handful of evaluations per 4,096-face Powell run, each of which can move one face's
// trajectory off scipy's.  Two different libraries cannot be made to agree bit for bit, but both can be held to the exact
// result: this cos is evaluated in double-double arithmetic (~104 bits) and rounded once, so it returns the correctly rounded
// value (a wrong rounding needs the true value within 2^-100 of a tie), and differs from glibc's (< 0.55 ulp, correctly rounded
// in all but ~1e-4 of the cases) only where glibc itself is not correctly rounded -- and then by one unit in the last place.
// tests/test_abi_and_host.py checks it on the CPU against 60-digit decimal arithmetic; tests/test_gpu_parity.py sweeps 1e6
// angles on the device against numpy.
//
//   |x| <= 2^20: k = rint(x * 2/pi), r = x - k*pi/2 with pi/2 as four doubles (212 bits), cos/sin of r (|r| <= pi/4 + eps) by
//                their Taylor series to r^30 / r^31 in double-double Horner form, quadrant fix-up;
//   otherwise (never the case for the f-vectors: |b*w + c| < 10) the library cos.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NLML_CR_HD __host__ __device__ __forceinline__
#else
#define NLML_CR_HD inline
#endif

namespace nlml {

struct DD {
  double hi, lo;
};

#if defined(__clang__)
#define NLML_CR_STRICT _Pragma("clang fp contract(off)")
#else
#define NLML_CR_STRICT
#endif

NLML_CR_HD DD dd_two_sum(double a, double b) {
  NLML_CR_STRICT
  const double s = a + b, bb = s - a;
  return DD{s, (a - (s - bb)) + (b - bb)};
}
NLML_CR_HD DD dd_quick_two_sum(double a, double b) {   // |a| >= |b|
  NLML_CR_STRICT
  const double s = a + b;
  return DD{s, b - (s - a)};
}
NLML_CR_HD DD dd_two_prod(double a, double b) {
  NLML_CR_STRICT
  const double p = a * b;
  return DD{p, fma(a, b, -p)};
}
NLML_CR_HD DD dd_add(DD a, DD b) {
  NLML_CR_STRICT
  DD s = dd_two_sum(a.hi, b.hi);
  const DD t = dd_two_sum(a.lo, b.lo);
  s.lo += t.hi;
  s = dd_quick_two_sum(s.hi, s.lo);
  s.lo += t.lo;
  return dd_quick_two_sum(s.hi, s.lo);
}
NLML_CR_HD DD dd_mul(DD a, DD b) {
  NLML_CR_STRICT
  DD p = dd_two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return dd_quick_two_sum(p.hi, p.lo);
}

// 1 / (2n)! and 1 / (2n+1)!, n = 0..15, as double-doubles (constexpr + fully unrolled use: immediates in the instruction stream;
// as run-time-indexed local arrays they were 30 dependent memory reads per value)
constexpr double CR_CC[16][2] = {
    {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.0000000000000p-1, 0x0.0p+0},
    {0x1.5555555555555p-5, 0x1.5555555555555p-59},
    {0x1.6c16c16c16c17p-10, -0x1.f49f49f49f49fp-65},
    {0x1.a01a01a01a01ap-16, 0x1.a01a01a01a01ap-76},
    {0x1.27e4fb7789f5cp-22, 0x1.cbbc05b4fa99ap-76},
    {0x1.1eed8eff8d898p-29, -0x1.2aec959e14c06p-83},
    {0x1.93974a8c07c9dp-37, 0x1.05d6f8a2efd1fp-92},
    {0x1.ae7f3e733b81fp-45, 0x1.1d8656b0ee8cbp-101},
    {0x1.6827863b97d97p-53, 0x1.eec01221a8b0bp-107},
    {0x1.e542ba4020225p-62, 0x1.ea72b4afe3c2fp-120},
    {0x1.0ce396db7f853p-70, -0x1.aebcdbd20331cp-124},
    {0x1.f2cf01972f578p-80, -0x1.9ada5fcc1ab14p-135},
    {0x1.88e85fc6a4e5ap-89, -0x1.71c37ebd16540p-143},
    {0x1.0a18a2635085dp-98, 0x1.b9e2e28e1aa54p-153},
    {0x1.3932c5047d60ep-108, 0x1.832b7b530a627p-162},
};
constexpr double CR_SC[16][2] = {
    {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.5555555555555p-3, 0x1.5555555555555p-57},
    {0x1.1111111111111p-7, 0x1.1111111111111p-63},
    {0x1.a01a01a01a01ap-13, 0x1.a01a01a01a01ap-73},
    {0x1.71de3a556c734p-19, -0x1.c154f8ddc6c00p-73},
    {0x1.ae64567f544e4p-26, -0x1.c062e06d1f209p-80},
    {0x1.6124613a86d09p-33, 0x1.f28e0cc748ebep-87},
    {0x1.ae7f3e733b81fp-41, 0x1.1d8656b0ee8cbp-97},
    {0x1.952c77030ad4ap-49, 0x1.ac981465ddc6cp-103},
    {0x1.2f49b46814157p-57, 0x1.2650f61dbdcb4p-112},
    {0x1.71b8ef6dcf572p-66, -0x1.d043ae40c4647p-120},
    {0x1.761b41316381ap-75, -0x1.3423c7d91404fp-130},
    {0x1.3f3ccdd165fa9p-84, -0x1.58ddadf344487p-139},
    {0x1.d1ab1c2dccea3p-94, 0x1.054d0c78aea14p-149},
    {0x1.259f98b4358adp-103, 0x1.eaf8c39dd9bc5p-157},
    {0x1.434d2e783f5bcp-113, 0x1.0b87b91be9affp-167},
};

NLML_CR_HD double cr_cos(double x) {
  NLML_CR_STRICT
  if (!(fabs(x) <= 1048576.0)) return cos(x);   // also NaN / Inf
  // argument reduction: r = x - k * pi/2, pi/2 = P1 + P2 + P3 + P4 (each exactly a double), |k| <= 2^20
  const double k = rint(x * 0x1.45f306dc9c883p-1);
  const double P1 = 0x1.921fb54442d18p+0, P2 = 0x1.1a62633145c07p-54, P3 = -0x1.f1976b7ed8fbcp-110, P4 = 0x1.4cf98e804177dp-164;
  DD r = dd_two_prod(-k, P1);                   // exact
  r = dd_add(DD{x, 0.0}, r);                    // x - k*P1: heavy cancellation, exact in double-double
  r = dd_add(r, dd_two_prod(-k, P2));
  r = dd_add(r, dd_two_prod(-k, P3));
  r = dd_add(r, DD{-k * P4, 0.0});
  const DD r2 = dd_mul(r, r);
  const long long q = (long long)k & 3;         // (two's complement: also right for negative k)
  // Horner in z = r^2 on the alternating series: s_15 = -c_15, s_n = (-1)^n c_n + z * s_{n+1}, s_0 = sum_n (-1)^n c_n z^n
  // +-cos(r) = sum (-1)^n r^(2n)/(2n)!;  +-sin(r) = r * sum (-1)^n r^(2n)/(2n+1)!
  // (one branch-free chain: the lanes of a wave hold different quadrants, so an if/else over the two series would run both;
  // the coefficients are immediates selected per lane)
  const bool odd = (q & 1) != 0;
  DD p = DD{odd ? -CR_SC[15][0] : -CR_CC[15][0], odd ? -CR_SC[15][1] : -CR_CC[15][1]};
#pragma unroll
  for (int n = 14; n >= 0; --n) {
    p = dd_mul(p, r2);
    const double sg = (n & 1) ? -1.0 : 1.0;
    p = dd_add(DD{odd ? sg * CR_SC[n][0] : sg * CR_CC[n][0], odd ? sg * CR_SC[n][1] : sg * CR_CC[n][1]}, p);
  }
  if ((q & 1) != 0) p = dd_mul(p, r);
  // cos(x) by quadrant: q = 0: cos r, 1: -sin r, 2: -cos r, 3: sin r
  const bool neg = (q == 1) || (q == 2);
  const double v = p.hi + p.lo;
  return neg ? -v : v;
}

// float32(a * cos(t) + d) with the cos correctly rounded -- the f-vector entry as the reference rounds it (TD_Tester.py:25-28,37) --
// without paying for the double-double chain (~450 dependent operations, 2 us of every Powell round) unless it can matter.  The
// library cos is within 2 ulp, i.e. within 2^-51 absolute of the correctly rounded one (|cos| <= 1), so v = a * cos + d is within
// |a| * 2^-51 + a few roundings of the value the slow path forms; if v - delta and v + delta round to the SAME float with
// delta = (|a| + |v|) * 2^-46 (32 times that bound), so does the slow path's value (rounding is monotonic) and it is returned;
// otherwise -- about once in 2^21 values, more often only where a * cos + d cancels -- the slow path decides.  NaN falls through.
NLML_CR_HD float cr_f32_a_cos_d(double a, double t, double d) {
  NLML_CR_STRICT
  const double v = a * cos(t) + d;
  const double delta = (fabs(a) + fabs(v)) * 0x1p-46;
  const float lo = (float)(v - delta), hi = (float)(v + delta);
  if (lo == hi) return lo;
  return (float)(a * cr_cos(t) + d);
}

}  // namespace nlml
