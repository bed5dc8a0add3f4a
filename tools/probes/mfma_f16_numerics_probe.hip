// Probe: the NUMERICS of v_mfma_f32_32x32x16_f16 on gfx950 -- how D = C + sum_{k<16} a_k * b_k is rounded.  The split-f16 kernel
// (csrc/encoder_heads_f16x2*.h) measures 1.36x the error a model with exact 16-term sums and ONE rounding per instruction
// predicts, so the instruction's internal accumulation matters.  Lane 0 owns A row 0 (k = 0..7) and B column 0; lane 32 the
// k = 8..15 halves; the result D[0][0] is register 0 of lane 0.
//   (1) targeted cases: is the final rounding to-nearest or truncating, are the 16 products summed exactly before they meet C,
//       how many bits below C's last place survive;
//   (2) statistics over random operands: signed error of D against the exact (f64) sum in units in the last place of D
//       (round-to-nearest once: mean 0, rms 0.29; truncation: mean -0.5 towards zero).
// Build: hipcc --offload-arch=gfx950 -O3 -w -o exp_libs/mfma_f16_numerics_probe tools/probes/mfma_f16_numerics_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// a[16], b[16] (f32 values exactly representable in f16), c -> D[0][0]
__global__ void one(const float* a, const float* b, const float* c, float* out, int n) {
  for (int t = 0; t < n; ++t) {
    h8 av, bv;
    const int h = threadIdx.x >> 5;
    for (int i = 0; i < 8; ++i) {
      av[i] = (threadIdx.x & 31) == 0 ? (_Float16)a[t * 16 + 8 * h + i] : (_Float16)0.f;
      bv[i] = (threadIdx.x & 31) == 0 ? (_Float16)b[t * 16 + 8 * h + i] : (_Float16)0.f;
    }
    f16v cv;
    for (int i = 0; i < 16; ++i) cv[i] = 0.f;
    if (threadIdx.x == 0) cv[0] = c[t];
    cv = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, cv, 0, 0, 0);
    if (threadIdx.x == 0) out[t] = cv[0];
  }
}

static float f16r(float v) { return (float)(_Float16)v; }

int main() {
  const int NT = 200000;
  float *ha = (float*)malloc(NT * 16 * 4), *hb = (float*)malloc(NT * 16 * 4), *hc = (float*)malloc(NT * 4), *ho = (float*)malloc(NT * 4);
  float *da, *db, *dc, *dout;
  hipMalloc(&da, NT * 16 * 4); hipMalloc(&db, NT * 16 * 4); hipMalloc(&dc, NT * 4); hipMalloc(&dout, NT * 4);
  auto run = [&](int n) {
    hipMemcpy(da, ha, n * 16 * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb, n * 16 * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, hc, n * 4, hipMemcpyHostToDevice);
    one<<<1, 64>>>(da, db, dc, dout, n);
    hipMemcpy(ho, dout, n * 4, hipMemcpyDeviceToHost);
  };
  // ---- targeted cases
  struct Case { const char* what; float c; float p[16]; };
  auto P = [](int e) { return ldexpf(1.0f, e); };
  Case cases[] = {
      {"C=1, one product 1.5*2^-24 (RN: 1+2^-23, truncation: 1)", 1.0f, {1.5f * P(-24)}},
      {"C=1, one product 2^-24 exactly (tie: RN-even 1)", 1.0f, {P(-24)}},
      {"C=1, one product 2^-24+2^-34 (just above the tie: RN 1+2^-23)", 1.0f, {P(-24) + P(-34)}},
      {"C=1, 16 products of 2^-25 (exact sum 2^-21: 1+2^-21 if summed before meeting C)", 1.0f,
       {P(-25), P(-25), P(-25), P(-25), P(-25), P(-25), P(-25), P(-25), P(-25), P(-25), P(-25), P(-25), P(-25), P(-25), P(-25), P(-25)}},
      {"C=1, 16 products of 2^-28 (exact sum 2^-24: tie -> 1; kept bits below the last place?)", 1.0f,
       {P(-28), P(-28), P(-28), P(-28), P(-28), P(-28), P(-28), P(-28), P(-28), P(-28), P(-28), P(-28), P(-28), P(-28), P(-28), P(-28)}},
      {"C=1, 16 products of 1.5*2^-28 (exact sum 1.5*2^-24 -> RN 1+2^-23)", 1.0f,
       {1.5f * P(-28), 1.5f * P(-28), 1.5f * P(-28), 1.5f * P(-28), 1.5f * P(-28), 1.5f * P(-28), 1.5f * P(-28), 1.5f * P(-28), 1.5f * P(-28),
        1.5f * P(-28), 1.5f * P(-28), 1.5f * P(-28), 1.5f * P(-28), 1.5f * P(-28), 1.5f * P(-28), 1.5f * P(-28)}},
      {"C=1, 16 products of 1.5*2^-32 (exact sum 1.5*2^-28: far below; then one of 2^-24: sum 2^-24+1.5*2^-28 -> RN 1+2^-23)", 1.0f,
       {P(-24), 1.5f * P(-32), 1.5f * P(-32), 1.5f * P(-32), 1.5f * P(-32), 1.5f * P(-32), 1.5f * P(-32), 1.5f * P(-32), 1.5f * P(-32),
        1.5f * P(-32), 1.5f * P(-32), 1.5f * P(-32), 1.5f * P(-32), 1.5f * P(-32), 1.5f * P(-32), 1.5f * P(-32)}},
      {"C=0, products 1 and 2^-24+2^-30 (sum needs 31 bits: RN 1+2^-23)", 0.0f, {1.0f, P(-24) + P(-30)}},
      {"C=0, products 1, -1, 2^-30 (cancellation: exact 2^-30)", 0.0f, {1.0f, -1.0f, P(-30)}},
      {"C=2^24, products sum 1.5 (RN-even: 2^24+2)", 16777216.0f, {1.0f, 0.5f}},
      {"C=-1, one product -1.5*2^-24 (RN: -(1+2^-23); truncation toward zero: -1)", -1.0f, {-1.5f * P(-24)}},
  };
  const int ncase = sizeof(cases) / sizeof(cases[0]);
  for (int t = 0; t < ncase; ++t) {
    for (int k = 0; k < 16; ++k) {   // product p = a*b with a = p's mantissa part, b = power of two
      const float p = cases[t].p[k];
      int e; const float m = p == 0.f ? 0.f : frexpf(p, &e);   // p = m * 2^e, m in [0.5,1): up to 11 significant bits wanted
      const int ea = p == 0.f ? 0 : e / 2, eb = p == 0.f ? 0 : e - e / 2;
      ha[t * 16 + k] = ldexpf(m, ea); hb[t * 16 + k] = p == 0.f ? 0.f : ldexpf(1.0f, eb);
      if (f16r(ha[t * 16 + k]) != ha[t * 16 + k] || f16r(hb[t * 16 + k]) != hb[t * 16 + k]) printf("  (case %d: operand %d not f16-exact)\n", t, k);
    }
    hc[t] = cases[t].c;
  }
  run(ncase);
  for (int t = 0; t < ncase; ++t) {
    double ex = cases[t].c;
    for (int k = 0; k < 16; ++k) ex += (double)ha[t * 16 + k] * hb[t * 16 + k];
    printf("%-110s D = %.10g (%a)   exact %.12g   RN(exact) %.10g\n", cases[t].what, ho[t], ho[t], ex, (double)(float)ex);
  }
  // ---- statistics: random operands, C of the magnitude of the running sum of a long dot product
  for (int mode = 0; mode < 3; ++mode) {
    srand(1234);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.0f - 1.0f; };
    for (int t = 0; t < NT; ++t) {
      for (int k = 0; k < 16; ++k) { ha[t * 16 + k] = f16r(rnd()); hb[t * 16 + k] = f16r(rnd() * (mode == 2 ? 4.9e-4f : 1.0f)); }
      hc[t] = mode == 0 ? 0.0f : rnd() * 20.0f;     // mode 1/2: |C| ~ 10: a partial sum of ~300 terms; mode 2: the products are a lo x hi term (2^-11 smaller)
    }
    run(NT);
    double mean = 0, rms = 0, meanabs_dir = 0; int n = 0, beyond = 0;
    for (int t = 0; t < NT; ++t) {
      double ex = hc[t];
      for (int k = 0; k < 16; ++k) ex += (double)ha[t * 16 + k] * hb[t * 16 + k];
      const double ulp = ldexp(1.0, ilogb(fabs((double)ho[t])) - 23);
      const double e = ((double)ho[t] - ex) / ulp;
      mean += e; rms += e * e; meanabs_dir += (ex >= 0 ? e : -e); ++n;
      if (fabs(e) > 0.5000001) ++beyond;
    }
    printf("random operands, %s: error of D in ulps of D: mean %+.4f  rms %.4f  mean towards-infinity-of-|D| %+.4f  |e| > 0.5 ulp in %.3f %% of %d\n",
           mode == 0 ? "C = 0, products O(1)" : (mode == 1 ? "|C| ~ 10, products O(1)" : "|C| ~ 10, products O(2^-11) (a lo x hi term)"),
           mean / n, sqrt(rms / n), meanabs_dir / n, 100.0 * beyond / n, n);
  }
  return 0;
}
