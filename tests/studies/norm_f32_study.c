/* Study (CPU): can the fused path's IPD normalisation  y = (float)(((double)x - r) / ipd)  be computed with f32 instructions only,
 * bit for bit, if a certificate sends the rare ambiguous elements to the f64 path?  (encoder_heads_f16x2.hip: the six f64
 * instructions per element are not hidden by f16 MFMAs and cost 7.5 % of a tile.)
 *   build: gcc -O2 -ffp-contract=off -o /tmp/norm_study tests/studies/norm_f32_study.c -lm
 * Prints the number of elements where the certified fast result differs from the truth (must be 0) and the fallback rate. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static uint64_t s = 88172645463325252ull;
static uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static float urand(float lo, float hi) { return lo + (hi - lo) * (float)((rnd() >> 40) * (1.0 / 16777216.0)); }

#ifndef THR
#define THR 0.99998474f   /* 1 - 2^-16 */
#endif
typedef struct { float dh, dl, y0; double d; } Face;

static Face face_of(double d) {
  Face f; f.d = d; f.dh = (float)d; f.dl = (float)(d - (double)f.dh); f.y0 = 1.0f / f.dh; return f;
}

/* returns 1 if certified, result in *y */
static int fast_norm(float x, float r, const Face* f, float* y) {
  /* TwoSum(x, -r): th + tl == x - r exactly */
  const float nr = -r;
  const float th = x + nr;
  const float bb = th - x;
  const float tl = (x - (th - bb)) + (nr - bb);
  const float q0 = th * f->y0;
  const float r0 = fmaf(-q0, f->dh, th);
  const float r1 = fmaf(-q0, f->dl, r0 + tl);
  const float corr = r1 * f->y0;
  const float yy = q0 + corr;
  const float e = (q0 - yy) + corr;                 /* FastTwoSum error: |q0| >= |corr| */
  uint32_t b; memcpy(&b, &yy, 4);
  const uint32_t ex = b & 0x7f800000u;
  uint32_t hb = ex - (24u << 23);                   /* half an ulp of yy: 2^(e-24) */
  float half; memcpy(&half, &hb, 4);
  const float thr = half * THR;                     /* (1 - 2^-k): anything within 2^-(k+1) ulp of a midpoint is sent to f64 */
  /* a power of two has a half-ulp of half the size below it: not worth a case, send it to f64 (one result in 2^23) */
  const int normal = ex > (40u << 23) && ex < (250u << 23) && (b & 0x007fffffu) != 0u;
  const int ok = (fabsf(e) <= thr) && normal;
  *y = yy;
  return ok || (th == 0.0f && tl == 0.0f && (*y = 0.0f, 1));
}

static float truth(float x, float r, double d) { return (float)(((double)x - (double)r) / d); }

int main(void) {
  long bad = 0, amb = 0, n = 0;
  for (int face = 0; face < 200000; ++face) {
    /* ipd as the kernel forms it: sqrt of a sum of squares of f32 differences, or tiny / huge faces */
    const float ax = urand(0, 1), ay = urand(0, 1), az = urand(0, 1), bx = urand(0, 1), by = urand(0, 1), bz = urand(0, 1);
    double sc = 1.0;
    if (face % 7 == 1) sc = 1e-3; else if (face % 7 == 2) sc = 1920.0; else if (face % 7 == 3) sc = 1e-6;
    const double dx = (double)(float)(ax * sc) - (double)(float)(bx * sc), dy = (double)(float)(ay * sc) - (double)(float)(by * sc),
                 dz = (double)(float)(az * sc) - (double)(float)(bz * sc);
    double d = sqrt(fma(dz, dz, fma(dy, dy, dx * dx)));
    if (d == 0.0) d = 1e-6;
    const Face f = face_of(d);
    const float r = (float)(urand(0, 1) * sc);
    for (int i = 0; i < 300; ++i) {
      float x = (float)(urand(0, 1) * sc);
      if (i % 50 == 0) x = r;                        /* exact zero */
      if (i % 50 == 1) x = nextafterf(r, 2 * r + 1); /* one ulp off */
      float y; const int ok = fast_norm(x, r, &f, &y);
      const float t = truth(x, r, d);
      ++n;
      if (!ok) { ++amb; continue; }
      uint32_t a, b; memcpy(&a, &y, 4); memcpy(&b, &t, 4);
      if (a != b && !(y == 0.0f && t == 0.0f)) { if (bad < 10) printf("MISMATCH x=%a r=%a d=%a fast=%a truth=%a\n", x, r, d, y, t); ++bad; }
    }
  }
  /* adversarial: quotients constructed next to f32 rounding midpoints: x - r = (m + 0.5 +- eps) * ulp * d */
  for (int k = 0; k < 2000000; ++k) {
    const float dh = urand(0.05f, 1.5f);
    const double d = (double)dh + (double)urand(-1, 1) * 1e-9 * dh;
    const Face f = face_of(d);
    float q = urand(0.01f, 40.0f);
    const float qn = nextafterf(q, 100.0f);
    const double mid = 0.5 * ((double)q + (double)qn) * (1.0 + ((double)(rnd() % 2001) - 1000.0) * 1e-16 * (double)(rnd() % 64));
    const double tt = mid * d;                       /* wanted x - r */
    const float r = urand(0, 1);
    const float x = (float)(tt + (double)r);
    float y; const int ok = fast_norm(x, r, &f, &y);
    const float t = truth(x, r, d);
    ++n;
    if (!ok) { ++amb; continue; }
    uint32_t a, b; memcpy(&a, &y, 4); memcpy(&b, &t, 4);
    if (a != b) { if (bad < 10) printf("MISMATCH(adv) x=%a r=%a d=%a fast=%a truth=%a\n", x, r, d, y, t); ++bad; }
  }
  /* adversarial 2: pick x, r, then the divisor that puts the quotient k * 2^-53 (relative) beside an f32 rounding midpoint */
  for (int k = 0; k < 20000000; ++k) {
    const float r = urand(0, 1), x = urand(0, 1);
    const double t = (double)x - (double)r;
    if (t == 0.0) continue;
    float q = urand(0.01f, 40.0f);
    const float qn = nextafterf(q, 100.0f);
    const double mid = 0.5 * ((double)q + (double)qn) * (t < 0 ? -1.0 : 1.0);
    /* +-(1 .. 2^22) ulp64 away from the midpoint, log-uniform: from far inside the certificate's margin (2^-41 relative)
     * to well outside it */
    const int sh = (int)(rnd() % 23);
    const long off = ((long)(rnd() % ((2l << sh) + 1)) - (1l << sh));
    double d = t / mid;
    d = d * (1.0 + (double)off * 1.1102230246251565e-16);
    if (!(d > 1e-7 && d < 1e4)) continue;
    const Face f = face_of(d);
    float y; const int ok = fast_norm(x, r, &f, &y);
    const float tr = truth(x, r, d);
    ++n;
    if (!ok) { ++amb; continue; }
    uint32_t a, b; memcpy(&a, &y, 4); memcpy(&b, &tr, 4);
    if (a != b) { if (bad < 10) printf("MISMATCH(adv2) x=%a r=%a d=%a fast=%a truth=%a\n", x, r, d, y, tr); ++bad; }
  }
  printf("elements %ld  certified-but-wrong %ld  sent to f64 %ld (%.3g %%)\n", n, bad, amb, 100.0 * amb / n);
  return bad != 0;
}
