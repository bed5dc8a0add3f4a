/* oracle.c -- plain-C CPU restatement of the NLML_HPE hot path.  TEST INFRASTRUCTURE ONLY
 * (see oracle/__init__.py): loaded by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg, never by the product package.
 *
 * Each function cites the reference lines it follows (paths under /root/reference).
 * Unlike the numpy oracle, summation ORDER is fixed here, and selectable to equal the order the
 * HIP kernel's MFMA chains use, so GPU results can be compared bit for bit wherever the
 * arithmetic is pure fma (everything up to the Tanh).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- helpers/FeatureExtractor.py:30-66 + the .float() at :101 ---------------------------- */
void oracle_normalize_ipd(const float* raw, int64_t B, int normalize, float* out) {
  for (int64_t b = 0; b < B; ++b) {
    const float* p = raw + b * 1404;
    float* o = out + b * 1404;
    if (!normalize) { memcpy(o, p, 1404 * sizeof(float)); continue; }
    const double ref[3] = {p[3], p[4], p[5]};                                /* landmark 1, :85-86 */
    const double dx = (double)p[99] - (double)p[789];                        /* 33 vs 263, :38-43 */
    const double dy = (double)p[100] - (double)p[790];
    const double dz = (double)p[101] - (double)p[791];
    double ipd = sqrt(fma(dz, dz, fma(dy, dy, dx * dx)));  /* np.linalg.norm == sqrt(ddot): fma chain on this BLAS */
    if (ipd == 0.0) ipd = 1e-6;                                              /* :47-48 */
    for (int k = 0; k < 1404; ++k) o[k] = (float)(((double)p[k] - ref[k % 3]) / ipd);   /* :55-61 */
  }
}

/* One Linear layer for one row.  order 0: k ascending; order 1: the HIP kernels' MFMA chain order -- within each group of 8:
 * k, k+4, k+1, k+5, k+2, k+6, k+3, k+7 (layout.h).  blk > 0 (a multiple of 8): the K-BLOCKED sum of the f32 kernel's layers 0
 * to 3 (encoder_heads.hip fold_block): one chain per block of blk k-values in the order above, the first from the bias, the
 * others from +0.0, and the block sums added up in block order starting from +0.0: ((0 + s_0) + s_1) + ... */
static float chain8(const float* w, const float* x, int k0, int k1, int K, float acc) {
  for (; k0 + 8 <= k1; k0 += 8)
    for (int j = 0; j < 4; ++j) {
      acc = fmaf(w[k0 + j], x[k0 + j], acc);
      acc = fmaf(w[k0 + 4 + j], x[k0 + 4 + j], acc);
    }
  if (k0 < k1) /* zero-padded tail step: padded products are exact zeros and change nothing */
    for (int j = 0; j < 4; ++j) {
      if (k0 + j < K) acc = fmaf(w[k0 + j], x[k0 + j], acc);
      if (k0 + 4 + j < K) acc = fmaf(w[k0 + 4 + j], x[k0 + 4 + j], acc);
    }
  return acc;
}

static void linear_row(const float* x, const float* W, const float* b, int K, int N, float* y, int order, int blk) {
  for (int n = 0; n < N; ++n) {
    const float* w = W + (size_t)n * K;
    if (order == 0) {
      float acc = b[n];
      for (int k = 0; k < K; ++k) acc = fmaf(w[k], x[k], acc);
      y[n] = acc;
    } else if (blk <= 0) {
      y[n] = chain8(w, x, 0, K, K, b[n]);
    } else {
      float tot = 0.0f;
      for (int k0 = 0; k0 < K; k0 += blk) {
        const int k1 = k0 + blk < K ? k0 + blk : K;
        tot += chain8(w, x, k0, k1, K, k0 == 0 ? b[n] : 0.0f);
      }
      y[n] = tot;
    }
  }
}

/* ---- NLML_HPE_Model_Builder.py:55-68 (encoder), :104-105 (heads), :115-126 (combined) -----
 * enc_w[i] [out,in] for widths F->1024->512->256->128->64->9; head_w[3*5] in (yaw,pitch,roll) x
 * layer order with widths 3->128->256->128->64->1.  out [B,3] radians; latent [B,9] and
 * pre_tanh [B,64] optional (NULL to skip). */
void oracle_encoder_heads_f32(const float* x, int64_t B, int F, const float* const* enc_w,
                              const float* const* enc_b, const float* const* head_w,
                              const float* const* head_b, int order, float* out, float* latent,
                              float* pre_tanh) {
  static const int EN[7] = {0, 1024, 512, 256, 128, 64, 9};
  static const int HN[6] = {3, 128, 256, 128, 64, 1};
#pragma omp parallel
  {
    float* a = (float*)malloc(sizeof(float) * (F > 1024 ? F : 1024));
    float* c = (float*)malloc(sizeof(float) * 1024);
#pragma omp for schedule(static)
    for (int64_t r = 0; r < B; ++r) {
      memcpy(a, x + r * F, sizeof(float) * F);
      int K = F;
      for (int l = 0; l < 6; ++l) {
        const int N = EN[l + 1];
        linear_row(a, enc_w[l], enc_b[l], K, N, c, order, (order == 2 && l < 4) ? 128 : 0);   /* order 2: layers 0..3 K-blocked */
        if (l == 4 && pre_tanh) memcpy(pre_tanh + r * 64, c, sizeof(float) * 64);
        for (int n = 0; n < N; ++n)
          a[n] = (l < 4) ? (c[n] < 0.0f ? 0.0f : c[n]) : (l == 4 ? tanhf(c[n]) : c[n]);   /* ReLU x4 (NaN propagates, like torch.relu), Tanh, none */
        K = N;
      }
      float lat[9];
      memcpy(lat, a, sizeof lat);
      if (latent) memcpy(latent + r * 9, lat, sizeof lat);
      for (int g = 0; g < 3; ++g) {                          /* latent[:, 3g:3g+3] -> head g, :118-124 */
        memcpy(a, lat + 3 * g, 3 * sizeof(float));
        int Kh = 3;
        for (int l = 0; l < 5; ++l) {
          const int N = HN[l + 1];
          linear_row(a, head_w[g * 5 + l], head_b[g * 5 + l], Kh, N, c, order, 0);
          for (int n = 0; n < N; ++n) a[n] = (l < 4) ? (c[n] < 0.0f ? 0.0f : c[n]) : c[n];
          Kh = N;
        }
        out[r * 3 + g] = a[0];
      }
    }
    free(a);
    free(c);
  }
}

/* ---- TD_Tester.py:25-28 (func), :31-58 (objective) ---------------------------------------
 * Wm f32[135,1404]; x f32[N,1404]; params f64[N,8]; cosp f64[3,3,4]; err f64[N]; xhat f64[N,1404]|NULL.
 * x_hat[m] = sum_q c[q]*Wm[q][m], q ascending in one fma chain (the HIP kernel's order). */
/* device_order != 0: the residual is summed in the HIP kernel's order (tucker_objective.hip /
 * tucker_powell.hip): the MFMA tiling's order, see residual_device_order; the result is then bit-identical to the
 * GPU's, which lets the device-side Powell run be replayed exactly on the CPU. */
static double residual_device_order(const float* xrow, const double* acc) {
  /* tucker_common.h: 8 waves x 176 columns from tcol0(w) = 176 w (the last wave: 1228; its first 4 columns belong to wave 6 and
   * are skipped); lane column c of wave w sums its 11 columns tcol0 + tlcol(c, mb) (fma chain, mb ascending), where
   * tlcol(c, mb) = 64 (mb/4) + 4 c + mb%4 for mb < 8 and 128 + 3 c + (mb - 8) above; xor butterfly over the 16 lanes (offsets
   * 1,2,4,8), then the 8 waves. */
  double red[8];
  for (int w = 0; w < 8; ++w) {
    const int base = w < 7 ? 176 * w : 1404 - 176;
    double v[16], n[16];
    for (int c = 0; c < 16; ++c) {
      double s = 0.0;
      for (int mb = 0; mb < 11; ++mb) {
        const int lc = mb < 8 ? 64 * (mb >> 2) + 4 * c + (mb & 3) : 128 + 3 * c + (mb - 8);
        const int m = base + lc;
        if (w < 7 || lc >= 8 * 176 - 1404) { const double r = (double)xrow[m] - acc[m]; s = fma(r, r, s); }
      }
      v[c] = s;
    }
    for (int off = 1; off < 16; off <<= 1) {
      for (int c = 0; c < 16; ++c) n[c] = v[c] + v[c ^ off];
      memcpy(v, n, sizeof v);
    }
    red[w] = v[0];
  }
  return 0.5 * (((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7])));
}

/* numpy's pairwise summation (numpy/_core/src/umath/loops_utils.h.src, DOUBLE_pairwise_sum; what np.sum at TD_Tester.py:49 runs on
 * a contiguous f64 vector): blocks of at most 128 elements are summed in eight strided partial sums, combined as
 * ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), the remainder added one by one; longer ranges split at n/2 rounded down to a multiple of 8. */
static double np_pairwise_sum(const double* a, int n) {
  if (n < 8) {
    double res = 0.0;
    for (int i = 0; i < n; ++i) res += a[i];
    return res;
  }
  if (n <= 128) {
    double r[8];
    int i;
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

/* The objective in the REFERENCE's own operation order (TD_Tester.py:46,49): np.einsum('ijklm,i,j,k,l->m') with five operands
 * runs numpy's generic sum-of-products loop -- for every (i,j,k,l), in that nesting order, and every m:
 *     x_hat[m] = ((((W[i,j,k,l,m] * u_i) * f_yj) * f_pk) * f_rl) + x_hat[m]
 * every operation rounded to f64 on its own (no fused multiply-add) -- then 0.5 * np.sum((x - x_hat)**2) with the pairwise sum
 * above.  Bit-identical to FX4's err and x_hat (tests/test_oracle_golden.py). */
void oracle_tucker_objective_reforder(const float* Wm, const float* x, const double* params, const double* cosp,
                                      int64_t N, double* err, double* xhat) {
#pragma omp parallel for schedule(static)
  for (int64_t n = 0; n < N; ++n) {
    const double* p = params + n * 8;
    double f[3][3];
    for (int a = 0; a < 3; ++a)
      for (int j = 0; j < 3; ++j) {
        const double* cp = cosp + (a * 3 + j) * 4;
        f[a][j] = (double)(float)(cp[0] * cos(cp[1] * p[a] + cp[2]) + cp[3]);
      }
    double acc[1404], d2[1404];
    for (int m = 0; m < 1404; ++m) acc[m] = 0.0;
    for (int q = 0; q < 135; ++q) {
      const double u = p[3 + q / 27], fy = f[0][(q / 9) % 3], fp = f[1][(q / 3) % 3], fr = f[2][q % 3];
      const float* w = Wm + (size_t)q * 1404;
      for (int m = 0; m < 1404; ++m) {
        double t = (double)w[m] * u;
        t = t * fy;
        t = t * fp;
        t = t * fr;
        acc[m] = t + acc[m];
      }
    }
    for (int m = 0; m < 1404; ++m) {
      const double d = (double)x[n * 1404 + m] - acc[m];
      d2[m] = d * d;
      if (xhat) xhat[n * 1404 + m] = acc[m];
    }
    err[n] = 0.5 * np_pairwise_sum(d2, 1404);
  }
}

void oracle_tucker_objective(const float* Wm, const float* x, const double* params, const double* cosp,
                             int64_t N, double* err, double* xhat, int device_order) {
#pragma omp parallel for schedule(static)
  for (int64_t n = 0; n < N; ++n) {
    const double* p = params + n * 8;
    double f[3][3];
    for (int a = 0; a < 3; ++a)
      for (int j = 0; j < 3; ++j) {
        const double* cp = cosp + (a * 3 + j) * 4;
        f[a][j] = (double)(float)(cp[0] * cos(cp[1] * p[a] + cp[2]) + cp[3]);   /* :26, .astype(f32) :37 */
      }
    double c[135];
    for (int q = 0; q < 135; ++q)
      c[q] = ((p[3 + q / 27] * f[0][(q / 9) % 3]) * f[1][(q / 3) % 3]) * f[2][q % 3];
    double s = 0.0;
    double accv[1404];
    for (int m = 0; m < 1404; ++m) {
      double acc = 0.0;
      for (int q = 0; q < 135; ++q) acc = fma(c[q], (double)Wm[(size_t)q * 1404 + m], acc);
      if (xhat) xhat[n * 1404 + m] = acc;
      accv[m] = acc;
      const double r = (double)x[n * 1404 + m] - acc;
      s += r * r;
    }
    err[n] = device_order ? residual_device_order(x + n * 1404, accv) : 0.5 * s;   /* :49 */
  }
}
