// tucker_ref.h -- the Tucker objective in the REFERENCE'S OWN OPERATION ORDER (NLML_TD_ORDER_REFERENCE).
//
// The fast path (tucker_common.h) evaluates x_hat = c^T Wm as a GEMM on the f64 matrix cores: c = ((u*f_y)*f_p)*f_r first, then one
// fma chain per output.  That agrees with the reference's objective to ~1e-15 relative -- and still moves the END POINT of the
// Powell minimisation by up to ~2e-2 degree, because the minimum is flat and Powell's termination is rounding-sensitive
// (tests/test_powell_sm.py measures scipy itself doing that).  This file evaluates the objective exactly as the reference does
// (TD_Tester.py:46,49), bit for bit:
//
//   np.einsum('ijklm,i,j,k,l->m', W, u, f_y, f_p, f_r)   numpy's generic sum-of-products loop: for (i,j,k,l) in nesting order,
//       for every m:  x_hat[m] = ((((W[i,j,k,l,m] * u_i) * f_yj) * f_pk) * f_rl) + x_hat[m],  each operation rounded on its own;
//   0.5 * np.sum((x - x_hat)**2)                         numpy's pairwise sum: 16 leaves of 80/88/92 elements, each as eight
//       strided partial sums combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus remainder, the leaves added as a balanced tree.
//
// The products cannot be shared between columns (W sits innermost), so this costs 5 f64 operations per (q, m) instead of one
// fma: ~948 kFLOP-equivalents per evaluation on the vector ALUs, ~25x the matrix-core path's time per evaluation.  It is the
// parity mode of the TD path (oracle: oracle/csrc/oracle.c oracle_tucker_objective_reforder, pinned to FX4 bit for bit);
// the matrix-core mode stays the default for throughput.
//
// One workgroup of 512 threads, thread t owns columns t, t+512, t+1024; NE (1, 2, 4 or 8) evaluations share every Wm load.
#pragma once
#include <hip/hip_runtime.h>

#include "tucker_common.h"

namespace nlml {

constexpr int TR_COLS = 3;                 // columns per thread: 3 * 512 = 1536 >= 1404
constexpr int TR_MAXE = 8;                 // evaluations per pass over Wm
constexpr int TR_LEAVES = 16;              // numpy pairwise tree for n = 1404 (tests/test_oracle_golden.py pins it to np.sum)

struct TuckerRefShared {
  double d2[TR_MAXE][TM + 4];              // squared residuals of the pass's evaluations
  double leaf8[TR_MAXE][TR_LEAVES][8];     // strided partial sums of every leaf
  double leaf[TR_MAXE][TR_LEAVES];
  double err[EV];                          // objective values of the round, by machine slot
};

__device__ __forceinline__ int tr_leaf_start(int L) { return L == 0 ? 0 : 80 + 88 * (L - 1); }
__device__ __forceinline__ int tr_leaf_len(int L) { return L == 0 ? 80 : (L == TR_LEAVES - 1 ? 92 : 88); }

// Evaluations ev[0..NE) (machine slots, f-vectors in sh.fvec[slot], parameters par(slot, k)) against the rows xe[i]; results
// into rs.err[ev[i]] (and x_hat rows if xh[i] != nullptr).  All 512 threads; barriers inside.  `store[i]` false = padding.
template <int NE, typename ParT>
__device__ __attribute__((noinline)) void tucker_ref_pass(const TuckerShared& sh, TuckerRefShared& rs, const float* __restrict__ Wm,
                                                          const ParT& par, const int (&ev)[NE], const float* const (&xe)[NE],
                                                          double* const (&xh)[NE], int tid) {
#pragma clang fp contract(off)
  int mc[TR_COLS];
  bool livec[TR_COLS];
#pragma unroll
  for (int c = 0; c < TR_COLS; ++c) {
    const int m = tid + TNT * c;
    livec[c] = m < TM;
    mc[c] = livec[c] ? m : TM - 1;
  }
  double acc[TR_COLS][NE];
#pragma unroll
  for (int c = 0; c < TR_COLS; ++c)
#pragma unroll
    for (int n = 0; n < NE; ++n) acc[c][n] = 0.0;

  // (i, j, k, l) in the einsum's nesting order; the factor of a level is re-read from LDS when that level advances
#pragma unroll 1
  for (int i = 0; i < 5; ++i) {
    double u[NE];
#pragma unroll
    for (int n = 0; n < NE; ++n) u[n] = par(ev[n], 3 + i);
#pragma unroll 1
    for (int j = 0; j < 3; ++j) {
      double fy[NE];
#pragma unroll
      for (int n = 0; n < NE; ++n) fy[n] = sh.fvec[ev[n]][0][j];
#pragma unroll 1
      for (int k = 0; k < 3; ++k) {
        double fp[NE];
#pragma unroll
        for (int n = 0; n < NE; ++n) fp[n] = sh.fvec[ev[n]][1][k];
        const int q0 = ((i * 3 + j) * 3 + k) * 3;
        float w[3][TR_COLS];
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
          for (int c = 0; c < TR_COLS; ++c) w[l][c] = gload<float>(Wm + (size_t)(q0 + l) * TM + mc[c]);
#pragma unroll
        for (int l = 0; l < 3; ++l) {
#pragma unroll
          for (int n = 0; n < NE; ++n) {
            const double fr = sh.fvec[ev[n]][2][l];
#pragma unroll
            for (int c = 0; c < TR_COLS; ++c) {
              double t = (double)w[l][c] * u[n];
              t = t * fy[n];
              t = t * fp[n];
              t = t * fr;
              acc[c][n] = t + acc[c][n];
            }
          }
        }
      }
    }
  }
  // residuals squared -> LDS (and x_hat out)
#pragma unroll
  for (int n = 0; n < NE; ++n)
#pragma unroll
    for (int c = 0; c < TR_COLS; ++c)
      if (livec[c]) {
        const double d = (double)gload<float>(xe[n] + mc[c]) - acc[c][n];
        rs.d2[n][mc[c]] = d * d;
        if (xh[n]) xh[n][mc[c]] = acc[c][n];
      }
  __syncthreads();
  // numpy's pairwise sum, level 1: eight strided partial sums per leaf
  for (int t = tid; t < NE * TR_LEAVES * 8; t += TNT) {
    const int n = t / (TR_LEAVES * 8), L = (t / 8) % TR_LEAVES, jj = t % 8;
    const double* a = rs.d2[n] + tr_leaf_start(L);
    const int len = tr_leaf_len(L), body = len - (len % 8);
    double r = a[jj];
    for (int i2 = 8 + jj; i2 < body; i2 += 8) r += a[i2];
    rs.leaf8[n][L][jj] = r;
  }
  __syncthreads();
  // level 2: combine the eight, then the remainder elements one by one
  for (int t = tid; t < NE * TR_LEAVES; t += TNT) {
    const int n = t / TR_LEAVES, L = t % TR_LEAVES;
    const double* r = rs.leaf8[n][L];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    const double* a = rs.d2[n] + tr_leaf_start(L);
    const int len = tr_leaf_len(L);
    for (int i2 = len - (len % 8); i2 < len; ++i2) res += a[i2];
    rs.leaf[n][L] = res;
  }
  __syncthreads();
  // level 3: the balanced tree over the 16 leaves, then * 0.5
  if (tid < NE) {
    const double* s = rs.leaf[tid];
    const double a0 = (s[0] + s[1]) + (s[2] + s[3]), a1 = (s[4] + s[5]) + (s[6] + s[7]);
    const double a2 = (s[8] + s[9]) + (s[10] + s[11]), a3 = (s[12] + s[13]) + (s[14] + s[15]);
    rs.err[ev[tid]] = 0.5 * ((a0 + a1) + (a2 + a3));
  }
  __syncthreads();
}

// f-vectors of all 16 slots into sh.fvec (tucker_coef's first half; the coefficient table is not used in this order)
template <typename ParT>
__device__ __forceinline__ void tucker_fvec(TuckerShared& sh, const ParT& par, const double (&cp4)[4], int tid) {
  if (tid < EV * 9) {
    const int e = tid / 9, a = (tid % 9) / 3;
    const double v = cp4[0] * cos(cp4[1] * par(e, a) + cp4[2]) + cp4[3];
    sh.fvec[e][a][tid % 3] = (double)(float)v;
  }
  __syncthreads();
}

// The evaluations whose bit is set in `mask` (machine slots), in passes of 8 / 4 / 2 / 1.  xrow(slot) -> that slot's x row;
// xhrow(slot) -> its x_hat output row or nullptr.
template <typename ParT, typename XRow, typename XhRow>
__device__ __forceinline__ void tucker_ref_eval(const TuckerShared& sh, TuckerRefShared& rs, const float* __restrict__ Wm,
                                                const ParT& par, int mask, XRow xrow, XhRow xhrow, int tid) {
  while (mask) {
    const int cnt = __popc(mask);
    int slots[TR_MAXE];
    int take = cnt >= 8 ? 8 : (cnt >= 4 ? 4 : (cnt >= 2 ? 2 : 1));
    for (int i = 0; i < take; ++i) {
      slots[i] = __ffs(mask) - 1;
      mask &= mask - 1;
    }
#define NLML_REF_CASE(K)                                                                   \
  case K: {                                                                                \
    int ek[K];                                                                             \
    const float* xk[K];                                                                    \
    double* hk[K];                                                                         \
    for (int i = 0; i < K; ++i) { ek[i] = slots[i]; xk[i] = xrow(slots[i]); hk[i] = xhrow(slots[i]); } \
    tucker_ref_pass<K>(sh, rs, Wm, par, ek, xk, hk, tid);                                  \
    break;                                                                                 \
  }
    switch (take) {
      NLML_REF_CASE(8) NLML_REF_CASE(4) NLML_REF_CASE(2) NLML_REF_CASE(1)
      default: break;
    }
#undef NLML_REF_CASE
  }
}

}  // namespace nlml
