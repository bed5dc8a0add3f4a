// tucker_common.h -- the shared body of K3 (tucker_objective.hip) and of every round of the
// device-side Powell minimiser (tucker_powell.hip): EV objective evaluations by one workgroup.
//
//   f-vectors   f32(a*cos(b*w+c)+d) in f64                      (TD_Tester.py:25-28,:36-43)
//   c[q][e]     = ((u_i * f_yj) * f_pk) * f_rl, q = ((i*3+j)*3+k)*3+l, in LDS
//   x_hat[m]    = sum_q c[q] * Wm[q][m]   one fma chain per (evaluation, column), q ascending   (:46)
//   err         = 0.5 * sum_m (x[m]-x_hat[m])^2   fixed-order reduction                         (:49)
//
// Workgroup = TNT (512) threads; thread t owns columns t, t+512, t+1024 (< 1404) for all EV = 8
// evaluations: 24 f64 accumulators.  Per row q of Wm the block reads the row once from L2 (coalesced
// dwords, f32 -> f64) and the 8 coefficients as an LDS broadcast, then issues 24 v_fma_f64: Wm
// traffic (758 KB per pass) is amortised over the 8 evaluations.  8 waves per workgroup and 2
// workgroups per CU give the 16 waves that hide the L2 latency of the dependent row loads.
//
// Reduction order (the C oracle's device_order mode replays it bit for bit):
//   thread: fma chain over its columns in ascending order; wave: xor butterfly, offsets 32..1;
//   block: ((w0+w1)+(w2+w3)) + ((w4+w5)+(w6+w7)); then * 0.5.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/nlml_hpe.h"

namespace nlml {

constexpr int TQ = NLML_TUCKER_Q;       // 135 rows of Wm
constexpr int TM = NLML_F_REFERENCE;    // 1404 columns
constexpr int EV = 8;                   // evaluations per workgroup
constexpr int TNT = 512;                // threads per workgroup
constexpr int CPT = (TM + TNT - 1) / TNT;  // 3 columns per thread
constexpr int TNW = TNT / 64;           // 8 waves

struct TuckerShared {
  double coef[TQ][EV];
  double fvec[EV][3][3];
  double red[TNW][EV];
};

// par: LDS or global, EV rows of 8 doubles (w_y, w_p, w_r, u_id[5]).  cp4: this thread's cosine row
// (threads < 72 only).  Leaves acc[e][j] = x_hat of evaluation e at column tid + TNT*j.
template <typename ParT>
__device__ __forceinline__ void tucker_xhat(TuckerShared& sh, const float* __restrict__ Wm, const ParT& par,
                                            const double (&cp4)[4], int tid, double (&acc)[EV][CPT]) {
  if (tid < EV * 9) {
    const int e = tid / 9, a = (tid % 9) / 3;
    const double v = cp4[0] * cos(cp4[1] * par(e, a) + cp4[2]) + cp4[3];
    sh.fvec[e][a][tid % 3] = (double)(float)v;                     // .astype(np.float32), :37,40,43
  }
  __syncthreads();
  for (int i = tid; i < TQ * EV; i += TNT) {
    const int q = i / EV, e = i % EV;
    const int ui = q / 27, j = (q / 9) % 3, k = (q / 3) % 3, l = q % 3;
    sh.coef[q][e] = ((par(e, 3 + ui) * sh.fvec[e][0][j]) * sh.fvec[e][1][k]) * sh.fvec[e][2][l];
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < EV; ++e)
#pragma unroll
    for (int j = 0; j < CPT; ++j) acc[e][j] = 0.0;
  const bool last_ok = tid + TNT * (CPT - 1) < TM;
#pragma unroll 3
  for (int q = 0; q < TQ; ++q) {
    const float* wr = Wm + (size_t)q * TM + tid;
    double w[CPT];
#pragma unroll
    for (int j = 0; j < CPT - 1; ++j) w[j] = (double)wr[TNT * j];
    w[CPT - 1] = last_ok ? (double)wr[TNT * (CPT - 1)] : 0.0;
#pragma unroll
    for (int e = 0; e < EV; ++e) {
      const double c = sh.coef[q][e];
#pragma unroll
      for (int j = 0; j < CPT; ++j) acc[e][j] = fma(c, w[j], acc[e][j]);
    }
  }
}

// Residual norms of the EV evaluations; xv[e][j] = x of evaluation e at this thread's column j.
// After the call (and the barrier inside) every thread may read err_of(e).
__device__ __forceinline__ void tucker_residual(TuckerShared& sh, const float (&xv)[EV][CPT],
                                                const double (&acc)[EV][CPT], int tid) {
#pragma unroll
  for (int e = 0; e < EV; ++e) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      if (tid + TNT * j < TM) {
        const double r = (double)xv[e][j] - acc[e][j];
        s = fma(r, r, s);
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((tid & 63) == 0) sh.red[tid >> 6][e] = s;
  }
  __syncthreads();
}

__device__ __forceinline__ double tucker_err(const TuckerShared& sh, int e) {
  return 0.5 * (((sh.red[0][e] + sh.red[1][e]) + (sh.red[2][e] + sh.red[3][e])) +
                ((sh.red[4][e] + sh.red[5][e]) + (sh.red[6][e] + sh.red[7][e])));
}

}  // namespace nlml
