// tucker_objective.hip -- K3: batched Tucker-einsum objective in f64.
//
// Replaces objective() (TD_Tester.py:31-58) for N face-evaluations at once:
//   f_y[j] = f32(a*cos(b*w_y+c)+d) etc.            (:25-28,:36-43; f64 cosine, rounded to f32)
//   x_hat[m] = sum_{ijkl} W[i,j,k,l,m] u_i f_yj f_pk f_rl   (:46, einsum promotes to f64)
//   err = 0.5 * sum_m (x[m] - x_hat[m])^2          (:49)
// restated as c = u (x) f_y (x) f_p (x) f_r (135 f64 coefficients per evaluation) and
// x_hat = c^T Wm, Wm = W.reshape(135,1404): a [N,135] x [135,1404] f64 GEMM fused with the
// residual norm.
//
// Workgroup = 256 threads = EV (8) evaluations x all 1404 columns; thread t owns columns
// t, t+256, ... (6 per thread) for all 8 evaluations (48 f64 accumulators).  Per row q of Wm
// the block reads the row once from L2 (coalesced dwords, f32 -> f64) and the 8 coefficients
// c[q][0..7] as an LDS broadcast, then issues 48 v_fma_f64: Wm traffic is amortised over 8
// evaluations, coefficients never leave the CU.  Sums run q = 0..134 ascending in one fma chain
// per (evaluation, column), and the residual reduction is a fixed-order tree, so results are
// run-to-run deterministic.
#include <hip/hip_runtime.h>

#include "../../include/nlml_hpe.h"
#include "abi_internal.h"

namespace nlml {

constexpr int TQ = NLML_TUCKER_Q;       // 135
constexpr int TM = NLML_F_REFERENCE;    // 1404
constexpr int EV = 8;                   // evaluations per workgroup
constexpr int CPT = (TM + 255) / 256;   // 6 columns per thread

__global__ __launch_bounds__(256) void tucker_objective_kernel(
    const float* __restrict__ Wm, const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ x_index,
    const double* __restrict__ params, const double* __restrict__ cosp, int64_t N, double* __restrict__ err,
    double* __restrict__ x_hat) {
  __shared__ __attribute__((aligned(16))) double coef[TQ][EV];   // c[q][e]
  __shared__ double fvec[EV][3][3];                              // f_y, f_p, f_r per evaluation
  __shared__ double red[4][EV];

  const int tid = threadIdx.x;
  const int64_t e0 = (int64_t)blockIdx.x * EV;

  // f-vectors: 8 evaluations x 3 angles x 3 cosine rows = 72 values
  if (tid < EV * 9) {
    const int e = tid / 9, a = (tid % 9) / 3, j = tid % 3;
    int64_t n = e0 + e;
    n = n < N ? n : N - 1;
    const double w = params[n * 8 + a];
    const double* cp = cosp + (a * 3 + j) * 4;                    // (a,b,c,d) row, TD_Tester.py:26
    const double v = cp[0] * cos(cp[1] * w + cp[2]) + cp[3];
    fvec[e][a][j] = (double)(float)v;                             // .astype(np.float32), :37,40,43
  }
  __syncthreads();
  for (int i = tid; i < TQ * EV; i += 256) {
    const int q = i / EV, e = i % EV;
    int64_t n = e0 + e;
    n = n < N ? n : N - 1;
    const int ui = q / 27, j = (q / 9) % 3, k = (q / 3) % 3, l = q % 3;
    coef[q][e] = ((params[n * 8 + 3 + ui] * fvec[e][0][j]) * fvec[e][1][k]) * fvec[e][2][l];
  }
  __syncthreads();

  double acc[EV][CPT];
#pragma unroll
  for (int e = 0; e < EV; ++e)
#pragma unroll
    for (int j = 0; j < CPT; ++j) acc[e][j] = 0.0;

  const bool last_ok = tid + 256 * (CPT - 1) < TM;
  for (int q = 0; q < TQ; ++q) {
    const float* wr = Wm + (size_t)q * TM + tid;
    double w[CPT];
#pragma unroll
    for (int j = 0; j < CPT - 1; ++j) w[j] = (double)wr[256 * j];
    w[CPT - 1] = last_ok ? (double)wr[256 * (CPT - 1)] : 0.0;
    double c[EV];
#pragma unroll
    for (int e = 0; e < EV; ++e) c[e] = coef[q][e];
#pragma unroll
    for (int e = 0; e < EV; ++e)
#pragma unroll
      for (int j = 0; j < CPT; ++j) acc[e][j] = fma(c[e], w[j], acc[e][j]);
  }

  // residual, optional x_hat store, per-evaluation reduction
  double part[EV];
#pragma unroll
  for (int e = 0; e < EV; ++e) {
    const int64_t n = e0 + e;
    const bool live = n < N;
    const int64_t nn = live ? n : N - 1;
    const int64_t xr = x_index ? (int64_t)x_index[nn] : nn;
    const float* xp = x + xr * ldx;
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const int m = tid + 256 * j;
      if (m < TM) {
        const double r = (double)xp[m] - acc[e][j];
        s = fma(r, r, s);
        if (x_hat && live) x_hat[n * TM + m] = acc[e][j];
      }
    }
    part[e] = s;
  }
#pragma unroll
  for (int e = 0; e < EV; ++e) {
    double s = part[e];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((tid & 63) == 0) red[tid >> 6][e] = s;
  }
  __syncthreads();
  if (tid < EV && e0 + tid < N) err[e0 + tid] = 0.5 * ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
}

int launch_tucker_objective(const float* Wm, const float* x, int64_t ldx, const int32_t* x_index,
                            const double* params, const double* cos_params, int64_t N, double* err,
                            double* x_hat, void* stream) {
  if (N == 0) return 0;
  const dim3 grid((unsigned)((N + EV - 1) / EV)), block(256);
  hipLaunchKernelGGL(tucker_objective_kernel, grid, block, 0, reinterpret_cast<hipStream_t>(stream), Wm, x, ldx,
                     x_index, params, cos_params, N, err, x_hat);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

}  // namespace nlml
