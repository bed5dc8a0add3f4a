// tucker_powell.hip -- TD end-to-end on device: lock-step Powell minimisation of the Tucker objective.
//
// Replaces Test() (TD_Tester.py:162-199) = scipy Powell over objective() (:31-58), 2-4 s per face on
// the reference's CPU path.  One workgroup owns EV = 8 faces for the whole minimisation:
//
//   repeat:  lanes 0..7 each resume their face's Powell/Brent state machine (powell.h) with the
//            objective value of the previous round and publish the next trial point (8 doubles);
//            all 256 threads then evaluate the 8 objectives together exactly as K3 does
//            (tucker_objective.hip: coefficients c = u (x) f_y (x) f_p (x) f_r in LDS, one pass over
//            Wm from L2 shared by the 8 evaluations, 48 f64 fma chains per thread, fixed-order
//            residual reduction);
//   until all 8 machines have finished.
//
// Faces are independent, so there is no grid-level synchronisation: a workgroup runs as many
// rounds as its slowest face needs (1.4k-3.3k).  The face's feature row is loaded once and stays in
// registers; the machine state lives in LDS.  Every round costs one Wm pass (758 KB from L2) per 8
// faces: ~383.7 kFLOP (f64) per face-evaluation, the same roofline as K3.
#include <hip/hip_runtime.h>

#include "../../include/nlml_hpe.h"
#include "abi_internal.h"
#include "powell.h"

namespace nlml {

constexpr int TQ = NLML_TUCKER_Q;
constexpr int TM = NLML_F_REFERENCE;
constexpr int EV = 8;
constexpr int CPT = (TM + 255) / 256;

__global__ __launch_bounds__(256, 2) void tucker_powell_kernel(
    const float* __restrict__ Wm, const float* __restrict__ x, int64_t ldx, const double* __restrict__ cosp,
    int64_t N, const double* __restrict__ x0, double* __restrict__ result, double* __restrict__ fval,
    int32_t* __restrict__ nfev, int32_t* __restrict__ nit, int32_t* __restrict__ status) {
  __shared__ PowellState st[EV];
  __shared__ __attribute__((aligned(16))) double coef[TQ][EV];
  __shared__ double fvec[EV][3][3];
  __shared__ double red[4][EV];
  __shared__ double par[EV][PW_N];
  __shared__ int need[EV];

  const int tid = threadIdx.x;
  const int64_t e0 = (int64_t)blockIdx.x * EV;

  // this thread's columns of the 8 feature rows, kept for the whole minimisation
  float xr[EV][CPT];
#pragma unroll
  for (int e = 0; e < EV; ++e) {
    int64_t n = e0 + e;
    n = n < N ? n : N - 1;
    const float* xp = x + n * ldx;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const int m = tid + 256 * j;
      xr[e][j] = m < TM ? xp[m] : 0.f;
    }
  }
  double cp[4] = {0, 0, 0, 0};       // cosine row of this thread's f-vector slot (threads < 72)
  if (tid < EV * 9) {
    const double* c4 = cosp + ((tid % 9) / 3 * 3 + tid % 3) * 4;
    cp[0] = c4[0]; cp[1] = c4[1]; cp[2] = c4[2]; cp[3] = c4[3];
  }

  if (tid < EV) {
    double z[PW_N];
    const bool live = e0 + tid < N;
#pragma unroll
    for (int k = 0; k < PW_N; ++k) z[k] = (x0 && live) ? x0[(e0 + tid) * PW_N + k] : 0.0;
    powell_init(st[tid], z);
    const bool nd = live && powell_step(st[tid], 0.0);
    need[tid] = nd ? 1 : 0;
#pragma unroll
    for (int k = 0; k < PW_N; ++k) par[tid][k] = st[tid].xeval[k];
  }
  __syncthreads();

  const bool last_ok = tid + 256 * (CPT - 1) < TM;
  for (int round = 0; round < PW_N * 1000 + 16; ++round) {
    int any = 0;
#pragma unroll
    for (int e = 0; e < EV; ++e) any |= need[e];
    if (!any) break;

    // ---- objective of the 8 trial points (TD_Tester.py:31-58), as in tucker_objective.hip ----
    if (tid < EV * 9) {
      const int e = tid / 9, a = (tid % 9) / 3;
      const double v = cp[0] * cos(cp[1] * par[e][a] + cp[2]) + cp[3];
      fvec[e][a][tid % 3] = (double)(float)v;
    }
    __syncthreads();
    for (int i = tid; i < TQ * EV; i += 256) {
      const int q = i / EV, e = i % EV;
      const int ui = q / 27, j = (q / 9) % 3, k = (q / 3) % 3, l = q % 3;
      coef[q][e] = ((par[e][3 + ui] * fvec[e][0][j]) * fvec[e][1][k]) * fvec[e][2][l];
    }
    __syncthreads();

    double acc[EV][CPT];
#pragma unroll
    for (int e = 0; e < EV; ++e)
#pragma unroll
      for (int j = 0; j < CPT; ++j) acc[e][j] = 0.0;
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
#pragma unroll
    for (int e = 0; e < EV; ++e) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < CPT; ++j) {
        if (tid + 256 * j < TM) {
          const double r = (double)xr[e][j] - acc[e][j];
          s = fma(r, r, s);
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      if ((tid & 63) == 0) red[tid >> 6][e] = s;
    }
    __syncthreads();

    // ---- resume the 8 state machines ----
    if (tid < EV && need[tid]) {
      const double f = 0.5 * ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
      const bool nd = powell_step(st[tid], f);
      need[tid] = nd ? 1 : 0;
      if (nd) {
#pragma unroll
        for (int k = 0; k < PW_N; ++k) par[tid][k] = st[tid].xeval[k];
      }
    }
    __syncthreads();
  }

  if (tid < EV && e0 + tid < N) {
    const int64_t n = e0 + tid;
#pragma unroll
    for (int k = 0; k < PW_N; ++k) result[n * PW_N + k] = st[tid].x[k];
    if (fval) fval[n] = st[tid].fval;
    if (nfev) nfev[n] = st[tid].nfev;
    if (nit) nit[n] = st[tid].iter;
    if (status) status[n] = need[tid] ? PW_RUNNING : st[tid].status;
  }
}

int launch_tucker_powell(const float* Wm, const float* x, int64_t ldx, const double* cos_params, int64_t N,
                         const double* x0, double* result, double* fval, int32_t* nfev, int32_t* nit,
                         int32_t* status, void* stream) {
  if (N == 0) return 0;
  const dim3 grid((unsigned)((N + EV - 1) / EV)), block(256);
  hipLaunchKernelGGL(tucker_powell_kernel, grid, block, 0, reinterpret_cast<hipStream_t>(stream), Wm, x, ldx,
                     cos_params, N, x0, result, fval, nfev, nit, status);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

}  // namespace nlml
