// tucker_objective.hip -- K3: batched Tucker-einsum objective in f64 on the matrix cores.
//
// Replaces objective() (TD_Tester.py:31-58) for N face-evaluations at once, restated as
// c = u (x) f_y (x) f_p (x) f_r (135 f64 coefficients per evaluation) and x_hat = c^T Wm,
// Wm = W.reshape(135,1404): a [N,135] x [135,1404] f64 GEMM fused with the residual norm.
// The per-workgroup body (16 evaluations, one pass over Wm, v_mfma_f64_16x16x4_f64) is tucker_common.h.
#include <hip/hip_runtime.h>

#include "abi_internal.h"
#include "tucker_common.h"
#include "tucker_ref.h"

namespace nlml {

#ifndef K3_XSTEP
#define K3_XSTEP 3     // the x rows are fetched with the loads of K step 34 - K3_XSTEP
#endif

struct GlobalPar {
  const double* p;   // params of this block's first evaluation
  int64_t left;      // evaluations available from there (>= 1)
  __device__ __forceinline__ double operator()(int e, int k) const {
    return gload<double>(p + (e < left ? e : left - 1) * 8 + k);
  }
};

__global__ __launch_bounds__(TNT, 2) void tucker_objective_kernel(
    const float* __restrict__ Wm, const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ x_index,
    const double* __restrict__ params, const double* __restrict__ cosp, int64_t N, double* __restrict__ err,
    double* __restrict__ x_hat) {
  __shared__ __attribute__((aligned(16))) TuckerShared sh;
  const int tid = threadIdx.x;
  const int64_t e0 = (int64_t)blockIdx.x * EV;

  double cp4[4] = {0, 0, 0, 0};
  if (tid < EV * 9) {
    const double* c4 = cosp + ((tid % 9) / 3 * 3 + tid % 3) * 4;     // (a,b,c,d) row, TD_Tester.py:26
    cp4[0] = c4[0]; cp4[1] = c4[1]; cp4[2] = c4[2]; cp4[3] = c4[3];
  }
  GlobalPar par{params + e0 * 8, N - e0};
  f64x4 acc[MBW];
#ifdef K3_STAMPS   // timing-only diagnostic build: s_memtime at the phase boundaries into x_hat (u64[grid][8 waves][8])
#define K3S(i) do { if (x_hat && (tid & 63) == 0) reinterpret_cast<unsigned long long*>(x_hat)[((size_t)blockIdx.x * 8 + (tid >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define K3S(i) do { } while (0)
#endif
  K3S(0);
  tucker_coef(sh, par, cp4, tid);
  K3S(1);
  // this lane's four x rows (evaluations (lane >> 4) + 4r); fetched K3_XSTEP steps before the end of the matrix-core phase
  const int lane = tid & 63, wv = tid >> 6, col = lane & 15;
  const float* xrow[4];
  bool xlive[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t n = e0 + (lane >> 4) + 4 * r;
    xlive[r] = n < N;
    const int64_t nn = xlive[r] ? n : N - 1;
    xrow[r] = x + (x_index ? (int64_t)x_index[nn] : nn) * ldx;
  }
  float xv[MBW][4];
  auto xfetch = [&](int qs) {
    if (qs != TQS - K3_XSTEP) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v[MBW];
      tucker_load_x(xrow[r], tid, v);
#pragma unroll
      for (int mb = 0; mb < MBW; ++mb) xv[mb][r] = v[mb];
    }
  };
  tucker_mfma(sh, Wm, tid, acc, xfetch);
  K3S(2);
#ifndef K3_STAMPS
  if (x_hat) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int mb = 0; mb < MBW; ++mb)
        if (xlive[r] && tcol_live(wv, col, mb)) x_hat[(e0 + (lane >> 4) + 4 * r) * TM + tcol0(wv) + tlcol(col, mb)] = acc[mb][r];
  }
#endif
  K3S(3);
  tucker_residual(sh, xv, acc, tid);
  K3S(4);
  if (tid < EV && e0 + tid < N) err[e0 + tid] = tucker_err(sh, tid);
  K3S(5);
}

// row functors of the reference-order pass (passed by value into the non-inlined pass)
struct ObjXRow {
  const float* x;
  const int32_t* x_index;
  int64_t ldx, e0;
  __device__ __forceinline__ const float* operator()(int slot) const {
    const int64_t n = e0 + slot;
    return x + (x_index ? (int64_t)x_index[n] : n) * ldx;
  }
};
struct ObjXhRow {
  double* x_hat;
  int64_t e0;
  __device__ __forceinline__ double* operator()(int slot) const { return x_hat ? x_hat + (e0 + slot) * TM : (double*)nullptr; }
};

// NLML_TD_ORDER_REFERENCE: the same 16 evaluations per workgroup in the reference's operation order (tucker_ref.h)
__global__ __launch_bounds__(TR_NT, 1) void tucker_objective_ref_kernel(
    const float* __restrict__ Wm, const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ x_index,
    const double* __restrict__ params, const double* __restrict__ cosp, int64_t N, double* __restrict__ err,
    double* __restrict__ x_hat) {
  __shared__ __attribute__((aligned(16))) TuckerShared sh;
  __shared__ __attribute__((aligned(16))) TuckerRefShared rs;
  const int tid = threadIdx.x;
  const int64_t e0 = (int64_t)blockIdx.x * EV;
  double cp4[4] = {0, 0, 0, 0};
  if (tid < EV * 9) {
    const double* c4 = cosp + ((tid % 9) / 3 * 3 + tid % 3) * 4;
    cp4[0] = c4[0]; cp4[1] = c4[1]; cp4[2] = c4[2]; cp4[3] = c4[3];
  }
  GlobalPar par{params + e0 * 8, N - e0};
#ifdef TR_STAMPS   // kernel-level stamps behind the per-pass ones (2 passes x 12 waves x 8 per workgroup): 8 words per workgroup at the end of x_hat
#define TKS(i) do { if (x_hat && tid == 0) reinterpret_cast<unsigned long long*>(x_hat)[(size_t)gridDim.x * 192 + (size_t)blockIdx.x * 8 + (i)] = (i) >= 4 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TKS(i) do { } while (0)
#endif
  TKS(0); TKS(4);
  tucker_fvec(sh, par, cp4, tid);
  TKS(1);
  const int64_t left = N - e0;
  const int mask = left >= EV ? 0xffff : ((1 << (int)left) - 1);
  tucker_ref_eval(sh, rs, Wm, par, mask, ObjXRow{x, x_index, ldx, e0}, ObjXhRow{x_hat, e0}, tid);
  TKS(2);
  if (tid < EV && e0 + tid < N) err[e0 + tid] = rs.err[tid];
  TKS(3); TKS(5);
#undef TKS
}

int launch_tucker_objective(const float* Wm, const float* x, int64_t ldx, const int32_t* x_index,
                            const double* params, const double* cos_params, int64_t N, double* err,
                            double* x_hat, int order, void* stream) {
  if (N == 0) return 0;
  const dim3 grid((unsigned)((N + EV - 1) / EV)), block(TNT);
  if (order == NLML_TD_ORDER_REFERENCE)
    hipLaunchKernelGGL(tucker_objective_ref_kernel, grid, dim3(TR_NT), 0, reinterpret_cast<hipStream_t>(stream), Wm, x, ldx,
                       x_index, params, cos_params, N, err, x_hat);
  else
  hipLaunchKernelGGL(tucker_objective_kernel, grid, block, 0, reinterpret_cast<hipStream_t>(stream), Wm, x, ldx,
                     x_index, params, cos_params, N, err, x_hat);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

}  // namespace nlml
