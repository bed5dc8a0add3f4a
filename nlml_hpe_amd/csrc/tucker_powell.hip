// tucker_powell.hip -- TD end-to-end on device: lock-step Powell minimisation of the Tucker objective.
//
// Replaces Test() (TD_Tester.py:162-199) = scipy Powell over objective() (:31-58), 2-4 s per face on
// the reference's CPU path.  One workgroup owns EV = 16 faces for the whole minimisation:
//
//   repeat:  lanes 0..15 each resume their face's Powell/Brent state machine (powell.h) with the
//            objective value of the previous round and publish the next trial point (8 doubles);
//            all 512 threads then evaluate the 16 objectives together exactly as K3 does
//            (tucker_common.h: one pass over Wm from L2 shared by the 16 evaluations, f64 MFMA);
//   until all 16 machines have finished.
//
// Faces are independent, so there is no grid-level synchronisation: a workgroup runs as many
// rounds as its slowest face needs (1.4k-7k).  The machine state lives in LDS.  Every round costs one Wm pass (758 KB from L2) per 8
// faces: ~383.7 kFLOP (f64) per face-evaluation, the same roofline as K3.  Rounds with at most PW_FEW live machines take
// the vector-ALU pass tucker_few (tucker_common.h): same bits, one Wm stream, no 16-wide MFMA work for dead machines.
#include <hip/hip_runtime.h>

#include "abi_internal.h"
#include "powell.h"
#include "tucker_common.h"
#include "tucker_ref.h"

namespace nlml {

// The state machine is a large switch; inlined into the kernel it inflates the register demand of the
// whole function (spills in the MFMA loop).  As a real call its registers are its own.
// The state is passed as an LDS pointer: through a generic one every access to it would be a flat_load / flat_store.
typedef __attribute__((address_space(3))) PowellState LdsPowellState;
__device__ __attribute__((noinline)) bool powell_step_call(LdsPowellState* s, double f) { return powell_step(*(PowellState*)s, f); }

// at most this many live machines: a few-machine pass instead of the 16-wide one -- 13 us (one machine, vector ALUs) or 15 us
// (2..4, 4x4x4 matrix instruction) a round against 30 us; BASELINE config 3 end to end: 0.117 s, 16-wide rounds only 0.24 s
#ifndef PW_FEW_N
#define PW_FEW_N 4
#endif
constexpr int PW_FEW = PW_FEW_N;

// A round with many live machines: all 16 evaluations on the 16x16x4 matrix cores, exactly K3's body.  Not inlined, like the
// few-machine passes: its 150 registers are then allocated on their own instead of across the state-machine code (inlined, the
// fully unrolled MFMA loop spilled).  The lane's 4 evaluations x 11 columns of the feature rows are re-read from L2 every round
// (44 dwords per lane against 374 of Wm).
__device__ __attribute__((noinline)) void tucker_round16(TuckerShared& sh, const float* __restrict__ Wm, const float* __restrict__ x,
                                                         int64_t ldx, int64_t e0, int64_t N, int tid) {
  f64x4 acc[MBW];
  tucker_mfma(sh, Wm, tid, acc);
  // (K3 fetches its x rows with the loads of one of the last K steps; here the 44 extra live registers spill: measured slower)
  float xv[MBW][4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int64_t n = e0 + ((tid & 63) >> 4) + 4 * r;
    n = n < N ? n : N - 1;
    float v[MBW];
    tucker_load_x(x + n * ldx, tid, v);
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb) xv[mb][r] = v[mb];
  }
  tucker_residual(sh, xv, acc, tid);
}

// The trial point of machine e IS its state's xeval field: the evaluation passes read the parameters straight from the machines
// (a separate copy per round cost ~450 cycles of one lane's LDS round trips).
struct LdsPar {
  const PowellState* st;
  __device__ __forceinline__ double operator()(int e, int k) const {
    return ((const __attribute__((address_space(3))) PowellState*)st)[e].xeval[k];   // a ds_ read, not a flat_ one
  }
};

// row functors of the reference-order pass (passed by value into the non-inlined pass)
struct PowellXRow {
  const float* x;
  int64_t ldx, e0, N;
  __device__ __forceinline__ const float* operator()(int slot) const {
    int64_t n = e0 + slot;
    n = n < N ? n : N - 1;
    return x + n * ldx;
  }
};
struct NoXhRow {
  static constexpr double* x_hat = nullptr;   // (the stamp build of the pass looks for this member)
  __device__ __forceinline__ double* operator()(int) const { return nullptr; }
};

// ORDER = NLML_TD_ORDER_FAST: the objective as a GEMM on the f64 matrix cores (tucker_common.h); NLML_TD_ORDER_REFERENCE: in the
// reference's own operation order (tucker_ref.h) -- then the machines receive the reference's objective values bit for bit and
// walk scipy's trajectory (FX5: same evaluation counts, same final angles).
template <int ORDER>
__global__ __launch_bounds__(ORDER == NLML_TD_ORDER_REFERENCE ? TR_NT : TNT, ORDER == NLML_TD_ORDER_REFERENCE ? 1 : 2) void tucker_powell_kernel(
    const float* __restrict__ Wm, const float* __restrict__ x, int64_t ldx, const double* __restrict__ cosp,
    int64_t N, const double* __restrict__ x0, double* __restrict__ result, double* __restrict__ fval,
    int32_t* __restrict__ nfev, int32_t* __restrict__ nit, int32_t* __restrict__ status) {
  __shared__ __attribute__((aligned(16))) TuckerShared sh;
  __shared__ PowellState st[EV];
  __shared__ int need[EV];

  const int tid = threadIdx.x;
  const int64_t e0 = (int64_t)blockIdx.x * EV;

  double cp4[4] = {0, 0, 0, 0};
  if (tid < EV * 9) {
    const double* c4 = cosp + ((tid % 9) / 3 * 3 + tid % 3) * 4;
    cp4[0] = c4[0]; cp4[1] = c4[1]; cp4[2] = c4[2]; cp4[3] = c4[3];
  }

  // machine e runs on lane e&1 of wave e>>1: two machines per wave, so the divergent state-machine code is at
  // most 2-way serialised and the 8 waves step their machines concurrently (16 machines on the lanes of one
  // wave would serialise up to 16 paths per round)
  const int me = ((tid & 63) < 2 && tid < 64 * (EV / 2)) ? 2 * (tid >> 6) + (tid & 63) : -1;   // (the reference-order workgroup has 12 waves)
  if (me >= 0) {
    double z[PW_N];
    const bool live = e0 + me < N;
#pragma unroll
    for (int k = 0; k < PW_N; ++k) z[k] = (x0 && live) ? x0[(e0 + me) * PW_N + k] : 0.0;
    powell_init(st[me], z);
#pragma unroll
    for (int k = 0; k < PW_N; ++k) st[me].xeval[k] = 0.0;   // slots beyond N are evaluated too (and ignored): defined parameters
    const bool nd = live && powell_step_call((LdsPowellState*)&st[me], 0.0);
    need[me] = nd ? 1 : 0;
  }
  __syncthreads();

  const LdsPar lp{st};
  // live machines as one word per round parity: the machines that want another evaluation OR their bit into the next round's
  // word (sixteen LDS reads per thread and round before)
  __shared__ int livew[2];
  if (tid == 0) {
    int m0 = 0;
    for (int e = 0; e < EV; ++e) m0 |= need[e] << e;
    livew[0] = m0;
    livew[1] = 0;
  }
  __syncthreads();
#ifdef PW_STAMPS   // timing-only diagnostic: cycles per phase summed over the rounds, written over fval[e0 .. e0+3] at the end
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp = __builtin_amdgcn_s_memtime();
#define PWS(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[i] += t_ - tp; tp = t_; } while (0)
#else
#define PWS(i) do { } while (0)
#endif
  for (int round = 0; round < PW_N * 1000 + 16; ++round) {
    const int live_mask = livew[round & 1];
    if (!live_mask) break;
    if (tid == 0) livew[(round + 1) & 1] = 0;     // last read a round ago; the barriers of the evaluation order this before the ORs below

    if constexpr (ORDER == NLML_TD_ORDER_REFERENCE) {
      __shared__ __attribute__((aligned(16))) TuckerRefShared rs;
      PWS(0);
      tucker_fvec(sh, lp, cp4, tid);
      PWS(1);
      tucker_ref_eval(sh, rs, Wm, lp, live_mask, PowellXRow{x, ldx, e0, N}, NoXhRow{}, tid);
      PWS(2);
      if (me >= 0 && need[me]) {
        const double fe = rs.err[me];
        PWS(4);
        const bool nd = powell_step_call((LdsPowellState*)&st[me], fe);
        PWS(5);
        need[me] = nd ? 1 : 0;
        if (nd) atomicOr(&livew[(round + 1) & 1], 1 << me);
        PWS(6);
      }
      __syncthreads();
      PWS(3);
    } else {   // NLML_TD_ORDER_FAST (discarded in the other instantiation: its LDS is the reference pass's)
    __shared__ __attribute__((aligned(16))) TuckerFewShared few;
    PWS(0);
    tucker_coef(sh, lp, cp4, tid);
    PWS(1);
    if (PW_FEW > 0 && __popc(live_mask) <= PW_FEW) {
      // the tail of a workgroup's run: one to a few machines left (a face that needs 6,000 evaluations next to fifteen
      // that needed 1,500).  Their evaluations share ONE pass over Wm on the vector ALUs, bit-identical to the MFMA pass.
      int ev[PW_FEW > 0 ? PW_FEW : 1];
      const float* xe[PW_FEW > 0 ? PW_FEW : 1];
      int ne = 0;
      for (int mleft = live_mask; mleft; mleft &= mleft - 1) {
        const int e = __ffs(mleft) - 1;
        int64_t n = e0 + e;
        n = n < N ? n : N - 1;
        ev[ne] = e;
        xe[ne++] = x + n * ldx;
      }
      // the count is a compile-time parameter of the pass: its accumulators live in registers
#define NLML_FEW_CASE(K)                                                  \
  case K: {                                                               \
    int ek[K];                                                            \
    const float* xk[K];                                                   \
    for (int i = 0; i < K; ++i) { ek[i] = ev[i]; xk[i] = xe[i]; }         \
    tucker_few<K>(sh, few, Wm, xk, ek, tid);                              \
    break;                                                                \
  }
#ifndef PW_MFMA4_FROM
#define PW_MFMA4_FROM 2          // rounds with this many live machines or more (up to 4) take the 4x4x4 matrix pass
#endif
      if (ne >= PW_MFMA4_FROM) {
        int e4[4];
        const float* x4[4];
        for (int i = 0; i < 4; ++i) { e4[i] = ev[i < ne ? i : 0]; x4[i] = xe[i < ne ? i : 0]; }
        tucker_mfma4(sh, Wm, x4, e4, ne, tid);
      } else {
      switch (ne) {
        NLML_FEW_CASE(1) NLML_FEW_CASE(2) NLML_FEW_CASE(3) NLML_FEW_CASE(4)
        default: break;   // unreachable: ne <= PW_FEW
      }
      }
#undef NLML_FEW_CASE
      __syncthreads();
    } else {
      tucker_round16(sh, Wm, x, ldx, e0, N, tid);
    }

    PWS(2);
    if (me >= 0 && need[me]) {   // resume the state machines with their objective values
      const double fe = tucker_err(sh, me);
      PWS(4);
      const bool nd = powell_step_call((LdsPowellState*)&st[me], fe);
      PWS(5);
      need[me] = nd ? 1 : 0;
      if (nd) atomicOr(&livew[(round + 1) & 1], 1 << me);
      PWS(6);
    }
    __syncthreads();
    PWS(3);
    }   // order
  }

  if (tid < EV && e0 + tid < N) {
    const int64_t n = e0 + tid;
#pragma unroll
    for (int k = 0; k < PW_N; ++k) result[n * PW_N + k] = st[tid].x[k];
    if (fval) fval[n] = st[tid].fval;
#ifdef PW_STAMPS
    for (int i = 0; i < 8; ++i) {   // lane 0 (machine 0) holds all eight sums
      const unsigned long long v = __shfl(ph[i], 0, 64);
      if (fval && tid == i) fval[n] = (double)v;
    }
#endif
    if (nfev) nfev[n] = st[tid].nfev;
    if (nit) nit[n] = st[tid].iter;
    if (status) status[n] = need[tid] ? PW_RUNNING : st[tid].status;
  }
}

int launch_tucker_powell(const float* Wm, const float* x, int64_t ldx, const double* cos_params, int64_t N,
                         const double* x0, double* result, double* fval, int32_t* nfev, int32_t* nit,
                         int32_t* status, int order, void* stream) {
  if (N == 0) return 0;
  const dim3 grid((unsigned)((N + EV - 1) / EV)), block(TNT);
  if (order == NLML_TD_ORDER_REFERENCE)
    hipLaunchKernelGGL((tucker_powell_kernel<NLML_TD_ORDER_REFERENCE>), grid, dim3(TR_NT), 0, reinterpret_cast<hipStream_t>(stream),
                       Wm, x, ldx, cos_params, N, x0, result, fval, nfev, nit, status);
  else
  hipLaunchKernelGGL((tucker_powell_kernel<NLML_TD_ORDER_FAST>), grid, block, 0, reinterpret_cast<hipStream_t>(stream), Wm, x, ldx,
                     cos_params, N, x0, result, fval, nfev, nit, status);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

}  // namespace nlml
