// tucker_common.h -- the shared body of K3 (tucker_objective.hip) and of every round of the
// device-side Powell minimiser (tucker_powell.hip): EV = 16 objective evaluations by one workgroup on
// the f64 matrix cores.
//
//   f-vectors   f32(a*cos(b*w+c)+d) in f64                      (TD_Tester.py:25-28,:36-43)
//   c[q][e]     = ((u_i * f_yj) * f_pk) * f_rl, q = ((i*3+j)*3+k)*3+l, in LDS (row 135 = 0 pads K to 136)
//   x_hat[e][m] = sum_q c[q][e] * Wm[q][m]     D[eval][column] += A[eval][q] * B[q][column] on
//                 v_mfma_f64_16x16x4_f64, q ascending (one fma chain per output)                 (:46)
//   err[e]      = 0.5 * sum_m (x[m]-x_hat[m])^2   fixed-order reduction                          (:49)
//
// Workgroup = 512 threads = 8 waves; the 1404 columns are 88 blocks of 16 and wave w owns blocks
// 11w .. 11w+10 (44 f64 accumulators per lane).  Per K step of 4 a wave reads its A fragment (16
// evaluations x 4 coefficients) from LDS once and, for each of its 11 blocks, one dword per lane of Wm
// (4 rows x 64 B, f32 -> f64 in the register) -- each row of Wm is read once per workgroup, i.e.
// once per 16 evaluations.  One MFMA (64 cycles) replaces 16 v_fma_f64 wave-instructions, so the
// loads, conversions and LDS reads hide in its shadow instead of competing for issue slots
// (the VALU form of this kernel reached 29 % of the f64 peak).
//
// MFMA operand / result maps (f64 16x16x4, cdna_hip_programming.md section 3): lane l supplies
// A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; result register r of lane l is D[(l>>4) + 4r][l&15].
//
// Reduction order (the C oracle's device_order mode replays it bit for bit):
//   lane:  for each of its 4 evaluations, fma chain over its 11 columns in ascending block order;
//   wave:  xor butterfly over the 16 lanes of a column group, offsets 1, 2, 4, 8;
//   block: ((w0+w1)+(w2+w3)) + ((w4+w5)+(w6+w7)); then * 0.5.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/nlml_hpe.h"

namespace nlml {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int TQ = NLML_TUCKER_Q;       // 135 rows of Wm
constexpr int TM = NLML_F_REFERENCE;    // 1404 columns
constexpr int EV = 16;                  // evaluations per workgroup = MFMA rows
constexpr int TNT = 512;                // threads per workgroup
constexpr int TNW = TNT / 64;           // 8 waves
constexpr int TQS = (TQ + 3) / 4;       // 34 K steps of 4 (K padded to 136)
constexpr int MBW = 11;                 // 16-column blocks per wave: 8 * 11 * 16 = 1408 >= 1404
constexpr int TRING = 3;                // Wm prefetch ring (K steps)

struct TuckerShared {
  double coef[TQS * 4][EV];   // [q][evaluation]; row 135 is zero
  double fvec[EV][3][3];
  double red[TNW][EV];
};

// par(e, k): parameter k (w_y, w_p, w_r, u_id[5]) of evaluation e.  cp4: this thread's cosine row
// (threads < 144 only).  Leaves acc[mb][r] = x_hat of evaluation (lane>>4) + 4r at column
// 16*(11*wave + mb) + (lane&15).
// Coefficient phase: f-vectors and c[q][e] of all 16 evaluations into LDS (two barriers inside).
template <typename ParT>
__device__ __forceinline__ void tucker_coef(TuckerShared& sh, const ParT& par, const double (&cp4)[4], int tid) {
  if (tid < EV * 9) {
    const int e = tid / 9, a = (tid % 9) / 3;
    const double v = cp4[0] * cos(cp4[1] * par(e, a) + cp4[2]) + cp4[3];
    sh.fvec[e][a][tid % 3] = (double)(float)v;                     // .astype(np.float32), :37,40,43
  }
  __syncthreads();
  for (int i = tid; i < TQS * 4 * EV; i += TNT) {
    const int q = i / EV, e = i % EV;
    const int ui = q / 27, j = (q / 9) % 3, k = (q / 3) % 3, l = q % 3;
    sh.coef[q][e] = q < TQ ? ((par(e, 3 + ui) * sh.fvec[e][0][j]) * sh.fvec[e][1][k]) * sh.fvec[e][2][l] : 0.0;
  }
  __syncthreads();
}

// Matrix-core phase (after tucker_coef): leaves acc[mb][r] = x_hat of evaluation (lane>>4) + 4r at column
// 16*(11*wave + mb) + (lane&15).
__device__ __forceinline__ void tucker_mfma(TuckerShared& sh, const float* __restrict__ Wm, int tid, f64x4 (&acc)[MBW]) {
  const int lane = tid & 63, wv = tid >> 6;
  const int kq = lane >> 4, col = lane & 15;
  // column of this lane in block mb: 16*(11*wv + mb) + col.  One base pointer + immediate offsets (64 B per
  // block); only the very last block (wave 7, block 10) reaches past column 1403: those lanes re-read 1403
  // (their x_hat is never used).
  const float* wbase = Wm + 16 * (MBW * wv) + col;
  const int last_off = (16 * (MBW * wv + MBW - 1) + col < TM) ? 16 * (MBW - 1) : (TM - 1) - (16 * MBW * wv + col);
  auto woff = [&](int mb) { return mb < MBW - 1 ? 16 * mb : last_off; };
#pragma unroll
  for (int mb = 0; mb < MBW; ++mb) acc[mb] = f64x4{0.0, 0.0, 0.0, 0.0};
  auto row_off = [&](int qs) {   // K step qs reads row 4*qs + kq; the padding row 135 re-reads row 134 (coefficient 0)
    const int q = 4 * qs + kq;
    return (size_t)(q < TQ ? q : TQ - 1) * TM;
  };
  float wr[TRING][MBW];
#pragma unroll
  for (int d = 0; d < TRING - 1; ++d) {
    const size_t ro = row_off(d);
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb) wr[d][mb] = wbase[ro + woff(mb)];
  }
  // 34 K steps = 11 groups of 3 + 1: ring slot = step % 3, static inside the unrolled group
  auto step = [&](int qs, int slot, bool prefetch) {
    if (prefetch) {
      const size_t ro = row_off(qs + TRING - 1 < TQS ? qs + TRING - 1 : TQS - 1);
#pragma unroll
      for (int mb = 0; mb < MBW; ++mb) wr[(slot + TRING - 1) % TRING][mb] = wbase[ro + woff(mb)];
    }
    const double a = sh.coef[4 * qs + kq][col];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
      acc[mb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, (double)wr[slot][mb], acc[mb], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  constexpr int GROUPS = TQS / TRING, TAIL = TQS % TRING;
  for (int g = 0; g < GROUPS; ++g) {
#pragma unroll
    for (int r = 0; r < TRING; ++r) step(g * TRING + r, r, true);
  }
#pragma unroll
  for (int r = 0; r < TAIL; ++r) step(GROUPS * TRING + r, r, false);
}

template <typename ParT>
__device__ __forceinline__ void tucker_xhat(TuckerShared& sh, const float* __restrict__ Wm, const ParT& par,
                                            const double (&cp4)[4], int tid, f64x4 (&acc)[MBW]) {
  tucker_coef(sh, par, cp4, tid);
  tucker_mfma(sh, Wm, tid, acc);
}

// NE (1..4) evaluations on the vector ALUs in ONE pass over Wm, bit-identical to what the matrix-core path leaves in
// sh.red[.][e] for them (after tucker_coef; the caller's barrier makes red visible).  Used by the Powell kernel when
// only a few of a workgroup's 16 machines are still running: the MFMA pass costs the same for 1 live evaluation as
// for 16, and either way a round cannot be shorter than streaming Wm (758 KB) through one CU (~11 us).
//   x_hat[m]  the same q-ascending fma chain from +0.0 that one MFMA output accumulates (the padded q = 135
//             step adds 0 * w and is skipped);
//   residual  lane (wave w, column group c = lane & 15) owns the columns 16*(11w + mb) + c, mb = 0..10: the four
//             16-lane quarters of the wave split the 11 blocks (quarter j takes mb = j, j+4, j+8), the differences
//             are gathered back by shuffles and squared-and-summed in ascending mb order, then the same xor butterfly.
template <int NE>
__device__ __attribute__((noinline)) void tucker_few(TuckerShared& sh, const float* __restrict__ Wm,
                                           const float* (&xe)[NE], const int (&ev)[NE], int tid) {
  const int lane = tid & 63, wv = tid >> 6, col = lane & 15, j = lane >> 4;
  int mc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int mb = j + 4 * i;                                   // quarter 3 has no third block: it recomputes mb 7
    const int m = 16 * (MBW * wv + (mb < MBW ? mb : MBW - 4)) + col;
    mc[i] = m < TM ? m : TM - 1;
  }
  double acc[3][NE];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int n = 0; n < NE; ++n) acc[i][n] = 0.0;
  // Wm streams through registers fifteen rows at a time (45 dwords per lane in flight, 92 KB per CU)
  constexpr int QB = 15;
  static_assert(TQ % QB == 0, "135 = 9 x 15");
  float w[2][QB][3];
#pragma unroll
  for (int qq = 0; qq < QB; ++qq)
#pragma unroll
    for (int i = 0; i < 3; ++i) w[0][qq][i] = Wm[(size_t)qq * TM + mc[i]];
#pragma unroll 1
  for (int qb = 0; qb < TQ / QB; qb += 2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int q0 = (qb + half) * QB;
      if (q0 < TQ) {
        const int qn = q0 + QB < TQ ? q0 + QB : q0;             // next block (the last one re-reads itself)
#pragma unroll
        for (int qq = 0; qq < QB; ++qq)
#pragma unroll
          for (int i = 0; i < 3; ++i) w[half ^ 1][qq][i] = Wm[(size_t)(qn + qq) * TM + mc[i]];
#pragma unroll
        for (int qq = 0; qq < QB; ++qq)
#pragma unroll
          for (int n = 0; n < NE; ++n) {
            const double c = sh.coef[q0 + qq][ev[n]];
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[i][n] = fma(c, (double)w[half][qq][i], acc[i][n]);
          }
      }
    }
  }
#pragma unroll
  for (int n = 0; n < NE; ++n) {
    double d[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) d[i] = (double)xe[n][mc[i]] - acc[i][n];
    double s = 0.0;
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb) {
      const double dv = __shfl(d[mb >> 2], ((mb & 3) << 4) | col, 64);
      const bool live = 16 * (MBW * wv + mb) + col < TM;
      s = live ? fma(dv, dv, s) : s;
    }
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) sh.red[wv][ev[n]] = s;
  }
}

// Residual norms of the 16 evaluations.  xv[mb][r] = x of evaluation (lane>>4) + 4r at this lane's column of
// block mb.  After the call (barrier inside) tucker_err(e) is valid for every thread.
__device__ __forceinline__ void tucker_residual(TuckerShared& sh, const float (&xv)[MBW][4], const f64x4 (&acc)[MBW],
                                                int tid) {
  const int lane = tid & 63, wv = tid >> 6, col = lane & 15;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int mb = 0; mb < MBW; ++mb) {
    const bool live = 16 * (MBW * wv + mb) + col < TM;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double d = (double)xv[mb][r] - acc[mb][r];
      s[r] = live ? fma(d, d, s[r]) : s[r];
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) s[r] += __shfl_xor(s[r], off, 64);
    if (col == 0) sh.red[wv][(lane >> 4) + 4 * r] = s[r];
  }
  __syncthreads();
}

__device__ __forceinline__ double tucker_err(const TuckerShared& sh, int e) {
  return 0.5 * (((sh.red[0][e] + sh.red[1][e]) + (sh.red[2][e] + sh.red[3][e])) +
                ((sh.red[4][e] + sh.red[5][e]) + (sh.red[6][e] + sh.red[7][e])));
}

}  // namespace nlml
