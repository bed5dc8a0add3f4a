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
// Workgroup = 512 threads = 8 waves; wave w owns the 176 columns from tcol0(w) = 176 w (the last wave: 1228, so that it ends
// at column 1403; its first four columns repeat wave 6's last four and are ignored) as 11 MFMA column blocks (44 f64
// accumulators per lane).  Block mb's MFMA column j is the wave's column
//     tlcol(j, mb) = 64 (mb/4) + 4 j + mb%4      for mb < 8      (two groups of four consecutive columns, 16 B aligned)
//                  = 128 + 3 j + (mb - 8)         for mb >= 8     (three consecutive columns)
// so a lane fetches its eleven B operands of a K step with three loads (16 + 16 + 12 bytes) and the 16 lanes of a row read
// CONTIGUOUS 256 + 256 + 192 bytes: a wave instruction touches 16 sixty-four-byte sectors.  History of this map, each step
// measured: blocks of 16 consecutive columns (eleven dword loads per lane: 256-byte wave instructions, the CU's address path
// -- not the matrix pipe -- set the pace: a timing-only build with 3 of the 11 loads ran at 66 % of the f64 peak instead of
// 56 %); eleven consecutive columns per lane (three loads at a 44-byte lane stride: every instruction touched ~46 sectors and
// the pass over Wm could not go faster than ~24 B/clk per CU -- the few-machine rounds of the Powell kernel were bound by
// exactly that); this one.  Per K step of 4 a wave reads its A fragment (16 evaluations x 4 coefficients) from LDS once; each
// row of Wm is read once per workgroup, i.e. once per 16 evaluations; f32 -> f64 in the register.  One MFMA (64 cycles)
// replaces 16 v_fma_f64 wave-instructions (the VALU form of this kernel reached 29 % of the f64 peak).
//
// MFMA operand / result maps (f64 16x16x4, cdna_hip_programming.md section 3): lane l supplies
// A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; result register r of lane l is D[(l>>4) + 4r][l&15].
//
// Reduction order (the C oracle's device_order mode replays it bit for bit):
//   lane:  for each of its 4 evaluations, fma chain over its 11 columns tlcol(j, mb), mb ascending, dead columns skipped;
//   wave:  xor butterfly over the 16 lanes of a column group, offsets 1, 2, 4, 8;
//   block: ((w0+w1)+(w2+w3)) + ((w4+w5)+(w6+w7)); then * 0.5.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/nlml_hpe.h"
#include "cr_cos.h"

namespace nlml {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int TQ = NLML_TUCKER_Q;       // 135 rows of Wm
constexpr int TM = NLML_F_REFERENCE;    // 1404 columns
constexpr int EV = 16;                  // evaluations per workgroup = MFMA rows
constexpr int TNT = 512;                // threads per workgroup
constexpr int TNW = TNT / 64;           // 8 waves
constexpr int TQS = (TQ + 3) / 4;       // 34 K steps of 4 (K padded to 136)
constexpr int MBW = 11;                 // MFMA column blocks per wave = consecutive columns per lane: 8 * 16 * 11 = 1408 >= 1404
constexpr int TWC = 16 * MBW;           // 176 columns per wave
#ifndef K3_TRING
#define K3_TRING 3
#endif
constexpr int TRING = K3_TRING;         // Wm prefetch ring (K steps)

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x3_t __attribute__((ext_vector_type(3)));

// first column of wave w; the wave's column of (lane column j, block mb); whether that is a column of its own (wave 7 starts 4 early)
__host__ __device__ __forceinline__ constexpr int tcol0(int w) { return w < TNW - 1 ? TWC * w : TM - TWC; }
__host__ __device__ __forceinline__ constexpr int tlcol(int j, int mb) { return mb < 8 ? 64 * (mb >> 2) + 4 * j + (mb & 3) : 128 + 3 * j + (mb - 8); }
__host__ __device__ __forceinline__ constexpr bool tcol_live(int w, int j, int mb) { return w < TNW - 1 || tlcol(j, mb) >= TNW * TWC - TM; }

// A load from DEVICE-GLOBAL memory through a generic pointer.  Inside a non-inlined device function the compiler cannot see
// that Wm / x are global and emits flat_load, which counts on lgkmcnt as well as vmcnt: every wait for an LDS read then also
// waits for the Wm prefetch issued just before it and the prefetch ring hides nothing.  global_load counts on vmcnt alone.
template <typename T>
__device__ __forceinline__ T gload(const void* p) {
  return *(const __attribute__((address_space(1))) T*)p;
}

// the eleven floats of lane column j from a wave's 176-column row segment p (global memory, 16-byte aligned): v[mb] = p[tlcol(j, mb)]
__device__ __forceinline__ void load11(const float* __restrict__ p, int j, float (&v)[MBW]) {
  const f32x4_t a = gload<f32x4_t>(p + 4 * j), b = gload<f32x4_t>(p + 64 + 4 * j);
  const f32x3_t c = gload<f32x3_t>(p + 128 + 3 * j);
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
  v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
  v[8] = c[0]; v[9] = c[1]; v[10] = c[2];
}

struct TuckerShared {
  double coef[TQS * 4][EV];   // [q][evaluation]; row 135 is zero
  double fvec[EV][3][3];
  double red[TNW][EV];
};

// Sum over the 16 lanes of a DPP row as the balanced tree ((v0+v1)+(v2+v3))+... -- the tree the xor butterfly
// v[c] += v[c^1], v[c^2], v[c^4], v[c^8] builds in every lane (the C oracle's device order) -- on the vector ALUs' DPP path
// (row_shr) instead of four ds_bpermute round trips per value.  The total is valid in lane 15 of the row only.
template <int N>
__device__ __forceinline__ double dpp_row_shr(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x110 + N, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x110 + N, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_tree_sum(double v) {
  v = v + dpp_row_shr<1>(v);
  v = v + dpp_row_shr<2>(v);
  v = v + dpp_row_shr<4>(v);
  v = v + dpp_row_shr<8>(v);
  return v;
}

// par(e, k): parameter k (w_y, w_p, w_r, u_id[5]) of evaluation e.  cp4: this thread's cosine row
// (threads < 144 only).  Leaves acc[mb][r] = x_hat of evaluation (lane>>4) + 4r at column
// 16*(11*wave + mb) + (lane&15).
// Coefficient phase: f-vectors and c[q][e] of all 16 evaluations into LDS (two barriers inside).
template <typename ParT>
__device__ __forceinline__ void tucker_coef(TuckerShared& sh, const ParT& par, const double (&cp4)[4], int tid) {
#pragma clang fp contract(off)   // numpy rounds b*w, + c, a*cos, + d separately (TD_Tester.py:25-28); the coefficient products too
  if (tid < EV * 9) {
    const int e = tid / 9, a = (tid % 9) / 3;
    const double v = cp4[0] * cr_cos(cp4[1] * par(e, a) + cp4[2]) + cp4[3];   // correctly rounded cos: cr_cos.h
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
// tcol0(wave) + tlcol(lane&15, mb).
struct NoHook {
  __device__ __forceinline__ void operator()(int) const {}
};
// `hook(qs)` runs with the loads of K step qs (before its matrix instructions): the callers fetch their x rows a few steps
// before the end, so that those loads' latency hides behind the last steps instead of following the loop.
template <typename Hook = NoHook>
__device__ __forceinline__ void tucker_mfma(TuckerShared& sh, const float* __restrict__ Wm, int tid, f64x4 (&acc)[MBW],
                                            const Hook& hook = Hook()) {
  const int lane = tid & 63, wv = tid >> 6;
  const int kq = lane >> 4, col = lane & 15;
  const float* wbase = Wm + tcol0(wv);                 // the wave's 176-column segment; always inside the row
#pragma unroll
  for (int mb = 0; mb < MBW; ++mb) acc[mb] = f64x4{0.0, 0.0, 0.0, 0.0};
  auto row_off = [&](int qs) {   // K step qs reads row 4*qs + kq; the padding row 135 re-reads row 134 (coefficient 0)
    const int q = 4 * qs + kq;
    return (size_t)(q < TQ ? q : TQ - 1) * TM;
  };
  float wr[TRING][MBW];
#pragma unroll
  for (int d = 0; d < TRING - 1; ++d) load11(wbase + row_off(d), col, wr[d]);
  // 34 K steps in groups of TRING: ring slot = step % TRING, static inside the unrolled group
  auto step = [&](int qs, int slot, bool prefetch) {
    if (prefetch) load11(wbase + row_off(qs + TRING - 1 < TQS ? qs + TRING - 1 : TQS - 1), col, wr[(slot + TRING - 1) % TRING]);
    hook(qs);
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
// for 16, and either way a round cannot be shorter than streaming Wm (758 KB) through one CU.
//   x_hat[m]  the same q-ascending fma chain from +0.0 that one MFMA output accumulates (the padded q = 135
//             step adds 0 * w and is skipped);
//   columns   lane L of wave w takes the three consecutive columns tcol0(w) + 3L .. + 2 (lanes 0..58): one 12-byte load per
//             row, 704 contiguous bytes per wave -- a third of the load instructions of the one-dword-per-block form;
//   residual  the differences go through `few` (LDS) so that lane c < 16 can square-and-sum the eleven columns
//             tcol0(w) + tlcol(c, mb), mb ascending, exactly as the MFMA path's lane c does, then the same xor butterfly.
struct TuckerFewShared {
  double d[4][TNW][TWC];       // residuals x - x_hat of up to 4 evaluations, by wave and column-in-wave
};

#ifndef K3_TRING4
#define K3_TRING4 3          // ring depth of the 4x4x4 pass (tucker_mfma4)
#endif
#ifndef K3_FEW_CPL
#define K3_FEW_CPL 4
#endif
template <int NE>
__device__ __attribute__((noinline)) void tucker_few(TuckerShared& sh, TuckerFewShared& few, const float* __restrict__ Wm,
                                                     const float* (&xe)[NE], const int (&ev)[NE], int tid) {
  const int lane = tid & 63, wv = tid >> 6;
  // CPL consecutive columns per lane: with 4, lanes 0..43 own columns 4L .. 4L+3 and every load is an ALIGNED 16-byte load (rows
  // are 5,616 B = 351 x 16, the wave bases 704 w and 4,912 B too); with 3, lanes 0..58 and 12-byte loads at 12-byte strides.
  constexpr int CPL = K3_FEW_CPL;
  static_assert(TWC % CPL == 0 || CPL == 3, "columns per lane");
  const int c0 = CPL * lane < TWC - CPL ? CPL * lane : TWC - CPL;   // lanes beyond the last owner repeat its loads
  const int own0 = CPL * lane < TWC ? CPL * lane : TWC;              // first column this lane writes
  const float* wb = Wm + tcol0(wv) + c0;
  double acc[CPL][NE];
#pragma unroll
  for (int i = 0; i < CPL; ++i)
#pragma unroll
    for (int n = 0; n < NE; ++n) acc[i][n] = 0.0;
  // Wm streams through registers fifteen rows at a time
#ifndef K3_FEW_QB
#define K3_FEW_QB 15
#endif
  constexpr int QB = K3_FEW_QB;
  static_assert(TQ % QB == 0, "135 = 9 x 15 = 5 x 27 = 3 x 45");
  float w[2][QB][CPL];
  auto loadc = [&](int q, float (&dst)[CPL]) {
    if constexpr (CPL == 4) {
      const f32x4_t t = gload<f32x4_t>(wb + (size_t)q * TM);
      dst[0] = t[0]; dst[1] = t[1]; dst[2] = t[2]; dst[3] = t[3];
    } else {
      const f32x3_t t = gload<f32x3_t>(wb + (size_t)q * TM);
      dst[0] = t[0]; dst[1] = t[1]; dst[2] = t[2];
    }
  };
#pragma unroll
  for (int qq = 0; qq < QB; ++qq) loadc(qq, w[0][qq]);
#pragma unroll 1
  for (int qb = 0; qb < TQ / QB; qb += 2) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int q0 = (qb + half) * QB;
      if (q0 < TQ) {
        const int qn = q0 + QB < TQ ? q0 + QB : q0;             // next block (the last one re-reads itself)
#pragma unroll
        for (int qq = 0; qq < QB; ++qq) loadc(qn + qq, w[half ^ 1][qq]);
#pragma unroll
        for (int qq = 0; qq < QB; ++qq)
#pragma unroll
          for (int n = 0; n < NE; ++n) {
            const double c = sh.coef[q0 + qq][ev[n]];
#pragma unroll
            for (int i = 0; i < CPL; ++i) acc[i][n] = fma(c, (double)w[half][qq][i], acc[i][n]);
          }
      }
    }
  }
#pragma unroll
  for (int n = 0; n < NE; ++n)
#pragma unroll
    for (int i = 0; i < CPL; ++i)
      if (c0 + i >= own0) few.d[n][wv][c0 + i] = (double)gload<float>(xe[n] + tcol0(wv) + c0 + i) - acc[i][n];
  __syncthreads();
  if (lane < 16) {
#pragma unroll
    for (int n = 0; n < NE; ++n) {
      double s = 0.0;
#pragma unroll
      for (int mb = 0; mb < MBW; ++mb) {
        const double dv = few.d[n][wv][tlcol(lane, mb)];
        s = tcol_live(wv, lane, mb) ? fma(dv, dv, s) : s;
      }
      s = row16_tree_sum(s);
      if (lane == 15) sh.red[wv][ev[n]] = s;
    }
  }
}

// Up to FOUR evaluations on v_mfma_f64_4x4x4_4b_f64 (the Powell kernel's rounds with 2..4 live machines).  That instruction
// (tools/probes/mfma_f64_4x4_probe.hip) has the 16x16x4 form's B and D lane maps restricted to four rows -- lane l supplies
// B[k = l>>4][column l&15] and receives D[evaluation l>>4][column l&15] -- takes A[evaluation l&3][k = l>>4], is the same
// k-ascending fma chain and issues in 17 cycles instead of 64: a pass over Wm for <= 4 evaluations costs a quarter of the matrix
// time of tucker_mfma and, unlike the vector-ALU pass (+3.7 us per machine), does not grow with the number of live machines.
// Same columns per lane, same chains, same reduction tree => the same bits as tucker_mfma + tucker_residual.
// ev[i] / xe[i], i < 4: machine slot and x row of evaluation i (entries at and beyond `ne` repeat a live one; not stored).
__device__ __attribute__((noinline)) void tucker_mfma4(TuckerShared& sh, const float* __restrict__ Wm, const float* const (&xe)[4],
                                                       const int (&ev)[4], int ne, int tid) {
  const int lane = tid & 63, wv = tid >> 6;
  const int kq = lane >> 4, col = lane & 15;
  const int ia = lane & 3;                                   // the evaluation whose coefficient this lane feeds (A operand)
  const int eva = ia == 0 ? ev[0] : (ia == 1 ? ev[1] : (ia == 2 ? ev[2] : ev[3]));
  const float* wbase = Wm + tcol0(wv);
  double acc[MBW];
#pragma unroll
  for (int mb = 0; mb < MBW; ++mb) acc[mb] = 0.0;
  auto row_off = [&](int qs) {
    const int q = 4 * qs + kq;
    return (size_t)(q < TQ ? q : TQ - 1) * TM;
  };
  // The f32 -> f64 conversion sits right before its MFMA: converting a step ahead between the matrix instructions was measured
  // slower -- on gfx950 nothing issues in the shadow of an f64 MFMA (tools/probes/mfma_f64_shadow_probe.hip: every vector
  // instruction next to one adds its own issue time), so a pass costs (MFMA + conversion) x 374 per wave whatever the order.
  constexpr int R4 = K3_TRING4;
  float wr[R4][MBW];
#pragma unroll
  for (int d = 0; d < R4 - 1; ++d) load11(wbase + row_off(d), col, wr[d]);
  double a = sh.coef[kq][eva];
  auto step = [&](int qs, int slot, bool prefetch) {
    if (prefetch) load11(wbase + row_off(qs + R4 - 1 < TQS ? qs + R4 - 1 : TQS - 1), col, wr[(slot + R4 - 1) % R4]);
    const double an = sh.coef[4 * (qs + 1 < TQS ? qs + 1 : qs) + kq][eva];   // next step's coefficient: its LDS latency hides here
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mb = 0; mb < MBW; ++mb)
      acc[mb] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, (double)wr[slot][mb], acc[mb], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    a = an;
  };
  constexpr int GROUPS = TQS / R4, TAIL = TQS % R4;
  for (int g = 0; g < GROUPS; ++g) {
#pragma unroll
    for (int r = 0; r < R4; ++r) step(g * R4 + r, r, true);
  }
#pragma unroll
  for (int r = 0; r < TAIL; ++r) step(GROUPS * R4 + r, r, false);
  // residual of evaluation id = lane >> 4 over this lane's eleven columns, then the butterfly over the 16 column lanes
  const int id = kq;
  const float* xrow = id == 0 ? xe[0] : (id == 1 ? xe[1] : (id == 2 ? xe[2] : xe[3]));
  const int evd = id == 0 ? ev[0] : (id == 1 ? ev[1] : (id == 2 ? ev[2] : ev[3]));
  float xv[MBW];
  load11(xrow + tcol0(wv), col, xv);
  double s = 0.0;
#pragma unroll
  for (int mb = 0; mb < MBW; ++mb) {
    const double d = (double)xv[mb] - acc[mb];
    s = tcol_live(wv, col, mb) ? fma(d, d, s) : s;
  }
  s = row16_tree_sum(s);
  if (col == 15 && id < ne) sh.red[wv][evd] = s;
}

// Residual norms of the 16 evaluations.  xv[mb][r] = x of evaluation (lane>>4) + 4r at this lane's column of
// block mb.  After the call (barrier inside) tucker_err(e) is valid for every thread.
__device__ __forceinline__ void tucker_residual(TuckerShared& sh, const float (&xv)[MBW][4], const f64x4 (&acc)[MBW],
                                                int tid) {
  const int lane = tid & 63, wv = tid >> 6, col = lane & 15;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int mb = 0; mb < MBW; ++mb) {
    const bool live = tcol_live(wv, col, mb);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double d = (double)xv[mb][r] - acc[mb][r];
      s[r] = live ? fma(d, d, s[r]) : s[r];
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    s[r] = row16_tree_sum(s[r]);
    if (col == 15) sh.red[wv][(lane >> 4) + 4 * r] = s[r];
  }
  __syncthreads();
}

// this lane's eleven x values of a row (the columns of its accumulators): xv[mb] = xrow[tcol0(wave) + tlcol(lane&15, mb)]
__device__ __forceinline__ void tucker_load_x(const float* __restrict__ xrow, int tid, float (&v)[MBW]) {
  load11(xrow + tcol0(tid >> 6), tid & 15, v);
}

__device__ __forceinline__ double tucker_err(const TuckerShared& sh, int e) {
  return 0.5 * (((sh.red[0][e] + sh.red[1][e]) + (sh.red[2][e] + sh.red[3][e])) +
                ((sh.red[4][e] + sh.red[5][e]) + (sh.red[6][e] + sh.red[7][e])));
}

}  // namespace nlml
