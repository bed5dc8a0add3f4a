// tucker_ref.h -- the Tucker objective in the REFERENCE'S OWN OPERATION ORDER (NLML_TD_ORDER_REFERENCE): the parity mode of the TD path.
//
// The matrix-core path (tucker_common.h) evaluates x_hat = c^T Wm as a GEMM: c = ((u*f_y)*f_p)*f_r first, then one fma chain per
// output.  That agrees with the reference's objective to ~1e-15 relative -- and still moves the END POINT of the Powell
// minimisation (the minimum is flat and Powell's termination is rounding-sensitive; tests/test_powell_sm.py measures scipy itself
// doing that).  This file evaluates the objective exactly as the reference does (TD_Tester.py:46,49), bit for bit:
//
//   np.einsum('ijklm,i,j,k,l->m', W, u, f_y, f_p, f_r)   numpy's generic sum-of-products loop: for (i,j,k,l) in nesting order,
//       for every m:  x_hat[m] = ((((W[i,j,k,l,m] * u_i) * f_yj) * f_pk) * f_rl) + x_hat[m],  each operation rounded on its own;
//   0.5 * np.sum((x - x_hat)**2)                         numpy's pairwise sum: 16 leaves of 80/88/92 elements, each as eight
//       strided partial sums combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus remainder, the leaves added as a balanced tree.
//
// W sits innermost in every product, so nothing can be shared between columns or evaluations: 5 separately rounded f64 operations
// per (q, m) and evaluation -- 135 * 1404 * 5 = 947,700 vector-ALU operations per evaluation against one fma per (q, m) on the
// matrix path.  The bound is the f64 vector issue rate (16 lanes per clock and SIMD: 39.3 T operations/s at 2.4 GHz, i.e.
// 41.5 M evaluations/s); oracle: oracle/csrc/oracle.c oracle_tucker_objective_reforder, pinned to FX4 bit for bit.
//
// One workgroup of 768 threads = 12 waves, three per SIMD; NE = 1..8 evaluations share every Wm load.  A one-evaluation pass puts thread t
// on columns t and t+768 (1404 of the 1536 lane slots live); passes of 2..8 evaluations deal (wave-column, half of the evaluations)
// units so that every SIMD issues the same number of chains (see "BALANCED passes" at tucker_ref_pass).  Three waves per SIMD because the
// waves of a SIMD do not advance together: the oldest takes every issue slot it can use, its siblings finish one after the other, and
// the last one runs alone -- at the 4.8 cycles per instruction ONE wave sustains -- for 1/3 of the pass (1/2 with two waves); since
// round 4 the waves lower their issue priority as they progress (s_setprio in the main loop), which keeps the three together.
// What keeps the vector ALUs fed (round 3; the round-2 form -- 512 threads x 3 columns -- ran at 0.46 of the issue rate, this one
// at 0.58 in a burst / 0.69 sustained, 0.63 / 0.74 with the balanced passes: DESIGN.md section 3 has the stamps and what bounds it now):
//   * Wm rows come through a three-slot register ring, the loads of block (i,j,k)+2 issued before the arithmetic of block
//     (i,j,k) (buffer loads: scalar row offset + the lane's column offset, no address arithmetic on the vector ALUs).  Round 2
//     loaded the nine values of a block at its top and used them at once: 45 exposed L2 round trips per pass;
//   * the innermost factor f_r[l] of every evaluation lives in scalar registers for the whole pass (it was an LDS read per (l, n)),
//     u / f_y / f_p are re-read from a compact per-pass LDS table only when their loop level advances, through ONE opaque base register;
//   * the five dependent operations of a (column, evaluation) pair advance TR_ILV evaluations x 2 columns at a time, stage by stage;
//   * every LDS access is a real ds_ instruction (address space 3): through the generic references this non-inlined function
//     receives they were flat_ loads, which count on vmcnt as well and so waited for the Wm prefetch just issued.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "tucker_common.h"

namespace nlml {

#ifndef TR_NT_N
#define TR_NT_N 768
#endif
constexpr int TR_NT = TR_NT_N;                // threads of a reference-order workgroup: 12 waves, three per SIMD
constexpr int TR_COLS = 2;                 // columns per thread: 2 * 768 = 1536 >= 1404
constexpr int TR_MAXE = 8;                 // evaluations per pass over Wm
constexpr int TR_LEAVES = 16;              // numpy pairwise tree for n = 1404 (tests/test_oracle_golden.py pins it to np.sum)
#ifndef TR_ILV_N
#define TR_ILV_N 4
#endif
constexpr int TR_ILV = TR_ILV_N;           // evaluations whose operation chains advance together (x 2 columns)
constexpr int TR_NFAC = 14;                // u[5], f_y[3], f_p[3], f_r[3] of one evaluation

// the (column, evaluation) pairs of a main-loop variant, evaluation-major (tucker_ref_pass has the variants)
template <int NE, int VAR>
struct TrPairs {
  int cnt;
  int c[2 * NE], n[2 * NE];
  constexpr TrPairs() : cnt(0), c{}, n{} {
    constexpr int SP = (NE + 1) / 2;
    for (int nn = 0; nn < NE; ++nn)
      for (int cc = 0; cc < 2; ++cc) {
        const bool on = cc == 0 || VAR == 0 || (VAR == 1 ? nn < SP : nn >= SP);
        if (on) {
          c[cnt] = cc;
          n[cnt] = nn;
          ++cnt;
        }
      }
  }
};

struct TuckerRefShared {
  double d2[TR_MAXE][TM + 4];              // squared residuals of the pass's evaluations
  double leaf8[TR_MAXE][TR_LEAVES][8];     // strided partial sums of every leaf
  double leaf[TR_MAXE][TR_LEAVES];
  double fac[TR_MAXE][TR_NFAC + 2];        // the pass's factors, by evaluation
  double err[EV];                          // objective values of the round, by machine slot
};

__device__ __forceinline__ int tr_leaf_start(int L) { return L == 0 ? 0 : 80 + 88 * (L - 1); }
__device__ __forceinline__ int tr_leaf_len(int L) { return L == 0 ? 80 : (L == TR_LEAVES - 1 ? 92 : 88); }

// LDS accesses through a generic pointer, as ds_ instructions
template <typename T>
__device__ __forceinline__ T lds_ld(const void* p) {
  return *(const __attribute__((address_space(3))) T*)p;
}
template <typename T>
__device__ __forceinline__ void lds_st(void* p, T v) {
  *(__attribute__((address_space(3))) T*)p = v;
}
// a value every lane holds alike, moved to scalar registers
__device__ __forceinline__ double uniform_f64(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// Evaluations of machine slots (slots >> 4n) & 15, n < NE (f-vectors in sh.fvec[slot], parameters par(slot, k)) against the rows
// xrow(slot); results into rs.err[slot] (and x_hat rows where xhrow(slot) != nullptr).  All TR_NT (768) threads; barriers inside.
// par / xrow / xhrow are small structs passed BY VALUE (a reference into the caller's frame would be scratch memory here).
//
// BALANCED passes (NE >= TR_BAL_FROM): with thread t on columns t and t + 768 the 22 wave-columns (1404 / 64) fall 6 / 6 / 5 / 5 on
// the four SIMDs (waves w, w + 4, w + 8 share a SIMD: tools/probes/wave_simd_map_probe.hip), so two SIMDs issue 48 (wave-column,
// evaluation) chains while the pass needs 44 per SIMD: 0.917 of the issue slots at best.  Here the unit of work is (wave-column,
// half of the evaluations): SIMD class s = w & 3 takes wave-columns 5s .. 5s+4 whole -- two each for its first two waves, one for
// the third -- and the third wave adds HALF of wave-column 20 + (s >> 1): evaluations [0, SP) for even s, [SP, NE) for odd s, SP = ceil(NE / 2).
// Every SIMD then issues 5 NE + SP (or + NE - SP) chains, 44 of 44 at NE = 8.  Same chains per (column, evaluation): same bits.
#ifndef TR_BAL_FROM
#define TR_BAL_FROM 2
#endif
template <int NE, typename ParT, typename XRow, typename XhRow>
__device__ __attribute__((noinline)) void tucker_ref_pass(const TuckerShared& sh, TuckerRefShared& rs, const float* __restrict__ Wm_,
                                                          const ParT par, const unsigned slots, const XRow xrow,
                                                          const XhRow xhrow, const int tid) {
#pragma clang fp contract(off)
  // LDS through address-space-3 pointers taken ONCE: every access is then a ds_ instruction with an immediate offset from one base
  // register (per-element casts of generic addresses cost an address register each -- ~30 of the 168 this function may use)
  typedef __attribute__((address_space(3))) TuckerRefShared LdsRef;
  typedef __attribute__((address_space(3))) const TuckerShared LdsSh;
  LdsRef* const rl = (LdsRef*)&rs;
  LdsSh* const sl = (LdsSh*)&sh;
#ifdef TR_STAMPS   // timing-only diagnostic build (tools/td_ref_stamps.py): s_memtime at the phase boundaries of each pass into the x_hat buffer
  unsigned long long* const stamp_base = reinterpret_cast<unsigned long long*>(xhrow.x_hat) +
                                         (((size_t)blockIdx.x * 2 + ((slots & 15) ? 1 : 0)) * 12 + (tid >> 6)) * 8;
#define TRS(i) do { if (xhrow.x_hat && (tid & 63) == 0) stamp_base[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TRS(i) do { } while (0)
#endif
  TRS(0);
  // Wm as a buffer resource: a load is (scalar row offset) + (the lane's column offset in a vector register), so the ring costs no
  // address arithmetic on the vector ALUs.  Every load stays INSIDE Wm by construction, not by the descriptor's range check (the
  // scalar offset is not part of that check): the two prefetches past the last block re-read row 134 (scalar min), and a dead
  // lane (column >= 1404) reads column 0 of its row; neither value is used.  Arguments of a non-inlined function arrive in vector
  // registers: the base is made uniform first.
  const float* Wm = reinterpret_cast<const float*>(
      ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)Wm_ >> 32)) << 32) |
      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint64_t)Wm_));
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)Wm, 0, TQ * TM * 4, 0x00027000);
  constexpr bool BAL = NE >= TR_BAL_FROM && TR_NT == 768 && TR_COLS == 2;
  unsigned mc[TR_COLS];
  bool livec[TR_COLS];
  int var = 0;                    // wave-uniform: 0 = both columns take every evaluation; 1 / 2 = column 1 takes [0, SP) / [SP, NE) only
  if constexpr (BAL) {
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), s4 = w & 3, r = w >> 2;
    const int wc0 = 5 * s4 + 2 * r, wc1 = r < 2 ? wc0 + 1 : 20 + (s4 >> 1);
    mc[0] = wc0 * 64 + (tid & 63);
    mc[1] = wc1 * 64 + (tid & 63);
    var = r < 2 ? 0 : 1 + (s4 & 1);
  } else {
#pragma unroll
    for (int c = 0; c < TR_COLS; ++c) mc[c] = tid + TR_NT * c;
  }
#pragma unroll
  for (int c = 0; c < TR_COLS; ++c) livec[c] = mc[c] < (unsigned)TM;
  constexpr int SP = (NE + 1) / 2;   // where the split wave-column's evaluations divide
  const int c1_lo = var == 2 ? SP : 0, c1_hi = var == 1 ? SP : NE;   // column 1's evaluations
  // Wm ring: block b = (i*3 + j)*3 + k holds rows 3b .. 3b+2 and lives in slot k; the loads of block b + 2 are issued before the
  // arithmetic of block b.  (Measured: a five- or nine-slot ring is no faster -- the L2 latency is covered; re-reading u and f_y
  // from LDS with every block is 7 % slower; holding f_p for all three k in registers spills at the 168 registers that three
  // waves per SIMD allow.)  f_p is read at the top of its block, u and f_y when their loop level advances.
  float wr[3][3][TR_COLS];
  unsigned voff[TR_COLS];
#pragma unroll
  for (int c = 0; c < TR_COLS; ++c) voff[c] = livec[c] ? mc[c] * 4 : 0u;
  auto wload = [&](int b, float (&dst)[3][TR_COLS]) {
#pragma unroll
    for (int l = 0; l < 3; ++l) {
      const int row = 3 * b + l < TQ ? 3 * b + l : TQ - 1;
#pragma unroll
      for (int c = 0; c < TR_COLS; ++c)
        dst[l][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrs, voff[c], row * (TM * 4), 0));
    }
  };
  // the pass's factors -> rs.fac[n][0..4] = u, [5..7] = f_y, [8..10] = f_p, [11..13] = f_r.  Their loads are issued first, the
  // first two blocks of the ring next (memory loads return in order), so the ring's latency overlaps the table and its barrier
  double facv = 0.0;
  if (tid < NE * TR_NFAC) {
    const int n = tid / TR_NFAC, f = tid % TR_NFAC;
    const int slot = (slots >> (4 * n)) & 15;
    facv = f < 5 ? par(slot, 3 + f) : sl->fvec[slot][(f - 5) / 3][(f - 5) % 3];
  }
  __builtin_amdgcn_sched_barrier(0);
  wload(0, wr[0]);
  wload(1, wr[1]);
#ifndef TR_NO_XPREFETCH
  // this thread's x values ride with the prologue's loads and wait in the d2 slots they will be subtracted in (as f32 in the low
  // half of the slot: written and read by the same thread) -- at the end of the main loop they used to be one exposed round trip
  // to global memory for the wave that finishes last
  {
    float xv[TR_COLS][NE];
#pragma unroll
    for (int n = 0; n < NE; ++n) {
      const float* xr = xrow((slots >> (4 * n)) & 15);
#pragma unroll
      for (int c = 0; c < TR_COLS; ++c)
        xv[c][n] = (livec[c] && (c == 0 || (n >= c1_lo && n < c1_hi))) ? gload<float>(xr + mc[c]) : 0.0f;
    }
#pragma unroll
    for (int n = 0; n < NE; ++n)
#pragma unroll
      for (int c = 0; c < TR_COLS; ++c)
        if (livec[c] && (c == 0 || (n >= c1_lo && n < c1_hi))) *(__attribute__((address_space(3))) float*)&rl->d2[n][mc[c]] = xv[c][n];
  }
#endif
  __builtin_amdgcn_sched_barrier(0);
  if (tid < NE * TR_NFAC) rl->fac[tid / TR_NFAC][tid % TR_NFAC] = facv;
  __syncthreads();

  // the factor table through ONE opaque base register: its LDS address is a compile-time constant beyond the 16-bit offset field,
  // and hipcc otherwise keeps every element's absolute address in a register of its own across the loop (24 of them)
  unsigned fac_addr = (unsigned)(uintptr_t)&rl->fac[0][0];
  asm volatile("" : "+v"(fac_addr));
  const __attribute__((address_space(3))) double* const fac = (const __attribute__((address_space(3))) double*)(uintptr_t)fac_addr;
  constexpr int FS = TR_NFAC + 2;          // row stride of rs.fac
  double fr[3][NE];
#pragma unroll
  for (int l = 0; l < 3; ++l)
#pragma unroll
    for (int n = 0; n < NE; ++n) fr[l][n] = uniform_f64(fac[n * FS + 11 + l]);

  double acc[TR_COLS][NE];
#pragma unroll
  for (int c = 0; c < TR_COLS; ++c)
#pragma unroll
    for (int n = 0; n < NE; ++n) acc[c][n] = 0.0;

#ifdef TR_FP_RESIDENT
  double fpr[3][NE];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int n = 0; n < NE; ++n) fpr[k][n] = fac[n * FS + 8 + k];
#endif
  TRS(1);

  // (i, j, k, l) in the einsum's nesting order.  VAR (compile time inside, chosen per wave): which evaluations column 1 takes
  auto main_loop = [&](auto var_c) {
  constexpr int VAR = decltype(var_c)::value;
  constexpr TrPairs<NE, VAR> PR{};
  constexpr int CNT = PR.cnt, NG = (CNT + 2 * TR_ILV - 1) / (2 * TR_ILV), G = (CNT + NG - 1) / NG;
#pragma unroll 1
  for (int i = 0; i < 5; ++i) {
    double u[NE];
#pragma unroll
    for (int n = 0; n < NE; ++n) u[n] = fac[n * FS + i];
#pragma unroll 1
    for (int j = 0; j < 3; ++j) {
      double fy[NE];
#pragma unroll
      for (int n = 0; n < NE; ++n) fy[n] = fac[n * FS + 5 + j];
      const int b0 = (i * 3 + j) * 3;
#ifndef TR_NO_PRIO
      {  // Issue priority falls with progress (s_setprio takes an immediate: four levels).  The arbiter serves the OLDEST wave of a SIMD
         // first, so its three waves finished one after the other and the last ran alone, at the 4.8 cycles per instruction one wave
         // sustains, for a third of the pass (main loop by wave: 59.7 k / 101.8 k / 131.5 k ticks; -DTR_NO_PRIO).  A wave that is
         // ahead now has the lower priority: it still fills the slots its siblings leave, but cannot run away -- only the last
         // segment's imbalance is left, hence segments of 8, 4, 2 and 1 of the 15 (i, j) steps: 119.8 k / 122.6 k / 121.6 k ticks
         // against 118.8 k of pure issue (135 rows x 44 chains x 5 operations x 4 cycles), the pass 143.3 k -> 132.5 k.
        const int ij = i * 3 + j;
        if (ij == 0) __builtin_amdgcn_s_setprio(3);
        else if (ij == 8) __builtin_amdgcn_s_setprio(2);
        else if (ij == 12) __builtin_amdgcn_s_setprio(1);
        else if (ij == 14) __builtin_amdgcn_s_setprio(0);
      }
#endif
#pragma unroll
      for (int k = 0; k < 3; ++k) {
#ifndef TR_ABL_NOLOAD
        wload(b0 + k + 2, wr[(k + 2) % 3]);
#endif
        double fp[NE];
#pragma unroll
#ifdef TR_FP_RESIDENT
        for (int n = 0; n < NE; ++n) fp[n] = fpr[k][n];
#else
        for (int n = 0; n < NE; ++n) fp[n] = fac[n * FS + 8 + k];
#endif
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int l = 0; l < 3; ++l) {
          double wd[TR_COLS];
#pragma unroll
          for (int c = 0; c < TR_COLS; ++c) wd[c] = (double)wr[k][l][c];
          // the five operations of a (column, evaluation) pair depend on each other; the sweep's pairs advance together in groups of
          // up to 2 * TR_ILV chains of equal size (16 pairs: 8 + 8, 12: 6 + 6), one operation each per stage (stages pinned by
          // sched_barrier: left alone, hipcc runs the chains one after the other through a single temporary)
#pragma unroll
          for (int g0 = 0; g0 < NG; ++g0) {
            double t[G];
#pragma unroll
            for (int q = 0; q < G; ++q)
              if (g0 * G + q < CNT) t[q] = wd[PR.c[g0 * G + q]] * u[PR.n[g0 * G + q]];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < G; ++q)
              if (g0 * G + q < CNT) t[q] = t[q] * fy[PR.n[g0 * G + q]];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < G; ++q)
              if (g0 * G + q < CNT) t[q] = t[q] * fp[PR.n[g0 * G + q]];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < G; ++q)
              if (g0 * G + q < CNT) t[q] = t[q] * fr[l][PR.n[g0 * G + q]];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < G; ++q)
              if (g0 * G + q < CNT) acc[PR.c[g0 * G + q]][PR.n[g0 * G + q]] = t[q] + acc[PR.c[g0 * G + q]][PR.n[g0 * G + q]];
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
  }
  };
  if constexpr (BAL) {
    if (var == 0) main_loop(std::integral_constant<int, 0>{});
    else if (var == 1) main_loop(std::integral_constant<int, 1>{});
    else main_loop(std::integral_constant<int, 2>{});
  } else {
    main_loop(std::integral_constant<int, 0>{});
  }
  TRS(2);
  // residuals squared -> LDS (and x_hat out)
#pragma unroll
  for (int n = 0; n < NE; ++n) {
    const int slot = (slots >> (4 * n)) & 15;
#ifdef TR_NO_XPREFETCH
    const float* xr = xrow(slot);
#endif
#ifdef TR_STAMPS
    double* xh = nullptr;
#else
    double* xh = xhrow(slot);
#endif
#pragma unroll
    for (int c = 0; c < TR_COLS; ++c)
      if (livec[c] && (c == 0 || (n >= c1_lo && n < c1_hi))) {
#ifndef TR_NO_XPREFETCH
        const double d = (double)*(const __attribute__((address_space(3))) float*)&rl->d2[n][mc[c]] - acc[c][n];
#else
        const double d = (double)gload<float>(xr + mc[c]) - acc[c][n];
#endif
        rl->d2[n][mc[c]] = d * d;
        if (xh) xh[mc[c]] = acc[c][n];
      }
  }
  __syncthreads();
  TRS(3);
  // numpy's pairwise sum, level 1: eight strided partial sums per leaf
  for (int t = tid; t < NE * TR_LEAVES * 8; t += TR_NT) {
    const int n = t / (TR_LEAVES * 8), L = (t / 8) % TR_LEAVES, jj = t % 8;
    const __attribute__((address_space(3))) double* a = rl->d2[n] + tr_leaf_start(L);
    const int len = tr_leaf_len(L), body = len - (len % 8);
    // 10 or 11 elements (leaves of 80 / 88 / 92): all reads first (a loop with a run-time trip count waits for every LDS read on
    // its own: 1.1 k ticks per sum), then the adds in numpy's order; the surplus read of a 10-element sum is a clamped re-read
    const int nq = body >> 3;
    double v[11];
#pragma unroll
    for (int q = 0; q < 11; ++q) v[q] = a[jj + 8 * (q < nq ? q : nq - 1)];
    double r = v[0];
#pragma unroll
    for (int q = 1; q < 11; ++q) {
      const double r1 = r + v[q];
      r = q < nq ? r1 : r;
    }
    rl->leaf8[n][L][jj] = r;
  }
  __syncthreads();
  TRS(4);
  // level 2: combine the eight, then the remainder elements one by one
  for (int t = tid; t < NE * TR_LEAVES; t += TR_NT) {
    const int n = t / TR_LEAVES, L = t % TR_LEAVES;
    const __attribute__((address_space(3))) double* r = rl->leaf8[n][L];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    const __attribute__((address_space(3))) double* a = rl->d2[n] + tr_leaf_start(L);
    const int len = tr_leaf_len(L);
    const int rem = len & 7, b8 = len - rem;            // 0 or 4 remainder elements
    double w4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) w4[q] = a[q < rem ? b8 + q : 0];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double r1 = res + w4[q];
      res = q < rem ? r1 : res;
    }
    rl->leaf[n][L] = res;
  }
  __syncthreads();
  TRS(5);
  // level 3: the balanced tree over the 16 leaves, then * 0.5
  if (tid < NE) {
    double s[TR_LEAVES];
#pragma unroll
    for (int L = 0; L < TR_LEAVES; ++L) s[L] = rl->leaf[tid][L];
    const double a0 = (s[0] + s[1]) + (s[2] + s[3]), a1 = (s[4] + s[5]) + (s[6] + s[7]);
    const double a2 = (s[8] + s[9]) + (s[10] + s[11]), a3 = (s[12] + s[13]) + (s[14] + s[15]);
    rl->err[(slots >> (4 * tid)) & 15] = 0.5 * ((a0 + a1) + (a2 + a3));
  }
  __syncthreads();
  TRS(6);
#undef TRS
}

// f-vectors of all 16 slots into sh.fvec (tucker_coef's first half; the coefficient table is not used in this order)
template <typename ParT>
__device__ __forceinline__ void tucker_fvec(TuckerShared& sh, const ParT& par, const double (&cp4)[4], int tid) {
#pragma clang fp contract(off)   // numpy rounds b*w, + c, a*cos, + d separately (TD_Tester.py:25-28)
  if (tid < EV * 9) {
    const int e = tid / 9, a = (tid % 9) / 3;
    // float32(a * cos(b * w + c) + d) with the cos correctly rounded, by the library cos wherever that cannot change the float (cr_cos.h)
    sh.fvec[e][a][tid % 3] = (double)cr_f32_a_cos_d(cp4[0], cp4[1] * par(e, a) + cp4[2], cp4[3]);
  }
  __syncthreads();
}

// The evaluations whose bit is set in `mask` (machine slots), up to eight per pass over Wm.  A pass costs what its evaluations
// cost on the vector ALUs (~6 us each at the issue rate) but never less than streaming Wm through the CU (~10 us), so nine or
// more live machines go as two passes of about the same size.  xrow(slot) -> that slot's x row; xhrow(slot) -> its x_hat output
// row or nullptr.
template <typename ParT, typename XRow, typename XhRow>
__device__ __forceinline__ void tucker_ref_eval(const TuckerShared& sh, TuckerRefShared& rs, const float* __restrict__ Wm,
                                                const ParT par, int mask, const XRow xrow, const XhRow xhrow, int tid) {
  while (mask) {
    const int cnt = __popc(mask);
    const int take = cnt <= TR_MAXE ? cnt : (cnt + 1) / 2;   // cnt <= EV = 16
    unsigned slots = 0;
    for (int i = 0; i < take; ++i) {
      slots |= (unsigned)(__ffs(mask) - 1) << (4 * i);
      mask &= mask - 1;
    }
    switch (take) {
      case 8: tucker_ref_pass<8>(sh, rs, Wm, par, slots, xrow, xhrow, tid); break;
      case 7: tucker_ref_pass<7>(sh, rs, Wm, par, slots, xrow, xhrow, tid); break;
      case 6: tucker_ref_pass<6>(sh, rs, Wm, par, slots, xrow, xhrow, tid); break;
      case 5: tucker_ref_pass<5>(sh, rs, Wm, par, slots, xrow, xhrow, tid); break;
      case 4: tucker_ref_pass<4>(sh, rs, Wm, par, slots, xrow, xhrow, tid); break;
      case 3: tucker_ref_pass<3>(sh, rs, Wm, par, slots, xrow, xhrow, tid); break;
      case 2: tucker_ref_pass<2>(sh, rs, Wm, par, slots, xrow, xhrow, tid); break;
      case 1: tucker_ref_pass<1>(sh, rs, Wm, par, slots, xrow, xhrow, tid); break;
      default: break;
    }
  }
}

}  // namespace nlml
