// encoder_heads_f16x2.hip -- K2 on the f16 matrix cores: the four-wave kernel of the opt-in fast mode (NLML_MODE_F16X2).  The default
// strict-fast mode (NLML_MODE_F16X2S: the same operands and instructions with split accumulators throughout) runs on the eight-wave
// kernel of encoder_heads_f16x2_w8.hip; both share encoder_heads_f16x2_dev.h.
//
// Same network, stages and jobs as the f32 parity kernel (encoder_heads.hip; reference:
// NLML_HPE_Model_Builder.py:33-53,76-92,115-126) and the same <=1e-4 degree bar, but the contraction runs on
// the f16 matrix cores: every f32 operand v is carried as two f16 pieces, v = hi + lo, hi = f16(v),
// lo = f16(v - hi) (v - hi is exact in f32 and lo keeps 11 bits of it: >= 22 significand bits), and a product is evaluated as
//
//     w*x  ~  w_hi*x_hi + w_hi*x_lo + w_lo*x_hi          (three v_mfma_f32_32x32x16_f16, f32 accumulate)
//
// The dropped w_lo*x_lo term is 2^-22 relative, the same order as f32 rounding; measured pose error against
// the f64 oracle is ~1e-5 degree, like the f32 kernel's (tests/test_gpu_parity.py, tests/studies/
// split_precision_study.py).  Three 32-cycle f16 MFMAs replace eight 64-cycle f32 MFMAs per 16 k values, so
// the matrix pipe has 5.3x less work for the same operand bytes (4 per weight, 4 per activation).  Measured:
// 80 M faces/s fused at B = 65,536 (2.8x the f32 kernel), at the board's power limit.
//
// Range: activations of 65520 and more do not fit f16; hi becomes inf, lo -inf and the MFMA path yields NaN for that
// face -- which the tile then re-evaluates in f32 on the vector ALUs from the same blob (encoder_heads_f16x2_rescue.h),
// so there is no input-range limit.  f16 subnormal pieces are kept by the MFMA (tools/probes/mfma_f16_probe.hip), so small
// values lose nothing beyond an absolute 2^-25.
//
// Structure: 64-face tiles, 4 waves; layer 0's 1024-wide output of 64 faces is 256 KB in hi+lo f16, so it reaches layer 1 in two
// halves of 512 neurons, each followed by one K half of layer 1 (ONE pass over x on 256 accumulators per lane, the second half
// waiting in registers; the eight-wave strict kernel: two passes over x of 128 + 128 accumulators, layer 1's parked in LDS meanwhile);
// x is staged f32 -> (optional f64 IPD normalisation) -> hi/lo f16 through three rotating 32-column LDS slabs; every K step
// issues ONE MFMA per slot with the step's fetches and the staging pieces spread behind them (step_fine); a stage's global
// fetches ride in the previous stage's epilogue; the heads run one at a time over both face blocks (encoder_heads_f16x2_dev.h).
#include <hip/hip_runtime.h>


#include "../../include/nlml_hpe.h"
#include "abi_internal.h"
#include "encoder_heads_f16x2_dev.h"
#include "layout.h"

namespace nlml {
namespace hx {

// operand types, MFMA steps, K loops, stores and the network's tail: encoder_heads_f16x2_dev.h

__device__ __forceinline__ double div_ipd(double n, double d, double y) {
  const double q = n * y;
  const double r = fma(-q, d, n);
  return fma(r, y, q);
}

// ------------------------------------------------------------------------------------------
// Layer 0 in ONE pass over x: x[64,F] f32 -> (optional IPD normalisation in f64) -> hi/lo f16 -> three rotating LDS slabs of
// 32 columns, staged ONCE per face; per K step this wave runs BOTH of its jobs -- neurons 128w.. (job w, accA) and 512+128w..
// (job 4+w, accB), 4 blocks x 2 face blocks each -- on the same x operands: 256 accumulators per lane = the whole AGPR file,
// which is why layer 1's accumulators do not exist yet (the caller stores accA, runs layer 1's first K half, then accB).
// Until round 2 this was two passes over x (one per job): the staging (global loads, normalisation, split, LDS writes) ran
// twice per face and cost 12 % of the launch (no-staging ablation); jobs, blob and per-accumulator MFMA order are unchanged,
// so the results are bit-identical to the two-pass form (which the layer-per-launch path still computes).
template <bool VEC4, bool NORM>
__device__ __forceinline__ void stage_e0(const Ctx& c, const Args& a, int64_t row0, int tid, f32x16 (&accA)[4][2],
                                         f32x16 (&accB)[4][2]) {
  constexpr int NB = 4, NFB = 2;
  const int F = a.F;
  const int nslab = (int)c.hdr.k8_e0 / XS_STEPS;   // even (pack.cpp)
  constexpr int SLAB_BYTES = 2 * P_XS;

  // staging role: row srow (0..63), 8 consecutive columns scol..scol+7 of every slab
  const int srow = tid >> 2, scol = (tid & 3) * 8;
  int64_t r = row0 + srow;
  const bool live = r < a.B;
  r = live ? r : a.B - 1;
  const float* p = a.x + r * a.ldx;
  // IPD normalisation (FeatureExtractor.py:30-66) in f64 exactly like K1 and the f32 kernel: the f32 value the
  // reference feeds the network is reproduced bit for bit (div_ipd == IEEE f64 division for these operands).  A
  // cheaper f32 form, (x - ref) * (1/ipd), is ~1.5 ulp off and that alone moved faces with a tiny IPD by up to
  // 3e-4 degree (65,536 random faces), so it is not used.
  double ipd = 1.0, rcp = 1.0, ra = 0.0, rb = 0.0, rc = 0.0;
  if (NORM) {
    const double dx = (double)p[99] - (double)p[789], dy = (double)p[100] - (double)p[790], dz = (double)p[101] - (double)p[791];
    ipd = sqrt(fma(dz, dz, fma(dy, dy, dx * dx)));
    if (ipd == 0.0) ipd = 1e-6;
    rcp = 1.0 / ipd;
    const double x0 = (double)p[3], y0 = (double)p[4], z0 = (double)p[5];
    const int ph = scol % 3;   // coordinate of this thread's first column; a slab later the phase is + 32 % 3 = + 2
    ra = ph == 0 ? x0 : (ph == 1 ? y0 : z0);
    rb = ph == 0 ? y0 : (ph == 1 ? z0 : x0);
    rc = ph == 0 ? z0 : (ph == 1 ? x0 : y0);
  }
  unsigned nzbits = 0u;

  // staging set: 8 consecutive columns of one row as scalars (each piece below touches single elements)
  struct Set { float v[8]; };
  auto gload_half = [&](int s, Set& st, int i) {   // columns scol + 4i .. + 3 of slab s
    s = s < nslab ? s : nslab - 1;
    const int k = s * XS_COLS + scol + 4 * i;
    if (VEC4) {
      const int kc = k < F ? k : (NORM ? k - 12 * ((k - F + 15) / 12) : F - 4);   // phase-preserving clamp
      const f32x4 t = *reinterpret_cast<const f32x4*>(p + kc);
#pragma unroll
      for (int e = 0; e < 4; ++e) st.v[4 * i + e] = t[e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) st.v[4 * i + e] = p[k + e < F ? k + e : F - 1];
    }
  };
  auto gload = [&](int s, Set& st) { gload_half(s, st, 0); gload_half(s, st, 1); };
  // staging of one slab in pieces, one per free MFMA slot of the slab's two K steps (lwrite() = all of them, prologue)
  auto lw_begin = [&](Set& st) {   // the set's loads must have landed: everything below consumes them
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(st.v[e]));
  };
  // Normalisation = six DEPENDENT f64 instructions per element (convert, subtract, multiply, two correction fmas, convert).
  // Issued as one chain behind one MFMA (the first form) they cost ~77 cycles per element: the chain's latency, ~13 cycles per
  // link, not its issue time -- tools/probes/mfma_f16_dp_probe.hip shows that up to three INDEPENDENT f64 instructions do hide
  // in the shadow of a 32x32x16 f16 MFMA.  So the loop form (lw_norm2) gives every free slot ONE link of the chains of TWO
  // elements: the next link of the same element comes a whole MFMA later.  lw_norm (prologue only) is the plain chain.
  auto lw_norm = [&](Set& st, int q) {   // element q (static)
    const int t = q % 3;
    const double rr = t == 0 ? ra : (t == 1 ? rb : rc);
    st.v[q] = (float)div_ipd((double)st.v[q] - rr, ipd, rcp);
  };
  double nA = 0.0, nB = 0.0, qA = 0.0, qB = 0.0;   // the pair in flight
  auto lw_norm2 = [&](Set& st, int pair, int link) {   // elements 2*pair, 2*pair+1 (static), link 0..5 of div_ipd's chain
    const int e0 = 2 * pair, e1 = e0 + 1;
    const double r0 = e0 % 3 == 0 ? ra : (e0 % 3 == 1 ? rb : rc), r1 = e1 % 3 == 0 ? ra : (e1 % 3 == 1 ? rb : rc);
    // (the empty asm pins each link to its slot: pure arithmetic is otherwise sunk towards its use, back into one cluster)
    if (link == 0) { nA = (double)st.v[e0]; nB = (double)st.v[e1]; asm volatile("" : "+v"(nA), "+v"(nB)); }
    if (link == 1) { nA = nA - r0; nB = nB - r1; asm volatile("" : "+v"(nA), "+v"(nB)); }
    if (link == 2) { qA = nA * rcp; qB = nB * rcp; asm volatile("" : "+v"(qA), "+v"(qB)); }
    if (link == 3) { nA = fma(-qA, ipd, nA); nB = fma(-qB, ipd, nB); asm volatile("" : "+v"(nA), "+v"(nB)); }
    if (link == 4) { qA = fma(nA, rcp, qA); qB = fma(nB, rcp, qB); asm volatile("" : "+v"(qA), "+v"(qB)); }
    if (link == 5) { st.v[e0] = (float)qA; st.v[e1] = (float)qB; asm volatile("" : "+v"(st.v[e0]), "+v"(st.v[e1])); }
  };
  auto lw_rotate = [&]() {   // next slab: columns + 32 => phase + 2
    const double t0 = rc; rc = rb; rb = ra; ra = t0;
  };
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  unsigned pend_hi[4], pend_lo[4];
  auto lw_split = [&](Set& st, int j, bool real_slab) {   // elements 2j, 2j+1 -> packed hi/lo f16 pairs
    const unsigned m = real_slab ? 0x7fffffffu : 0u;
    // split2 takes the f32 VALUES (asm operands), so the compiler cannot fold (f16)(f32)double into one f64 -> f16 conversion
    // as hipcc 7.2 does for the C form (~20 integer instructions per element and a different rounding)
    nzbits |= (__float_as_uint(st.v[2 * j]) | __float_as_uint(st.v[2 * j + 1])) & m;
    split2(st.v[2 * j], st.v[2 * j + 1], pend_hi[j], pend_lo[j]);
  };
  auto lw_store = [&](int buf_off, int piece) {
    char* d = c.lds + O_XS + buf_off + (srow * S_XS + scol) * 2;
    if (piece == 0) *reinterpret_cast<u4*>(d) = u4{pend_hi[0], pend_hi[1], pend_hi[2], pend_hi[3]};
    else *reinterpret_cast<u4*>(d + P_XS) = u4{pend_lo[0], pend_lo[1], pend_lo[2], pend_lo[3]};
  };
  auto lwrite = [&](int buf_off, Set& st, bool real_slab) {
    lw_begin(st);
    if (NORM) {
#pragma unroll
      for (int q = 0; q < 8; ++q) lw_norm(st, q);
      lw_rotate();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) lw_split(st, j, real_slab);
    lw_store(buf_off, 0);
    lw_store(buf_off, 1);
  };

  load_bias<NB, NFB>(accA, c.blob4 + c.hdr.b_off(ST_E0) + c.wv * (NB * 8), c.h);
  load_bias<NB, NFB>(accB, c.blob4 + c.hdr.b_off(ST_E0) + (4 + c.wv) * (NB * 8), c.h);
  const h8* wA = c.blob8 + c.hdr.w_off(ST_E0) + (size_t)c.wv * c.hdr.job_w16(ST_E0) + c.lane;
  const h8* wB = wA + (size_t)4 * c.hdr.job_w16(ST_E0);
  // the weight stream in consumption order: half-step hs = 2 * (K step) + (0: job A, 1: job B)
  auto wfrag = [&](int hs) { return ((hs & 1) ? wB : wA) + (size_t)(hs >> 1) * (NB * 2 * 64); };

  // TWO staging register sets (8 floats per thread each), one per slab parity: slab s+2 is written to LDS during
  // slab s from set[s & 1], which is refilled at once with the loads of slab s+4.  vmcnt counts in issue
  // order, so a set must be older than every weight load still wanted in flight when it is waited for: two slabs
  // (4 K steps, 32 weight loads) lie between its loads and its use, and in the prologue the sets are loaded BEFORE
  // the weight ring so that the loop header sees the same distance on entry as on the back edge.
  Set set[2];
  constexpr int R0 = 4, D0 = R0 - 1;   // weight ring: half-step hs in slot hs % 4 = 2 * (step of the slab) + job
  static_assert(2 * XS_STEPS == R0, "one slab == four half-steps == ring slots");
  h8 wr[R0][NB][2];
  gload(0, set[0]);
  gload(1, set[1]);
  lwrite(0, set[0], true);
  gload(2, set[0]);
  lwrite(SLAB_BYTES, set[1], true);
  gload(3, set[1]);
#pragma unroll
  for (int d = 0; d < D0; ++d)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) wr[d][nb][pp] = wfrag(d)[(nb * 2 + pp) * 64];
  __syncthreads();

  const int lane_off = (c.f * S_XS + 8 * c.h) * 2;   // bytes
  constexpr int FB = 32 * S_XS * 2;                  // face block stride inside a plane
  h8 xr[2][NFB][2];
#pragma unroll
  for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) xr[0][fb][pp] = *reinterpret_cast<const h8*>(c.lds + O_XS + lane_off + pp * P_XS + fb * FB);

  int o0 = 0, o1 = SLAB_BYTES, o2 = 2 * SLAB_BYTES;   // buffers of slabs s, s+1, s+2
  auto slab = [&](int s, auto par_c) {
    constexpr int PAR = decltype(par_c)::value;
    const char* xrow = c.lds + O_XS + o0 + lane_off;
    const char* xnext = c.lds + O_XS + o1 + lane_off;
    const bool real = s + 2 < nslab;
#pragma unroll
    for (int kk = 0; kk < XS_STEPS; ++kk) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int slot = 2 * kk + half;                       // == half-step % 4: a slab is exactly one turn of the ring
        const int hs = 2 * (s * XS_STEPS + kk) + half;
        step_fine<NB, NFB>(half ? accB : accA, half ? accB : accA, wr[slot], xr[kk & 1], wr[(slot + D0) % R0], wfrag(hs + D0), true,
                [&](int fb, int pp) {                         // the next K step's x operands, fetched during job A's half
                  if (half) return;
                  xr[(kk + 1) & 1][fb][pp] = (kk + 1 < XS_STEPS)
                                                 ? *reinterpret_cast<const h8*>(xrow + pp * P_XS + fb * FB + 32 * (kk + 1))
                                                 : *reinterpret_cast<const h8*>(xnext + pp * P_XS + fb * FB);
                },
                [&](int m) {   // slab s+2's staging in the free (odd) slots of the slab's four half-steps
#ifdef HX_ABL_NOSTAGE
                  return;      // timing-only ablation (wrong results)
#endif
                  if ((m & 1) == 0) return;
                  const int j = 12 * slot + (m >> 1);          // free slot 0..47 of this slab
                  // 48 free slots per slab: 24 normalisation links (4 pairs x 6), rotate, 4 splits, 2 LDS stores, 2 reloads
                  if (j == 0) lw_begin(set[PAR]);
#ifdef HX_NORM_CHAIN   // the first form, kept for A/B timing: one whole element per slot
                  if (NORM && j >= 2 && j < 18 && (j & 1) == 0) lw_norm(set[PAR], (j - 2) >> 1);
                  if (NORM && j == 18) lw_rotate();
#else
                  if (NORM && j < 24) lw_norm2(set[PAR], j / 6, j % 6);
                  if (NORM && j == 24) lw_rotate();
#endif
                  if (j >= 26 && j < 34 && (j & 1) == 0) lw_split(set[PAR], (j - 26) >> 1, real);
                  if (j == 34) lw_store(o2, 0);
                  if (j == 36) lw_store(o2, 1);
                  if (j == 38) gload_half(s + 4, set[PAR], 0);
                  if (j == 40) gload_half(s + 4, set[PAR], 1);
                });
      }
    }
#ifndef HX_ABL_NOBAR
    __syncthreads();
#endif
    const int t0 = o0;   // rotate: (o0, o1, o2) <- (o1, o2, o0)
    o0 = o1; o1 = o2; o2 = t0;
  };
  for (int s = 0; s < nslab; s += 2) {
    slab(s, std::integral_constant<int, 0>{});
    slab(s + 1, std::integral_constant<int, 1>{});
  }
  if (a.valid) {  // all-zero feature row == "no face" (FeatureExtractor.py:105-106); 4 lanes share a row
    const unsigned long long m = __ballot(nzbits != 0u);
    if ((tid & 3) == 0 && live) a.valid[row0 + srow] = ((m >> (c.lane & 60)) & 0xFull) ? 1 : 0;
  }
}

// ------------------------------------------------------------------------------------------
// NLML_MODE_F16X2: layer 0 in one pass over x on single accumulators, so is layer 1's first K half (no register is free while layer
// 0's second half waits); split accumulators from layer 1's second K half on.  (Until the end of round 4 a template parameter SPLIT
// also made this kernel the strict-fast mode's -- two passes of layer 0 with split accumulators, layer 1's parked in LDS: the
// eight-wave kernel of encoder_heads_f16x2_w8.hip computes the same bits 15 % faster and is the only strict-fast kernel now.)
template <bool VEC4, bool NORM>
__global__ __launch_bounds__(256, 1) void encoder_heads_f16x2_kernel(Args a) {
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];

  const int tid = threadIdx.x;
  Ctx c;
  c.blob8 = reinterpret_cast<const h8*>(a.blob);
  c.blob4 = reinterpret_cast<const f32x4*>(a.blob);
  c.hdr = load_hdr(reinterpret_cast<const Header*>(a.blob));
  c.lds = lds;
  c.lane = tid & 63;
  c.f = c.lane & 31;
  c.h = c.lane >> 5;
  c.wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wv = c.wv;
  const int64_t row0 = (int64_t)blockIdx.x * TILE_FACES;

  f32x16 acc2[2][2];
  h8 wr2[ring_slots(2, 2)][2][2];
  {  // E0 (one pass over x, both neuron halves) then the two K halves of E1
    f32x16 acc1[4][2];
    const h8* w1 = c.blob8 + c.hdr.w_off(ST_E1) + (size_t)wv * c.hdr.job_w16(ST_E1) + c.lane;
    const float inv0 = c.hdr.inv_scale[ST_E0];
    HXS(0);
    HXS_WALL(30);
    {
      f32x16 acc0a[4][2], acc0b[4][2];
      stage_e0<VEC4, NORM>(c, a, row0, tid, acc0a, acc0b);
      HXS(1);
      // acc0b (128 registers) has to survive layer 1's first K half next to layer 1's 128 accumulators: the AGPR file is full
      // and the compiler spilled one tile of it to scratch -- sixteen reloads, each waited for on its own (11.7 k cycles for
      // this store in the fused kernel against 5.2 k without the spill).  The x slabs' LDS region is free from here on: one
      // tile is parked there by hand (four 16-byte LDS stores and loads per lane).
      f32x4* stash = reinterpret_cast<f32x4*>(c.lds + O_XS) + tid * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        stash[q] = f32x4{acc0b[3][1][4 * q], acc0b[3][1][4 * q + 1], acc0b[3][1][4 * q + 2], acc0b[3][1][4 * q + 3]};
      // layer 1's accumulators start to exist here (bias): until now layer 0 held every accumulator register
      load_bias<4, 2>(acc1, c.blob4 + c.hdr.b_off(ST_E1) + wv * (4 * 8), c.h);
      job_store<4, 2, ACT_RELU>(c, acc0a, O_H1H, P_H1H, S_H1H, 128 * wv, 0, inv0);
      __syncthreads();
      HXS(2);
      kloop<4, 2, 32>(acc1, acc1, w1, c.lds + O_H1H + (c.f * S_H1H + 8 * c.h) * 2, P_H1H, 32 * S_H1H * 2);
      HXS(3);
      __syncthreads();   // H1H is free again: for the second half of layer 0's output
      HXS(4);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 t = stash[q];
        acc0b[3][1][4 * q] = t[0]; acc0b[3][1][4 * q + 1] = t[1]; acc0b[3][1][4 * q + 2] = t[2]; acc0b[3][1][4 * q + 3] = t[3];
      }
      job_store<4, 2, ACT_RELU>(c, acc0b, O_H1H, P_H1H, S_H1H, 128 * wv, 0, inv0);
      HXS(29);
    }
    __syncthreads();
    HXS(5);
    {  // layer 0's second half is stored: 128 registers are free, and layer 1's second K half runs with the small products of each
       // step in accumulators of their own (see step_fine; FX3c p50 2.08e-5 -> 1.86e-5 deg together with layer 2's, at no cost)
      f32x16 acc1s[4][2];
      zero_acc<4, 2>(acc1s);
      kloop<4, 2, 32>(acc1, acc1s, w1 + (size_t)32 * (4 * 2 * 64), c.lds + O_H1H + (c.f * S_H1H + 8 * c.h) * 2, P_H1H, 32 * S_H1H * 2);
      add_acc<4, 2>(acc1, acc1s);
    }
    HXS(6);
    __syncthreads();     // H1H is free again: for H2
    HXS(7);
    HXS(8);
    // E2's global fetches (bias, first ring slots) ride in E1's epilogue (store_lds hook, encoder_heads_f16x2_dev.h)
    job_store<4, 2, ACT_RELU>(c, acc1, O_H2, P_H2, S_H2, 128 * wv, 0, c.hdr.inv_scale[ST_E1],
                              fetch_hook<16, 2, 2, ST_E2>(c, wv, acc2, wr2));
  }
  __syncthreads();
  HXS(9);
  f32x16 acc3[1][2];
  h8 wr3[ring_slots(1, 2)][1][2];
  // E2: 512 -> 256, ReLU; h3 overwrites h2 => barrier between the K loop and the store
  job_run_split<2, 2, ST_E2>(c, wv, acc2, wr2, O_H2, P_H2, S_H2, 0, 0);   // (split accumulators)
  __syncthreads();
  job_store<2, 2, ACT_RELU>(c, acc2, O_H3, P_H3, S_H3, 64 * wv, 0, c.hdr.inv_scale[ST_E2],
                            fetch_hook<8, 1, 2, ST_E3>(c, wv, acc3, wr3));
  __syncthreads();
  HXS(10);
  tail_stages<false, 64>(c, a, row0, acc3, wr3);
  HXS_WALL(31);
}

}  // namespace hx

int launch_encoder_heads_f16x2(const float* x, int64_t ldx, const float* raw, int normalize, int64_t B, int F,
                               const void* blob, float* out, float* latent, uint8_t* valid, int split, void* stream) {
  if (B == 0) return 0;
  // the strict-fast mode runs on the eight-wave kernel (encoder_heads_f16x2_w8.hip)
  if (split) return launch_encoder_heads_f16x2_w8(x, ldx, raw, normalize, B, F, blob, out, latent, valid, stream);
  hx::Args a;
  a.B = B; a.F = F; a.blob = blob; a.out = out; a.latent = latent; a.valid = valid; a.norm = 0;
  if (raw) {
    a.x = raw; a.ldx = NLML_F_REFERENCE; a.norm = normalize ? 1 : 0;
  } else {
    a.x = x; a.ldx = ldx;
  }
  const bool vec4 = (F % 4 == 0) && (a.ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0);
  const dim3 grid((unsigned)((B + TILE_FACES - 1) / TILE_FACES)), block(256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a.norm) {
    if (vec4) hipLaunchKernelGGL((hx::encoder_heads_f16x2_kernel<true, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((hx::encoder_heads_f16x2_kernel<false, true>), grid, block, 0, st, a);
  } else {
    if (vec4) hipLaunchKernelGGL((hx::encoder_heads_f16x2_kernel<true, false>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((hx::encoder_heads_f16x2_kernel<false, false>), grid, block, 0, st, a);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
  return 0;
}

}  // namespace nlml
