// encoder_heads_f16x2_w8.hip -- the strict-fast mode (NLML_MODE_F16X2S) with EIGHT waves per workgroup (two per SIMD).
//
// Same network, blob, LDS images, jobs and -- per accumulator -- the same MFMA sequence as round 3's four-wave strict kernel (a SPLIT
// instantiation of encoder_heads_f16x2.hip's kernel, tested bit-identical to this one and then deleted; reference:
// NLML_HPE_Model_Builder.py:33-53,76-92,115-126) and as the layer-per-launch path, which the tests compare it with bit for bit.
// What changed against the four-wave form is who computes what: a job of the trunk (layer 0: four neuron blocks; layer 1: four; layer 2:
// two) is shared by a PAIR of waves, each taking half of its blocks with both face blocks.  Why:
//   * split accumulators need 2 x (512 neurons x 64 faces) registers per layer-0 pass whatever the wave count; with four waves that is
//     256 of a wave's 512 registers next to a 128-register weight ring, so layer 1's accumulators had to be parked and everything else
//     squeezed (110 spilled registers, 444 B/lane of scratch).  Eight waves hold 128 accumulator registers each; nothing spills.
//   * the L2 -> CU weight stream, not the matrix pipe, bounds layer 0 (9.6 MB per 64-face tile): four waves take in 35-41 B/clk, eight
//     46-51 (tools/probes/l1_stream_probe.hip, profiles/r03_l1_stream_probe.txt) -- more waves asking, more bytes in flight.
//   * two waves per SIMD hide each other's dependent chains: the x staging (f64 normalisation, hi/lo split, LDS writes) and the
//     epilogues of one wave run under the other's MFMAs without hand-placing every link.
// The tail (E3, E4, E5, the three heads; 10 % of the FLOPs, stream-bound short stages) is the four-wave code of
// encoder_heads_f16x2_dev.h run by waves 0-3.  Round 4 let waves 4-7 END after layer 2's store (S_BARRIER waits only for the waves of
// the workgroup that have not terminated: gfx9 ISA, not the HIP programming model); since round 5 they STAY and take blocks 4..7 of every
// head's H1 stage (tail_helper_w8: the same number of barriers as the main path), so nothing relies on that rule any more.  The gain is
// small -- 0.918 against 0.921 ms per 65,536 faces, same box, five alternating runs: H1 is bound by its weight stream, not by its
// MFMAs -- and the bits are unchanged (-DW8_TAIL_EXIT builds round 4's form for A/B).
#include <hip/hip_runtime.h>

#include "../../include/nlml_hpe.h"
#include "abi_internal.h"
#include "encoder_heads_f16x2_dev.h"
#include "layout.h"

// -DW8_TAIL_EXIT (round 4's form, A/B only): waves 4-7 of a workgroup end while waves 0-3 go on through barriers.  That is defined by
// the gfx9 ISA's S_BARRIER (terminated waves are not waited for), not by the HIP programming model -- so that form builds for the
// targets where the rule was read and tested, and nowhere else (an architecture with split or named barriers would hang instead of failing).
#if defined(W8_TAIL_EXIT) && defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "-DW8_TAIL_EXIT relies on S_BARRIER ignoring terminated waves (gfx942 / gfx950)"
#endif

namespace nlml {
namespace hx {

__device__ __forceinline__ double div_ipd_w8(double n, double d, double y) {   // == IEEE n / d for these operands (K1)
  const double q = n * y;
  const double r = fma(-q, d, n);
  return fma(r, y, q);
}

// One K step of a layer-0 pass (two neuron blocks x two face blocks, split accumulators): step_fine's products in step_fine's order
// -- (w_lo, x_hi) and (w_hi, x_lo) into accS, (w_hi, x_hi) into acc, one MFMA per slot -- with the x operands in registers of their
// own: the hi pieces double-buffered (read by the first and the last four MFMAs), the LO pieces SINGLE-buffered: their last reader is
// MFMA 7 of 12, so the next step's lo pieces are fetched into the same registers behind MFMAs 8 and 9 -- eight registers fewer than
// two full operand sets (at 256 registers per wave that is the difference between 93 and 87 spilled registers outside the K loops, and
// +0.6 % faces/s on one box).
template <typename XHi, typename XLo, typename Extra>
__device__ __forceinline__ void step_w8(f32x16 (&acc)[2][2], f32x16 (&accS)[2][2], const h8 (&wcur)[2][2], const h8 (&xh)[2], h8 (&xl)[2],
                                        h8 (&wnext_hi)[2][2], const h8* __restrict__ wp_hi, h8 (&wnext_lo)[2][2], const h8* __restrict__ wp_lo,
                                        XHi xload_hi, XLo xload_lo, Extra extra) {
#pragma unroll
  for (int m = 0; m < 12; ++m) {
    const int t = m / 4, nb = (m % 4) / 2, fb = m % 2;
    if (t == 0) accS[nb][fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wcur[nb][1], xh[fb], accS[nb][fb], 0, 0, 0);
    else if (t == 1) accS[nb][fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wcur[nb][0], xl[fb], accS[nb][fb], 0, 0, 0);
    else acc[nb][fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wcur[nb][0], xh[fb], acc[nb][fb], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (m == 0) xload_hi(0);
    if (m == 1) xload_hi(1);
    if (m == 2) wnext_hi[0][0] = wp_hi[0];          // (the hi and the lo pieces may run a different number of steps ahead)
    if (m == 4) wnext_lo[0][1] = wp_lo[64];
    if (m == 6) wnext_hi[1][0] = wp_hi[128];
    if (m == 8) xload_lo(0);          // (xl's last reader was MFMA 7)
    if (m == 9) xload_lo(1);
    if (m == 10) wnext_lo[1][1] = wp_lo[192];
    extra(m);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// One pass of layer 0 for 512 threads: x[64,F] f32 -> (optional IPD normalisation in f64, FeatureExtractor.py:30-66) -> hi/lo f16 ->
// three rotating LDS slabs of 32 columns (thread = row tid/8, columns 4*(tid%8) .. +3); this wave computes neuron blocks 2*nbh, 2*nbh+1
// of job 4*pass + jw for both face blocks with split accumulators (step_fine: `acc` takes w_hi*x_hi, `accS` the two small products).
template <bool VEC4, bool NORM>
__device__ __forceinline__ void stage_e0_pass_w8(const Ctx& c, const Args& a, int64_t row0, int tid, int pass, int jw, int nbh,
                                                 f32x16 (&acc)[2][2], f32x16 (&accS)[2][2]) {
  constexpr int NB = 2, NFB = 2;
  constexpr int WSTEP = 4 * 2 * 64;                 // a layer-0 job has four blocks x two pieces per K step
  const int F = a.F;
  const int nslab = (int)c.hdr.k8_e0 / XS_STEPS;   // even (pack.cpp)

  const int srow = tid >> 3, scol = (tid & 7) * 4;
  int64_t r = row0 + srow;
  const bool live = r < a.B;
  r = live ? r : a.B - 1;
  const float* p = a.x + r * a.ldx;
  double ipd = 1.0, rcp = 1.0, ra = 0.0, rb = 0.0, rc = 0.0;
  if (NORM) {   // exactly K1's arithmetic: the f32 value the reference feeds the network, bit for bit
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

  // ONE L2 request per 128-byte line of x -- built, bit-identical, measured SLOWER here, kept behind -DW8_XLINE.  A row is 16-byte but not
  // 128-byte aligned (1,404 floats = 43.9 lines), so the 128 bytes a row contributes to a slab straddle two lines, and the line between
  // slabs q and q+1 is requested twice, a slab apart (by then the weight stream -- 64 KB per slab -- has pushed it out of the 32-KB L1):
  // 88 requests per row and pass for 45 lines, 6 % of all the kernel's L2 requests (profiles/r05l1_summary.md).  With -DW8_XLINE the
  // threads whose 16 bytes lie in a slab's FIRST line (`adv`) load one slab ahead: one load instruction then asks, per row, for exactly
  // one whole line, and an `adv` thread's value waits one slab in `carry`.  Same values into the same LDS bytes.  Measured on one box,
  // alternating: 0.880 against 0.844 ms with the weight ring three steps ahead (the four carry registers push the spills from 30 to 52),
  // 0.834 against 0.826 ms with the ring two steps ahead (21 against 15 spills): the saved requests do not pay for the registers.  The
  // bf16 kernel, which has the registers, keeps the scheme (+0.8 %).
#ifndef W8_XLINE
  const bool adv = false;
#else
  const int xphase = (int)((reinterpret_cast<uintptr_t>(p) >> 4) & 7);   // the row's first 16-byte unit within its line
  const bool adv = VEC4 && xphase != 0 && xphase + (tid & 7) < 8;
#endif

  struct Set { float v[4]; };
  [[maybe_unused]] Set carry;
  auto gload_at = [&](int s, Set& st) {   // columns scol .. scol+3 of slab s
    s = s < nslab ? s : nslab - 1;
#ifdef W8_ABL_XHOT   // timing-only ablation (wrong results): every slab's x comes from the row's first 128 bytes (L1 / L2 hot)
    s = 0;
#endif
    const int k = s * XS_COLS + scol;
    if (VEC4) {
      const int kc = k < F ? k : (NORM ? k - 12 * ((k - F + 15) / 12) : F - 4);   // phase-preserving clamp (zero weights there)
#ifdef W8_X_NT   // x lines are used once per pass: ask L2 not to keep them in the weights' way (experiment)
      const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + kc));
#else
      const f32x4 t = *reinterpret_cast<const f32x4*>(p + kc);
#endif
#pragma unroll
      for (int e = 0; e < 4; ++e) st.v[e] = t[e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) st.v[e] = p[k + e < F ? k + e : F - 1];
    }
  };
  auto gload = [&](int s, Set& st) { gload_at(s + (adv ? 1 : 0), st); };   // (slab s+1 in the threads that run a slab ahead)
  auto lw_begin = [&](Set& st) {   // the set's loads must have landed: everything below consumes them
#pragma unroll
    for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(st.v[e]));
#ifdef W8_XLINE
    if (VEC4) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {   // a thread that loads a slab ahead stages what it loaded a slab ago
        const float t = st.v[e];
        st.v[e] = adv ? carry.v[e] : t;
        carry.v[e] = t;
      }
    }
#endif
  };
  auto lw_norm = [&](Set& st, int q) {   // element q (static): the plain chain (prologue)
    const int t = q % 3;
    const double rr = t == 0 ? ra : (t == 1 ? rb : rc);
    st.v[q] = (float)div_ipd_w8((double)st.v[q] - rr, ipd, rcp);
  };
  auto lw_rotate = [&]() {   // next slab: columns + 32 => phase + 2
    const double t0 = rc; rc = rb; rb = ra; ra = t0;
  };
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  unsigned pend_hi[2], pend_lo[2];
  auto lw_split = [&](Set& st, int j, bool real_slab) {   // elements 2j, 2j+1 -> packed hi/lo f16 pairs
    const unsigned m = real_slab ? 0x7fffffffu : 0u;
    nzbits |= (__float_as_uint(st.v[2 * j]) | __float_as_uint(st.v[2 * j + 1])) & m;
    split2(st.v[2 * j], st.v[2 * j + 1], pend_hi[j], pend_lo[j]);
  };
  // x slabs of this kernel: 32 columns = 64 bytes per row and plane, NO padding, the 16-byte chunk c of row r stored at chunk
  // c ^ ((r >> 2) & 3): the 16 lanes of a ds_read_b128 group (rows distinct mod 16, one logical chunk) then cover all sixteen
  // 16-byte slots of the 256-byte bank row -- conflict-free without the pad.  Four slabs (2 x 32 KB... 4 x 8 KB) sit in the LAST
  // 32 KB of LDS, so that the parked layer-1 accumulators (128 KB from offset 0) and the slabs live side by side; the 2 KB by which
  // the h1 / h2 images reach into that range are never live together with a slab (barriers on both sides of every layer-0 pass).
  // The slabs form TWO buffers of 64 columns: an iteration of the main loop reads one (four K steps) while the other is being
  // written for the next iteration -- ONE barrier per four K steps instead of per two (the slab barrier cost 14-16 k of a pass's
  // 96-109 k cycles: every wave waits for the slowest of eight each time; -DW8_ABL_NOBAR stamps).
  constexpr int XW_PLANE = 64 * 64, XW_SLAB = 2 * XW_PLANE, XW_BUF = 2 * XW_SLAB, O_XW = LDS_BYTES - 2 * XW_BUF;
  static_assert(O_XW >= 16 * 512 * 16, "slabs behind the parked accumulators");
  const int wr_off = srow * 64 + ((((tid & 7) >> 1) ^ ((srow >> 2) & 3)) << 4) + 8 * (tid & 1);   // this thread's 8 bytes of a plane
  auto lw_store = [&](int slab_off, int piece) {
    char* d = c.lds + O_XW + slab_off + wr_off;
    if (piece == 0) *reinterpret_cast<u2*>(d) = u2{pend_hi[0], pend_hi[1]};
    else *reinterpret_cast<u2*>(d + XW_PLANE) = u2{pend_lo[0], pend_lo[1]};
  };
  auto lwrite = [&](int slab_off, Set& st, bool real_slab) {
    lw_begin(st);
    if (NORM) {
#pragma unroll
      for (int q = 0; q < 4; ++q) lw_norm(st, q);
      lw_rotate();
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) lw_split(st, j, real_slab);
    lw_store(slab_off, 0);
    lw_store(slab_off, 1);
  };

  const int job = 4 * pass + jw;
  load_bias<NB, NFB>(acc, c.blob4 + c.hdr.b_off(ST_E0) + job * (4 * 8) + (2 * nbh) * 8, c.h);
  zero_acc<NB, NFB>(accS);
  const h8* w = c.blob8 + c.hdr.w_off(ST_E0) + (size_t)job * c.hdr.job_w16(ST_E0) + (2 * nbh * 2) * 64 + c.lane;
#ifdef W8_ABL_WHOT   // timing-only ablation (wrong results): every K step reads the weights of steps 0..3 -- always L2-resident, all CUs the same lines
  auto wfrag = [&](int ks) { return w + (size_t)(ks & 3) * WSTEP; };
#elif defined(W8_ABL_WL2)   // timing-only ablation (wrong results): a small L2-resident region per CU (no shared hot lines)
  auto wfrag = [&](int ks) { return w + (size_t)((ks & 3) + 4 * (blockIdx.x % 20)) * WSTEP; };
#else
  auto wfrag = [&](int ks) { return w + (size_t)ks * WSTEP; };   // K step ks of this wave's two blocks
#endif

  // TWO staging register sets (4 floats per thread each), slab q in set[q & 1]: a slab's global loads are issued one iteration (four K
  // steps) before it is written to LDS.  In pass 0 x comes from HBM, and a load that has not returned holds back every weight load
  // behind it in the in-order return queue: pass 0 takes ~10 k cycles longer than pass 1 (x from L2).  FOUR sets (loads two iterations
  // ahead) fit the register budget since step_w8 single-buffers the lo operands, but need the loop unrolled by two for static set
  // indices, and hipcc's allocation of that 96-MFMA body spills 166 registers, 30 reloads inside the loop: 1.25 ms against 0.84 (measured).
  Set set[2];
#ifndef W8_D0_HI
#define W8_D0_HI 2
#endif
#ifndef W8_D0_LO
#define W8_D0_LO 2
#endif
  // weight ring: K step ks in slot ks % 4 = its step within the iteration, its hi / lo pieces requested DH / DL steps ahead
  constexpr int R0 = 4, DH = W8_D0_HI, DL = W8_D0_LO;
  h8 wr[R0][NB][2];
#ifdef W8_XLINE
  if (VEC4) gload_at(0, carry);   // slab 0 for the threads that run a slab ahead (their first regular load is slab 1's)
#endif
  gload(0, set[0]);
  gload(1, set[1]);
  lwrite(0, set[0], true);
  lwrite(XW_SLAB, set[1], true);
  gload(2, set[0]);
  gload(3, set[1]);
#pragma unroll
  for (int d = 0; d < (DH > DL ? DH : DL); ++d)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int pp = 0; pp < 2; ++pp)
        if (d < (pp ? DL : DH)) wr[d][nb][pp] = wfrag(d)[(nb * 2 + pp) * 64];
  __syncthreads();

  // reader: lane (f, h) of face block fb reads row 32*fb + f, logical chunk 2*(step & 1) + h of slab (step >> 1): two lane
  // addresses per buffer (step parity 0 / 1), formed once per iteration; everything else is an immediate offset
  const int rd0 = c.f * 64 + ((c.h ^ ((c.f >> 2) & 3)) << 4);
  typedef __attribute__((address_space(3))) const char LdsC;   // (address space 3 kept through the opaque copies below: ds_read, not flat_load)
  LdsC* const xb0 = (LdsC*)(c.lds + O_XW + rd0);
  LdsC* const xb1 = (LdsC*)(c.lds + O_XW + (rd0 ^ 32));
  h8 xh[2][NFB], xl[NFB];
  auto xread = [&](LdsC* base, int t, int fb, int pp) {     // x operand (fb, piece pp) of step t (0..3) of a buffer
    return *reinterpret_cast<const __attribute__((address_space(3))) h8*>(base + (t >> 1) * XW_SLAB + pp * XW_PLANE + fb * (32 * 64));
  };
  const int niter = nslab / 2;
#pragma unroll
  for (int fb = 0; fb < NFB; ++fb) {
    xh[0][fb] = xread(xb0, 0, fb, 0);
    xl[fb] = xread(xb0, 0, fb, 1);
  }
#ifdef W8_XTOUCH
  // EXPERIMENT (-DW8_XTOUCH; bit-identical, measured 2.8 % SLOWER: 0.890 against 0.866 ms on one box): x lines pulled into L2 through the
  // SCALAR cache, two iterations before the vector loads ask for them.  Why it loses: scalar loads count in lgkmcnt with the LDS reads, so
  // every `s_waitcnt lgkmcnt(k)` for an x operand of the next MFMAs also waits until all but k of the 16 touches -- each 300-500 cycles
  // away -- have come back.  The idea as it was:  A vector load of
  // a line that is still in HBM (pass A) or the Infinity Cache (pass B) sits 300-500 cycles at the head of the L1's in-order return queue
  // with every younger weight line behind it (-DW8_ABL_XHOT: 23 k cycles per tile); a touch through the vector path costs the same.
  // s_load_dword goes CU -> scalar cache -> L2 on a path of its own.  One touch per row and slab: the slab's last dword, i.e. the line
  // it shares with the next slab.  The destination register is never read; it is kept live (`+s`) across the iteration's barrier, whose
  // lgkmcnt(0) all touches have passed before the next ones are issued.
  unsigned xt_sink = 0;
  auto xtouch = [&](int k, int slab0) {   // touch k of 16: row k & 7 of this wave's eight rows, slab slab0 + (k >> 3)
    int64_t rr = row0 + 8 * c.wv + (k & 7);
    rr = rr < a.B ? rr : a.B - 1;
    int col = (slab0 + (k >> 3)) * XS_COLS + XS_COLS - 1;
    col = col < F ? col : F - 1;
    const float* q = a.x + rr * a.ldx + col;
    asm volatile("s_load_dword %0, %1, 0x0" : "+s"(xt_sink) : "s"(q) : "memory");
  };
#endif
  int cur = 0, nxt = XW_BUF;   // byte offsets of the buffer being read / written
  for (int it = 0; it < niter; ++it) {
    const int s = 2 * it;
    LdsC* xc[2] = {xb0 + cur, xb1 + cur};   // this iteration's buffer, step parity 0 / 1
    LdsC* xn = xb0 + nxt;                   // the next iteration's first step
    asm volatile("" : "+v"(xc[0]), "+v"(xc[1]), "+v"(xn));   // (formed here, not again in front of every read)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int ks = 4 * it + t;
      // The iteration's ONE barrier stands before step 3: by then the buffer being written is complete (slab s+2 is staged during
      // step 0, slab s+3 during step 1) and the buffer being read has been read to its end (step 3's operands were fetched during
      // step 2) -- so step 3 fetches the next iteration's first operands from the new buffer under its own MFMAs, and the next
      // iteration may overwrite the old one from its first slot on.
#ifndef W8_ABL_NOBAR
      if (t == 3) __syncthreads();
#endif
#ifdef W8_XTOUCH
      if (t == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(xt_sink));   // (free behind the barrier: the last iteration's touches have landed)
#endif
      step_w8(acc, accS, wr[t], xh[t & 1], xl, wr[(t + DH) % R0], wfrag(ks + DH), wr[(t + DL) % R0], wfrag(ks + DL),
              [&](int fb) { xh[(t + 1) & 1][fb] = t < 3 ? xread(xc[(t + 1) & 1], t + 1, fb, 0) : xread(xn, 0, fb, 0); },
              [&](int fb) { xl[fb] = t < 3 ? xread(xc[(t + 1) & 1], t + 1, fb, 1) : xread(xn, 0, fb, 1); },
              [&](int m) {   // the staging of slab s+2 (step 0) and s+3 (step 1), one piece behind an MFMA
#ifdef W8_ABL_NOSTAGE
                return;      // timing-only ablation (wrong results)
#endif
#ifdef W8_XTOUCH
                if (t == 3) {   // slabs s+6 and s+7: their vector loads are issued in the next iteration
                  if (m < 4) { xtouch(2 * m, s + 6); xtouch(2 * m + 1, s + 6); }
                  else xtouch(4 + m, s + 6);
                }
#endif
                if (t >= 2) return;
                Set& st = set[t & 1];
                const int j = m;
                const bool real = s + t + 2 < nslab;
                if (j == 0) lw_begin(st);
                // an element's whole division chain in one slot: the partner wave's MFMAs cover its latency (+1 % over link-by-link)
                if (NORM && j >= 1 && j < 5) lw_norm(st, j - 1);
                if (NORM && j == 5) lw_rotate();
                if (j == 6 || j == 7) lw_split(st, j - 6, real);
                if (j == 8) lw_store(nxt + t * XW_SLAB, 0);
                if (j == 9) lw_store(nxt + t * XW_SLAB, 1);
                // (both slabs' loads of an iteration issued back to back -- one far-request episode in the L1's in-order queue instead of
                // two -- measured 2-3 % slower, behind step 1 or behind the barrier: 29 instead of 15 spilled registers either way)
                if (j == 11) gload(s + t + 4, st);
              });
    }
    const int t0 = cur; cur = nxt; nxt = t0;
  }
#ifdef W8_XTOUCH
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(xt_sink));
#endif
  if (pass == 0 && a.valid) {  // all-zero feature row == "no face" (FeatureExtractor.py:105-106); 8 lanes share a row
    const unsigned long long m = __ballot(nzbits != 0u);
    if ((tid & 7) == 0 && live) a.valid[row0 + srow] = ((m >> (c.lane & 56)) & 0xFFull) ? 1 : 0;
  }
}

// Timing-only diagnostic build (-DHX_STAMPS, tools/w8_stage_shares.py): per-wave s_memtime stamps at the trunk's stage boundaries into
// the buffer passed as `latent` (16 slots per wave, 8 waves per tile), the tail's own stamps (encoder_heads_f16x2_dev.h) behind them.
#ifdef HX_STAMPS
#define W8S(i)                                                                                                          \
  do {                                                                                                                  \
    if (a.latent && c.lane == 0)                                                                                        \
      reinterpret_cast<unsigned long long*>(a.latent)[((size_t)blockIdx.x * 8 + c.wv) * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define W8S_WALL(i)                                                                                                     \
  do {                                                                                                                  \
    if (a.latent && c.lane == 0)                                                                                        \
      reinterpret_cast<unsigned long long*>(a.latent)[((size_t)blockIdx.x * 8 + c.wv) * 16 + (i)] = wall_clock64();     \
  } while (0)
#else
#define W8S(i) do { } while (0)
#define W8S_WALL(i) do { } while (0)
#endif

// TRUNK = true (round 5, second half): the launch ends with layer 2 -- its output goes to a.h3ws as MFMA operand fragments
// ([32-face block][K16 step][piece][lane] x 16 bytes: 1 KB per face) and the rest of the network runs as its own launch with the
// weights streaming through LDS ONCE per 256 faces (encoder_heads_f16x2_tailws.hip).  All eight waves work to the end of this launch.
template <bool VEC4, bool NORM, bool TRUNK = false>
__global__ __launch_bounds__(512) void encoder_heads_f16x2_w8_kernel(Args a) {
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
  c.wv = __builtin_amdgcn_readfirstlane(tid >> 6);       // 0..7
  const int jw = c.wv >> 1, nbh = c.wv & 1;              // the job this wave shares with its partner, and its half of the job's blocks
  const int64_t row0 = (int64_t)blockIdx.x * TILE_FACES;

  {  // E0 (two passes of 512 neurons, split accumulators) interleaved with the two K halves of E1
    constexpr int WSTEP1 = 4 * 2 * 64;
    const h8* w1 = c.blob8 + c.hdr.w_off(ST_E1) + (size_t)jw * c.hdr.job_w16(ST_E1) + (2 * nbh * 2) * 64 + c.lane;
    const float inv0 = c.hdr.inv_scale[ST_E0];
    // Layer 1's accumulators (neurons 128*jw + 64*nbh .. +63, both face blocks) are parked in LDS while a layer-0 pass runs (the h1
    // half image is dead then; lane-private 16-byte pieces, conflict-free) -- round 3's four-wave strict kernel's structure, at half the registers.
    f32x16 acc1[2][2];
    f32x4* const park = reinterpret_cast<f32x4*>(c.lds + O_H1H) + tid;   // piece i of this lane at park[512 * i]
    static_assert(16 * 512 * 16 <= 2 * P_H1H, "the parked layer-1 accumulators fit the h1 half image");
    load_bias<2, 2>(acc1, c.blob4 + c.hdr.b_off(ST_E1) + jw * (4 * 8) + (2 * nbh) * 8, c.h);
    auto park_acc1 = [&]() {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int fb = 0; fb < 2; ++fb)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            park[512 * ((nb * 2 + fb) * 4 + q)] = f32x4{acc1[nb][fb][4 * q], acc1[nb][fb][4 * q + 1], acc1[nb][fb][4 * q + 2], acc1[nb][fb][4 * q + 3]};
    };
    park_acc1();
    W8S(0);
    W8S_WALL(14);
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      {
        f32x16 acc0[2][2], acc0s[2][2];
        stage_e0_pass_w8<VEC4, NORM>(c, a, row0, tid, pass, jw, nbh, acc0, acc0s);
        add_acc<2, 2>(acc0, acc0s);
        W8S(1 + 4 * pass);
        // layer 1's accumulators back from LDS before the h1 half image is (over)written
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int fb = 0; fb < 2; ++fb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4 t = park[512 * ((nb * 2 + fb) * 4 + q)];
              acc1[nb][fb][4 * q] = t[0]; acc1[nb][fb][4 * q + 1] = t[1]; acc1[nb][fb][4 * q + 2] = t[2]; acc1[nb][fb][4 * q + 3] = t[3];
            }
        __syncthreads();
        job_store<2, 2, ACT_RELU>(c, acc0, O_H1H, P_H1H, S_H1H, 128 * jw + 64 * nbh, 0, inv0);
      }
      __syncthreads();
      W8S(2 + 4 * pass);
      {  // layer 1 over this K half; its small products are added at the end of the half
        f32x16 acc1s[2][2];
        zero_acc<2, 2>(acc1s);
        kloop<2, 2, 32, WSTEP1>(acc1, acc1s, w1 + (size_t)pass * 32 * WSTEP1, c.lds + O_H1H + (c.f * S_H1H + 8 * c.h) * 2, P_H1H,
                                32 * S_H1H * 2);
        add_acc<2, 2>(acc1, acc1s);
      }
      W8S(3 + 4 * pass);
      __syncthreads();   // H1H is free again (pass 0: for the parked set and pass 1's store; pass 1: for H2)
      if (pass == 0) park_acc1();
      W8S(4 + 4 * pass);
    }
    job_store<2, 2, ACT_RELU>(c, acc1, O_H2, P_H2, S_H2, 128 * jw + 64 * nbh, 0, c.hdr.inv_scale[ST_E1]);
  }
  __syncthreads();
  W8S(9);
  // E2: 512 -> 256, ReLU, split accumulators; job jw's block nbh.  h3 overwrites h2 => barrier between the K loop and the store
  Ctx ct = c;
  ct.wv = c.wv & 3;                                       // the tail's wave index (waves 4-7 only fetch with it, then end)
  f32x16 acc3[1][2];
  h8 wr3[ring_slots(1, 2)][1][2];
  {
    constexpr int WSTEP2 = 2 * 2 * 64;
    f32x16 acc2[1][2], acc2s[1][2];
    load_bias<1, 2>(acc2, c.blob4 + c.hdr.b_off(ST_E2) + jw * (2 * 8) + nbh * 8, c.h);
    zero_acc<1, 2>(acc2s);
    const h8* w2 = c.blob8 + c.hdr.w_off(ST_E2) + (size_t)jw * c.hdr.job_w16(ST_E2) + (nbh * 2) * 64 + c.lane;
    kloop<1, 2, 32, WSTEP2>(acc2, acc2s, w2, c.lds + O_H2 + (c.f * S_H2 + 8 * c.h) * 2, P_H2, 32 * S_H2 * 2);
    add_acc<1, 2>(acc2, acc2s);
    W8S(10);
    if constexpr (TRUNK) {   // h3 = relu(E2) straight from the accumulators to the hand-over buffer (no LDS image, no barrier)
      typedef unsigned u4_ __attribute__((ext_vector_type(4)));
      u4_* const dst = reinterpret_cast<u4_*>(a.h3ws) + (size_t)blockIdx.x * (2 * 16 * 2 * 64) + c.lane;
#pragma unroll
      for (int fb = 0; fb < 2; ++fb) {
        h8 fr[2][2];
        frags_from_acc<ACT_RELU>(acc2[0][fb], c.hdr.inv_scale[ST_E2], fr);
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int pc = 0; pc < 2; ++pc)
            dst[((fb * 16 + 4 * jw + 2 * nbh + p) * 2 + pc) * 64] = __builtin_bit_cast(u4_, fr[p][pc]);
      }
      return;
    }
    tail_pre_e3<false>(ct, acc3, wr3);                    // E3's global fetches in front of the store and the barriers
    __syncthreads();
    job_store<1, 2, ACT_RELU>(c, acc2, O_H3, P_H3, S_H3, 64 * jw + 32 * nbh, 0, c.hdr.inv_scale[ST_E2]);
  }
  __syncthreads();
  W8S(11);
  W8S_WALL(15);
#ifdef W8_TAIL_EXIT   // round 4's form (A/B): waves 4-7 end here; the tail's barriers wait only for the surviving waves (see the header)
  if (c.wv >= 4) return;
  constexpr bool H1W8 = false;
#else                  // round 5: waves 4-7 stay for the heads' H1 stage (blocks 4..7 of each head's eight), tail_helper_w8
  ct.helper = c.wv >= 4 ? 1 : 0;
  constexpr bool H1W8 = true;
#endif
  // (Measured and dropped: on their way out these waves TOUCHED the input rows of the tile that starts one tile time later -- one dword
  // per line, so that its pass 0 finds x in the Infinity Cache instead of HBM: 0.852 ms against 0.834 without, same box.)
#ifdef HX_STAMPS
  Args at = a;             // the tail's stamps (32 slots per wave, 4 waves per tile) behind the trunk's
  if (a.latent) at.latent = a.latent + (size_t)gridDim.x * 8 * 16 * 2;
  tail_stages<false, STRICT_INKERNEL_RESCUE_MAX, H1W8>(ct, at, row0, acc3, wr3);
#else
  tail_stages<false, STRICT_INKERNEL_RESCUE_MAX, H1W8>(ct, a, row0, acc3, wr3);
#endif
}

}  // namespace hx

// ---- the strict-fast forward as trunk launch + streamed tail launch (+ the f32 re-evaluation launch) -------------------------------
bool tailws_supported(const float* x, int64_t ldx, int F) {
  return (F % 4 == 0) && (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
}
size_t tailws_workspace_bytes(int64_t B, int F) {
  if (B <= 0 || F <= 0) return 0;
  return (size_t)((B + TILE_FACES - 1) / TILE_FACES) * (2 * 16 * 2 * 1024);   // 1 KB per face of whole 64-face tiles
}

int launch_encoder_heads_f16x2_tailws(const float* x, int64_t ldx, const float* raw, int normalize, int64_t B, int F,
                                      const void* blob, float* out, float* latent, uint8_t* valid, void* workspace,
                                      size_t ws_bytes, void* stream) {
  if (B == 0) return 0;
  if (!workspace || ws_bytes < tailws_workspace_bytes(B, F)) return fail(NLML_E_BADARG, "streamed-tail path: workspace too small");
  if (reinterpret_cast<uintptr_t>(workspace) & 15) return fail(NLML_E_BADARG, "streamed-tail path: workspace must be 16-byte aligned");
  hx::Args a;
  a.B = B; a.F = F; a.blob = blob; a.out = out; a.latent = latent; a.valid = valid; a.norm = 0; a.h3ws = workspace;
  if (raw) {
    a.x = raw; a.ldx = NLML_F_REFERENCE; a.norm = normalize ? 1 : 0;
  } else {
    a.x = x; a.ldx = ldx;
  }
  if (!tailws_supported(a.x, a.ldx, F)) return fail(NLML_E_BADARG, "streamed-tail path: x must be 16-byte aligned with F and ldx multiples of 4");
  const int64_t ntiles = (B + TILE_FACES - 1) / TILE_FACES;
  const dim3 grid((unsigned)ntiles), block(512);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a.norm) hipLaunchKernelGGL((hx::encoder_heads_f16x2_w8_kernel<true, true, true>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((hx::encoder_heads_f16x2_w8_kernel<true, false, true>), grid, block, 0, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
  if (int rc = launch_tail_ws(blob, workspace, 2 * ntiles, B, out, latent, stream)) return rc;
  return launch_encoder_heads_f32(x, ldx, raw, normalize, B, F, static_cast<const char*>(blob) + strict_f32_image_offset(F), out, latent,
                                  nullptr, nullptr, nullptr, stream, STRICT_INKERNEL_RESCUE_MAX);
}

int launch_encoder_heads_f16x2_w8(const float* x, int64_t ldx, const float* raw, int normalize, int64_t B, int F,
                                  const void* blob, float* out, float* latent, uint8_t* valid, void* stream) {
  if (B == 0) return 0;
  hx::Args a;
  a.B = B; a.F = F; a.blob = blob; a.out = out; a.latent = latent; a.valid = valid; a.norm = 0;
  if (raw) {
    a.x = raw; a.ldx = NLML_F_REFERENCE; a.norm = normalize ? 1 : 0;
  } else {
    a.x = x; a.ldx = ldx;
  }
  const bool vec4 = (F % 4 == 0) && (a.ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0);
  const dim3 grid((unsigned)((B + TILE_FACES - 1) / TILE_FACES)), block(512);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a.norm) {
    if (vec4) hipLaunchKernelGGL((hx::encoder_heads_f16x2_w8_kernel<true, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((hx::encoder_heads_f16x2_w8_kernel<false, true>), grid, block, 0, st, a);
  } else {
    if (vec4) hipLaunchKernelGGL((hx::encoder_heads_f16x2_w8_kernel<true, false>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((hx::encoder_heads_f16x2_w8_kernel<false, false>), grid, block, 0, st, a);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
  // tiles with faces beyond f16's range (more than STRICT_INKERNEL_RESCUE_MAX = 0 of them): the whole tile again on the f32 matrix cores, from the
  // f32 image behind the split-f16 one; every other tile of that launch ends after one 768-byte read (encoder_heads.hip)
  return launch_encoder_heads_f32(x, ldx, raw, normalize, B, F, static_cast<const char*>(blob) + strict_f32_image_offset(F), out, latent,
                                  nullptr, nullptr, nullptr, stream, STRICT_INKERNEL_RESCUE_MAX);
}

}  // namespace nlml
