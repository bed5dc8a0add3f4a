// encoder_heads_f16x2_tailws.hip -- the strict-fast mode's TAIL (layers E3, E4, E5 and the three heads: NLML_HPE_Model_Builder.py:45-53,
// 76-92,115-126) as a launch of its own with the roles of weights and faces exchanged.
//
// In the fused kernel (encoder_heads_f16x2_w8.hip) a 64-face tile stays on its CU through the whole network, so the tail's 1.06 MB of
// split-f16 weights stream L2 -> CU once per 64 faces: thirteen short, barrier-separated stages on four of the CU's eight waves, bound by that
// stream and by their own start-up latencies (63.5 k of a tile's 361 k cycles at 43 % of the matrix pipe).  Here a WAVE
// owns one 32-face block for the whole tail and keeps its activations in REGISTERS: faces are MFMA columns and a lane's accumulators
// belong to one face, so a layer's output becomes the next layer's B operand by the scale / activation / hi-lo split of store_lds and one
// half-wave exchange (frags_from_acc) -- no LDS image, no barrier between layers.  The WEIGHTS go through LDS: a workgroup of eight waves
// (256 faces) walks the blob's tail unit by unit -- 31 units of 8 .. 64 KB, each one contiguous in the blob in exactly the fragment order
// the MFMAs want -- copying unit u+1 global -> LDS by LDS-DMA (global_load_lds_dwordx4, no staging registers) while all eight waves
// compute unit u from the other buffer.  A wave always runs TWO independent accumulator chains (two jobs, or the two blocks of an H1 job),
// the schedule of the big jobs is pinned (tw_job2p), and the two waves of a SIMD run half a unit apart by where their barrier stands (see
// TW_MID / TW_END): one wave's epilogue runs under its partner's MFMAs.  The tail's weight stream is paid once per 256 faces instead of
// once per 64 and one barrier per unit and wave is all the synchronisation there is.
//
// Measured (round 5, B = 65,536): trunk launch + this launch + re-evaluation launch against the fused kernel, same box, alternating:
// +1.4 ... +2.2 % (0.801 against 0.815 ms; tools/ab_streamed.py: +1.40 / +1.54 / +1.54 % where the round's first form -- one chain per wave,
// 39 units of <= 32 KB, a job's epilogue sliced into the next job's MFMA shadows, common barriers -- reads +0.89 / +0.95 / +1.26 %).
// rocprofv3 (profiles/r05st2_*): trunk 776 us on that box, this kernel 100.0 us (204 k cycles at 2.05 GHz, MFMA pipe 0.51 busy, no LDS bank
// conflicts), re-evaluation 5.3 us.  The trunk alone costs 0.87-0.90 of the fused kernel -- its share of the L1 -> L2 requests (78.8 M of
// 88.5 M): the fused kernel's time IS its L2 traffic, so taking the tail out buys the tail's bytes and no more.
// This kernel's own cycles (-DTW_STAMPS, tools/tw_stamps.py, profiles/r05st2_tail_stamps.txt): 183 k per 256 faces = prologue 23 k (h3:
// 64 MB chip-wide at ~5.7 TB/s) + 160 k for 1,632 MFMAs per 32 faces (104 k at the pipe's rate) and 62 epilogue blocks per wave.
// Forms measured and dropped, all bit-identical: one chain per wave in lock step 215 k / with sliced epilogues 200 k;
// two chains compiler-scheduled 206 k (hipcc issues each ds_read_b128 in front of its MFMA with lgkmcnt(0)); two chains with the previous
// job's epilogue in the MFMA shadows 255 k; hand-counted lgkmcnt waits 202 k; the half-unit skew as two code instances (spills).
//
// Bits: per accumulator the same bias, the same K-ascending MFMA sequence ((w_lo, x_hi), (w_hi, x_lo), (w_hi, x_hi) per K step, single
// accumulators as in the fused tail), the same epilogue arithmetic => bit-identical to tail_stages() (tests: the fused kernel and the
// layer-per-launch path against this one, every batch shape).  Input: layer 2's output as operand fragments, written by the trunk-only
// instantiation of the eight-wave kernel: h3[32-face block][K16 step][piece][lane] x 16 bytes.
#include <hip/hip_runtime.h>

#include "../../include/nlml_hpe.h"
#include "abi_internal.h"
#include "encoder_heads_f16x2_dev.h"
#include "layout.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "global_load_lds_dwordx4: gfx950"
#endif

namespace nlml {
namespace hx {

typedef unsigned u4 __attribute__((ext_vector_type(4)));
#ifdef TW_NO_PRIO
#define TW_PRIO(p) do { } while (0)
#else
#define TW_PRIO(p) __builtin_amdgcn_s_setprio(p)
#endif
typedef __attribute__((address_space(3))) const char LdsC;
typedef __attribute__((address_space(3))) char LdsW;

struct TwArgs {
  const void* blob;
  const u4* h3;
  float* out;
  float* latent;
  int64_t B, nfb;   // faces; 32-face blocks present in h3 (whole 64-face tiles)
};

// LDS: two unit buffers | the tail's biases (stage order, accumulator-register order as in the blob) | per wave the latent's fragments
// (their non-zero part: lanes 0..31 hold k = 0..3 of head g's K step in 8 bytes per piece; everything else of the fragment is exact zeros)
constexpr int TW_SLOT = 65536;
constexpr int tw_bias_n16(int s) { return kStages[s].jobs * kStages[s].nb * 8; }   // 16-byte units
constexpr int tw_bias_off(int s) {                                                  // bytes from TW_O_BIAS
  int o = 0;
  for (int t = ST_E3; t < s; ++t) o += tw_bias_n16(t) * 16;
  return o;
}
constexpr int TW_O_BIAS = 2 * TW_SLOT;
constexpr int TW_O_LAT = TW_O_BIAS + ((tw_bias_off(NUM_STAGES) + 511) / 512) * 512;
constexpr int TW_LAT_WAVE = 3 * 2 * 256;
constexpr int TW_LDS = TW_O_LAT + 8 * TW_LAT_WAVE;
static_assert(TW_LDS <= 163840, "LDS map");

// The units, in the order they are consumed -- each one a contiguous piece of the blob of at most 64 KB holding TWO independent
// accumulator chains' worth of work at a time (a dependent v_mfma_f32_32x32x16_f16 chain issues every ~62 cycles, the pipe takes one
// every 32: stamped with one chain per wave, 3.0 k cycles for the older wave of a SIMD and 4.6 k for the younger per 48 MFMAs each):
//   E3 jobs (0,1) | E3 jobs (2,3) | E4 (both jobs) | E5 (one job of two blocks) |
//   per head g: H0 (four jobs) | H1 job 0 | H1 job 1 | H1 job 2 | H1 job 3 (a job's two blocks = the two chains) | H2 jobs (0,1) | H2 jobs (2,3) |
//   H3 (both jobs) | H4
constexpr int TW_HEAD_UNITS = 9, TW_UNITS = 4 + 3 * TW_HEAD_UNITS;
__device__ __forceinline__ void tw_unit(const HdrRegs& H, int idx, uint32_t& off, uint32_t& n16) {
  constexpr uint32_t J3 = stage_job_w16_const(ST_E3), J4 = stage_job_w16_const(ST_E4), J5 = stage_job_w16_const(ST_E5);
  constexpr uint32_t K0 = stage_job_w16_const(ST_H0), K1 = stage_job_w16_const(ST_H1), K2 = stage_job_w16_const(ST_H2);
  constexpr uint32_t K3 = stage_job_w16_const(ST_H3), K4 = stage_job_w16_const(ST_H4);
  static_assert(2 * J3 * 16 == TW_SLOT && 2 * J4 * 16 <= TW_SLOT && J5 * 16 <= TW_SLOT && 4 * K0 * 16 <= TW_SLOT && K1 * 16 <= TW_SLOT &&
                    2 * K2 * 16 == TW_SLOT && 2 * K3 * 16 <= TW_SLOT && K4 * 16 <= TW_SLOT,
                "a unit fits a buffer");
  if (idx < 2) { n16 = 2 * J3; off = H.w_off(ST_E3) + idx * 2 * J3; }
  else if (idx == 2) { n16 = 2 * J4; off = H.w_off(ST_E4); }
  else if (idx == 3) { n16 = J5; off = H.w_off(ST_E5); }
  else {
    const int k = idx - 4, g = k / TW_HEAD_UNITS, r = k - TW_HEAD_UNITS * g;
    if (r == 0) { n16 = 4 * K0; off = H.w_off(ST_H0) + 4 * g * K0; }
    else if (r <= 4) { n16 = K1; off = H.w_off(ST_H1) + (4 * g + (r - 1)) * K1; }
    else if (r <= 6) { n16 = 2 * K2; off = H.w_off(ST_H2) + (4 * g + 2 * (r - 5)) * K2; }
    else if (r == 7) { n16 = 2 * K3; off = H.w_off(ST_H3) + 2 * g * K3; }
    else { n16 = K4; off = H.w_off(ST_H4) + g * K4; }
  }
}

// TWO one-block accumulator chains for this wave's face block, K16 steps each, over the same input operands (registers): weights from the
// LDS copy of the unit (`w0`, `w1`: each chain's first hi fragment + this lane's 16 bytes; `wstep` bytes between K steps), bias from the
// LDS table (each chain's block + 64 * h).  Per accumulator: the bias, then per K step (w_lo, x_hi), (w_hi, x_lo), (w_hi, x_hi) --
// mma_step's order; consecutive MFMAs alternate between the chains.
template <int K16, int NCH = 2>
__device__ __forceinline__ void tw_job2(f32x16 (&acc)[2], LdsC* w0, LdsC* w1, int wstep, LdsC* b0, LdsC* b1, const h8 (&in)[K16][2]) {
  typedef const __attribute__((address_space(3))) h8 LdsH8;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((c ? b1 : b0) + q * 16);
      acc[c][4 * q + 0] = v[0];
      acc[c][4 * q + 1] = v[1];
      acc[c][4 * q + 2] = v[2];
      acc[c][4 * q + 3] = v[3];
    }
  TW_PRIO(2);   // (a wave in its MFMAs goes before its SIMD partner's epilogue at the issue port)
#pragma unroll
  for (int ks = 0; ks < K16; ++ks) {
    h8 wf[NCH][2];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int p = 0; p < 2; ++p) wf[c][p] = *reinterpret_cast<LdsH8*>((c ? w1 : w0) + ks * wstep + p * 1024);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[c][t == 0 ? 1 : 0], in[ks][t == 1 ? 1 : 0], acc[c], 0, 0, 0);
  }
  TW_PRIO(0);
}

// tw_job2 with the schedule PINNED (the heads): hipcc, at 246 registers, issues each weight fragment's ds_read_b128 directly in front of the
// MFMA that needs it with a full lgkmcnt(0) wait -- one exposed LDS latency per K step and chain, the same on every wave of the SIMD at
// once (stamped: 63 % of the matrix pipe).  Here the next K step's four fragments are fetched into a second operand set, one read behind
// each of the step's first four MFMAs, and sched_barrier keeps every slot where it is (as step_fine does in the fused kernel).
template <int K16>
__device__ __forceinline__ void tw_job2p(f32x16 (&acc)[2], LdsC* w0, LdsC* w1, int wstep, LdsC* b0, LdsC* b1, const h8 (&in)[K16][2]) {
  typedef const __attribute__((address_space(3))) h8 LdsH8;
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((c ? b1 : b0) + q * 16);
      acc[c][4 * q + 0] = v[0];
      acc[c][4 * q + 1] = v[1];
      acc[c][4 * q + 2] = v[2];
      acc[c][4 * q + 3] = v[3];
    }
  h8 wf[2][2][2];   // [K step parity][chain][piece]
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int p = 0; p < 2; ++p) wf[0][c][p] = *reinterpret_cast<LdsH8*>((c ? w1 : w0) + p * 1024);
  TW_PRIO(2);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ks = 0; ks < K16; ++ks) {
#pragma unroll
    for (int m = 0; m < 6; ++m) {
      const int t = m >> 1, c = m & 1;
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks & 1][c][t == 0 ? 1 : 0], in[ks][t == 1 ? 1 : 0], acc[c], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // the lo pieces are read first (t == 0 uses them); slot m < 4 fetches (chain m & 1, piece m < 2 ? lo : hi) of step ks + 1
      if (m < 4 && ks + 1 < K16)
        wf[(ks + 1) & 1][m & 1][m < 2 ? 1 : 0] = *reinterpret_cast<LdsH8*>(((m & 1) ? w1 : w0) + (ks + 1) * wstep + (m < 2 ? 1024 : 0));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  TW_PRIO(0);
}

__global__ __launch_bounds__(512) void tail_ws_kernel(TwArgs a) {
  __shared__ __attribute__((aligned(1024))) char lds[TW_LDS];
  const int tid = threadIdx.x, lane = tid & 63, f = lane & 31, h = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const HdrRegs H = load_hdr(reinterpret_cast<const Header*>(a.blob));
  const h8* const blob8 = reinterpret_cast<const h8*>(a.blob);
  const f32x4* const blob4 = reinterpret_cast<const f32x4*>(a.blob);

  // this wave's 32-face block; a wave beyond the batch computes the last block again and stores nothing
  int64_t fbg = (int64_t)blockIdx.x * 8 + wv;
  const bool wave_live = fbg < a.nfb;
  fbg = wave_live ? fbg : a.nfb - 1;
  const int64_t row = fbg * 32 + f;
  const bool row_live = wave_live && row < a.B;

  LdsW* const ring = (LdsW*)(lds);
  auto issue = [&](int idx, int slot_off) {   // unit idx: global -> LDS by DMA, 1 KB per wave-instruction, pieces dealt round-robin to the waves
    if (idx >= TW_UNITS) return;
    uint32_t off, n16;
    tw_unit(H, idx, off, n16);
    const int npieces = (int)(n16 >> 6);
#pragma unroll
    for (int i = 0; i < TW_SLOT / 8192; ++i) {
      const int pc = wv + 8 * i;
      if (pc < npieces)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(blob8 + off + pc * 64 + lane),
                                         (__attribute__((address_space(3))) void*)(ring + slot_off + pc * 1024), 16, 0, 0);
    }
  };
  // Timing-only diagnostic build (-DTW_STAMPS, tools/tw_stamps.py): per-wave s_memtime stamps into the buffer passed as `latent` (96 slots
  // per wave: 0 = kernel entry, 1 = prologue done, 2 + 2u = unit u computed, 3 + 2u = past unit u's barrier), no latent then.
#ifdef TW_STAMPS
  int tw_u = 0;
#define TWS(i)                                                                                                                \
  do {                                                                                                                        \
    if (a.latent && lane == 0)                                                                                                \
      reinterpret_cast<unsigned long long*>(a.latent)[((size_t)blockIdx.x * 8 + wv) * 96 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define TW_STAMP_MID() TWS(2 + 2 * tw_u)
#define TW_STAMP_END() do { TWS(3 + 2 * tw_u); ++tw_u; } while (0)
#else
#define TWS(i) do { } while (0)
#define TW_STAMP_MID() do { } while (0)
#define TW_STAMP_END() do { } while (0)
#endif
  TWS(0);
  issue(0, 0);

  // the tail's biases -> LDS, once
  {
    f32x4* const bl = reinterpret_cast<f32x4*>(lds + TW_O_BIAS);
#pragma unroll
    for (int s = ST_E3; s < NUM_STAGES; ++s) {
      const f32x4* src = blob4 + H.b_off(s);
      for (int i = tid; i < tw_bias_n16(s); i += 512) bl[tw_bias_off(s) / 16 + i] = src[i];
    }
  }
  // this block's h3 (layer 2's output) as operand fragments: 16 K steps x (hi, lo)
  h8 x3[16][2];
  {
    const u4* src = a.h3 + (size_t)fbg * (16 * 2 * 64) + lane;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
      for (int p = 0; p < 2; ++p) x3[ks][p] = __builtin_bit_cast(h8, src[(ks * 2 + p) * 64]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  TWS(1);

  int cur = 0;   // byte offset of the buffer holding the current unit
  LdsC* const bias0 = (LdsC*)(lds + TW_O_BIAS) + 64 * h;
  LdsC* const wl0 = (LdsC*)(lds) + lane * 16;
  LdsW* const latw = (LdsW*)(lds + TW_O_LAT) + wv * TW_LAT_WAVE + f * 8;
  // THE TWO WAVES OF A SIMD RUN HALF A UNIT APART -- by where their barrier stands, in ONE instruction stream.  A unit is M (its MFMAs)
  // then E (its last epilogue: scale, activation, hi/lo split, exchange -- vector ALU only).  Waves 0-3 pass the unit's barrier behind E,
  // waves 4-7 ("late") behind M: between two barriers the early wave of a SIMD runs M_k E_k, its late partner E_(k-1) M_k -- one's epilogue
  // under the other's MFMAs, where with one common barrier both compute together and then both sit in their epilogues (stamped: the
  // epilogues ADD to a unit's MFMA time, 8.0 k + 2.1 k cycles for a 6.1 k-cycle H1 unit).  Every wave executes one barrier per unit, so
  // the counts match; unit k is read by both kinds between barriers k-1 and k, so the two-buffer DMA schedule is unchanged (unit k + 1 is
  // requested right behind barrier k - 1 by every wave and waited for -- vmcnt(0) -- in front of barrier k).  The conditional wraps nothing
  // but the barrier: no register is live on one side only (two code INSTANCES of the body made hipcc spill 99 registers).
  // This is the gfx9 S_BARRIER's rule (it counts the workgroup's waves that have arrived, wherever each one's barrier instruction stands),
  // not the HIP programming model's (every thread at the same __syncthreads()): the file refuses to build for anything but gfx950 (above),
  // the bit-identity tests are the gate, and -DTW_NO_SKEW builds the model-conforming form (every wave's barrier behind E; same bits).
#ifdef TW_NO_SKEW
  const bool late = false;
#else
  const bool late = wv >= 4;
#endif
#define TW_BEGIN(idx) issue((idx) + 1, cur ^ TW_SLOT); LdsC* const wl = wl0 + cur
#define TW_MID()                                      \
  TW_STAMP_MID();                                     \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    \
  if (late) __syncthreads()
#define TW_END()                                      \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    \
  if (!late) __syncthreads();                         \
  TW_STAMP_END();                                     \
  cur ^= TW_SLOT

  f32x16 acc[2];
  auto emit = [&](auto act, const f32x16& ac, float inv, h8 (&d0)[2], h8 (&d1)[2]) {
    h8 fr[2][2];
    frags_from_acc<decltype(act)::value>(ac, inv, fr);
    d0[0] = fr[0][0]; d0[1] = fr[0][1]; d1[0] = fr[1][0]; d1[1] = fr[1][1];
  };
  constexpr std::integral_constant<int, ACT_RELU> RELU{};
  constexpr std::integral_constant<int, ACT_TANH> TANH{};
  const float inv3 = H.inv_scale[ST_E3], inv4 = H.inv_scale[ST_E4], inv5 = H.inv_scale[ST_E5];
  constexpr int JB3 = (int)(stage_job_w16_const(ST_E3) * 16), JB4 = (int)(stage_job_w16_const(ST_E4) * 16);

  // ---- E3: 256 -> 128, ReLU (two jobs per unit)
  h8 x4[8][2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    TW_BEGIN(u);
    LdsC* const b = bias0 + tw_bias_off(ST_E3) + (2 * u) * 128;
    tw_job2p<16>(acc, wl, wl + JB3, 2048, b, b + 128, x3);
    TW_MID();
    emit(RELU, acc[0], inv3, x4[4 * u], x4[4 * u + 1]);
    emit(RELU, acc[1], inv3, x4[4 * u + 2], x4[4 * u + 3]);
    TW_END();
  }
  // ---- E4: 128 -> 64, Tanh (both jobs)
  h8 x5[4][2];
  {
    TW_BEGIN(2);
    LdsC* const b = bias0 + tw_bias_off(ST_E4);
    tw_job2<8>(acc, wl, wl + JB4, 2048, b, b + 128, x4);
    TW_MID();
    emit(TANH, acc[0], inv4, x5[0], x5[1]);
    emit(TANH, acc[1], inv4, x5[2], x5[3]);
    TW_END();
  }
  // ---- E5: 64 -> 9, latent n = 3g + c on row 16g + c (one job of two blocks); K step g of its output is head g's input
  {
    TW_BEGIN(3);
    LdsC* const b = bias0 + tw_bias_off(ST_E5);
    tw_job2<4>(acc, wl, wl + 2048, 4096, b, b + 128, x5);
    TW_MID();
#ifndef TW_STAMPS
    if (a.latent && row_live) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int rowi = 32 * nb + (q & 3) + 8 * (q >> 2) + 4 * h, g = rowi >> 4, cc = rowi & 15;
          if (g < 3 && cc < 3) a.latent[row * NLML_LATENT + 3 * g + cc] = acc[nb][q] * inv5;
        }
    }
#endif
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      h8 fr[2][2];
      frags_from_acc<ACT_NONE>(acc[nb], inv5, fr);
      // Of K step 2 nb + p only k = 0..2 are non-zero (the latent's three values of head g = 2 nb + p; zero weights and zero bias make the
      // other rows exact zeros): lanes 0..31 keep k = 0..3 (8 bytes per piece), the rest of the fragment is rebuilt as zeros.
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc)
          if (2 * nb + p < 3 && h == 0) {
            const u4 t = __builtin_bit_cast(u4, fr[p][pc]);
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<__attribute__((address_space(3))) u2*>(latw + ((2 * nb + p) * 2 + pc) * 256) = u2{t[0], t[1]};
          }
    }
    TW_END();
  }
  // ---- the three heads, one after the other
#pragma unroll 1
  for (int g = 0; g < 3; ++g) {
    const int u0 = 4 + TW_HEAD_UNITS * g;
    const float i0 = H.inv_scale[ST_H0], i1 = H.inv_scale[ST_H1], i2 = H.inv_scale[ST_H2], i3 = H.inv_scale[ST_H3];
    h8 xa[8][2];
    {  // H0_g: 3 -> 128 (K padded to 16), ReLU; four jobs in one unit
      TW_BEGIN(u0);
      h8 xin[1][2];
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        const u2 t = *reinterpret_cast<const __attribute__((address_space(3))) u2*>((LdsC*)latw + (g * 2 + pc) * 256);
        xin[0][pc] = __builtin_bit_cast(h8, u4{h == 0 ? t[0] : 0u, h == 0 ? t[1] : 0u, 0u, 0u});
      }
      LdsC* const b = bias0 + tw_bias_off(ST_H0) + (4 * g) * 128;
      constexpr int JB = (int)(stage_job_w16_const(ST_H0) * 16);
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        tw_job2<1>(acc, wl + (2 * jj) * JB, wl + (2 * jj + 1) * JB, 2048, b + (2 * jj) * 128, b + (2 * jj + 1) * 128, xin);
        if (jj == 1) { TW_MID(); }
        emit(RELU, acc[0], i0, xa[4 * jj], xa[4 * jj + 1]);
        emit(RELU, acc[1], i0, xa[4 * jj + 2], xa[4 * jj + 3]);
      }
      TW_END();
    }
    h8 xb[16][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // H1_g: 128 -> 256, ReLU; one job of two blocks per unit, the blocks as the two chains
      TW_BEGIN(u0 + 1 + j);
      LdsC* const b = bias0 + tw_bias_off(ST_H1) + (4 * g + j) * 256;
      tw_job2p<8>(acc, wl, wl + 2048, 4096, b, b + 128, xa);
      TW_MID();
      emit(RELU, acc[0], i1, xb[4 * j], xb[4 * j + 1]);
      emit(RELU, acc[1], i1, xb[4 * j + 2], xb[4 * j + 3]);
      TW_END();
    }
    h8 xc[8][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {  // H2_g: 256 -> 128, ReLU; two jobs per unit
      TW_BEGIN(u0 + 5 + u);
      constexpr int JB = (int)(stage_job_w16_const(ST_H2) * 16);
      LdsC* const b = bias0 + tw_bias_off(ST_H2) + (4 * g + 2 * u) * 128;
      tw_job2p<16>(acc, wl, wl + JB, 2048, b, b + 128, xb);
      TW_MID();
      emit(RELU, acc[0], i2, xc[4 * u], xc[4 * u + 1]);
      emit(RELU, acc[1], i2, xc[4 * u + 2], xc[4 * u + 3]);
      TW_END();
    }
    h8 xd[4][2];
    {  // H3_g: 128 -> 64, ReLU; both jobs
      TW_BEGIN(u0 + 7);
      LdsC* const b = bias0 + tw_bias_off(ST_H3) + (2 * g) * 128;
      tw_job2p<8>(acc, wl, wl + (int)(stage_job_w16_const(ST_H3) * 16), 2048, b, b + 128, xc);
      TW_MID();
      emit(RELU, acc[0], i3, xd[0], xd[1]);
      emit(RELU, acc[1], i3, xd[2], xd[3]);
      TW_END();
    }
    {  // H4_g: 64 -> 1 (one chain); the neuron is accumulator row 0 = register 0 of lanes 0..31
      TW_BEGIN(u0 + 8);
      LdsC* const b = bias0 + tw_bias_off(ST_H4) + g * 128;
      tw_job2<4, 1>(acc, wl, wl, 2048, b, b, xd);
      TW_MID();
      if (h == 0 && row_live) a.out[row * 3 + g] = acc[0][0] * H.inv_scale[ST_H4];   // non-finite = beyond f16's range: the re-evaluation launch's flag
      TW_END();
    }
  }
#undef TW_BEGIN
#undef TW_MID
#undef TW_END
}

}  // namespace hx

int launch_tail_ws(const void* blob, const void* h3, int64_t nfb, int64_t B, float* out, float* latent, void* stream) {
  if (B == 0) return 0;
  hx::TwArgs a;
  a.blob = blob; a.h3 = reinterpret_cast<const hx::u4*>(h3); a.out = out; a.latent = latent; a.B = B; a.nfb = nfb;
  hipLaunchKernelGGL(hx::tail_ws_kernel, dim3((unsigned)((nfb + 7) / 8)), dim3(512), 0, reinterpret_cast<hipStream_t>(stream), a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
  return 0;
}

}  // namespace nlml
