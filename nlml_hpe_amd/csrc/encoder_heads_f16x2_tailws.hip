// encoder_heads_f16x2_tailws.hip -- the strict-fast mode's TAIL (layers E3, E4, E5 and the three heads: NLML_HPE_Model_Builder.py:45-53,
// 76-92,115-126) as a launch of its own with the roles of weights and faces exchanged.
//
// In the fused kernel (encoder_heads_f16x2_w8.hip) a 64-face tile stays on its CU through the whole network, so the tail's 1.06 MB of
// split-f16 weights stream L2 -> CU once per 64 faces: thirteen short, barrier-separated stages on four of the CU's eight waves, bound by that
// stream and by their own start-up latencies (63.5 k of a tile's 361 k cycles at 43 % of the matrix pipe).  Here a WAVE
// owns one 32-face block for the whole tail and keeps its activations in REGISTERS: faces are MFMA columns and a lane's accumulators
// belong to one face, so a layer's output becomes the next layer's B operand by the scale / activation / hi-lo split of store_lds and one
// half-wave exchange (frags_from_acc) -- no LDS image, no barrier between layers.  The WEIGHTS go through LDS: a workgroup of eight waves
// (256 faces) walks the blob's tail job by job -- 39 units of 8 .. 32 KB, each one contiguous in the blob in exactly the fragment order
// the MFMAs want -- copying unit u+1 global -> LDS by LDS-DMA (global_load_lds_dwordx4, no staging registers) while all eight waves
// compute unit u from the other buffer.  The tail's weight stream is paid once per 256 faces instead of once per 64, every wave of the CU
// has matrix work all the time, and one barrier per unit is all the synchronisation there is.
//
// Measured (round 5, B = 65,536, same box, alternating): trunk launch + this launch + re-evaluation launch 0.805 ms against the fused
// kernel's 0.815 ms (+1.2 %); rocprofv3: trunk 739 us, this kernel 107 -> 95 us, re-evaluation 5 us.  The trunk alone costs 0.90 of the
// fused kernel -- exactly its share of the L2 -> CU bytes (9.3 of 10.3 MB per tile): the fused kernel's time IS its L2 traffic, so taking
// the tail out buys the tail's bytes and no more.  This kernel's own cycles (-DTW_STAMPS, tools/tw_stamps.py): 200 k
// per 256 faces = prologue 24 k (h3: 64 MB chip-wide at ~5.7 TB/s) + compute 138 k + barrier skew 38 k, against 104 k of pure MFMA time.
// Variants measured and dropped: two accumulator chains per wave over 64-KB units (compiler-scheduled 206 k; pinned 207 k; pinned with
// the previous job's epilogue in the MFMA shadows 255 k: the slices' dependent vector-ALU chains are longer than an MFMA slot); the two
// waves of a SIMD half a job apart as two code instances (hipcc keeps 99 registers of one instance alive across the other: spills); the
// same half-unit skew in ONE instruction stream by moving only the barrier (waves 4-7 pass a unit's barrier behind its MFMAs, waves 0-3
// behind its epilogue; two chains, 64-KB units; bit-identical) 193-194 k, with s_setprio(2) around the MFMA phases the same.  Every form
// lands at 193-207 k cycles -- also with every weight fragment read two K steps ahead by ds_read_b128 in inline asm and hand-counted
// `s_waitcnt lgkmcnt(3)` in place of hipcc's `lgkmcnt(0)` in front of each MFMA (78 full drains per head), the encoder part pinned the same
// way: 202 k.  What stays is the pair of waves on a SIMD: the older one is served first and finishes a 48-MFMA job in 2.5-3.0 k cycles
// (52-62 per MFMA: its dependent chain), the younger in 4.2-4.6 k, and the unit lasts as long as the younger.
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
constexpr int TW_SLOT = 32768;
constexpr int tw_bias_n16(int s) { return kStages[s].jobs * kStages[s].nb * 8; }   // 16-byte units
constexpr int tw_bias_off(int s) {                                                  // bytes from TW_O_BIAS
  int o = 0;
  for (int t = ST_E3; t < s; ++t) o += tw_bias_n16(t) * 16;
  return o;
}
constexpr int TW_O_BIAS = 2 * TW_SLOT;
constexpr int TW_O_LAT = TW_O_BIAS + ((tw_bias_off(NUM_STAGES) + 1023) / 1024) * 1024;
constexpr int TW_LAT_WAVE = 3 * 2 * 1024;
constexpr int TW_LDS = TW_O_LAT + 8 * TW_LAT_WAVE;
static_assert(TW_LDS <= 163840, "LDS map");

// The units, in the order they are consumed: E3 job 0..3 | E4 (both jobs) | E5 | per head g: H0 (four jobs), H1 job 0..3, H2 job 0..3,
// H3 (both jobs), H4.  A unit = `n16` consecutive 16-byte fragments-per-lane x 64 lanes... i.e. n16 16-byte units from `off` in the blob.
constexpr int TW_HEAD_UNITS = 11, TW_UNITS = 6 + 3 * TW_HEAD_UNITS;
__device__ __forceinline__ void tw_unit(const HdrRegs& H, int idx, uint32_t& off, uint32_t& n16) {
  constexpr uint32_t J3 = stage_job_w16_const(ST_E3), J4 = stage_job_w16_const(ST_E4), J5 = stage_job_w16_const(ST_E5);
  constexpr uint32_t K0 = stage_job_w16_const(ST_H0), K1 = stage_job_w16_const(ST_H1), K2 = stage_job_w16_const(ST_H2);
  constexpr uint32_t K3 = stage_job_w16_const(ST_H3), K4 = stage_job_w16_const(ST_H4);
  static_assert(J3 * 16 == TW_SLOT && 2 * J4 * 16 == TW_SLOT && J5 * 16 <= TW_SLOT && 4 * K0 * 16 <= TW_SLOT && K1 * 16 == TW_SLOT &&
                    K2 * 16 == TW_SLOT && 2 * K3 * 16 == TW_SLOT && K4 * 16 <= TW_SLOT,
                "a unit fits a buffer");
  if (idx < 4) { n16 = J3; off = H.w_off(ST_E3) + idx * J3; }
  else if (idx == 4) { n16 = 2 * J4; off = H.w_off(ST_E4); }
  else if (idx == 5) { n16 = J5; off = H.w_off(ST_E5); }
  else {
    const int k = idx - 6, g = k / TW_HEAD_UNITS, r = k - TW_HEAD_UNITS * g;
    if (r == 0) { n16 = 4 * K0; off = H.w_off(ST_H0) + 4 * g * K0; }
    else if (r <= 4) { n16 = K1; off = H.w_off(ST_H1) + (4 * g + r - 1) * K1; }
    else if (r <= 8) { n16 = K2; off = H.w_off(ST_H2) + (4 * g + r - 5) * K2; }
    else if (r == 9) { n16 = 2 * K3; off = H.w_off(ST_H3) + 2 * g * K3; }
    else { n16 = K4; off = H.w_off(ST_H4) + g * K4; }
  }
}

// One job for this wave's face block: NB neuron blocks, K16 steps, weights from the LDS copy of the job's stream (`w`: the stream's
// first fragment + this lane's 16 bytes), bias from the LDS table (`bias`: the job's first block + 64 * h), input operands in registers.
template <int NB, int K16>
__device__ __forceinline__ void tw_job(f32x16 (&acc)[2], LdsC* w, LdsC* bias, const h8 (&in)[K16][2]) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(bias + nb * 128 + q * 16);
      acc[nb][4 * q + 0] = v[0];
      acc[nb][4 * q + 1] = v[1];
      acc[nb][4 * q + 2] = v[2];
      acc[nb][4 * q + 3] = v[3];
    }
#pragma unroll
  for (int ks = 0; ks < K16; ++ks) {
    h8 wf[NB][2];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        wf[nb][p] = *reinterpret_cast<const __attribute__((address_space(3))) h8*>(w + ((ks * NB + nb) * 2 + p) * 1024);
#pragma unroll
    for (int t = 0; t < 3; ++t) {   // (w_lo, x_hi), (w_hi, x_lo), (w_hi, x_hi): mma_step's order
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[nb][t == 0 ? 1 : 0], in[ks][t == 1 ? 1 : 0], acc[nb], 0, 0, 0);
    }
  }
}


// Slice s (0..7) of one block's epilogue = frags_from_acc cut into eight pieces: values 8p + 2k, 8p + 2k + 1 (p = s >> 2, k = s & 3) -> scale,
// activation, hi/lo split; the fourth piece of a half also does the half-wave exchange and hands the two fragments of K step p over.
template <int ACT>
__device__ __forceinline__ void epi_slice(const f32x16& ac, float inv, unsigned (&hi)[4], unsigned (&lo)[4], h8 (&d0)[2], h8 (&d1)[2], int s) {
  const int p = s >> 2, k = s & 3;
  split2(activate<ACT>(ac[8 * p + 2 * k] * inv), activate<ACT>(ac[8 * p + 2 * k + 1] * inv), hi[k], lo[k]);
  if (k == 3) {
    swap_halves(hi[0], hi[2]);
    swap_halves(hi[1], hi[3]);
    swap_halves(lo[0], lo[2]);
    swap_halves(lo[1], lo[3]);
    h8 (&d)[2] = p ? d1 : d0;
    d[0] = __builtin_bit_cast(h8, u4{hi[0], hi[1], hi[2], hi[3]});
    d[1] = __builtin_bit_cast(h8, u4{lo[0], lo[1], lo[2], lo[3]});
  }
}

// One ONE-BLOCK job with the previous job's epilogue in its MFMA shadows: every MFMA is followed by its share of the next K step's two
// operand reads and of the S pending slices (`pend(i)`), pinned by sched_barrier as in step_fine.  The slices are all placed within the
// first K16 - 2 steps: when the pending block is the last one of the previous layer, its two fragments are this job's last two K steps.
// WSTEP: bytes between the K steps of this block's stream (2 KB; 4 KB in a two-block job).
template <int K16, int WSTEP, int S, typename Pend>
__device__ __forceinline__ void tw_job1(f32x16& acc, LdsC* w, LdsC* bias, const h8 (&in)[K16][2], Pend pend) {
  typedef const __attribute__((address_space(3))) h8 LdsH8;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(bias + q * 16);
    acc[4 * q + 0] = v[0];
    acc[4 * q + 1] = v[1];
    acc[4 * q + 2] = v[2];
    acc[4 * q + 3] = v[3];
  }
  h8 wf[2][2];
  wf[0][0] = *reinterpret_cast<LdsH8*>(w);
  wf[0][1] = *reinterpret_cast<LdsH8*>(w + 1024);
  constexpr int MD = 3 * (K16 > 2 ? K16 - 2 : 1);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ks = 0; ks < K16; ++ks) {
#pragma unroll
    for (int m = 0; m < 3; ++m) {   // (w_lo, x_hi), (w_hi, x_lo), (w_hi, x_hi): mma_step's order
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks & 1][m == 0 ? 1 : 0], in[ks][m == 1 ? 1 : 0], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (m < 2 && ks + 1 < K16) wf[(ks + 1) & 1][m] = *reinterpret_cast<LdsH8*>(w + (ks + 1) * WSTEP + m * 1024);
      if constexpr (S > 0) {
#pragma unroll
        for (int i = 0; i < S; ++i)
          if ((i * MD) / S == ks * 3 + m) pend(i);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
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
    for (int i = 0; i < 4; ++i) {
      const int pc = wv + 8 * i;
      if (pc < npieces)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(blob8 + off + pc * 64 + lane),
                                         (__attribute__((address_space(3))) void*)(ring + slot_off + pc * 1024), 16, 0, 0);
    }
  };
#ifdef TW_STAMPS
  if (a.latent && lane == 0)
    reinterpret_cast<unsigned long long*>(a.latent)[((size_t)blockIdx.x * 8 + wv) * 96 + 0] = __builtin_amdgcn_s_memtime();
#endif
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
  int cur = 0;   // byte offset of the buffer holding the current unit
  LdsC* const bias0 = (LdsC*)(lds + TW_O_BIAS) + 64 * h;
  LdsC* const wl0 = (LdsC*)(lds) + lane * 16;
  LdsW* const latw = (LdsW*)(lds + TW_O_LAT) + wv * TW_LAT_WAVE + lane * 16;
  // Timing-only diagnostic build (-DTW_STAMPS, tools/tw_stamps.py): per-wave s_memtime stamps into the buffer passed as `latent` (64 slots
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
#define TW_BEGIN(idx) issue((idx) + 1, cur ^ TW_SLOT); LdsC* const wl = wl0 + cur
#define TW_END()                                      \
  TW_STAMP_MID();                                     \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    \
  __syncthreads();                                    \
  TW_STAMP_END();                                     \
  cur ^= TW_SLOT

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

  f32x16 acc[2];
  auto emit = [&](auto act, const f32x16& ac, float inv, h8 (&d0)[2], h8 (&d1)[2]) {
    h8 fr[2][2];
    frags_from_acc<decltype(act)::value>(ac, inv, fr);
    d0[0] = fr[0][0]; d0[1] = fr[0][1]; d1[0] = fr[1][0]; d1[1] = fr[1][1];
  };
  constexpr std::integral_constant<int, ACT_RELU> RELU{};
  constexpr std::integral_constant<int, ACT_TANH> TANH{};
  const float inv3 = H.inv_scale[ST_E3], inv4 = H.inv_scale[ST_E4], inv5 = H.inv_scale[ST_E5];

  // ---- E3: 256 -> 128, ReLU (one job per unit).  The encoder part runs job, then epilogue (with x3 resident there are no registers for
  // the pipelined form the heads use below).
  h8 x4[8][2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    TW_BEGIN(j);
    tw_job<1, 16>(acc, wl, bias0 + tw_bias_off(ST_E3) + j * 128, x3);
    emit(RELU, acc[0], inv3, x4[2 * j], x4[2 * j + 1]);
    TW_END();
  }
  // ---- E4: 128 -> 64, Tanh (both jobs in one unit)
  h8 x5[4][2];
  {
    TW_BEGIN(4);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      tw_job<1, 8>(acc, wl + j * (int)(stage_job_w16_const(ST_E4) * 16), bias0 + tw_bias_off(ST_E4) + j * 128, x4);
      emit(TANH, acc[0], inv4, x5[2 * j], x5[2 * j + 1]);
    }
    TW_END();
  }
  // ---- E5: 64 -> 9, latent n = 3g + c on row 16g + c (two blocks); its fragments (K step g = head g's input) go to this wave's LDS corner
  {
    TW_BEGIN(5);
    tw_job<2, 4>(acc, wl, bias0 + tw_bias_off(ST_E5), x5);
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
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc)
          if (2 * nb + p < 3) *reinterpret_cast<__attribute__((address_space(3))) h8*>(latw + ((2 * nb + p) * 2 + pc) * 1024) = fr[p][pc];
    }
    TW_END();
  }
  // ---- the three heads, one after the other: nineteen one-block jobs per head, SOFTWARE-PIPELINED -- a job's epilogue (scale, ReLU, hi/lo
  // split, half-wave exchange: ~75 vector-ALU instructions per block) runs in eight slices behind the NEXT job's MFMAs (tw_job1), on
  // accumulators of its own (aA / aB alternate).  In lock step -- all eight waves computing, then all eight in their epilogues -- a
  // 96-MFMA unit took 4.05 k cycles of which 3.07 k are MFMAs (stamped); the two waves of a SIMD pass every barrier together, so nothing
  // but the instruction order can put one's epilogue under the other's MFMAs.
#pragma unroll 1
  for (int g = 0; g < 3; ++g) {
    const int u0 = 6 + TW_HEAD_UNITS * g;
    const float i0 = H.inv_scale[ST_H0], i1 = H.inv_scale[ST_H1], i2 = H.inv_scale[ST_H2], i3 = H.inv_scale[ST_H3];
    f32x16 aA, aB;
    unsigned ehi[4], elo[4];
    auto none = [](int) {};
    auto pend = [&](const f32x16& ac, float inv, h8 (&d0)[2], h8 (&d1)[2]) {
      return [&, inv](int s) { epi_slice<ACT_RELU>(ac, inv, ehi, elo, d0, d1, s); };
    };
    h8 xa[8][2];
    {  // H0_g: 3 -> 128 (K padded to 16), ReLU; four jobs in one unit
      TW_BEGIN(u0);
      h8 xin[1][2];
#pragma unroll
      for (int pc = 0; pc < 2; ++pc)
        xin[0][pc] = *reinterpret_cast<const __attribute__((address_space(3))) h8*>((LdsC*)latw + (g * 2 + pc) * 1024);
      LdsC* const b = bias0 + tw_bias_off(ST_H0) + (4 * g) * 128;
      constexpr int JB = (int)(stage_job_w16_const(ST_H0) * 16);
      tw_job1<1, 2048, 0>(aA, wl, b, xin, none);
      tw_job1<1, 2048, 8>(aB, wl + JB, b + 128, xin, pend(aA, i0, xa[0], xa[1]));
      tw_job1<1, 2048, 8>(aA, wl + 2 * JB, b + 256, xin, pend(aB, i0, xa[2], xa[3]));
      tw_job1<1, 2048, 8>(aB, wl + 3 * JB, b + 384, xin, pend(aA, i0, xa[4], xa[5]));
      TW_END();
    }
    h8 xb[16][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // H1_g: 128 -> 256, ReLU; a job's two neuron blocks one after the other (same stream, K step = 4 KB)
      TW_BEGIN(u0 + 1 + j);
      LdsC* const b = bias0 + tw_bias_off(ST_H1) + (4 * g + j) * 256;
      const int jp = j > 0 ? j - 1 : 0;
      if (j == 0) tw_job1<8, 4096, 8>(aA, wl, b, xa, pend(aB, i0, xa[6], xa[7]));
      else tw_job1<8, 4096, 8>(aA, wl, b, xa, pend(aB, i1, xb[4 * jp + 2], xb[4 * jp + 3]));
      tw_job1<8, 4096, 8>(aB, wl + 2048, b + 128, xa, pend(aA, i1, xb[4 * j], xb[4 * j + 1]));
      TW_END();
    }
    h8 xc[8][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // H2_g: 256 -> 128, ReLU
      TW_BEGIN(u0 + 5 + j);
      LdsC* const b = bias0 + tw_bias_off(ST_H2) + (4 * g + j) * 128;
      const int jp = j > 0 ? j - 1 : 0;
      f32x16& cur_acc = (j & 1) ? aB : aA;
      f32x16& prev_acc = (j & 1) ? aA : aB;
      if (j == 0) tw_job1<16, 2048, 8>(cur_acc, wl, b, xb, pend(prev_acc, i1, xb[14], xb[15]));
      else tw_job1<16, 2048, 8>(cur_acc, wl, b, xb, pend(prev_acc, i2, xc[2 * jp], xc[2 * jp + 1]));
      TW_END();
    }
    h8 xd[4][2];
    {  // H3_g: 128 -> 64, ReLU; both jobs in one unit
      TW_BEGIN(u0 + 9);
      LdsC* const b = bias0 + tw_bias_off(ST_H3) + (2 * g) * 128;
      tw_job1<8, 2048, 8>(aA, wl, b, xc, pend(aB, i2, xc[6], xc[7]));
      tw_job1<8, 2048, 8>(aB, wl + (int)(stage_job_w16_const(ST_H3) * 16), b + 128, xc, pend(aA, i3, xd[0], xd[1]));
      TW_END();
    }
    {  // H4_g: 64 -> 1; the neuron is accumulator row 0 = register 0 of lanes 0..31
      TW_BEGIN(u0 + 10);
      tw_job1<4, 2048, 8>(aA, wl, bias0 + tw_bias_off(ST_H4) + g * 128, xd, pend(aB, i3, xd[2], xd[3]));
      if (h == 0 && row_live) a.out[row * 3 + g] = aA[0] * H.inv_scale[ST_H4];   // non-finite = beyond f16's range: the re-evaluation launch's flag
      TW_END();
    }
  }
#undef TW_BEGIN
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
