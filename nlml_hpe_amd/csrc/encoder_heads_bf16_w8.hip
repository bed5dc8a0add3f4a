// encoder_heads_bf16_w8.hip -- K2 in THROUGHPUT mode (NLML_MODE_BF16) with EIGHT waves per workgroup (two per SIMD): bf16 storage of
// weights and activations, v_mfma_f32_32x32x16_bf16 with f32 accumulation.  BASELINE config 2 names this dtype; it is NOT a parity path
// (bf16 operands put the pose ~0.1 degree from the reference) -- the measured error is reported, never claimed as parity.
//
// Same network, blob, LDS images and -- per accumulator -- the same K-ascending MFMA sequence as round 1's four-wave kernel
// (encoder_heads_bf16.hip: tested bit-identical to this one on seven shapes and the fused-landmarks path, 0.379 -> 0.360 ms per 65,536 faces on
// one box, then removed; reference: NLML_HPE_Model_Builder.py:33-53,76-92,115-126).
// What changed is who computes what and how the loads are issued -- the round-4 structure of the strict-fast kernel
// (encoder_heads_f16x2_w8.hip) applied to the mode whose matrix time per weight byte is a third of that kernel's:
//   * eight waves: a job of the trunk is shared by a PAIR of waves, each taking half of its neuron blocks with both face blocks (layer 0:
//     4 of 8 blocks = 128 accumulator registers; layer 1: 2 of 4; layer 2: 1 of 2), E3 by (neuron block, face block), and the heads run
//     BOTH 32-face blocks at once (wave group w >> 2 = face block): half the sequential stages of the four-wave form;
//   * the kernel is bound by the L2 -> CU weight stream (4.7 MB per 64-face tile: one 16-byte fragment feeds ONE 32-cycle MFMA per face
//     block): eight waves keep twice the loads in flight (profiles/r03_l1_stream_probe.txt: 35-41 B/clk with four waves, 46-51 with eight);
//   * one load per MFMA slot: every MFMA is followed by its share of the step's fetches (LDS reads of the next step's operands first, then
//     the weights D steps ahead) instead of a block of loads in front of a block of MFMAs.
#include <hip/hip_runtime.h>

#include "../../include/nlml_hpe.h"
#include "abi_internal.h"
#include "encoder_heads_bf16_dev.h"
#include "layout.h"

namespace nlml {
namespace bf {
namespace w8 {

// ------------------------------------------------------------------------------------------
// Layer 0 (single pass) for 512 threads: x[64,F] f32 -> (optional IPD normalisation in f64, FeatureExtractor.py:30-66) -> bf16 -> three
// rotating LDS slabs of 64 columns (thread = row tid/8, columns 8*(tid%8) .. +7); this wave computes 4 of job jw's 8 neuron blocks for
// both face blocks.
template <bool VEC4, bool NORM>
__device__ __forceinline__ void stage_e0(const Ctx& c, const Args& a, int64_t row0, int tid, int jw, int nbh, f32x16 (&acc)[4][2]) {
  constexpr int NB = 4, NFB = 2, WSTEP = 8 * 64;
  const int F = a.F;
  const int nslab = (int)c.hdr->k8_e0 / XS_STEPS;   // even (pack.cpp)
  constexpr int SLAB_BYTES = 64 * S_XS * 2;

  const int srow = tid >> 3, scol = (tid & 7) * 8;
  int64_t r = row0 + srow;
  const bool live = r < a.B;
  r = live ? r : a.B - 1;
  const float* p = a.x + r * a.ldx;
  double ipd = 1.0, rcp = 1.0, ra = 0.0, rb = 0.0, rc = 0.0;
  if (NORM) {
    const double dx = (double)p[99] - (double)p[789], dy = (double)p[100] - (double)p[790], dz = (double)p[101] - (double)p[791];
    ipd = sqrt(fma(dz, dz, fma(dy, dy, dx * dx)));
    if (ipd == 0.0) ipd = 1e-6;
    rcp = 1.0 / ipd;
    const double x0 = (double)p[3], y0 = (double)p[4], z0 = (double)p[5];
    const int ph = scol % 3;   // coordinate of this thread's first column; a slab later the phase is + 64 % 3 = + 1
    ra = ph == 0 ? x0 : (ph == 1 ? y0 : z0);
    rb = ph == 0 ? y0 : (ph == 1 ? z0 : x0);
    rc = ph == 0 ? z0 : (ph == 1 ? x0 : y0);
  }
  unsigned nzbits = 0u;

  // ONE L2 request per 128-byte line of x (the strict-fast kernel's scheme, encoder_heads_f16x2_w8.hip): a row is 16-byte but not
  // 128-byte aligned, so the 256 bytes it contributes to a slab touch three lines and the one shared with the next slab used to be
  // requested twice, a slab apart.  The 16-byte units that lie in a slab's FIRST line (`adv[i]`) are loaded one slab ahead -- one load
  // instruction then asks for whole lines only -- and wait one slab in `carry`.  Same values into the same LDS bytes.
#ifdef BF8_NO_XLINE
  const bool adv[2] = {false, false};
#else
  const int xphase = (int)((reinterpret_cast<uintptr_t>(p) >> 4) & 7);   // the row's first 16-byte unit within its line
  const bool adv[2] = {VEC4 && xphase != 0 && xphase + 2 * (tid & 7) < 8, VEC4 && xphase != 0 && xphase + 2 * (tid & 7) + 1 < 8};
#endif
  f32x4 set[2];   // ONE staging register set (8 floats per thread), refilled as soon as it has been written to LDS
  f32x4 carry[2];
  auto gload_at = [&](int s0, f32x4 (&dst)[2], bool ahead) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int s = s0 + ((ahead && adv[i]) ? 1 : 0);
      s = s < nslab ? s : nslab - 1;
      const int k = s * XS_COLS + scol + 4 * i;
      if (VEC4) {
        const int kc = k < F ? k : (NORM ? k - 12 * ((k - F + 15) / 12) : F - 4);   // phase-preserving clamp (zero weights there)
        dst[i] = *reinterpret_cast<const f32x4*>(p + kc);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[i][e] = p[k + e < F ? k + e : F - 1];
      }
    }
  };
  auto gload = [&](int s) { gload_at(s, set, true); };
  // the staging of a slab in pieces (one per MFMA slot): 0..7 normalise element q, 8 convert + store
  auto lpiece = [&](int piece, int buf_off, bool real_slab) {
    if (piece == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(set[i]));
#ifndef BF8_NO_XLINE
      if (VEC4) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {   // a unit that is loaded a slab ahead stages what it loaded a slab ago
          const f32x4 t = set[i];
#pragma unroll
          for (int e = 0; e < 4; ++e) set[i][e] = adv[i] ? carry[i][e] : t[e];
          carry[i] = t;
        }
      }
#endif
    }
    if (piece < 8) {
      if (NORM) {
        const int i = piece >> 2, e = piece & 3, t = piece % 3;
        const double rr = t == 0 ? ra : (t == 1 ? rb : rc);
        set[i][e] = (float)div_ipd((double)set[i][e] - rr, ipd, rcp);
      }
    } else {
      if (NORM) {
        // keep the f32 value: otherwise (bf16)(f32)double may be folded into one software f64 -> bf16 conversion
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(set[i]));
        const double t0 = ra; ra = rb; rb = rc; rc = t0;   // next slab: columns + 64 => phase + 1
      }
      const unsigned m = real_slab ? 0x7fffffffu : 0u;
      bf16x8 v;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          nzbits |= __float_as_uint(set[i][e]) & m;
          v[4 * i + e] = (__bf16)set[i][e];
        }
      *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(c.lds + O_XS + buf_off) + srow * S_XS + scol) = v;
    }
  };
  auto lwrite = [&](int buf_off, bool real_slab) {
#pragma unroll
    for (int pc = 0; pc < 9; ++pc) lpiece(pc, buf_off, real_slab);
  };

  const int job = jw;
  const bf16x8* w = c.blob8 + c.hdr->w_off[ST_E0] + (size_t)job * c.hdr->job_w16[ST_E0] + (size_t)(4 * nbh) * 64 + c.lane;
#ifndef BF8_D0
#define BF8_D0 3
#endif
  constexpr int R0 = 4, D0 = BF8_D0;   // weight ring: K step ks in slot ks % 4 (a slab holds 4 steps), loaded D0 steps ahead
  static_assert(XS_STEPS == R0, "slab steps == ring slots");
  bf16x8 wr[R0][NB];
  // prologue: slabs 0 and 1 in LDS, slab 2 in the registers; the ring and the bias are requested LAST (so that at the loop's entry no
  // load is younger than the loop's own steady state: hipcc merges the two entries' pending-load states conservatively)
#ifndef BF8_NO_XLINE
  if (VEC4) gload_at(0, carry, false);   // slab 0 for the units that run a slab ahead
#endif
  gload(0);
  lwrite(0, true);
  gload(1);
  lwrite(SLAB_BYTES, true);
  gload(2);
#pragma unroll
  for (int d = 0; d < D0; ++d)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) wr[d][nb] = w[(size_t)d * WSTEP + nb * 64];
  load_bias<NB, NFB>(acc, c.blob4 + c.hdr->b_off[ST_E0] + job * (8 * 8) + (4 * nbh) * 8, c.h);
  __syncthreads();

  const int lane_off = (c.f * S_XS + 8 * c.h) * 2;   // bytes
  bf16x8 xr[2][NFB];
#pragma unroll
  for (int fb = 0; fb < NFB; ++fb) xr[0][fb] = *reinterpret_cast<const bf16x8*>(c.lds + O_XS + lane_off + fb * (32 * S_XS * 2));

  int o0 = 0, o1 = SLAB_BYTES, o2 = 2 * SLAB_BYTES;   // buffers of slabs s, s+1, s+2
  for (int s = 0; s < nslab; ++s) {
    const char* xrow = c.lds + O_XS + o0 + lane_off;
    const char* xnext = c.lds + O_XS + o1 + lane_off;
    const bool real = s + 2 < nslab;
#pragma unroll
    for (int kk = 0; kk < XS_STEPS; ++kk) {
      const int ks = s * XS_STEPS + kk;
      step8<NB, NFB>(acc, wr[kk], xr[kk & 1], wr[(kk + D0) % R0], w + (size_t)(ks + D0) * WSTEP, true,
                     [&](int fb) {
                       xr[(kk + 1) & 1][fb] = (kk + 1 < XS_STEPS) ? *reinterpret_cast<const bf16x8*>(xrow + fb * (32 * S_XS * 2) + 32 * (kk + 1))
                                                                  : *reinterpret_cast<const bf16x8*>(xnext + fb * (32 * S_XS * 2));
                     },
                     [&](int m) {   // slab s+2 is staged during steps 0 and 1 (pieces 0..8 over 16 slots), its registers refilled at once
                       const int slot = 8 * kk + m;
                       if (slot < 9) lpiece(slot, o2, real);
                       if (slot == 9) gload(s + 3);
                     });
    }
    __syncthreads();
    const int t0 = o0;   // rotate: (o0, o1, o2) <- (o1, o2, o0)
    o0 = o1; o1 = o2; o2 = t0;
  }
  if (a.valid) {  // all-zero feature row == "no face" (FeatureExtractor.py:105-106); 8 lanes share a row
    const unsigned long long m = __ballot(nzbits != 0u);
    if ((tid & 7) == 0 && live) a.valid[row0 + srow] = ((m >> (c.lane & 56)) & 0xFFull) ? 1 : 0;
  }
}

// ------------------------------------------------------------------------------------------
template <bool VEC4, bool NORM>
__global__ __launch_bounds__(512) void encoder_heads_bf16_w8_kernel(Args a) {
  __shared__ __attribute__((aligned(16))) char lds[163840];

  const int tid = threadIdx.x;
  Ctx c;
  c.blob8 = reinterpret_cast<const bf16x8*>(a.blob);
  c.blob4 = reinterpret_cast<const f32x4*>(a.blob);
  c.hdr = reinterpret_cast<const Header*>(a.blob);
  c.lds = lds;
  c.lane = tid & 63;
  c.f = c.lane & 31;
  c.h = c.lane >> 5;
  c.wv = __builtin_amdgcn_readfirstlane(tid >> 6);       // 0..7
  const int wv = c.wv, jw = wv >> 1, nbh = wv & 1;        // the trunk job this wave shares with its partner, and its half of the job's blocks
  const int64_t row0 = (int64_t)blockIdx.x * TILE_FACES;

  {  // E0: F -> 1024, ReLU (one pass; wave wv owns neurons 128*wv .. +127)
    f32x16 acc[4][2];
    stage_e0<VEC4, NORM>(c, a, row0, tid, jw, nbh, acc);
    store_lds<4, 2, ACT_RELU>(acc, img(c, O_H1) + c.f * S_H1 + 128 * wv + 4 * c.h, 32 * S_H1);
  }
  __syncthreads();
  {  // E1: 1024 -> 512, ReLU; job jw's blocks 2*nbh, 2*nbh+1; h2 overwrites h1 => barrier between the K loop and the store
    f32x16 acc[2][2];
    load_bias<2, 2>(acc, c.blob4 + c.hdr->b_off[ST_E1] + jw * (4 * 8) + (2 * nbh) * 8, c.h);
    const bf16x8* w = c.blob8 + c.hdr->w_off[ST_E1] + (size_t)jw * c.hdr->job_w16[ST_E1] + (size_t)(2 * nbh) * 64 + c.lane;
    kloop8<2, 2, kStages[ST_E1].k8, 4 * 64>(acc, w, img(c, O_H1) + c.f * S_H1 + 8 * c.h, 32 * S_H1);
    __syncthreads();
    store_lds<2, 2, ACT_RELU>(acc, img(c, O_H2) + c.f * S_H2 + 64 * wv + 4 * c.h, 32 * S_H2);
  }
  __syncthreads();
  {  // E2: 512 -> 256, ReLU; job jw's block nbh
    f32x16 acc[1][2];
    load_bias<1, 2>(acc, c.blob4 + c.hdr->b_off[ST_E2] + jw * (2 * 8) + nbh * 8, c.h);
    const bf16x8* w = c.blob8 + c.hdr->w_off[ST_E2] + (size_t)jw * c.hdr->job_w16[ST_E2] + (size_t)nbh * 64 + c.lane;
    kloop8<1, 2, kStages[ST_E2].k8, 2 * 64>(acc, w, img(c, O_H2) + c.f * S_H2 + 8 * c.h, 32 * S_H2);
    store_lds<1, 2, ACT_RELU>(acc, img(c, O_H3) + c.f * S_H3 + 32 * wv + 4 * c.h, 32 * S_H3);
  }
  __syncthreads();
  tail_stages_bf16(c, a, row0);
}

}  // namespace w8
}  // namespace bf

int launch_encoder_heads_bf16(const float* x, int64_t ldx, const float* raw, int normalize, int64_t B, int F,
                              const void* blob, float* out, float* latent, uint8_t* valid, void* stream) {
  if (B == 0) return 0;
  bf::w8::Args a;
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
    if (vec4) hipLaunchKernelGGL((bf::w8::encoder_heads_bf16_w8_kernel<true, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((bf::w8::encoder_heads_bf16_w8_kernel<false, true>), grid, block, 0, st, a);
  } else {
    if (vec4) hipLaunchKernelGGL((bf::w8::encoder_heads_bf16_w8_kernel<true, false>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((bf::w8::encoder_heads_bf16_w8_kernel<false, false>), grid, block, 0, st, a);
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

}  // namespace nlml
