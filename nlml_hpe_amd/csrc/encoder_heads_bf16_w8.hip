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
#include "layout.h"

namespace nlml {
namespace bf {   // (stage table, strides and LDS offsets of the bf16 mode: layout.h)
namespace w8 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };

struct Args {
  const float* x;
  int64_t ldx, B;
  int F, norm;
  const void* blob;
  float* out;
  float* latent;
  uint8_t* valid;
};

template <int ACT>
__device__ __forceinline__ float activate(float v) {
  if (ACT == ACT_RELU) return v < 0.0f ? 0.0f : v;
  if (ACT == ACT_TANH) return tanhf(v);
  return v;
}

template <int NB, int NFB>
__device__ __forceinline__ void load_bias(f32x16 (&acc)[NB][NFB], const f32x4* __restrict__ b, int h) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const f32x4* p = b + (nb * 2 + h) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = p[q];
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb) {
        acc[nb][fb][4 * q + 0] = v[0];
        acc[nb][fb][4 * q + 1] = v[1];
        acc[nb][fb][4 * q + 2] = v[2];
        acc[nb][fb][4 * q + 3] = v[3];
      }
    }
  }
}

// One K step, one load per MFMA slot: MFMA m (nb-major over the face blocks, the order the four-wave kernel gives every accumulator) is
// followed by item m of the step's fetches -- the NFB LDS reads of the next step's x operands, then the NB weight fragments D steps ahead
// -- and by the caller's extra(m).
template <int NB, int NFB, typename XLoad, typename Extra>
__device__ __forceinline__ void step8(f32x16 (&acc)[NB][NFB], const bf16x8 (&wc)[NB], const bf16x8 (&xc)[NFB], bf16x8 (&wn)[NB],
                                      const bf16x8* __restrict__ wp, bool prefetch, XLoad xload, Extra extra) {
  constexpr int M = NB * NFB, I = NFB + NB;
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int nb = m / NFB, fb = m % NFB;
    acc[nb][fb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[nb], xc[fb], acc[nb][fb], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < I; ++i) {
      if ((i * M) / I != m) continue;                       // item i lives in slot floor(i * M / I)
      if (i < NFB) xload(i);
      else if (prefetch) wn[i - NFB] = wp[(i - NFB) * 64];
    }
    extra(m);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// K loop over an LDS-resident bf16 image.  `w`: this lane's fragment of the wave's first block at step 0; WSTEP: fragments (bf16x8 units
// per lane) between consecutive K steps of the job's stream (a wave that takes NB of a job's blocks passes the job's full stride);
// `in`: this lane's (face row of block 0, k = 8h).
template <int NB, int NFB, int K16, int WSTEP>
__device__ __forceinline__ void kloop8(f32x16 (&acc)[NB][NFB], const bf16x8* __restrict__ w, const __bf16* in, int fb_stride) {
  constexpr int R = 4, D = R - 1;
  static_assert(K16 % R == 0, "K steps in whole ring rounds");
  bf16x8 wr[R][NB], xr[2][NFB];
#pragma unroll
  for (int fb = 0; fb < NFB; ++fb) xr[0][fb] = *reinterpret_cast<const bf16x8*>(in + fb * fb_stride);
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) wr[d][nb] = w[(size_t)d * WSTEP + nb * 64];
  for (int g = 0; g < K16 / R; ++g) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int s = g * R + r, sp = s + D < K16 ? s + D : K16 - 1, sx = s + 1 < K16 ? s + 1 : K16 - 1;
      step8<NB, NFB>(acc, wr[r], xr[r & 1], wr[(r + D) % R], w + (size_t)sp * WSTEP, true,
                     [&](int fb) { xr[(r + 1) & 1][fb] = *reinterpret_cast<const bf16x8*>(in + fb * fb_stride + 16 * sx); }, [](int) {});
    }
  }
}

// Grouped form for the heads: NJ jobs (own input each) of one 32-face block in lock step through one ring.
template <int NJ, int NB, int K16>
__device__ __forceinline__ void kloop_grouped(f32x16 (&acc)[NJ][NB][1], const bf16x8* __restrict__ w0, size_t job_stride,
                                              const __bf16* const (&in)[NJ]) {
  constexpr int R = 4, D = R - 1, M = NJ * NB, I = NJ + NJ * NB;
  bf16x8 wr[R][NJ][NB], xr[2][NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) xr[0][j] = *reinterpret_cast<const bf16x8*>(in[j]);
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (d < K16) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) wr[d][j][nb] = w0[j * job_stride + (d * NB + nb) * 64];
    }
  }
  auto step = [&](int r, int s, bool prefetch) {
    const int sp = s + D, sx = s + 1 < K16 ? s + 1 : K16 - 1;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int j = m / NB, nb = m % NB;
      acc[j][nb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[r][j][nb], xr[r & 1][j], acc[j][nb][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < I; ++i) {
        if ((i * M) / I != m) continue;
        if (i < NJ) xr[(r + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(in[i] + 16 * sx);
        else if (prefetch) {
          const int q = i - NJ, jj = q / NB, nn = q % NB;
          wr[(r + D) % R][jj][nn] = w0[jj * job_stride + ((size_t)sp * NB + nn) * 64];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  constexpr int GROUPS = K16 / R, TAIL = K16 % R;
  for (int g = 0; g < GROUPS; ++g) {
#pragma unroll
    for (int r = 0; r < R; ++r) step(r, g * R + r, g * R + r + D < K16);
  }
#pragma unroll
  for (int r = 0; r < TAIL; ++r) step(r, GROUPS * R + r, false);
}

// accumulators -> activation -> bf16 -> LDS image [face][neuron]; `out`: lane's (face row, col0 + 4h)
template <int NB, int NFB, int ACT>
__device__ __forceinline__ void store_lds(const f32x16 (&acc)[NB][NFB], __bf16* out, int fb_stride) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bf16x4 v;
        v[0] = (__bf16)activate<ACT>(acc[nb][fb][4 * q + 0]);
        v[1] = (__bf16)activate<ACT>(acc[nb][fb][4 * q + 1]);
        v[2] = (__bf16)activate<ACT>(acc[nb][fb][4 * q + 2]);
        v[3] = (__bf16)activate<ACT>(acc[nb][fb][4 * q + 3]);
        *reinterpret_cast<bf16x4*>(out + fb * fb_stride + 32 * nb + 8 * q) = v;
      }
}

struct Ctx {
  const bf16x8* blob8;
  const f32x4* blob4;
  const Header* hdr;
  char* lds;
  int lane, f, h, wv;
};

__device__ __forceinline__ __bf16* img(const Ctx& c, int off_bytes) { return reinterpret_cast<__bf16*>(c.lds + off_bytes); }

__device__ __forceinline__ double div_ipd(double n, double d, double y) {   // == IEEE n / d for these operands (K1)
  const double q = n * y;
  const double r = fma(-q, d, n);
  return fma(r, y, q);
}

// the heads' images for BOTH 32-face blocks side by side (byte offsets): HA / HC per block, then HB / HD per block, below the latent
constexpr int P_HA = 32 * S_HA * 2, P_HB = 32 * S_HB * 2;
constexpr int O8_HA = 0, O8_HB = O8_HA + 2 * P_HA;
static_assert(O8_HB + 2 * P_HB <= O_LAT, "LDS map (heads, both face blocks)");

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

  f32x4 set[2];   // ONE staging register set (8 floats per thread), refilled as soon as it has been written to LDS
  auto gload = [&](int s) {
    s = s < nslab ? s : nslab - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int k = s * XS_COLS + scol + 4 * i;
      if (VEC4) {
        const int kc = k < F ? k : (NORM ? k - 12 * ((k - F + 15) / 12) : F - 4);   // phase-preserving clamp (zero weights there)
        set[i] = *reinterpret_cast<const f32x4*>(p + kc);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) set[i][e] = p[k + e < F ? k + e : F - 1];
      }
    }
  };
  // the staging of a slab in pieces (one per MFMA slot): 0..7 normalise element q, 8 convert + store
  auto lpiece = [&](int piece, int buf_off, bool real_slab) {
    if (piece == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(set[i]));
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
  constexpr int R0 = 4, D0 = R0 - 1;   // weight ring: K step ks in slot ks % 4; a slab holds 4 steps
  static_assert(XS_STEPS == R0, "slab steps == ring slots");
  bf16x8 wr[R0][NB];
  // prologue: slabs 0 and 1 in LDS, slab 2 in the registers; the ring and the bias are requested LAST (so that at the loop's entry no
  // load is younger than the loop's own steady state: hipcc merges the two entries' pending-load states conservatively)
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
  {  // E3: 256 -> 128, ReLU; neuron block wv & 3, face block wv >> 2
    const int nb = wv & 3, face0 = 32 * (wv >> 2);
    f32x16 acc[1][1];
    load_bias<1, 1>(acc, c.blob4 + c.hdr->b_off[ST_E3] + nb * 8, c.h);
    const bf16x8* w = c.blob8 + c.hdr->w_off[ST_E3] + (size_t)nb * c.hdr->job_w16[ST_E3] + c.lane;
    kloop8<1, 1, kStages[ST_E3].k8, 64>(acc, w, img(c, O_H3) + (face0 + c.f) * S_H3 + 8 * c.h, 0);
    store_lds<1, 1, ACT_RELU>(acc, img(c, O_H4) + (face0 + c.f) * S_H4 + 32 * nb + 4 * c.h, 0);
  }
  __syncthreads();
  if (wv < 4) {  // E4: 128 -> 64, Tanh; neuron block wv&1, face block wv>>1
    const int nb = wv & 1, face0 = 32 * (wv >> 1);
    f32x16 acc[1][1];
    load_bias<1, 1>(acc, c.blob4 + c.hdr->b_off[ST_E4] + nb * 8, c.h);
    const bf16x8* w = c.blob8 + c.hdr->w_off[ST_E4] + (size_t)nb * c.hdr->job_w16[ST_E4] + c.lane;
    kloop8<1, 1, kStages[ST_E4].k8, 64>(acc, w, img(c, O_H4) + (face0 + c.f) * S_H4 + 8 * c.h, 0);
    store_lds<1, 1, ACT_TANH>(acc, img(c, O_H5) + (face0 + c.f) * S_H5 + 32 * nb + 4 * c.h, 0);
  }
  __syncthreads();
  if (wv < 2) {  // E5: 64 -> 9, latent n = 3g+c on row 16g+c (2 blocks), other rows exact zeros; face block wv
    f32x16 acc[2][1];
    load_bias<2, 1>(acc, c.blob4 + c.hdr->b_off[ST_E5], c.h);
    const bf16x8* w = c.blob8 + c.hdr->w_off[ST_E5] + c.lane;
    kloop8<2, 1, kStages[ST_E5].k8, 2 * 64>(acc, w, img(c, O_H5) + (32 * wv + c.f) * S_H5 + 8 * c.h, 0);
    if (a.latent && row0 + 32 * wv + c.f < a.B) {   // f32 latent straight from the accumulators
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int rowi = 32 * nb + (q & 3) + 8 * (q >> 2) + 4 * c.h, g = rowi >> 4, cc = rowi & 15;
          if (g < 3 && cc < 3) a.latent[(row0 + 32 * wv + c.f) * NLML_LATENT + 3 * g + cc] = acc[nb][0][q];
        }
    }
    store_lds<2, 1, ACT_NONE>(acc, img(c, O_LAT) + (32 * wv + c.f) * S_LAT + 4 * c.h, 0);
  }
  __syncthreads();
  // ---- heads: BOTH 32-face blocks at once -- wave group wv >> 2 takes face block wv >> 2, wave wq = wv & 3 of the group the jobs 3 wq .. 3 wq + 2
  // of every stage, in lock step through one ring (kloop_grouped), as the four-wave kernel did one block after the other
  {
    const int fb = wv >> 2, wq = wv & 3, face0 = 32 * fb;
    __bf16* const HA = img(c, O8_HA + fb * P_HA);   // also HC
    __bf16* const HB = img(c, O8_HB + fb * P_HB);   // also HD
    {  // H0: 3 -> 128 (K padded to 16 with zeros), ReLU
      constexpr int ST = ST_H0;
      f32x16 acc[3][1][1];
      const __bf16* in[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wq * 3 + j;
        load_bias<1, 1>(acc[j], c.blob4 + c.hdr->b_off[ST] + job * 8, c.h);
        in[j] = img(c, O_LAT) + (face0 + c.f) * S_LAT + 16 * (job >> 2) + 8 * c.h;
      }
      kloop_grouped<3, 1, kStages[ST].k8>(acc, c.blob8 + c.hdr->w_off[ST] + (size_t)(wq * 3) * c.hdr->job_w16[ST] + c.lane, c.hdr->job_w16[ST], in);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wq * 3 + j;
        store_lds<1, 1, ACT_RELU>(acc[j], HA + c.f * S_HA + 128 * (job >> 2) + 32 * (job & 3) + 4 * c.h, 0);
      }
    }
    __syncthreads();
    {  // H1: 128 -> 256, ReLU
      constexpr int ST = ST_H1;
      f32x16 acc[3][2][1];
      const __bf16* in[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wq * 3 + j;
        load_bias<2, 1>(acc[j], c.blob4 + c.hdr->b_off[ST] + job * 16, c.h);
        in[j] = HA + c.f * S_HA + 128 * (job >> 2) + 8 * c.h;
      }
      kloop_grouped<3, 2, kStages[ST].k8>(acc, c.blob8 + c.hdr->w_off[ST] + (size_t)(wq * 3) * c.hdr->job_w16[ST] + c.lane, c.hdr->job_w16[ST], in);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wq * 3 + j;
        store_lds<2, 1, ACT_RELU>(acc[j], HB + c.f * S_HB + 256 * (job >> 2) + 64 * (job & 3) + 4 * c.h, 0);
      }
    }
    __syncthreads();
    {  // H2: 256 -> 128, ReLU; its output (HC) lies over HA, dead since the barrier above
      constexpr int ST = ST_H2;
      f32x16 acc[3][1][1];
      const __bf16* in[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wq * 3 + j;
        load_bias<1, 1>(acc[j], c.blob4 + c.hdr->b_off[ST] + job * 8, c.h);
        in[j] = HB + c.f * S_HB + 256 * (job >> 2) + 8 * c.h;
      }
      kloop_grouped<3, 1, kStages[ST].k8>(acc, c.blob8 + c.hdr->w_off[ST] + (size_t)(wq * 3) * c.hdr->job_w16[ST] + c.lane, c.hdr->job_w16[ST], in);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wq * 3 + j;
        store_lds<1, 1, ACT_RELU>(acc[j], HA + c.f * S_HC + 128 * (job >> 2) + 32 * (job & 3) + 4 * c.h, 0);
      }
    }
    __syncthreads();
    if (wq < 3) {  // H3: 128 -> 64, ReLU: waves 0..2 of the group take the two blocks of head wq; output HD over HB
      constexpr int ST = ST_H3;
      f32x16 acc[2][1][1];
      const __bf16* in[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        load_bias<1, 1>(acc[j], c.blob4 + c.hdr->b_off[ST] + (wq * 2 + j) * 8, c.h);
        in[j] = HA + c.f * S_HC + 128 * wq + 8 * c.h;
      }
      kloop_grouped<2, 1, kStages[ST].k8>(acc, c.blob8 + c.hdr->w_off[ST] + (size_t)(wq * 2) * c.hdr->job_w16[ST] + c.lane, c.hdr->job_w16[ST], in);
#pragma unroll
      for (int j = 0; j < 2; ++j) store_lds<1, 1, ACT_RELU>(acc[j], HB + c.f * S_HD + 64 * wq + 32 * j + 4 * c.h, 0);
    }
    __syncthreads();
    if (wq < 3) {  // H4: 64 -> 1; neuron on accumulator row 0 = register 0 of lanes 0..31
      f32x16 acc[1][1];
      load_bias<1, 1>(acc, c.blob4 + c.hdr->b_off[ST_H4] + wq * 8, c.h);
      const bf16x8* w = c.blob8 + c.hdr->w_off[ST_H4] + (size_t)wq * c.hdr->job_w16[ST_H4] + c.lane;
      kloop8<1, 1, kStages[ST_H4].k8, 64>(acc, w, HB + c.f * S_HD + 64 * wq + 8 * c.h, 0);
      if (c.h == 0 && row0 + face0 + c.f < a.B) a.out[(row0 + face0 + c.f) * 3 + wq] = acc[0][0][0];
    }
  }
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
