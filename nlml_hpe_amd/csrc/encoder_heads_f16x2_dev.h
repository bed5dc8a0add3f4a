// encoder_heads_f16x2_dev.h -- device helpers of the split-f16 kernels (encoder_heads_f16x2.hip: the fused kernel;
// encoder_heads_f16x2_small.hip: the layer-per-launch path, whose last launch runs tail_stages() below): operand types,
// the three-product MFMA steps, the K loops over LDS images, the hi/lo store, and the network's tail (layers E3, E4, E5
// and the three heads) as one function over a 64-face tile whose E2 output sits in the H3 LDS image.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/nlml_hpe.h"
#include "layout.h"

namespace nlml {
namespace hx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };

struct Args {
  const float* x;
  int64_t ldx, B;
  int F, norm;
  const void* blob;
  float* out;
  float* latent;
  uint8_t* valid;
  void* h3ws = nullptr;   // trunk-only launch (encoder_heads_f16x2_w8.hip, TRUNK = true): E2's output as MFMA operand fragments, read by
                          // the streamed tail (encoder_heads_f16x2_tailws.hip)
};

template <int ACT>
__device__ __forceinline__ float activate(float v) {
  if (ACT == ACT_RELU) return v < 0.0f ? 0.0f : v;   // NaN-propagating like torch.relu
  if (ACT == ACT_TANH) return tanhf(v);
  return v;
}

// (v0, v1) -> packed f16 pairs hi = f16(v), lo = f16(v - hi) in THREE instructions: v_cvt_pk_f16_f32, then one mixed-precision
// fma per element (-hi * 1.0 + v evaluated in f32 -- exact -- and rounded once to f16 into the low / high half).  Bit-identical
// to the plain C form (five conversions, two subtractions, a pack) on 4 M values incl. subnormal, overflowing and non-finite ones
// (tools/probes/split_probe.hip); the epilogues run on the vector ALU with the matrix pipe idle, so their length is wall time.
__device__ __forceinline__ void split2(float v0, float v1, unsigned& hi, unsigned& lo) {
#ifdef HX_SPLIT_C
  typedef _Float16 h2_ __attribute__((ext_vector_type(2)));
  asm volatile("" : "+v"(v0), "+v"(v1));
  h2_ a, b;
  a[0] = (_Float16)v0; a[1] = (_Float16)v1;
  b[0] = (_Float16)(v0 - (float)a[0]); b[1] = (_Float16)(v1 - (float)a[1]);
  hi = __builtin_bit_cast(unsigned, a); lo = __builtin_bit_cast(unsigned, b);
  return;
#endif
  asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
               "v_fma_mixlo_f16 %1, -%0, 1.0, %2 op_sel_hi:[1,0,0]\n\t"
               "v_fma_mixhi_f16 %1, -%0, 1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
               : "=&v"(hi), "=&v"(lo) : "v"(v0), "v"(v1));
}

// lanes 32-63 of `a` <-> lanes 0-31 of `b` (v_permlane32_swap): the half-wave exchange that turns two 8-byte row pieces per lane
// into one 16-byte piece
__device__ __forceinline__ void swap_halves(unsigned& a, unsigned& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}

// The accumulators of ONE 32-neuron block for one 32-face block -> the NEXT layer's MFMA B operands, in registers: store_lds's arithmetic
// (scale, activation, hi/lo split: the same values bit for bit) followed by the half-wave exchange of its 16-byte form, after which lane
// (f, h) holds neurons 16p + 8h .. + 7 of face f -- exactly what a v_mfma_f32_32x32x16_f16 reads as its B operand in K step p of the
// block (the bytes a consumer's ds_read_b128 would find at (row f, column 16p + 8h) of the LDS image).  frag[p][0] = hi, [p][1] = lo.
template <int ACT>
__device__ __forceinline__ void frags_from_acc(const f32x16& acc, float inv, h8 (&frag)[2][2]) {
  typedef unsigned u4_ __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = activate<ACT>(acc[8 * p + e] * inv);
    unsigned hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split2(v[2 * j], v[2 * j + 1], hi[j], lo[j]);
    swap_halves(hi[0], hi[2]);
    swap_halves(hi[1], hi[3]);
    swap_halves(lo[0], lo[2]);
    swap_halves(lo[1], lo[3]);
    frag[p][0] = __builtin_bit_cast(h8, u4_{hi[0], hi[1], hi[2], hi[3]});
    frag[p][1] = __builtin_bit_cast(h8, u4_{lo[0], lo[1], lo[2], lo[3]});
  }
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

// the three products of one K step for NB x NFB accumulators; the two small terms first.  Consecutive MFMAs
// go to different accumulators.
template <int NB, int NFB>
__device__ __forceinline__ void mma_step(f32x16 (&acc)[NB][NFB], const h8 (&w)[NB][2], const h8 (&x)[NFB][2]) {
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int wp = t == 0 ? 1 : 0, xp = t == 1 ? 1 : 0;   // (lo,hi), (hi,lo), (hi,hi)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb)
        acc[nb][fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[nb][wp], x[fb][xp], acc[nb][fb], 0, 0, 0);
  }
}

// One K step, interleaved one MFMA at a time.  A 32x32x16 MFMA holds the matrix pipe for 32 cycles but the SIMD's issue port
// for only 8, so ~24 cycles of OTHER instructions (5-6 VALU, or one vector-memory instruction) issue for free behind each MFMA
// -- but only behind each: with four MFMAs issued back to back and then a block of loads (the previous shape of this step),
// just the last MFMA's shadow is used and the block's remainder idles the pipe (stage stamps: 86 % busy in layer 1 with
// nothing but the step's own 8 weight loads and 4 LDS reads, 60-70 % in layer 0 with the x staging on top).  Here every
// MFMA m of the step (M = 3 * NB * NFB, order unchanged: (lo,hi), (hi,lo), (hi,hi) over the accumulators) is followed by its
// share of the step's I = 2*NFB + 2*NB operand fetches -- first the LDS reads of the NEXT step's x operands (they are needed
// soonest), then the global loads of the weights D steps ahead -- and by whatever the caller drops into slot m via `extra(m)`
// (layer 0: pieces of the x staging).  sched_barrier(0) pins every slot.
//
// SPLIT ACCUMULATORS (round 3): `accS` receives the two SMALL products (w_lo*x_hi, w_hi*x_lo), `acc` only w_hi*x_hi; the caller adds
// accS to acc once at the end (layer 1: also at its K midpoint).  v_mfma_f32_32x32x16_f16 aligns C and its 16 products to the
// largest exponent among them with ~3 guard bits and truncates each aligned addend (tools/probes/mfma_f16_numerics_probe.hip):
// against a large running sum one instruction costs 0.41 ulp rms of the sum.  With all three products in one accumulator that is
// 3 x 88 such costs per layer-0 dot product, and the kernel with single accumulators throughout sat 1.23x further from the exact result than the reference's own f32
// GEMM at the reference's operating range (FX3c).  Now the big accumulator takes ONE instruction per 16 k, the small one sums
// values 2^-11 the size (its truncations are 2^-11 the size too): measured p50 2.08e-5 -> 1.49e-5 deg, 12 % INSIDE the reference's
// distance in p50, p99 and max.  Pass the same array twice for a single accumulator (the tail layers E3.. and the heads).
template <int NB, int NFB, typename XLoad, typename Extra>
__device__ __forceinline__ void step_fine(f32x16 (&acc)[NB][NFB], f32x16 (&accS)[NB][NFB], const h8 (&wcur)[NB][2], const h8 (&xcur)[NFB][2],
                                          h8 (&wnext)[NB][2], const h8* __restrict__ wp, bool prefetch, XLoad xload, Extra extra) {
  constexpr int M = 3 * NB * NFB, I = 2 * NFB + 2 * NB;
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int t = m / (NB * NFB), nb = (m % (NB * NFB)) / NFB, fb = m % NFB;
    const int wp_ = t == 0 ? 1 : 0, xp_ = t == 1 ? 1 : 0;   // (lo,hi), (hi,lo), (hi,hi)
    if (t < 2) accS[nb][fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wcur[nb][wp_], xcur[fb][xp_], accS[nb][fb], 0, 0, 0);
    else acc[nb][fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wcur[nb][wp_], xcur[fb][xp_], acc[nb][fb], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < I; ++i) {
      if ((i * M) / I != m) continue;                       // item i lives in slot floor(i * M / I)
      if (i < 2 * NFB) xload(i >> 1, i & 1);                // x operand (face block, piece) of the next step
      else if (prefetch) wnext[(i - 2 * NFB) >> 1][(i - 2 * NFB) & 1] = wp[(i - 2 * NFB) * 64];   // (non-temporal: 70 M instead of 81 M)
    }
    extra(m);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// K loop over an LDS-resident hi/lo image; K16 static, no runtime branch in the body (see the f32 kernel).
// `w`: this lane's hi fragment of block 0, step 0 (fragment (step, nb, piece) at ((step*NB + nb)*2 + piece)*64);
// `in`: this lane's (face row of block 0, k = 8h) in the hi plane; the lo plane is `plane` bytes further.
// Split in two so that a stage's GLOBAL fetches (kloop_pro: the first D ring slots; the bias goes with them, job_pre) can be
// issued before the PREVIOUS stage's store and barrier -- the short stages of the tail otherwise expose one L2 latency and the
// issue time of ~40 loads each -- while the LDS reads and the MFMAs (kloop_run) wait for the barrier.
constexpr int ring_slots(int nb, int nfb) { return (nb * nfb >= 4) ? 4 : 6; }
// K steps a ring runs ahead of the MFMAs (at most slots - 1).  -DHX_KLOOP_D4=<n> / -DHX_KLOOP_D6=<n>: experiments with a shallower look-ahead.
#ifndef HX_KLOOP_D4
#define HX_KLOOP_D4 3
#endif
#ifndef HX_KLOOP_D6
#define HX_KLOOP_D6 5
#endif
constexpr int ring_ahead(int slots) { return slots == 4 ? HX_KLOOP_D4 : HX_KLOOP_D6; }

// WSTEP: fragments (16-byte units per lane, i.e. h8 elements / 64 lanes) between consecutive K steps of the stream `w` points into.
// The default is a job's own stream (NB blocks x 2 pieces); a wave that takes only NB of a job's blocks (the eight-wave kernel:
// two of four, one of two) passes the job's full stride and a `w` that points at its first block.
template <int NB, int NFB, int K16, int WSTEP = NB * 2 * 64>
__device__ __forceinline__ void kloop_pro(h8 (&wr)[ring_slots(NB, NFB)][NB][2], const h8* __restrict__ w) {
  constexpr int R = ring_slots(NB, NFB), D = ring_ahead(R);
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (d < K16) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int p = 0; p < 2; ++p) wr[d][nb][p] = w[d * WSTEP + (nb * 2 + p) * 64];
    }
  }
}

template <int NB, int NFB, int K16, int WSTEP = NB * 2 * 64>
__device__ __forceinline__ void kloop_run(f32x16 (&acc)[NB][NFB], f32x16 (&accS)[NB][NFB], h8 (&wr)[ring_slots(NB, NFB)][NB][2],
                                          const h8* __restrict__ w, const char* in, int plane, int fb_stride) {
  constexpr int R = ring_slots(NB, NFB), D = ring_ahead(R);
  h8 xr[2][NFB][2];
#pragma unroll
  for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
    for (int p = 0; p < 2; ++p) xr[0][fb][p] = *reinterpret_cast<const h8*>(in + p * plane + fb * fb_stride);
  auto step = [&](int r, int xs, int sp, int sx, bool prefetch) {
    const int sxc = sx < K16 ? sx : K16 - 1;
    step_fine<NB, NFB>(acc, accS, wr[r], xr[xs], wr[(r + D) % R], w + (size_t)sp * WSTEP, prefetch,
                       [&](int fb, int p) { xr[xs ^ 1][fb][p] = *reinterpret_cast<const h8*>(in + p * plane + fb * fb_stride + 32 * sxc); },
                       [](int) {});
  };
  // R is even, so the x double buffer slot (step & 1) is static inside the unrolled group
  static_assert(R % 2 == 0, "ring size even");
  constexpr int GROUPS = K16 / R, TAIL = K16 % R;
  for (int g = 0; g < GROUPS; ++g) {
#pragma unroll
    for (int r = 0; r < R; ++r) step(r, r & 1, g * R + r + D, g * R + r + 1, true);
  }
#pragma unroll
  for (int r = 0; r < TAIL; ++r) step(r, r & 1, 0, GROUPS * R + r + 1, false);
}

template <int NB, int NFB, int K16, int WSTEP = NB * 2 * 64>
__device__ __forceinline__ void kloop(f32x16 (&acc)[NB][NFB], f32x16 (&accS)[NB][NFB], const h8* __restrict__ w, const char* in,
                                      int plane, int fb_stride) {
  h8 wr[ring_slots(NB, NFB)][NB][2];
  kloop_pro<NB, NFB, K16, WSTEP>(wr, w);
  kloop_run<NB, NFB, K16, WSTEP>(acc, accS, wr, w, in, plane, fb_stride);
}

// small accumulator helpers
template <int NB, int NFB>
__device__ __forceinline__ void zero_acc(f32x16 (&a)[NB][NFB]) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
      for (int q = 0; q < 16; ++q) a[nb][fb][q] = 0.0f;
}
template <int NB, int NFB>
__device__ __forceinline__ void add_acc(f32x16 (&a)[NB][NFB], const f32x16 (&b)[NB][NFB]) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) a[nb][fb] += b[nb][fb];
}

// Grouped form for the heads: NJ jobs (own input each) in lock step through one ring (32-face block); the same split.
template <int NJ, int NB, int K16>
__device__ __forceinline__ void gloop_pro(h8 (&wr)[4][NJ][NB][2], const h8* __restrict__ w0, size_t job_stride) {
  constexpr int R = 4, D = R - 1;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (d < K16) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int p = 0; p < 2; ++p) wr[d][j][nb][p] = w0[j * job_stride + ((d * NB + nb) * 2 + p) * 64];
    }
  }
}

template <int NJ, int NB, int K16>
__device__ __forceinline__ void gloop_run(f32x16 (&acc)[NJ][NB][1], h8 (&wr)[4][NJ][NB][2], const h8* __restrict__ w0,
                                          size_t job_stride, const char* const (&in)[NJ], int plane) {
  constexpr int R = 4, D = R - 1;
  h8 xr[2][NJ][1][2];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int p = 0; p < 2; ++p) xr[0][j][0][p] = *reinterpret_cast<const h8*>(in[j] + p * plane);
  auto step = [&](int r, int xs, int sp, int sx, bool prefetch) {   // one MFMA at a time, as step_fine
    const int sxc = sx < K16 ? sx : K16 - 1;
    constexpr int M = 3 * NJ * NB, I = 2 * NJ + 2 * NJ * NB;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int t = m / (NJ * NB), j = (m % (NJ * NB)) / NB, nb = m % NB;
      const int wp = t == 0 ? 1 : 0, xp = t == 1 ? 1 : 0;
      acc[j][nb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[r][j][nb][wp], xr[xs][j][0][xp], acc[j][nb][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < I; ++i) {
        if ((i * M) / I != m) continue;
        if (i < 2 * NJ) {
          xr[xs ^ 1][i >> 1][0][i & 1] = *reinterpret_cast<const h8*>(in[i >> 1] + (i & 1) * plane + 32 * sxc);
        } else if (prefetch) {
          const int q = i - 2 * NJ, jj = q / (2 * NB), nn = (q / 2) % NB, pp = q & 1;
          wr[(r + D) % R][jj][nn][pp] = w0[jj * job_stride + (((size_t)sp * NB + nn) * 2 + pp) * 64];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  constexpr int GROUPS = K16 / R, TAIL = K16 % R;
  for (int g = 0; g < GROUPS; ++g) {
#pragma unroll
    for (int r = 0; r < R; ++r) step(r, r & 1, g * R + r + D, g * R + r + 1, true);
  }
#pragma unroll
  for (int r = 0; r < TAIL; ++r) step(r, r & 1, 0, GROUPS * R + r + 1, false);
}

// accumulators * inv -> activation -> hi/lo f16 -> LDS image [piece][face][neuron].  `out`: the lane's hi-plane address of
// (its face row, the job's first column); columns at or beyond MAXCOL are not written (latent image).
// A lane holds rows 8q + 4h .. + 3 (q < 4) of its face column: 8-byte pieces, and as ds_write_b64 they were 2-way bank
// conflicted (26 % of all LDS cycles, profiles/r01_pmc_summary.md).  Register q = 2p of the upper half-wave is exchanged with
// register q = 2p + 1 of the lower one, after which lanes 0-31 hold rows 16p .. 16p+7 and lanes 32-63 rows 16p+8 .. 16p+15:
// one conflict-free 16-byte store per plane (row strides are 4 mod 32 dwords: the 8 lanes of a ds_write_b128 group tile all
// 32 banks).  `hook(i)` runs after group i of NB * NFB * 2 groups: the NEXT stage's global fetches are dropped in there one or
// two at a time, because a 1-KiB load costs ~64 cycles of issue when all four waves of the CU are fetching (the CU takes in
// 64 B/clk) and the epilogue's ~130 VALU cycles per group hide it.
struct NoHook { __device__ __forceinline__ void operator()(int) const {} };

template <int NB, int NFB, int ACT, int MAXCOL = 1 << 30, typename Hook = NoHook>
__device__ __forceinline__ void store_lds(const f32x16 (&acc)[NB][NFB], char* out, int h, int plane, int fb_stride, float inv,
                                          Hook hook = Hook{}) {
  hook(-1);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        if (32 * nb + 16 * p < MAXCOL) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = activate<ACT>(acc[nb][fb][8 * p + e] * inv);
          unsigned hi[4], lo[4];      // [0,1] = register q = 2p (rows 4h .. +3), [2,3] = register q = 2p + 1 (rows 8 + 4h .. +3)
#pragma unroll
          for (int j = 0; j < 4; ++j) split2(v[2 * j], v[2 * j + 1], hi[j], lo[j]);
#ifndef HX_STORE_B128
          typedef unsigned u2 __attribute__((ext_vector_type(2)));
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
            const int col = 32 * nb + 16 * p + 8 * qq + 4 * h;
            if (MAXCOL >= 32 * NB || 32 * nb + 16 * p + 8 * qq < MAXCOL) {
              char* d = out + fb * fb_stride + col * 2;
              *reinterpret_cast<u2*>(d) = u2{hi[2 * qq], hi[2 * qq + 1]};
              *reinterpret_cast<u2*>(d + plane) = u2{lo[2 * qq], lo[2 * qq + 1]};
            }
          }
#else
          swap_halves(hi[0], hi[2]);
          swap_halves(hi[1], hi[3]);
          swap_halves(lo[0], lo[2]);
          swap_halves(lo[1], lo[3]);
          const int col = 32 * nb + 16 * p + 8 * h;
          if (MAXCOL >= 32 * NB || col < MAXCOL) {
            char* d = out + fb * fb_stride + col * 2;
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<u4*>(d) = u4{hi[0], hi[1], hi[2], hi[3]};
            *reinterpret_cast<u4*>(d + plane) = u4{lo[0], lo[1], lo[2], lo[3]};
          }
#endif
        }
        if constexpr (!std::is_same<Hook, NoHook>::value) {   // pin the fetches between the groups; without a hook the
          __builtin_amdgcn_sched_barrier(0);                 // scheduler is free to overlap the groups' dependency chains
          hook((nb * NFB + fb) * 2 + p);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
}

// The blob header's per-stage fields, without reading them at their points of use: read through the header pointer they are
// re-fetched after every global store of the kernel (the blob pointer is not provably distinct from the output pointers) -- as
// VECTOR loads with a full memory latency in front of every address that depends on them: ~600 exposed cycles in front of each
// prefetch group of the head stages.  The offsets are arithmetic in layer 0's step count (the only size that depends on F):
// stage s starts at  W0[s] + W1[s] * k16_e0  (16-byte units), exactly as pack.cpp lays the blob out (it checks this formula
// against the header it writes); only k16_e0 and the eleven power-of-two scales are data, read once at kernel entry into SGPRs.
struct StageOff { int w0, w1; };   // w_off = w0 + w1 * k16_e0
constexpr int stage_job_w16_const(int s) { return s == ST_E0 ? 0 : kStages[s].k8 * kStages[s].nb * 64 * PIECES; }
constexpr int stage_job_w16_k(int s) { return s == ST_E0 ? kStages[s].nb * 64 * PIECES : 0; }
constexpr StageOff stage_w_off(int s) {
  StageOff o{(int)(sizeof(Header) / 16), 0};
  for (int t = 0; t < s; ++t) {
    o.w0 += kStages[t].jobs * (stage_job_w16_const(t) + kStages[t].nb * 8);
    o.w1 += kStages[t].jobs * stage_job_w16_k(t);
  }
  return o;
}

struct HdrRegs {
  uint32_t k8_e0;
  float inv_scale[NUM_STAGES];
  __device__ __forceinline__ uint32_t job_w16(int s) const { return stage_job_w16_const(s) + stage_job_w16_k(s) * k8_e0; }
  __device__ __forceinline__ uint32_t w_off(int s) const { return stage_w_off(s).w0 + stage_w_off(s).w1 * k8_e0; }
  __device__ __forceinline__ uint32_t b_off(int s) const { return w_off(s) + kStages[s].jobs * job_w16(s); }
};

__device__ __forceinline__ HdrRegs load_hdr(const Header* hdr) {
  HdrRegs r;
#pragma unroll
  for (int s = 0; s < NUM_STAGES; ++s)
    r.inv_scale[s] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, hdr->inv_scale[s])));
  r.k8_e0 = __builtin_amdgcn_readfirstlane(hdr->k8_e0);
  return r;
}

struct Ctx {
  const h8* blob8;
  const f32x4* blob4;
  HdrRegs hdr;
  char* lds;
  int lane, f, h, wv;
  int helper = 0;   // eight-wave kernel, tail_stages<.., H1W8 = true>: waves 4..7 (wv = wave & 3, helper = 1) stay for the heads' H1 stage
};

}  // namespace hx
}  // namespace nlml
#include "encoder_heads_f16x2_rescue.h"
namespace nlml {
namespace hx {

// "this face's pose came out non-finite": one f16 slot per face in the latent image's hi plane, column 48 -- E5's store
// writes exact zeros there (columns 48..55 belong to zero-padded accumulator rows) and no head reads them.
__device__ __forceinline__ unsigned short* rescue_flag(char* lds, int face) {
  return reinterpret_cast<unsigned short*>(lds + O_LAT + (face * S_LAT + 48) * 2);
}

// One job whose input image (hi plane at byte offset in_off, row stride in_stride f16) is in LDS, in two halves:
// job_pre = its global fetches (bias into the accumulators, first ring slots), pinned where the caller puts it;
// job_run = LDS reads + MFMAs.  job_compute = both (stages with nothing to overlap with).
// number of global fetches of a job's first half: 4 * NB bias quads + the first min(D, K16) ring slots of NB x 2 fragments
template <int NB, int NFB, int STAGE>
constexpr int job_pre_items() {
  constexpr int D = ring_slots(NB, NFB) - 1, K16 = kStages[STAGE].k8;
  return 4 * NB + (D < K16 ? D : K16) * NB * 2;
}

// fetch number i (static) of job_pre: lets a caller spread the fetches over another stage's epilogue (store_lds hook)
template <int NB, int NFB, int STAGE>
__device__ __forceinline__ void job_pre_item(const Ctx& c, int job, f32x16 (&acc)[NB][NFB], h8 (&wr)[ring_slots(NB, NFB)][NB][2], int i) {
  static_assert(kStages[STAGE].nb == NB, "job shape");
  if (i < 4 * NB) {
    const int nb = i / 4, q = i % 4;
    const f32x4 v = (c.blob4 + c.hdr.b_off(STAGE) + job * (NB * 8))[(nb * 2 + c.h) * 4 + q];
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) {
      acc[nb][fb][4 * q + 0] = v[0];
      acc[nb][fb][4 * q + 1] = v[1];
      acc[nb][fb][4 * q + 2] = v[2];
      acc[nb][fb][4 * q + 3] = v[3];
    }
  } else if (i < job_pre_items<NB, NFB, STAGE>()) {
    const int j = i - 4 * NB, d = j / (NB * 2), nb = (j / 2) % NB, p = j & 1;
    const h8* w = c.blob8 + c.hdr.w_off(STAGE) + (size_t)job * c.hdr.job_w16(STAGE) + c.lane;
    wr[d][nb][p] = w[((d * NB + nb) * 2 + p) * 64];
  }
}

template <int NB, int NFB, int STAGE>
__device__ __forceinline__ void job_pre(const Ctx& c, int job, f32x16 (&acc)[NB][NFB], h8 (&wr)[ring_slots(NB, NFB)][NB][2]) {
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < job_pre_items<NB, NFB, STAGE>(); ++i) job_pre_item<NB, NFB, STAGE>(c, job, acc, wr, i);
  __builtin_amdgcn_sched_barrier(0);
}

// hook for store_lds: the fetches of job (NB, NFB, STAGE) spread evenly over the GROUPS epilogue groups of the storing job
template <int GROUPS, int NB, int NFB, int STAGE>
__device__ __forceinline__ auto fetch_hook(const Ctx& c, int job, f32x16 (&acc)[NB][NFB], h8 (&wr)[ring_slots(NB, NFB)][NB][2]) {
  // g = -1 (before the first group): the bias quads, all at once -- they land in accumulator registers, and a copy that has to
  // wait for its load inside a pinned hook region would serialise one L2 latency per quad; g >= 0: the ring fragments
  return [&c, job, &acc, &wr](int g) {
    constexpr int NBIAS = 4 * NB, N = job_pre_items<NB, NFB, STAGE>() - NBIAS;
#pragma unroll
    for (int i = 0; i < NBIAS; ++i)
      if (g < 0) job_pre_item<NB, NFB, STAGE>(c, job, acc, wr, i);
#pragma unroll
    for (int i = 0; i < N; ++i)
      if ((i * GROUPS) / N == g) job_pre_item<NB, NFB, STAGE>(c, job, acc, wr, NBIAS + i);
  };
}

template <int NB, int NFB, int STAGE>
__device__ __forceinline__ void job_run(const Ctx& c, int job, f32x16 (&acc)[NB][NFB], h8 (&wr)[ring_slots(NB, NFB)][NB][2],
                                        int in_off, int plane, int in_stride, int in_col, int face0) {
  const h8* w = c.blob8 + c.hdr.w_off(STAGE) + (size_t)job * c.hdr.job_w16(STAGE) + c.lane;
  kloop_run<NB, NFB, kStages[STAGE].k8>(acc, acc, wr, w, c.lds + in_off + ((face0 + c.f) * in_stride + in_col + 8 * c.h) * 2, plane,
                                        32 * in_stride * 2);
}

// the same with the two small products in an accumulator of their own, added to `acc` at the end (see step_fine)
template <int NB, int NFB, int STAGE>
__device__ __forceinline__ void job_run_split(const Ctx& c, int job, f32x16 (&acc)[NB][NFB], h8 (&wr)[ring_slots(NB, NFB)][NB][2],
                                              int in_off, int plane, int in_stride, int in_col, int face0) {
  const h8* w = c.blob8 + c.hdr.w_off(STAGE) + (size_t)job * c.hdr.job_w16(STAGE) + c.lane;
  f32x16 accS[NB][NFB];
  zero_acc<NB, NFB>(accS);
  kloop_run<NB, NFB, kStages[STAGE].k8>(acc, accS, wr, w, c.lds + in_off + ((face0 + c.f) * in_stride + in_col + 8 * c.h) * 2, plane,
                                        32 * in_stride * 2);
  add_acc<NB, NFB>(acc, accS);
}

template <int NB, int NFB, int STAGE>
__device__ __forceinline__ void job_compute(const Ctx& c, int job, f32x16 (&acc)[NB][NFB], int in_off, int plane,
                                            int in_stride, int in_col, int face0) {
  h8 wr[ring_slots(NB, NFB)][NB][2];
  job_pre<NB, NFB, STAGE>(c, job, acc, wr);
  job_run<NB, NFB, STAGE>(c, job, acc, wr, in_off, plane, in_stride, in_col, face0);
}

// The same halves for a head stage: NJ jobs job0 .. job0+NJ-1 of stage ST through one ring.
template <int NJ, int NB, int ST>
__device__ __forceinline__ void heads_pre(const Ctx& c, int job0, f32x16 (&acc)[NJ][NB][1], h8 (&wr)[4][NJ][NB][2]) {
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < NJ; ++j) load_bias<NB, 1>(acc[j], c.blob4 + c.hdr.b_off(ST) + (job0 + j) * (NB * 8), c.h);
  gloop_pro<NJ, NB, kStages[ST].k8>(wr, c.blob8 + c.hdr.w_off(ST) + (size_t)job0 * c.hdr.job_w16(ST) + c.lane, c.hdr.job_w16(ST));
  __builtin_amdgcn_sched_barrier(0);
}

template <int NJ, int NB, int ST>
__device__ __forceinline__ void heads_run(const Ctx& c, int job0, f32x16 (&acc)[NJ][NB][1], h8 (&wr)[4][NJ][NB][2],
                                          const char* const (&in)[NJ], int plane) {
  gloop_run<NJ, NB, kStages[ST].k8>(acc, wr, c.blob8 + c.hdr.w_off(ST) + (size_t)job0 * c.hdr.job_w16(ST) + c.lane,
                                    c.hdr.job_w16(ST), in, plane);
}

template <int NB, int NFB, int ACT, int MAXCOL = 1 << 30, typename Hook = NoHook>
__device__ __forceinline__ void job_store(const Ctx& c, const f32x16 (&acc)[NB][NFB], int out_off, int plane,
                                          int out_stride, int out_col, int face0, float inv, Hook hook = Hook{}) {
  store_lds<NB, NFB, ACT, MAXCOL>(acc, c.lds + out_off + ((face0 + c.f) * out_stride + out_col) * 2, c.h, plane,
                                  32 * out_stride * 2, inv, hook);
}

// Timing-only diagnostic build (-DHX_STAMPS, tools/hx_stage_shares.py): per-wave s_memtime stamps at the stage
// boundaries are written into the buffer passed as `latent` (which then carries no latent).
#ifdef HX_STAMPS
#define HXS(i)                                                                                                 \
  do {                                                                                                         \
    if (a.latent && c.lane == 0)                                                                               \
      reinterpret_cast<unsigned long long*>(a.latent)[((size_t)blockIdx.x * 4 + wv) * 32 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define HXS_WALL(i)                                                                                            \
  do {                                                                                                         \
    if (a.latent && c.lane == 0)                                                                               \
      reinterpret_cast<unsigned long long*>(a.latent)[((size_t)blockIdx.x * 4 + wv) * 32 + (i)] = __builtin_readcyclecounter() * 0 + wall_clock64(); \
  } while (0)
#define HXS_G0(i) do { if (g == 0) HXS(i); } while (0)
#else
#define HXS(i) do { } while (0)
#define HXS_WALL(i) do { } while (0)
#define HXS_G0(i) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------
// Layers E3 .. H4 of one 64-face tile (rows row0 .. row0+63): input = the H3 image (E2's output, hi/lo planes at O_H3),
// output = poses (and the latent) in global memory.  All 256 threads of the workgroup call it after a barrier.
// ONEFB = true (small-batch path): the workgroup handles only the 32-face block `fbsel` of the tile (another workgroup
// takes the other one): the same MFMAs per face block -- faces are MFMA columns -- in fewer sequential stages.
//
// The thirteen stages are short (4 to 144 MFMAs per wave), so what a stage fetches from global memory -- bias and the first
// ring slots of its weights -- is issued BEFORE the previous stage's store and barrier (job_pre / heads_pre, pinned by
// sched_barrier): the L2 latency and the issue time of those loads then run under the store's VALU work and the barrier
// instead of in front of the stage's first MFMA.  The E3 fetch of the first stage is the caller's (tail_pre_e3) for the same
// reason: it goes in front of E2's store.
// The three heads of a 64-face tile, ONE HEAD AT A TIME: H0_g .. H4_g for g = yaw, pitch, roll.  The head layers stream 4 bytes
// of weights per MAC-column like every other layer, but with the three heads side by side their activations for 64 faces do
// not fit LDS (H1's output alone is 3 x 256 x 64 x 4 B = 196 KB), which is why the heads used to run per 32-face block: every
// weight fragment fetched twice, 3 MFMAs per fragment -- and at ~36 B/clk/CU the L2 -> CU stream, not the matrix pipe, set
// their time (10 stages, 77 k of a tile's 370 k cycles at 6-45 % MFMA-busy).  One head's images for 64 faces are 121 KB, so a
// fragment is fetched once and feeds both face blocks (6 MFMAs), as in the trunk.  Same jobs, same blob, and per accumulator
// the same K-ascending MFMA sequence as before: the results are bit-identical to the per-block form (the small-batch tail).
//   images (f16 hi plane, lo plane behind it; row = face 0..63): HA_g 128 (+8), HB_g 256 (+8), HC_g 128 (+8) over HA_g, HD_g 64 (+8)
constexpr int S_G128 = 136, S_G256 = 264, S_G64 = 72;
constexpr int P_G128 = 64 * S_G128 * 2, P_G256 = 64 * S_G256 * 2, P_G64 = 64 * S_G64 * 2;
constexpr int O_GA = 0, O_GB = O_GA + 2 * P_G128, O_GC = O_GA, O_GD = O_GB + 2 * P_G256;
static_assert(O_GD + 2 * P_G64 <= O_LAT, "LDS map (heads, one at a time)");

// H1 of head g by NEURON BLOCK over eight waves: block b (0..7) of the head's eight = job 4g + (b >> 1), block b & 1 of
// that job, both face blocks -- a weight fragment is still fetched once per tile (no duplicate fetch, unlike the face-block split that was
// measured slower in round 4), and per accumulator the MFMAs are step_fine's in step_fine's order: the same bits.
constexpr int H1_WSTEP = 2 * 2 * 64;   // an H1 job streams two blocks x two pieces per K step
__device__ __forceinline__ void h1_block_pre(const Ctx& c, int g, int b, f32x16 (&acc)[1][2], h8 (&wr)[ring_slots(1, 2)][1][2]) {
  const int job = 4 * g + (b >> 1), nbh = b & 1;
  __builtin_amdgcn_sched_barrier(0);
  load_bias<1, 2>(acc, c.blob4 + c.hdr.b_off(ST_H1) + job * (2 * 8) + nbh * 8, c.h);
  kloop_pro<1, 2, kStages[ST_H1].k8, H1_WSTEP>(wr, c.blob8 + c.hdr.w_off(ST_H1) + (size_t)job * c.hdr.job_w16(ST_H1) + (nbh * 2) * 64 + c.lane);
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void h1_block_run(const Ctx& c, int g, int b, f32x16 (&acc)[1][2], h8 (&wr)[ring_slots(1, 2)][1][2], int o_ga, int p_ga, int s_ga) {
  const int job = 4 * g + (b >> 1), nbh = b & 1;
  const h8* w = c.blob8 + c.hdr.w_off(ST_H1) + (size_t)job * c.hdr.job_w16(ST_H1) + (nbh * 2) * 64 + c.lane;
  kloop_run<1, 2, kStages[ST_H1].k8, H1_WSTEP>(acc, acc, wr, w, c.lds + o_ga + (c.f * s_ga + 8 * c.h) * 2, p_ga, 32 * s_ga * 2);
}

template <bool H1W8 = false>
__device__ __forceinline__ void heads_by_head(const Ctx& c, const Args& a, int64_t row0) {
  const int wv = c.wv;
#pragma unroll 1
  for (int g = 0; g < 3; ++g) {
    // opaque copy of the lane coordinates: keeps the loop-invariant LDS addresses and weight pointers of all five stages from
    // being hoisted out of the loop and held (then spilled) across the stages that need every register
    Ctx cl = c;
    asm volatile("" : "+v"(cl.lane), "+v"(cl.f), "+v"(cl.h));
    const int job = 4 * g + wv;
    f32x16 acc0[1][2], acc1[2][2], acc2[1][2], acc3[1][1], acc4[1][1];
    h8 wr0[6][1][2], wr1[4][2][2], wr2[6][1][2], wr3[6][1][2], wr4[6][1][2];
    // H0_g: 3 -> 128 (K padded to 16), ReLU; wave = neuron block
    job_pre<1, 2, ST_H0>(cl, job, acc0, wr0);
    HXS_G0(22);
    job_run<1, 2, ST_H0>(cl, job, acc0, wr0, O_LAT, P_LAT, S_LAT, 16 * g, 0);
    HXS_G0(23);
    HXS_G0(24);
    f32x16 acc1h[1][2];          // (H1W8: this wave's ONE block of H1_g's eight; waves 4..7 take the other four, tail_helper_w8)
    h8 wr1h[ring_slots(1, 2)][1][2];
    if constexpr (H1W8) {
      h1_block_pre(cl, g, wv, acc1h, wr1h);
      job_store<1, 2, ACT_RELU>(cl, acc0, O_GA, P_G128, S_G128, 32 * wv, 0, cl.hdr.inv_scale[ST_H0]);
    } else {
      job_store<1, 2, ACT_RELU>(cl, acc0, O_GA, P_G128, S_G128, 32 * wv, 0, cl.hdr.inv_scale[ST_H0],
                                fetch_hook<4, 2, 2, ST_H1>(cl, job, acc1, wr1));
    }
    HXS_G0(25);
    __syncthreads();
    HXS(13 + 3 * g);
    // H1_g: 128 -> 256, ReLU; wave = two neuron blocks (H1W8: one of eight)
    if constexpr (H1W8) {
      h1_block_run(cl, g, wv, acc1h, wr1h, O_GA, P_G128, S_G128);
      job_store<1, 2, ACT_RELU>(cl, acc1h, O_GB, P_G256, S_G256, 32 * wv, 0, cl.hdr.inv_scale[ST_H1],
                                fetch_hook<4, 1, 2, ST_H2>(cl, job, acc2, wr2));
    } else {
    job_run<2, 2, ST_H1>(cl, job, acc1, wr1, O_GA, P_G128, S_G128, 0, 0);
    HXS_G0(26);
    HXS_G0(27);
    job_store<2, 2, ACT_RELU>(cl, acc1, O_GB, P_G256, S_G256, 64 * wv, 0, cl.hdr.inv_scale[ST_H1],
                              fetch_hook<8, 1, 2, ST_H2>(cl, job, acc2, wr2));
    }
    HXS_G0(28);
    __syncthreads();
    HXS(14 + 3 * g);
    // H2_g: 256 -> 128, ReLU; its output image lies over HA_g (dead since the barrier above)
    job_run<1, 2, ST_H2>(cl, job, acc2, wr2, O_GB, P_G256, S_G256, 0, 0);
    const int nb3 = wv & 1, fb3 = wv >> 1;                      // H3_g: wave = (neuron block, face block)
    job_store<1, 2, ACT_RELU>(cl, acc2, O_GC, P_G128, S_G128, 32 * wv, 0, cl.hdr.inv_scale[ST_H2],
                              fetch_hook<4, 1, 1, ST_H3>(cl, 2 * g + nb3, acc3, wr3));
    __syncthreads();
    // H3_g: 128 -> 64, ReLU
    job_run<1, 1, ST_H3>(cl, 2 * g + nb3, acc3, wr3, O_GC, P_G128, S_G128, 0, 32 * fb3);
    // every wave fetches H4's operands (see tail_stages, E5); waves 0, 1 run it
    job_store<1, 1, ACT_RELU>(cl, acc3, O_GD, P_G64, S_G64, 32 * nb3, 32 * fb3, cl.hdr.inv_scale[ST_H3],
                              fetch_hook<2, 1, 1, ST_H4>(cl, g, acc4, wr4));
    __syncthreads();
    // H4_g: 64 -> 1; neuron on accumulator row 0 = register 0 of lanes 0..31; wave = face block
    if (wv < 2) {
      job_run<1, 1, ST_H4>(cl, g, acc4, wr4, O_GD, P_G64, S_G64, 0, 32 * wv);
      const int face = 32 * wv + cl.f;
      if (cl.h == 0 && row0 + face < a.B) {
        const float pose = acc4[0][0][0] * cl.hdr.inv_scale[ST_H4];
        a.out[(row0 + face) * 3 + g] = pose;
        if (!__builtin_isfinite(pose)) *rescue_flag(cl.lds, face) = 1;   // an activation left f16's range (or the input is NaN/Inf)
      }
    }
    __syncthreads();   // HD_g and HC_g are free for the next head
    HXS(15 + 3 * g);
  }
}

template <bool ONEFB>
struct TailE3 {
  static constexpr int NFB = ONEFB ? 1 : 2;
};

template <bool ONEFB>
__device__ __forceinline__ void tail_pre_e3(const Ctx& c, f32x16 (&acc)[1][TailE3<ONEFB>::NFB],
                                            h8 (&wr)[ring_slots(1, TailE3<ONEFB>::NFB)][1][2]) {
  job_pre<1, TailE3<ONEFB>::NFB, ST_E3>(c, c.wv, acc, wr);
}

// RESCUE_UP_TO: the workgroup re-evaluates its non-finite faces itself (encoder_heads_f16x2_rescue.h: vector ALUs, four faces at a
// time) when there are at most this many; with more it leaves them non-finite for the f32 re-evaluation launch that follows the
// strict-fast kernels (encoder_heads.hip, reeval_over).  64 = always here (NLML_MODE_F16X2, which has no f32 image in its blob).
// The helper waves' side of tail_stages<false, .., H1W8 = true> (eight-wave kernel, waves 4..7): they take blocks 4..7 of every head's H1 and
// otherwise only keep the workgroup's barrier count -- the SAME sequence of __syncthreads() as the main path below (three in E3..E5, five per
// head), or the workgroup deadlocks.
__device__ __forceinline__ void tail_helper_w8(const Ctx& c) {
  __syncthreads();   // E3 stored
  __syncthreads();   // E4 stored
  __syncthreads();   // E5 stored (latent image)
#pragma unroll 1
  for (int g = 0; g < 3; ++g) {
    Ctx cl = c;
    asm volatile("" : "+v"(cl.lane), "+v"(cl.f), "+v"(cl.h));
    f32x16 acc1h[1][2];
    h8 wr1h[ring_slots(1, 2)][1][2];
    h1_block_pre(cl, g, 4 + cl.wv, acc1h, wr1h);
    __syncthreads();   // H0_g stored
    h1_block_run(cl, g, 4 + cl.wv, acc1h, wr1h, O_GA, P_G128, S_G128);
    job_store<1, 2, ACT_RELU>(cl, acc1h, O_GB, P_G256, S_G256, 32 * (4 + cl.wv), 0, cl.hdr.inv_scale[ST_H1]);
    __syncthreads();   // H1_g stored
    __syncthreads();   // H2_g stored
    __syncthreads();   // H3_g stored
    __syncthreads();   // H4_g done
  }
}

template <bool ONEFB = false, int RESCUE_UP_TO = 64, bool H1W8 = false>
__device__ __forceinline__ void tail_stages(const Ctx& c, const Args& a, int64_t row0, f32x16 (&acc3)[1][TailE3<ONEFB>::NFB],
                                            h8 (&wr3)[ring_slots(1, TailE3<ONEFB>::NFB)][1][2], int fbsel = 0) {
  static_assert(!H1W8 || (!ONEFB && RESCUE_UP_TO == 0), "H1 over eight waves: the strict eight-wave kernel's 64-face tail only");
  if constexpr (H1W8) {
    if (c.helper) {
      tail_helper_w8(c);
      return;
    }
  }
  const int wv = c.wv;
  constexpr int NFB3 = TailE3<ONEFB>::NFB;
  const bool do4 = !ONEFB || wv < 2;                      // E4: neuron block wv&1, face block wv>>1 (ONEFB: waves 0,1 on block fbsel)
  const bool do5 = ONEFB ? wv == 0 : wv < 2;              // E5: wave = face block (ONEFB: wave 0 on block fbsel)
  const int nb4 = wv & 1, face4 = ONEFB ? 32 * fbsel : 32 * (wv >> 1);
  const int fb5 = ONEFB ? fbsel : wv;
  f32x16 acc4[1][1], acc5[2][1];
  h8 wr4[6][1][2], wr5[6][2][2];
  // ---- E3: 256 -> 128, ReLU
  job_run<1, NFB3, ST_E3>(c, wv, acc3, wr3, O_H3, P_H3, S_H3, 0, ONEFB ? 32 * fbsel : 0);
  // (fetches are unconditional, see below)
  job_store<1, NFB3, ACT_RELU>(c, acc3, O_H4, P_H4, S_H4, 32 * wv, ONEFB ? 32 * fbsel : 0, c.hdr.inv_scale[ST_E3],
                               fetch_hook<2 * NFB3, 1, 1, ST_E4>(c, nb4, acc4, wr4));
  __syncthreads();
  HXS(11);
  // ---- E4: 128 -> 64, Tanh
  if (do4) job_run<1, 1, ST_E4>(c, nb4, acc4, wr4, O_H4, P_H4, S_H4, 0, face4);
  // every wave fetches, also the ones that will not run the stage: an accumulator array defined under one `if` and used under
  // a later one becomes a loop-carried / partially defined value that the register allocator spills to scratch
  job_pre<2, 1, ST_E5>(c, 0, acc5, wr5);
  if (do4) job_store<1, 1, ACT_TANH>(c, acc4, O_H5, P_H5, S_H5, 32 * nb4, face4, c.hdr.inv_scale[ST_E4]);
  __syncthreads();
  // ---- E5: 64 -> 9, latent n = 3g+c on row 16g+c (2 blocks), other rows exact zeros
  if (do5) job_run<2, 1, ST_E5>(c, 0, acc5, wr5, O_H5, P_H5, S_H5, 0, 32 * fb5);
  if (do5) {
    const float inv = c.hdr.inv_scale[ST_E5];
#ifndef HX_STAMPS
    if (a.latent && row0 + 32 * fb5 + c.f < a.B) {   // f32 latent straight from the accumulators
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int rowi = 32 * nb + (q & 3) + 8 * (q >> 2) + 4 * c.h, g = rowi >> 4, cc = rowi & 15;
          if (g < 3 && cc < 3) a.latent[(row0 + 32 * fb5 + c.f) * NLML_LATENT + 3 * g + cc] = acc5[nb][0][q] * inv;
        }
    }
#endif
    job_store<2, 1, ACT_NONE, S_LAT>(c, acc5, O_LAT, P_LAT, S_LAT, 0, 32 * fb5, inv);
  }
  __syncthreads();
  HXS(12);
  // ---- heads.  Fused kernel (64 faces in the workgroup): ONE HEAD AT A TIME over both face blocks (heads_by_head below);
  // small-batch tail (ONEFB, 32 faces): the three heads together, the jobs a wave owns in lock step (gloop_run).
  if constexpr (!ONEFB) {
    heads_by_head<H1W8>(c, a, row0);
  } else {
  // ---- heads, one 32-face block at a time; the jobs a wave owns run together (gloop_run)
  const int fb_first = ONEFB ? fbsel : 0, fb_last = ONEFB ? fbsel : 1;
#pragma unroll 1
  for (int fb = fb_first; fb <= fb_last; ++fb) {
    int face0 = 32 * fb;
    // opaque copy of the lane coordinates: without it ~40 loop-invariant LDS addresses and weight pointers are hoisted out of
    // this loop, held in AGPRs and spilled to scratch around the H1 stage (which needs every register)
    Ctx cl = c;
    asm volatile("" : "+v"(cl.lane), "+v"(cl.f), "+v"(cl.h));
    // nothing prefetched is carried into or around the loop (the register allocator spills loop-carried rings): H0, the
    // cheapest stage to start cold (one K step), fetches its own operands here
    f32x16 accA[3][1][1];        // H0, then H2
    h8 wrA[4][3][1][2];
    f32x16 accB[3][2][1];        // H1
    h8 wrB[4][3][2][2];
    heads_pre<3, 1, ST_H0>(cl, wv * 3, accA, wrA);
    {  // H0: 3 -> 128 (K padded to 16 with zeros), ReLU
      constexpr int ST = ST_H0;
      const char* in[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) in[j] = cl.lds + O_LAT + ((face0 + cl.f) * S_LAT + 16 * ((wv * 3 + j) >> 2) + 8 * cl.h) * 2;
      heads_run<3, 1, ST>(cl, wv * 3, accA, wrA, in, P_LAT);
      heads_pre<3, 2, ST_H1>(cl, wv * 3, accB, wrB);
      const float inv = cl.hdr.inv_scale[ST];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wv * 3 + j;
        store_lds<1, 1, ACT_RELU>(accA[j], cl.lds + O_HA + (cl.f * S_HA + 128 * (job >> 2) + 32 * (job & 3)) * 2, cl.h, P_HA, 0, inv);
      }
    }
    __syncthreads();
    HXS(13 + 5 * fb);
    {  // H1: 128 -> 256, ReLU
      constexpr int ST = ST_H1;
      const char* in[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) in[j] = cl.lds + O_HA + (cl.f * S_HA + 128 * ((wv * 3 + j) >> 2) + 8 * cl.h) * 2;
      heads_run<3, 2, ST>(cl, wv * 3, accB, wrB, in, P_HA);
      heads_pre<3, 1, ST_H2>(cl, wv * 3, accA, wrA);
      const float inv = cl.hdr.inv_scale[ST];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wv * 3 + j;
        store_lds<2, 1, ACT_RELU>(accB[j], cl.lds + O_HB + (cl.f * S_HB + 256 * (job >> 2) + 64 * (job & 3)) * 2, cl.h, P_HB, 0, inv);
      }
    }
    __syncthreads();
    HXS(14 + 5 * fb);
    f32x16 accC[2][1][1];        // H3
    h8 wrC[4][2][1][2];
    {  // H2: 256 -> 128, ReLU
      constexpr int ST = ST_H2;
      const char* in[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) in[j] = cl.lds + O_HB + (cl.f * S_HB + 256 * ((wv * 3 + j) >> 2) + 8 * cl.h) * 2;
      heads_run<3, 1, ST>(cl, wv * 3, accA, wrA, in, P_HB);
      heads_pre<2, 1, ST_H3>(cl, (wv < 3 ? wv : 2) * 2, accC, wrC);   // wave 3 idles in H3/H4: it fetches wave 2's operands (see E5)
      const float inv = cl.hdr.inv_scale[ST];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wv * 3 + j;
        store_lds<1, 1, ACT_RELU>(accA[j], cl.lds + O_HC + (cl.f * S_HC + 128 * (job >> 2) + 32 * (job & 3)) * 2, cl.h, P_HC, 0, inv);
      }
    }
    __syncthreads();
    HXS(15 + 5 * fb);
    f32x16 accD[1][1];           // H4
    h8 wrD[6][1][2];
    job_pre<1, 1, ST_H4>(cl, wv < 3 ? wv : 2, accD, wrD);
    if (wv < 3) {  // H3: 128 -> 64, ReLU: waves 0..2 take the two blocks of head wv
      constexpr int ST = ST_H3;
      const char* in[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) in[j] = cl.lds + O_HC + (cl.f * S_HC + 128 * wv + 8 * cl.h) * 2;
      heads_run<2, 1, ST>(cl, wv * 2, accC, wrC, in, P_HC);
      const float inv = cl.hdr.inv_scale[ST];
#pragma unroll
      for (int j = 0; j < 2; ++j)
        store_lds<1, 1, ACT_RELU>(accC[j], cl.lds + O_HD + (cl.f * S_HD + 64 * wv + 32 * j) * 2, cl.h, P_HD, 0, inv);
    }
    __syncthreads();
    HXS(16 + 5 * fb);
    if (wv < 3)   // H4: 64 -> 1; neuron on accumulator row 0 = register 0 of lanes 0..31
      job_run<1, 1, ST_H4>(cl, wv, accD, wrD, O_HD, P_HD, S_HD, 64 * wv, 0);
    if (wv < 3) {
      if (cl.h == 0 && row0 + face0 + cl.f < a.B) {
        const float pose = accD[0][0][0] * cl.hdr.inv_scale[ST_H4];
        a.out[(row0 + face0 + cl.f) * 3 + wv] = pose;
        if (!__builtin_isfinite(pose)) *rescue_flag(cl.lds, face0 + cl.f) = 1;   // an activation left f16's range (or the input is NaN/Inf)
      }
    }
    __syncthreads();
    HXS(17 + 5 * fb);
  }
  }   // ONEFB heads
#if !defined(HX_STAMPS) && !defined(HX_NO_RESCUE)
  {  // faces whose pose is non-finite take the f32 slow path (encoder_heads_f16x2_rescue.h); normally none
    const int fl = ONEFB ? 32 * fbsel + c.f : c.lane;
    const bool mine = ONEFB ? c.h == 0 : true;
    const unsigned long long m = __ballot(mine && row0 + fl < a.B && *rescue_flag(c.lds, fl) != 0);
    const unsigned long long mask = ONEFB ? (m & 0xffffffffull) << (32 * fbsel) : m;
    if constexpr (RESCUE_UP_TO > 0) {   // (the strict kernels, RESCUE_UP_TO == 0, carry no slow-path code: their flagged faces go to the f32 re-evaluation launch)
      if (mask && __popcll(mask) <= RESCUE_UP_TO) rescue_tile(a.x, a.ldx, a.F, a.norm, a.blob, a.out, a.latent, c.lds, row0, mask);
    }
  }
#endif
}

}  // namespace hx
}  // namespace nlml
