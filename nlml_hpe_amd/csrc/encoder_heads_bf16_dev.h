// encoder_heads_bf16_dev.h -- device helpers of the bf16 throughput-mode kernel (encoder_heads_bf16_w8.hip: the fused eight-wave kernel; a
// 128-face-tile path built on them in round 5 measured slower and was deleted): operand types, the one-load-per-
// MFMA-slot K steps, the K loops over LDS images, the bf16 store, and the network's tail (E3, E4, E5 and the three heads on both 32-face
// blocks at once) as one function over a 64-face tile whose E2 output sits in the H3 LDS image.  512 threads.  NOT a parity path.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/nlml_hpe.h"
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
#ifndef BF8_KD
#define BF8_KD 3
#endif
  constexpr int R = 4, D = BF8_KD;
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
// Layers E3 .. H4 of one 64-face tile (rows row0 .. row0+63): input = the H3 image (E2's output at O_H3), output = poses (and the latent)
// in global memory.  All 512 threads of the workgroup call it after a barrier.
__device__ __forceinline__ void tail_stages_bf16(const Ctx& c, const Args& a, int64_t row0) {
  const int wv = c.wv;
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
}  // namespace nlml
