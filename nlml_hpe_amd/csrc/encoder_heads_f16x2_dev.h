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
};

template <int ACT>
__device__ __forceinline__ float activate(float v) {
  if (ACT == ACT_RELU) return v < 0.0f ? 0.0f : v;   // NaN-propagating like torch.relu
  if (ACT == ACT_TANH) return tanhf(v);
  return v;
}

// v -> (hi, lo) f16 pieces, four values at a time
__device__ __forceinline__ void split4(const float (&v)[4], h4& hi, h4& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hi[e] = (_Float16)v[e];
    lo[e] = (_Float16)(v[e] - (float)hi[e]);
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

// One K step of the 4x2 shape (24 MFMAs = 768 cycles) cut into six sub-groups of four MFMAs; everything else the step
// has to issue -- the eight weight loads of the step D ahead (gaps 0-3, one neuron block each), the LDS reads of the
// next step's x operands (gap 4) and whatever the caller drops into gap j through `between(j)` (layer 0: a piece of
// the x staging) -- goes into the gaps, so that no block of non-MFMA instructions idles the matrix pipe for its whole
// length (stage stamps: a 24-MFMA step with all of it in front ran at 55 % busy in layer 0).
template <typename XLoad, typename Between>
__device__ __forceinline__ void step_il(f32x16 (&acc)[4][2], const h8 (&wcur)[4][2], const h8 (&xcur)[2][2],
                                        h8 (&wnext)[4][2], const h8* __restrict__ wp, XLoad xload, Between between) {
#pragma unroll
  for (int g = 0; g < 6; ++g) {
    if (g < 4) {
      wnext[g][0] = wp[(g * 2 + 0) * 64];
      wnext[g][1] = wp[(g * 2 + 1) * 64];
    }
    if (g == 4) xload();
    between(g);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 4 * g; m < 4 * g + 4; ++m) {
      const int t = m / 8, nb = (m % 8) / 2, fb = m % 2;
      const int wp_ = t == 0 ? 1 : 0, xp_ = t == 1 ? 1 : 0;   // (lo,hi), (hi,lo), (hi,hi)
      acc[nb][fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wcur[nb][wp_], xcur[fb][xp_], acc[nb][fb], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// K loop over an LDS-resident hi/lo image; K16 static, no runtime branch in the body (see the f32 kernel).
// `w`: this lane's hi fragment of block 0, step 0 (fragment (step, nb, piece) at ((step*NB + nb)*2 + piece)*64);
// `in`: this lane's (face row of block 0, k = 8h) in the hi plane; the lo plane is `plane` bytes further.
template <int NB, int NFB, int K16>
__device__ __forceinline__ void kloop(f32x16 (&acc)[NB][NFB], const h8* __restrict__ w, const char* in, int plane,
                                      int fb_stride) {
  constexpr int R = (NB * NFB >= 4) ? 4 : 6, D = R - 1;
  h8 wr[R][NB][2], xr[2][NFB][2];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (d < K16) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int p = 0; p < 2; ++p) wr[d][nb][p] = w[((d * NB + nb) * 2 + p) * 64];
    }
  }
#pragma unroll
  for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
    for (int p = 0; p < 2; ++p) xr[0][fb][p] = *reinterpret_cast<const h8*>(in + p * plane + fb * fb_stride);
  auto step = [&](int r, int xs, int sp, int sx, bool prefetch) {
    const int sxc = sx < K16 ? sx : K16 - 1;
    if constexpr (NB == 4 && NFB == 2) {
      if (prefetch) {
        step_il(acc, wr[r], xr[xs], wr[(r + D) % R], w + (size_t)sp * (NB * 2 * 64),
                [&]() {
#pragma unroll
                  for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                      xr[xs ^ 1][fb][p] = *reinterpret_cast<const h8*>(in + p * plane + fb * fb_stride + 32 * sxc);
                },
                [](int) {});
        return;
      }
    }
    if (prefetch) {
      const h8* wp = w + (size_t)sp * (NB * 2 * 64);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int p = 0; p < 2; ++p) wr[(r + D) % R][nb][p] = wp[(nb * 2 + p) * 64];
    }
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        xr[xs ^ 1][fb][p] = *reinterpret_cast<const h8*>(in + p * plane + fb * fb_stride + 32 * sxc);
    __builtin_amdgcn_sched_barrier(0);
    mma_step<NB, NFB>(acc, wr[r], xr[xs]);
    __builtin_amdgcn_sched_barrier(0);
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

// Grouped form for the heads: NJ jobs (own input each) in lock step through one ring (32-face block).
template <int NJ, int NB, int K16>
__device__ __forceinline__ void kloop_grouped(f32x16 (&acc)[NJ][NB][1], const h8* __restrict__ w0, size_t job_stride,
                                              const char* const (&in)[NJ], int plane) {
  constexpr int R = 4, D = R - 1;
  h8 wr[R][NJ][NB][2], xr[2][NJ][1][2];
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
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int p = 0; p < 2; ++p) xr[0][j][0][p] = *reinterpret_cast<const h8*>(in[j] + p * plane);
  auto step = [&](int r, int xs, int sp, int sx, bool prefetch) {
    if (prefetch) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int p = 0; p < 2; ++p)
            wr[(r + D) % R][j][nb][p] = w0[j * job_stride + (((size_t)sp * NB + nb) * 2 + p) * 64];
    }
    const int sxc = sx < K16 ? sx : K16 - 1;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int p = 0; p < 2; ++p) xr[xs ^ 1][j][0][p] = *reinterpret_cast<const h8*>(in[j] + p * plane + 32 * sxc);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int wp = t == 0 ? 1 : 0, xp = t == 1 ? 1 : 0;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[j][nb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[r][j][nb][wp], xr[xs][j][0][xp], acc[j][nb][0], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  constexpr int GROUPS = K16 / R, TAIL = K16 % R;
  for (int g = 0; g < GROUPS; ++g) {
#pragma unroll
    for (int r = 0; r < R; ++r) step(r, r & 1, g * R + r + D, g * R + r + 1, true);
  }
#pragma unroll
  for (int r = 0; r < TAIL; ++r) step(r, r & 1, 0, GROUPS * R + r + 1, false);
}

// accumulators * inv -> activation -> hi/lo f16 -> LDS image [piece][face][neuron]; `out`: lane's hi-plane
// address of (face row, col0 + 4h); columns at or beyond MAXCOL are not written (latent image)
template <int NB, int NFB, int ACT, int MAXCOL = 1 << 30>
__device__ __forceinline__ void store_lds(const f32x16 (&acc)[NB][NFB], char* out, int plane, int fb_stride, float inv) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (32 * nb + 8 * q >= MAXCOL) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = activate<ACT>(acc[nb][fb][4 * q + e] * inv);
        h4 hi, lo;
        split4(v, hi, lo);
        char* d = out + fb * fb_stride + (32 * nb + 8 * q) * 2;
        *reinterpret_cast<h4*>(d) = hi;
        *reinterpret_cast<h4*>(d + plane) = lo;
      }
}

struct Ctx {
  const h8* blob8;
  const f32x4* blob4;
  const Header* hdr;
  char* lds;
  int lane, f, h, wv;
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

// bias + K loop of one job whose input image (hi plane at byte offset in_off, row stride in_stride f16) is in LDS
template <int NB, int NFB, int STAGE>
__device__ __forceinline__ void job_compute(const Ctx& c, int job, f32x16 (&acc)[NB][NFB], int in_off, int plane,
                                            int in_stride, int in_col, int face0) {
  static_assert(kStages[STAGE].nb == NB, "job shape");
  load_bias<NB, NFB>(acc, c.blob4 + c.hdr->b_off[STAGE] + job * (NB * 8), c.h);
  const h8* w = c.blob8 + c.hdr->w_off[STAGE] + (size_t)job * c.hdr->job_w16[STAGE] + c.lane;
  kloop<NB, NFB, kStages[STAGE].k8>(acc, w, c.lds + in_off + ((face0 + c.f) * in_stride + in_col + 8 * c.h) * 2, plane,
                                    32 * in_stride * 2);
}

template <int NB, int NFB, int ACT, int MAXCOL = 1 << 30>
__device__ __forceinline__ void job_store(const Ctx& c, const f32x16 (&acc)[NB][NFB], int out_off, int plane,
                                          int out_stride, int out_col, int face0, float inv) {
  store_lds<NB, NFB, ACT, MAXCOL>(acc, c.lds + out_off + ((face0 + c.f) * out_stride + out_col + 4 * c.h) * 2, plane,
                                  32 * out_stride * 2, inv);
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
#else
#define HXS(i) do { } while (0)
#define HXS_WALL(i) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------
// Layers E3 .. H4 of one 64-face tile (rows row0 .. row0+63): input = the H3 image (E2's output, hi/lo planes at O_H3),
// output = poses (and the latent) in global memory.  All 256 threads of the workgroup call it after a barrier.
// ONEFB = true (small-batch path): the workgroup handles only the 32-face block `fbsel` of the tile (another workgroup
// takes the other one): the same MFMAs per face block -- faces are MFMA columns -- in fewer sequential stages.
template <bool ONEFB = false>
__device__ __forceinline__ void tail_stages(const Ctx& c, const Args& a, int64_t row0, int fbsel = 0) {
  const int wv = c.wv;
  if constexpr (ONEFB) {  // E3: 256 -> 128, ReLU, this face block only
    f32x16 acc[1][1];
    job_compute<1, 1, ST_E3>(c, wv, acc, O_H3, P_H3, S_H3, 0, 32 * fbsel);
    job_store<1, 1, ACT_RELU>(c, acc, O_H4, P_H4, S_H4, 32 * wv, 32 * fbsel, c.hdr->inv_scale[ST_E3]);
  } else {  // E3: 256 -> 128, ReLU
    f32x16 acc[1][2];
    job_compute<1, 2, ST_E3>(c, wv, acc, O_H3, P_H3, S_H3, 0, 0);
    job_store<1, 2, ACT_RELU>(c, acc, O_H4, P_H4, S_H4, 32 * wv, 0, c.hdr->inv_scale[ST_E3]);
  }
  __syncthreads();
  HXS(11);
  if (!ONEFB || wv < 2) {  // E4: 128 -> 64, Tanh; neuron block wv&1, face block wv>>1 (ONEFB: waves 0,1 on block fbsel)
    const int nb = wv & 1, face0 = ONEFB ? 32 * fbsel : 32 * (wv >> 1);
    f32x16 acc[1][1];
    job_compute<1, 1, ST_E4>(c, nb, acc, O_H4, P_H4, S_H4, 0, face0);
    job_store<1, 1, ACT_TANH>(c, acc, O_H5, P_H5, S_H5, 32 * nb, face0, c.hdr->inv_scale[ST_E4]);
  }
  __syncthreads();
  const int fb5 = ONEFB ? fbsel : wv;   // E5's face block: wave wv of waves 0,1; ONEFB: wave 0 on block fbsel
  if (ONEFB ? wv == 0 : wv < 2) {  // E5: 64 -> 9, latent n = 3g+c on row 16g+c (2 blocks), other rows exact zeros
    f32x16 acc[2][1];
    job_compute<2, 1, ST_E5>(c, 0, acc, O_H5, P_H5, S_H5, 0, 32 * fb5);
    const float inv = c.hdr->inv_scale[ST_E5];
#ifndef HX_STAMPS
    if (a.latent && row0 + 32 * fb5 + c.f < a.B) {   // f32 latent straight from the accumulators
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int rowi = 32 * nb + (q & 3) + 8 * (q >> 2) + 4 * c.h, g = rowi >> 4, cc = rowi & 15;
          if (g < 3 && cc < 3) a.latent[(row0 + 32 * fb5 + c.f) * NLML_LATENT + 3 * g + cc] = acc[nb][0][q] * inv;
        }
    }
#endif
    job_store<2, 1, ACT_NONE, S_LAT>(c, acc, O_LAT, P_LAT, S_LAT, 0, 32 * fb5, inv);
  }
  __syncthreads();
  HXS(12);
  // ---- heads, one 32-face block at a time; the jobs a wave owns run together (kloop_grouped)
#pragma unroll 1
  for (int fb = ONEFB ? fbsel : 0; fb < (ONEFB ? fbsel + 1 : 2); ++fb) {
    const int face0 = 32 * fb;
    {  // H0: 3 -> 128 (K padded to 16 with zeros), ReLU
      constexpr int ST = ST_H0;
      f32x16 acc[3][1][1];
      const char* in[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wv * 3 + j;
        load_bias<1, 1>(acc[j], c.blob4 + c.hdr->b_off[ST] + job * 8, c.h);
        in[j] = c.lds + O_LAT + ((face0 + c.f) * S_LAT + 16 * (job >> 2) + 8 * c.h) * 2;
      }
      kloop_grouped<3, 1, kStages[ST].k8>(acc, c.blob8 + c.hdr->w_off[ST] + (size_t)(wv * 3) * c.hdr->job_w16[ST] + c.lane,
                                          c.hdr->job_w16[ST], in, P_LAT);
      const float inv = c.hdr->inv_scale[ST];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wv * 3 + j;
        store_lds<1, 1, ACT_RELU>(acc[j], c.lds + O_HA + (c.f * S_HA + 128 * (job >> 2) + 32 * (job & 3) + 4 * c.h) * 2, P_HA, 0, inv);
      }
    }
    __syncthreads();
    HXS(13 + 5 * fb);
    {  // H1: 128 -> 256, ReLU
      constexpr int ST = ST_H1;
      f32x16 acc[3][2][1];
      const char* in[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wv * 3 + j;
        load_bias<2, 1>(acc[j], c.blob4 + c.hdr->b_off[ST] + job * 16, c.h);
        in[j] = c.lds + O_HA + (c.f * S_HA + 128 * (job >> 2) + 8 * c.h) * 2;
      }
      kloop_grouped<3, 2, kStages[ST].k8>(acc, c.blob8 + c.hdr->w_off[ST] + (size_t)(wv * 3) * c.hdr->job_w16[ST] + c.lane,
                                          c.hdr->job_w16[ST], in, P_HA);
      const float inv = c.hdr->inv_scale[ST];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wv * 3 + j;
        store_lds<2, 1, ACT_RELU>(acc[j], c.lds + O_HB + (c.f * S_HB + 256 * (job >> 2) + 64 * (job & 3) + 4 * c.h) * 2, P_HB, 0, inv);
      }
    }
    __syncthreads();
    HXS(14 + 5 * fb);
    {  // H2: 256 -> 128, ReLU
      constexpr int ST = ST_H2;
      f32x16 acc[3][1][1];
      const char* in[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wv * 3 + j;
        load_bias<1, 1>(acc[j], c.blob4 + c.hdr->b_off[ST] + job * 8, c.h);
        in[j] = c.lds + O_HB + (c.f * S_HB + 256 * (job >> 2) + 8 * c.h) * 2;
      }
      kloop_grouped<3, 1, kStages[ST].k8>(acc, c.blob8 + c.hdr->w_off[ST] + (size_t)(wv * 3) * c.hdr->job_w16[ST] + c.lane,
                                          c.hdr->job_w16[ST], in, P_HB);
      const float inv = c.hdr->inv_scale[ST];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wv * 3 + j;
        store_lds<1, 1, ACT_RELU>(acc[j], c.lds + O_HC + (c.f * S_HC + 128 * (job >> 2) + 32 * (job & 3) + 4 * c.h) * 2, P_HC, 0, inv);
      }
    }
    __syncthreads();
    HXS(15 + 5 * fb);
    if (wv < 3) {  // H3: 128 -> 64, ReLU: waves 0..2 take the two blocks of head wv
      constexpr int ST = ST_H3;
      f32x16 acc[2][1][1];
      const char* in[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        load_bias<1, 1>(acc[j], c.blob4 + c.hdr->b_off[ST] + (wv * 2 + j) * 8, c.h);
        in[j] = c.lds + O_HC + (c.f * S_HC + 128 * wv + 8 * c.h) * 2;
      }
      kloop_grouped<2, 1, kStages[ST].k8>(acc, c.blob8 + c.hdr->w_off[ST] + (size_t)(wv * 2) * c.hdr->job_w16[ST] + c.lane,
                                          c.hdr->job_w16[ST], in, P_HC);
      const float inv = c.hdr->inv_scale[ST];
#pragma unroll
      for (int j = 0; j < 2; ++j)
        store_lds<1, 1, ACT_RELU>(acc[j], c.lds + O_HD + (c.f * S_HD + 64 * wv + 32 * j + 4 * c.h) * 2, P_HD, 0, inv);
    }
    __syncthreads();
    HXS(16 + 5 * fb);
    if (wv < 3) {  // H4: 64 -> 1; neuron on accumulator row 0 = register 0 of lanes 0..31
      f32x16 acc[1][1];
      job_compute<1, 1, ST_H4>(c, wv, acc, O_HD, P_HD, S_HD, 64 * wv, 0);
      if (c.h == 0 && row0 + face0 + c.f < a.B) {
        const float pose = acc[0][0][0] * c.hdr->inv_scale[ST_H4];
        a.out[(row0 + face0 + c.f) * 3 + wv] = pose;
        if (!__builtin_isfinite(pose)) *rescue_flag(c.lds, face0 + c.f) = 1;   // an activation left f16's range (or the input is NaN/Inf)
      }
    }
    __syncthreads();
    HXS(17 + 5 * fb);
  }
#if !defined(HX_STAMPS) && !defined(HX_NO_RESCUE)
  {  // faces whose pose is non-finite take the f32 slow path (encoder_heads_f16x2_rescue.h); normally none
    const int fl = ONEFB ? 32 * fbsel + c.f : c.lane;
    const bool mine = ONEFB ? c.h == 0 : true;
    const unsigned long long m = __ballot(mine && row0 + fl < a.B && *rescue_flag(c.lds, fl) != 0);
    const unsigned long long mask = ONEFB ? (m & 0xffffffffull) << (32 * fbsel) : m;
    if (mask) rescue_tile(a.x, a.ldx, a.F, a.norm, a.blob, a.out, a.latent, c.lds, row0, mask);
  }
#endif
}

}  // namespace hx
}  // namespace nlml
