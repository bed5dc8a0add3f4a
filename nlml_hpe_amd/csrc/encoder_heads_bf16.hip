// encoder_heads_bf16.hip -- K2 in THROUGHPUT mode (NLML_MODE_BF16): bf16 storage of weights and
// activations, v_mfma_f32_32x32x16_bf16 with f32 accumulation.
//
// Same network, same stages and jobs as the f32 parity kernel (encoder_heads.hip; reference:
// NLML_HPE_Model_Builder.py:33-53,76-92,115-126), but NOT a parity path: bf16 operands put the pose
// ~0.1 degree from the reference -- the measured error is reported by the tests and by
// bench.py, never claimed as parity.
//
// What changes against the f32 kernel:
//   * a K step is 16: one 16-byte weight fragment per lane (8 bf16) feeds ONE MFMA per face block
//     (32 cycles) instead of four f32 MFMAs (256 cycles), so per byte of weights the matrix pipe has
//     8x less work: this kernel is bound by the L2 -> CU weight stream, not by the matrix cores;
//   * activations are bf16 in LDS, so layer 0's output for the 64 faces of a tile is 128 KB and fits:
//     layer 0 runs in ONE pass (8 neuron blocks per wave, 256 accumulator registers) and x is read once;
//   * accumulators -> activation -> bf16 (v_cvt_pk_bf16_f32) -> LDS; the last stage writes f32 poses.
#include <hip/hip_runtime.h>

#include "../../include/nlml_hpe.h"
#include "abi_internal.h"
#include "layout.h"

namespace nlml {
namespace bf {

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

// K loop over an LDS-resident bf16 image; K16 static, no runtime branch in the body (see the f32 kernel).
// `w`: this lane's first fragment of the first step; `in`: this lane's (face row of block 0, k = 8h).
template <int NB, int NFB, int K16>
__device__ __forceinline__ void kloop(f32x16 (&acc)[NB][NFB], const bf16x8* __restrict__ w, const __bf16* in,
                                      int fb_stride) {
  constexpr int R = (NB >= 8) ? 3 : ((NB * NFB >= 4) ? 4 : 8), D = R - 1;
  bf16x8 wr[R][NB], xr[R][NFB];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (d < K16) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) wr[d][nb] = w[(d * NB + nb) * 64];
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb) xr[d][fb] = *reinterpret_cast<const bf16x8*>(in + fb * fb_stride + 16 * d);
    }
  }
  auto step = [&](int r, int sp, bool prefetch) {
    if (prefetch) {
      const bf16x8* wp = w + (size_t)sp * (NB * 64);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) wr[(r + D) % R][nb] = wp[nb * 64];
      const int spx = sp < K16 ? sp : K16 - 1;
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb)
        xr[(r + D) % R][fb] = *reinterpret_cast<const bf16x8*>(in + fb * fb_stride + 16 * spx);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb)
        acc[nb][fb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[r][nb], xr[r][fb], acc[nb][fb], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  constexpr int GROUPS = K16 / R, TAIL = K16 % R;
  for (int g = 0; g < GROUPS; ++g) {
#pragma unroll
    for (int r = 0; r < R; ++r) step(r, g * R + r + D, true);
  }
#pragma unroll
  for (int r = 0; r < TAIL; ++r) step(r, 0, false);
}

// Grouped form for the heads: NJ jobs (own input each) in lock step through one ring (32-face block).
template <int NJ, int NB, int K16>
__device__ __forceinline__ void kloop_grouped(f32x16 (&acc)[NJ][NB][1], const bf16x8* __restrict__ w0,
                                              size_t job_stride, const __bf16* const (&in)[NJ]) {
  constexpr int R = (NJ * NB >= 6) ? 3 : 4, D = R - 1;
  bf16x8 wr[R][NJ][NB], xr[R][NJ];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (d < K16) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) wr[d][j][nb] = w0[j * job_stride + (d * NB + nb) * 64];
        xr[d][j] = *reinterpret_cast<const bf16x8*>(in[j] + 16 * d);
      }
    }
  }
  auto step = [&](int r, int sp, bool prefetch) {
    if (prefetch) {
      const int spx = sp < K16 ? sp : K16 - 1;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) wr[(r + D) % R][j][nb] = w0[j * job_stride + ((size_t)sp * NB + nb) * 64];
        xr[(r + D) % R][j] = *reinterpret_cast<const bf16x8*>(in[j] + 16 * spx);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        acc[j][nb][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[r][j][nb], xr[r][j], acc[j][nb][0], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  constexpr int GROUPS = K16 / R, TAIL = K16 % R;
  for (int g = 0; g < GROUPS; ++g) {
#pragma unroll
    for (int r = 0; r < R; ++r) step(r, g * R + r + D, true);
  }
#pragma unroll
  for (int r = 0; r < TAIL; ++r) step(r, 0, false);
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

template <int NB, int NFB, int STAGE>
__device__ __forceinline__ void job_compute(const Ctx& c, int job, f32x16 (&acc)[NB][NFB], const __bf16* in_img,
                                            int in_stride, int in_col, int face0) {
  static_assert(kStages[STAGE].nb == NB, "job shape");
  load_bias<NB, NFB>(acc, c.blob4 + c.hdr->b_off[STAGE] + job * (NB * 8), c.h);
  const bf16x8* w = c.blob8 + c.hdr->w_off[STAGE] + (size_t)job * c.hdr->job_w16[STAGE] + c.lane;
  kloop<NB, NFB, kStages[STAGE].k8>(acc, w, in_img + (face0 + c.f) * in_stride + in_col + 8 * c.h, 32 * in_stride);
}

template <int NB, int NFB, int ACT>
__device__ __forceinline__ void job_store(const Ctx& c, const f32x16 (&acc)[NB][NFB], __bf16* out_img, int out_stride,
                                          int out_col, int face0) {
  store_lds<NB, NFB, ACT>(acc, out_img + (face0 + c.f) * out_stride + out_col + 4 * c.h, 32 * out_stride);
}

__device__ __forceinline__ double div_ipd(double n, double d, double y) {
  const double q = n * y;
  const double r = fma(-q, d, n);
  return fma(r, y, q);
}

// ------------------------------------------------------------------------------------------
// Layer 0 (single pass): x[64,F] f32 -> (optional IPD normalisation in f64) -> bf16 -> three rotating LDS
// slabs of 64 columns; this wave computes 256 neurons (8 blocks) for both face blocks.
template <bool VEC4, bool NORM>
__device__ __forceinline__ void stage_e0(const Ctx& c, const Args& a, int64_t row0, int tid, f32x16 (&acc)[8][2]) {
  constexpr int NB = 8, NFB = 2;
  const int F = a.F;
  const int nslab = (int)c.hdr->k8_e0 / XS_STEPS;   // even (pack.cpp)
  constexpr int SLAB_BYTES = 64 * S_XS * 2;

  // staging role: row srow (0..63), 16 consecutive columns scol..scol+15 of every slab
  const int srow = tid >> 2, scol = (tid & 3) * 16;
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

  auto gload = [&](int s, f32x4 (&st)[4]) {
    s = s < nslab ? s : nslab - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = s * XS_COLS + scol + 4 * i;
      if (VEC4) {
        const int kc = k < F ? k : (NORM ? k - 12 * ((k - F + 15) / 12) : F - 4);   // phase-preserving clamp
        st[i] = *reinterpret_cast<const f32x4*>(p + kc);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) st[i][e] = p[k + e < F ? k + e : F - 1];
      }
    }
  };
  auto lwrite = [&](int buf_off, f32x4 (&st)[4], bool real_slab) {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(st[i]));
    if (NORM) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int t = (4 * i + e) % 3;
          const double rr = t == 0 ? ra : (t == 1 ? rb : rc);
          st[i][e] = (float)div_ipd((double)st[i][e] - rr, ipd, rcp);
        }
      // keep the f32 value: otherwise (bf16)(f32)double may be folded into one software f64 -> bf16 conversion
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(st[i]));
      const double t0 = ra; ra = rb; rb = rc; rc = t0;   // next slab: columns + 64 => phase + 1
    }
    const unsigned m = real_slab ? 0x7fffffffu : 0u;
    bf16x8 lo, hi;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        nzbits |= __float_as_uint(st[i][e]) & m;
        if (i < 2) lo[4 * i + e] = (__bf16)st[i][e]; else hi[4 * (i - 2) + e] = (__bf16)st[i][e];
      }
    __bf16* d = reinterpret_cast<__bf16*>(c.lds + O_XS + buf_off) + srow * S_XS + scol;
    *reinterpret_cast<bf16x8*>(d) = lo;
    *reinterpret_cast<bf16x8*>(d + 8) = hi;
  };

  load_bias<NB, NFB>(acc, c.blob4 + c.hdr->b_off[ST_E0] + c.wv * (NB * 8), c.h);
  const bf16x8* w = c.blob8 + c.hdr->w_off[ST_E0] + (size_t)c.wv * c.hdr->job_w16[ST_E0] + c.lane;

  // ONE staging register set (16 floats per thread): slab s+2 is written in the middle of slab s and the set is
  // refilled at once with the loads of slab s+3, which then have a whole slab (~9k cycles, weight-stream
  // bound) to arrive.  Two sets pushed the NORM variant over the register file (spills into the hot loop).
  f32x4 set[4];
  constexpr int R0 = 4, D0 = R0 - 1;   // weight ring: K step ks in slot ks % 4; a slab holds 4 steps
  static_assert(XS_STEPS == R0, "slab steps == ring slots");
  bf16x8 wr[R0][NB];
#pragma unroll
  for (int d = 0; d < D0; ++d)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) wr[d][nb] = w[(d * NB + nb) * 64];
  gload(0, set);
  lwrite(0, set, true);
  gload(1, set);
  lwrite(SLAB_BYTES, set, true);
  gload(2, set);
  __syncthreads();

  const int lane_off = (c.f * S_XS + 8 * c.h) * 2;   // bytes
  bf16x8 xr[2][NFB];
#pragma unroll
  for (int fb = 0; fb < NFB; ++fb) xr[0][fb] = *reinterpret_cast<const bf16x8*>(c.lds + O_XS + lane_off + fb * (32 * S_XS * 2));

  int o0 = 0, o1 = SLAB_BYTES, o2 = 2 * SLAB_BYTES;   // buffers of slabs s, s+1, s+2
  for (int s = 0; s < nslab; ++s) {
    const char* xrow = c.lds + O_XS + o0 + lane_off;
    const char* xnext = c.lds + O_XS + o1 + lane_off;
#pragma unroll
    for (int kk = 0; kk < XS_STEPS; ++kk) {
      const bf16x8* wp = w + (size_t)(s * XS_STEPS + kk + D0) * (NB * 64);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) wr[(kk + D0) % R0][nb] = wp[nb * 64];
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb)
        xr[(kk + 1) & 1][fb] = (kk + 1 < XS_STEPS)
                                   ? *reinterpret_cast<const bf16x8*>(xrow + fb * (32 * S_XS * 2) + 32 * (kk + 1))
                                   : *reinterpret_cast<const bf16x8*>(xnext + fb * (32 * S_XS * 2));
      if (kk == 1) {
        lwrite(o2, set, s + 2 < nslab);
        gload(s + 3, set);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb)
          acc[nb][fb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[kk % R0][nb], xr[kk & 1][fb], acc[nb][fb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    const int t0 = o0;   // rotate: (o0, o1, o2) <- (o1, o2, o0)
    o0 = o1; o1 = o2; o2 = t0;
  }
  if (a.valid) {  // all-zero feature row == "no face" (FeatureExtractor.py:105-106); 4 lanes share a row
    const unsigned long long m = __ballot(nzbits != 0u);
    if ((tid & 3) == 0 && live) a.valid[row0 + srow] = ((m >> (c.lane & 60)) & 0xFull) ? 1 : 0;
  }
}

// ------------------------------------------------------------------------------------------
template <bool VEC4, bool NORM>
__global__ __launch_bounds__(256, 1) void encoder_heads_bf16_kernel(Args a) {
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
  c.wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wv = c.wv;
  const int64_t row0 = (int64_t)blockIdx.x * TILE_FACES;

  {  // E0: F -> 1024, ReLU (one pass; wave wv owns neurons 256*wv .. +255)
    f32x16 acc[8][2];
    stage_e0<VEC4, NORM>(c, a, row0, tid, acc);
    job_store<8, 2, ACT_RELU>(c, acc, img(c, O_H1), S_H1, 256 * wv, 0);
  }
  __syncthreads();
  {  // E1: 1024 -> 512, ReLU; h2 overwrites h1 => barrier between the K loop and the store
    f32x16 acc[4][2];
    job_compute<4, 2, ST_E1>(c, wv, acc, img(c, O_H1), S_H1, 0, 0);
    __syncthreads();
    job_store<4, 2, ACT_RELU>(c, acc, img(c, O_H2), S_H2, 128 * wv, 0);
  }
  __syncthreads();
  {  // E2: 512 -> 256, ReLU
    f32x16 acc[2][2];
    job_compute<2, 2, ST_E2>(c, wv, acc, img(c, O_H2), S_H2, 0, 0);
    job_store<2, 2, ACT_RELU>(c, acc, img(c, O_H3), S_H3, 64 * wv, 0);
  }
  __syncthreads();
  {  // E3: 256 -> 128, ReLU
    f32x16 acc[1][2];
    job_compute<1, 2, ST_E3>(c, wv, acc, img(c, O_H3), S_H3, 0, 0);
    job_store<1, 2, ACT_RELU>(c, acc, img(c, O_H4), S_H4, 32 * wv, 0);
  }
  __syncthreads();
  {  // E4: 128 -> 64, Tanh; neuron block wv&1, face block wv>>1
    const int nb = wv & 1, face0 = 32 * (wv >> 1);
    f32x16 acc[1][1];
    job_compute<1, 1, ST_E4>(c, nb, acc, img(c, O_H4), S_H4, 0, face0);
    job_store<1, 1, ACT_TANH>(c, acc, img(c, O_H5), S_H5, 32 * nb, face0);
  }
  __syncthreads();
  if (wv < 2) {  // E5: 64 -> 9, latent n = 3g+c on row 16g+c (2 blocks), other rows exact zeros; face block wv
    f32x16 acc[2][1];
    job_compute<2, 1, ST_E5>(c, 0, acc, img(c, O_H5), S_H5, 0, 32 * wv);
    if (a.latent && row0 + 32 * wv + c.f < a.B) {   // f32 latent straight from the accumulators
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int rowi = 32 * nb + (q & 3) + 8 * (q >> 2) + 4 * c.h, g = rowi >> 4, cc = rowi & 15;
          if (g < 3 && cc < 3) a.latent[(row0 + 32 * wv + c.f) * NLML_LATENT + 3 * g + cc] = acc[nb][0][q];
        }
    }
    job_store<2, 1, ACT_NONE>(c, acc, img(c, O_LAT), S_LAT, 0, 32 * wv);
  }
  __syncthreads();
  // ---- heads, one 32-face block at a time; the jobs a wave owns run together (kloop_grouped)
#pragma unroll 1
  for (int fb = 0; fb < 2; ++fb) {
    const int face0 = 32 * fb;
    {  // H0: 3 -> 128 (K padded to 16 with zeros), ReLU
      constexpr int ST = ST_H0;
      f32x16 acc[3][1][1];
      const __bf16* in[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wv * 3 + j;
        load_bias<1, 1>(acc[j], c.blob4 + c.hdr->b_off[ST] + job * 8, c.h);
        in[j] = img(c, O_LAT) + (face0 + c.f) * S_LAT + 16 * (job >> 2) + 8 * c.h;
      }
      kloop_grouped<3, 1, kStages[ST].k8>(acc, c.blob8 + c.hdr->w_off[ST] + (size_t)(wv * 3) * c.hdr->job_w16[ST] + c.lane,
                                          c.hdr->job_w16[ST], in);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wv * 3 + j;
        store_lds<1, 1, ACT_RELU>(acc[j], img(c, O_HA) + c.f * S_HA + 128 * (job >> 2) + 32 * (job & 3) + 4 * c.h, 0);
      }
    }
    __syncthreads();
    {  // H1: 128 -> 256, ReLU
      constexpr int ST = ST_H1;
      f32x16 acc[3][2][1];
      const __bf16* in[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wv * 3 + j;
        load_bias<2, 1>(acc[j], c.blob4 + c.hdr->b_off[ST] + job * 16, c.h);
        in[j] = img(c, O_HA) + c.f * S_HA + 128 * (job >> 2) + 8 * c.h;
      }
      kloop_grouped<3, 2, kStages[ST].k8>(acc, c.blob8 + c.hdr->w_off[ST] + (size_t)(wv * 3) * c.hdr->job_w16[ST] + c.lane,
                                          c.hdr->job_w16[ST], in);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wv * 3 + j;
        store_lds<2, 1, ACT_RELU>(acc[j], img(c, O_HB) + c.f * S_HB + 256 * (job >> 2) + 64 * (job & 3) + 4 * c.h, 0);
      }
    }
    __syncthreads();
    {  // H2: 256 -> 128, ReLU
      constexpr int ST = ST_H2;
      f32x16 acc[3][1][1];
      const __bf16* in[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wv * 3 + j;
        load_bias<1, 1>(acc[j], c.blob4 + c.hdr->b_off[ST] + job * 8, c.h);
        in[j] = img(c, O_HB) + c.f * S_HB + 256 * (job >> 2) + 8 * c.h;
      }
      kloop_grouped<3, 1, kStages[ST].k8>(acc, c.blob8 + c.hdr->w_off[ST] + (size_t)(wv * 3) * c.hdr->job_w16[ST] + c.lane,
                                          c.hdr->job_w16[ST], in);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wv * 3 + j;
        store_lds<1, 1, ACT_RELU>(acc[j], img(c, O_HC) + c.f * S_HC + 128 * (job >> 2) + 32 * (job & 3) + 4 * c.h, 0);
      }
    }
    __syncthreads();
    if (wv < 3) {  // H3: 128 -> 64, ReLU: waves 0..2 take the two blocks of head wv
      constexpr int ST = ST_H3;
      f32x16 acc[2][1][1];
      const __bf16* in[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        load_bias<1, 1>(acc[j], c.blob4 + c.hdr->b_off[ST] + (wv * 2 + j) * 8, c.h);
        in[j] = img(c, O_HC) + c.f * S_HC + 128 * wv + 8 * c.h;
      }
      kloop_grouped<2, 1, kStages[ST].k8>(acc, c.blob8 + c.hdr->w_off[ST] + (size_t)(wv * 2) * c.hdr->job_w16[ST] + c.lane,
                                          c.hdr->job_w16[ST], in);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        store_lds<1, 1, ACT_RELU>(acc[j], img(c, O_HD) + c.f * S_HD + 64 * wv + 32 * j + 4 * c.h, 0);
    }
    __syncthreads();
    if (wv < 3) {  // H4: 64 -> 1; neuron on accumulator row 0 = register 0 of lanes 0..31
      f32x16 acc[1][1];
      job_compute<1, 1, ST_H4>(c, wv, acc, img(c, O_HD), S_HD, 64 * wv, 0);
      if (c.h == 0 && row0 + face0 + c.f < a.B) a.out[(row0 + face0 + c.f) * 3 + wv] = acc[0][0][0];
    }
    __syncthreads();
  }
}

}  // namespace bf

int launch_encoder_heads_bf16(const float* x, int64_t ldx, const float* raw, int normalize, int64_t B, int F,
                              const void* blob, float* out, float* latent, uint8_t* valid, void* stream) {
  if (B == 0) return 0;
  bf::Args a;
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
    if (vec4) hipLaunchKernelGGL((bf::encoder_heads_bf16_kernel<true, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((bf::encoder_heads_bf16_kernel<false, true>), grid, block, 0, st, a);
  } else {
    if (vec4) hipLaunchKernelGGL((bf::encoder_heads_bf16_kernel<true, false>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((bf::encoder_heads_bf16_kernel<false, false>), grid, block, 0, st, a);
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

}  // namespace nlml
