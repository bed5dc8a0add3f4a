// encoder_heads_f16x2.hip -- K2 in SPLIT-F16 PARITY mode (NLML_MODE_F16X2).
//
// Same network, stages and jobs as the f32 parity kernel (encoder_heads.hip; reference:
// NLML_HPE_Model_Builder.py:33-53,76-92,115-126) and the same <=1e-4 degree bar, but the contraction runs on
// the f16 matrix cores: every f32 operand v is carried as two f16 pieces, v = hi + lo, hi = f16(v),
// lo = f16(v - hi) (22 significand bits), and a product is evaluated as
//
//     w*x  ~  w_hi*x_hi + w_hi*x_lo + w_lo*x_hi          (three v_mfma_f32_32x32x16_f16, f32 accumulate)
//
// The dropped w_lo*x_lo term is 2^-22 relative, the same order as f32 rounding; measured pose error against
// the f64 oracle is ~1e-5 degree, like the f32 kernel's (tests/test_gpu_parity.py, tools/probes/
// split_precision_study.py).  Three 32-cycle f16 MFMAs replace eight 64-cycle f32 MFMAs per 16 k values, so
// the matrix pipe has 5.3x less work for the same operand bytes (4 per weight, 4 per activation): this kernel
// is bound by the L2 -> CU weight stream like the bf16 kernel, not by the matrix cores.
//
// Range: activations above 65504 do not fit f16; hi becomes inf, lo -inf, and the pose of that face comes out
// NaN (never a silently wrong number).  f16 subnormal pieces are kept by the MFMA (tools/probes/
// mfma_f16_probe.hip), so small values lose nothing beyond an absolute 2^-25.
//
// Structure: 64-face tiles, 4 waves; layer 0 in two passes of 512 neurons interleaved with the two K halves
// of layer 1 (as in the f32 kernel: the 1024-wide layer-0 output of 64 faces is 256 KB in hi+lo f16);
// x is staged f32 -> (optional f64 IPD normalisation) -> hi/lo f16 through three rotating 32-column LDS slabs.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/nlml_hpe.h"
#include "abi_internal.h"
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
    if (prefetch) {
      const h8* wp = w + (size_t)sp * (NB * 2 * 64);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int p = 0; p < 2; ++p) wr[(r + D) % R][nb][p] = wp[(nb * 2 + p) * 64];
    }
    const int sxc = sx < K16 ? sx : K16 - 1;
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

__device__ __forceinline__ double div_ipd(double n, double d, double y) {
  const double q = n * y;
  const double r = fma(-q, d, n);
  return fma(r, y, q);
}

// ------------------------------------------------------------------------------------------
// One pass of layer 0: x[64,F] f32 -> (optional IPD normalisation in f64) -> hi/lo f16 -> three rotating LDS
// slabs of 32 columns; this wave computes 128 neurons (4 blocks, job 4*pass + wave) for both face blocks.
template <bool VEC4, bool NORM>
__device__ __forceinline__ void stage_e0_pass(const Ctx& c, const Args& a, int64_t row0, int tid, int pass,
                                              f32x16 (&acc)[4][2]) {
  constexpr int NB = 4, NFB = 2;
  const int F = a.F;
  const int nslab = (int)c.hdr->k8_e0 / XS_STEPS;   // even (pack.cpp)
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

  auto gload = [&](int s, f32x4 (&st)[2]) {
    s = s < nslab ? s : nslab - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
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
  auto lwrite = [&](int buf_off, f32x4 (&st)[2], bool real_slab) {
#pragma unroll
    for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(st[i]));
    if (NORM) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int t = (4 * i + e) % 3;
          const double rr = t == 0 ? ra : (t == 1 ? rb : rc);
          st[i][e] = (float)div_ipd((double)st[i][e] - rr, ipd, rcp);
        }
      // the f32 value must exist as such: without this fence hipcc 7.2 folds (f16)(f32)double into ONE f64 -> f16
      // conversion, done in ~20 integer instructions per element and rounded differently from the two-step path
#pragma unroll
      for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(st[i]));
      const double t0 = rc; rc = rb; rb = ra; ra = t0;   // next slab: columns + 32 => phase + 2
    }
    const unsigned m = real_slab ? 0x7fffffffu : 0u;
    h8 hi, lo;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = st[i][e];
        nzbits |= __float_as_uint(v) & m;
        const _Float16 hv = (_Float16)v;
        hi[4 * i + e] = hv;
        lo[4 * i + e] = (_Float16)(v - (float)hv);
      }
    char* d = c.lds + O_XS + buf_off + (srow * S_XS + scol) * 2;
    *reinterpret_cast<h8*>(d) = hi;
    *reinterpret_cast<h8*>(d + P_XS) = lo;
  };

  const int job = 4 * pass + c.wv;
  load_bias<NB, NFB>(acc, c.blob4 + c.hdr->b_off[ST_E0] + job * (NB * 8), c.h);
  const h8* w = c.blob8 + c.hdr->w_off[ST_E0] + (size_t)job * c.hdr->job_w16[ST_E0] + c.lane;

  // TWO staging register sets (8 floats per thread each), one per slab parity: slab s+2 is written to LDS at the
  // start of slab s from set[s & 1], which is refilled at once with the loads of slab s+4.  vmcnt counts in issue
  // order, so a set must be older than every weight load still wanted in flight when it is waited for: two slabs
  // (4 K steps, 32 weight loads) lie between its loads and its use, and in the prologue the sets are loaded BEFORE
  // the weight ring so that the loop header sees the same distance on entry as on the back edge.
  f32x4 set[2][2];
  constexpr int R0 = 4, D0 = R0 - 1;   // weight ring: K step ks in slot ks % 4 = 2*(slab & 1) + step of the slab
  static_assert(2 * XS_STEPS == R0, "two slabs == ring slots");
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
      for (int pp = 0; pp < 2; ++pp) wr[d][nb][pp] = w[((d * NB + nb) * 2 + pp) * 64];
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
#pragma unroll
    for (int kk = 0; kk < XS_STEPS; ++kk) {
      const int slot = 2 * PAR + kk;
      const h8* wp = w + (size_t)(s * XS_STEPS + kk + D0) * (NB * 2 * 64);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) wr[(slot + D0) % R0][nb][pp] = wp[(nb * 2 + pp) * 64];
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
          xr[(kk + 1) & 1][fb][pp] = (kk + 1 < XS_STEPS)
                                         ? *reinterpret_cast<const h8*>(xrow + pp * P_XS + fb * FB + 32 * (kk + 1))
                                         : *reinterpret_cast<const h8*>(xnext + pp * P_XS + fb * FB);
      if (kk == 0) {
        lwrite(o2, set[PAR], s + 2 < nslab);
        gload(s + 4, set[PAR]);
      }
      __builtin_amdgcn_sched_barrier(0);
      mma_step<NB, NFB>(acc, wr[slot], xr[kk & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    const int t0 = o0;   // rotate: (o0, o1, o2) <- (o1, o2, o0)
    o0 = o1; o1 = o2; o2 = t0;
  };
  for (int s = 0; s < nslab; s += 2) {
    slab(s, std::integral_constant<int, 0>{});
    slab(s + 1, std::integral_constant<int, 1>{});
  }
  if (pass == 0 && a.valid) {  // all-zero feature row == "no face" (FeatureExtractor.py:105-106); 4 lanes share a row
    const unsigned long long m = __ballot(nzbits != 0u);
    if ((tid & 3) == 0 && live) a.valid[row0 + srow] = ((m >> (c.lane & 60)) & 0xFull) ? 1 : 0;
  }
}

// ------------------------------------------------------------------------------------------
template <bool VEC4, bool NORM>
__global__ __launch_bounds__(256, 1) void encoder_heads_f16x2_kernel(Args a) {
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];

  const int tid = threadIdx.x;
  Ctx c;
  c.blob8 = reinterpret_cast<const h8*>(a.blob);
  c.blob4 = reinterpret_cast<const f32x4*>(a.blob);
  c.hdr = reinterpret_cast<const Header*>(a.blob);
  c.lds = lds;
  c.lane = tid & 63;
  c.f = c.lane & 31;
  c.h = c.lane >> 5;
  c.wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wv = c.wv;
  const int64_t row0 = (int64_t)blockIdx.x * TILE_FACES;

  {  // E0 (two passes of 512 neurons) interleaved with the two K halves of E1
    f32x16 acc1[4][2];
    load_bias<4, 2>(acc1, c.blob4 + c.hdr->b_off[ST_E1] + wv * (4 * 8), c.h);
    const h8* w1 = c.blob8 + c.hdr->w_off[ST_E1] + (size_t)wv * c.hdr->job_w16[ST_E1] + c.lane;
    const float inv0 = c.hdr->inv_scale[ST_E0];
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      {
        f32x16 acc0[4][2];
        stage_e0_pass<VEC4, NORM>(c, a, row0, tid, pass, acc0);
        // the last barrier of the slab loop also says: every wave is done reading H1H (previous pass's E1 half)
        job_store<4, 2, ACT_RELU>(c, acc0, O_H1H, P_H1H, S_H1H, 128 * wv, 0, inv0);
      }
      __syncthreads();
      kloop<4, 2, 32>(acc1, w1 + (size_t)pass * 32 * (4 * 2 * 64), c.lds + O_H1H + (c.f * S_H1H + 8 * c.h) * 2, P_H1H,
                      32 * S_H1H * 2);
      __syncthreads();   // H1H is free again (pass 0: for pass 1's store; pass 1: for H2)
    }
    job_store<4, 2, ACT_RELU>(c, acc1, O_H2, P_H2, S_H2, 128 * wv, 0, c.hdr->inv_scale[ST_E1]);
  }
  __syncthreads();
  {  // E2: 512 -> 256, ReLU; h3 overwrites h2 => barrier between the K loop and the store
    f32x16 acc[2][2];
    job_compute<2, 2, ST_E2>(c, wv, acc, O_H2, P_H2, S_H2, 0, 0);
    __syncthreads();
    job_store<2, 2, ACT_RELU>(c, acc, O_H3, P_H3, S_H3, 64 * wv, 0, c.hdr->inv_scale[ST_E2]);
  }
  __syncthreads();
  {  // E3: 256 -> 128, ReLU
    f32x16 acc[1][2];
    job_compute<1, 2, ST_E3>(c, wv, acc, O_H3, P_H3, S_H3, 0, 0);
    job_store<1, 2, ACT_RELU>(c, acc, O_H4, P_H4, S_H4, 32 * wv, 0, c.hdr->inv_scale[ST_E3]);
  }
  __syncthreads();
  {  // E4: 128 -> 64, Tanh; neuron block wv&1, face block wv>>1
    const int nb = wv & 1, face0 = 32 * (wv >> 1);
    f32x16 acc[1][1];
    job_compute<1, 1, ST_E4>(c, nb, acc, O_H4, P_H4, S_H4, 0, face0);
    job_store<1, 1, ACT_TANH>(c, acc, O_H5, P_H5, S_H5, 32 * nb, face0, c.hdr->inv_scale[ST_E4]);
  }
  __syncthreads();
  if (wv < 2) {  // E5: 64 -> 9, latent n = 3g+c on row 16g+c (2 blocks), other rows exact zeros; face block wv
    f32x16 acc[2][1];
    job_compute<2, 1, ST_E5>(c, 0, acc, O_H5, P_H5, S_H5, 0, 32 * wv);
    const float inv = c.hdr->inv_scale[ST_E5];
    if (a.latent && row0 + 32 * wv + c.f < a.B) {   // f32 latent straight from the accumulators
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int rowi = 32 * nb + (q & 3) + 8 * (q >> 2) + 4 * c.h, g = rowi >> 4, cc = rowi & 15;
          if (g < 3 && cc < 3) a.latent[(row0 + 32 * wv + c.f) * NLML_LATENT + 3 * g + cc] = acc[nb][0][q] * inv;
        }
    }
    job_store<2, 1, ACT_NONE, S_LAT>(c, acc, O_LAT, P_LAT, S_LAT, 0, 32 * wv, inv);
  }
  __syncthreads();
  // ---- heads, one 32-face block at a time; the jobs a wave owns run together (kloop_grouped)
#pragma unroll 1
  for (int fb = 0; fb < 2; ++fb) {
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
    if (wv < 3) {  // H4: 64 -> 1; neuron on accumulator row 0 = register 0 of lanes 0..31
      f32x16 acc[1][1];
      job_compute<1, 1, ST_H4>(c, wv, acc, O_HD, P_HD, S_HD, 64 * wv, 0);
      if (c.h == 0 && row0 + face0 + c.f < a.B) a.out[(row0 + face0 + c.f) * 3 + wv] = acc[0][0][0] * c.hdr->inv_scale[ST_H4];
    }
    __syncthreads();
  }
}

}  // namespace hx

int launch_encoder_heads_f16x2(const float* x, int64_t ldx, const float* raw, int normalize, int64_t B, int F,
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
  return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

}  // namespace nlml
