// encoder_heads.hip -- K2: fused LandmarkEncoder + 3 AnglePredictionNetwork heads, f32 mode: the strict parity mode.
//
// Replaces CombinedAnglePredictionModel.forward (NLML_HPE_Model_Builder.py:115-126):
//   encoder  Linear(F,1024) ReLU Linear(1024,512) ReLU Linear(512,256) ReLU Linear(256,128) ReLU
//            Linear(128,64) Tanh Linear(64,9)                                   (:33-53)
//   heads    3 x [Linear(3,128) ReLU Linear(128,256) ReLU Linear(256,128) ReLU
//                 Linear(128,64) ReLU Linear(64,1)]                             (:76-92)
// ~21 ATen launches per call in the reference; here ONE launch, activations never leave the CU.
//
// Shape of the computation on CDNA4 (gfx950):
//   * one workgroup (4 waves, one per SIMD) = one tile of 64 faces through the WHOLE network;
//   * every layer is D[neuron][face] += W[neuron][k] * act[k][face] on v_mfma_f32_32x32x2_f32
//     (exact f32: a k-ordered fmaf chain per accumulator), neurons on MFMA rows, faces on MFMA columns: the tile
//     is two column blocks of 32 faces that SHARE every weight fragment;
//   * waves split the NEURONS of a layer, so weights are private to a wave and stream
//     global -> VGPR in the pre-packed fragment order of layout.h (one coalesced 1-KiB
//     dwordx4 load per 32x8 weight block = 8 MFMAs, L2/MALL resident, prefetched through a
//     register ring several K steps ahead);
//   * activations are shared by the 4 waves and live in LDS as [face][k] rows (stride 4*odd
//     floats => conflict-free ds_read_b128 / ds_write_b128); each lane reads 16 B = the four
//     k values of its face for the four MFMAs of a K step;
//   * bias is the accumulator's initial value; ReLU/Tanh are applied on the way to LDS;
//   * layer 0's output for 64 faces (256 KB) does not fit the 160 KB LDS, so layers 0 and 1 are
//     interleaved in two passes: pass A computes neurons 0..511 of layer 0 into LDS and layer 1
//     accumulates over that K half; pass B does neurons 512..1023 in place and layer 1 finishes;
//   * layers 0 to 3 sum in BLOCKS of 128 k (a chain per block, block sums added in order in a second accumulator set:
//     fold_block): closer to the exact result than the reference's own GEMM (FX3c, tests/test_gpu_parity.py).  Layer 0 then
//     holds two sets of 128 registers, so layer 1's set is parked in LDS (the dead h1 half image) while layer 0 runs;
//   * layer 0 streams x through three rotating 64x32 LDS slabs (coalesced 128-B row segments; global
//     loads 3 slabs ahead, LDS write under the previous slab's MFMAs, first operands of the next slab
//     read before the barrier), optionally applying the IPD normalisation (FeatureExtractor.py:30-66)
//     in f64 on the way in, one element per MFMA gap, so normalised features never exist in HBM;
//   * E4, E5 and the heads (10 % of the FLOPs) run per 32-face block with the 32-face LDS images; the
//     jobs a wave owns in a head stage run in lock step through one weight ring (kloop_grouped);
//   * the hot loops contain no runtime branch, no exec-masked load and no use of a loaded value ahead of
//     the MFMAs it is prefetched for: any of these makes hipcc 7.2 emit s_waitcnt vmcnt(0) and drain the
//     prefetch ring every step ("What it took").
#include <hip/hip_runtime.h>

#include "../../include/nlml_hpe.h"
#include "abi_internal.h"
#include "layout.h"

namespace nlml {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };

struct EncArgs {
  const float* x;       // [B, ldx] features, or raw landmarks [B,1404] when norm != 0
  int64_t ldx;
  int64_t B;
  int F;
  int norm;             // 1: x is raw landmarks, apply IPD normalisation while staging
  const void* blob;
  float* out;           // [B,3]
  float* latent;        // [B,9] or null
  uint8_t* valid;       // [B] or null
  float* pre_tanh;      // [B,64] or null: E4 accumulators before the Tanh (test hook, DBG build only)
  unsigned long long* stamps;  // [tiles,4 waves,16] s_memtime at stage boundaries (DBG build only)
  int reeval_over;      // >= 0: RE-EVALUATION launch behind a split-f16 launch (NLML_MODE_F16X2S): a tile is computed only if more
                        // than this many of its faces hold a non-finite pose in `out`, and only those faces are written; -1: off
};

template <int ACT>
__device__ __forceinline__ float activate(float v) {
  if (ACT == ACT_RELU) return v < 0.0f ? 0.0f : v;   // NaN propagates like torch.relu (fmaxf would swallow it)
  if (ACT == ACT_TANH) return tanhf(v);
  return v;
}

// acc[nb][fb]: neuron block nb x face block fb.  The bias depends on the neuron only.
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

template <int NB, int NFB>
__device__ __forceinline__ void mfma_step(f32x16 (&acc)[NB][NFB], const f32x4 (&w)[NB], const f32x4 (&x)[NFB]) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb)
        acc[nb][fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[nb][j], x[fb][j], acc[nb][fb], 0, 0, 0);
}

// One K step with its prefetch folded in: the NB weight loads of a later step (into ring slot `wnext`) and
// the x fragment reads are issued one at a time in the gaps BETWEEN the four sub-steps' MFMA groups, each
// gap pinned by sched_barrier.  Issued as a block ahead of the 32 MFMAs they cost ~100 cycles of idle
// matrix pipe per step; one load per gap hides under the 64 cycles the previous MFMA is still executing.
// `between(j)` lets the caller drop extra work (layer 0's x staging) into gap j.
template <int NB, int NFB, typename XLoad, typename Between>
__device__ __forceinline__ void step_interleaved(f32x16 (&acc)[NB][NFB], const f32x4 (&wcur)[NB],
                                                 const f32x4 (&xcur)[NFB], f32x4 (&wnext)[NB],
                                                 const f32x4* __restrict__ wp, XLoad xload, Between between) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
      if ((nb * 4) / NB == j) wnext[nb] = wp[nb * 64];   // static: NB 4 -> one load per gap
    if (j == 0) xload();
    between(j);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb)
        acc[nb][fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[nb][j], xcur[fb][j], acc[nb][fb], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Prefetch ring.  The compiler, left alone, sinks the weight loads of step s+1 below the MFMAs of
// step s and then waits for them at once (34 % of wave cycles parked in s_waitcnt, profiles/r01):
// so the K loops are unrolled over a ring of R register buffers with the loads of step s+D
// (D = R-1) pinned ABOVE the MFMAs of step s by sched_barrier.  A K step is NB*NFB*256 cycles of
// MFMA; D steps must cover an L2 miss (Infinity-Cache round trip, ~2 us under load).
template <int NB, int NFB> struct Ring { static constexpr int R = (NB * NFB >= 8) ? 4 : 8; };

// K loop with the input image resident in LDS; K8 (steps of 8) is static, so the loop has NO runtime
// branch: a conditional inside the unrolled body makes the compiler lose the exact load count at the
// join and fall back to s_waitcnt vmcnt(0), which drains the whole ring every step.
// `w` points at this lane's first fragment of the FIRST K step to run, `in` at this lane's (face row of
// block 0, k-half) of the input image at that step; face block fb is fb_stride floats further.
// Weight loads run up to D steps past the last step (next job / tail pad of the blob: harmless).
template <int NB, int NFB, int K8>
__device__ __forceinline__ void kloop_lds(f32x16 (&acc)[NB][NFB], const f32x4* __restrict__ w,
                                          const float* in, int fb_stride) {
  constexpr int R = Ring<NB, NFB>::R, D = R - 1;
  f32x4 wr[R][NB];
  f32x4 xr[R][NFB];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (d < K8) {  // static
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) wr[d][nb] = w[(d * NB + nb) * 64];
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb) xr[d][fb] = *reinterpret_cast<const f32x4*>(in + fb * fb_stride + 8 * d);
    }
  }
  constexpr int GROUPS = K8 / R, TAIL = K8 % R;
  for (int g = 0; g < GROUPS; ++g) {
    const int s0 = g * R;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int sp = s0 + r + D;
      const int spx = sp < K8 ? sp : K8 - 1;   // the LDS read stays inside the image
      step_interleaved<NB, NFB>(
          acc, wr[r], xr[r], wr[(r + D) % R], w + (size_t)sp * (NB * 64),
          [&]() {
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb)
              xr[(r + D) % R][fb] = *reinterpret_cast<const f32x4*>(in + fb * fb_stride + 8 * spx);
          },
          [](int) {});
    }
  }
#pragma unroll
  for (int r = 0; r < TAIL; ++r) {  // static tail (only jobs shorter than the ring: H0)
    __builtin_amdgcn_sched_barrier(0);
    mfma_step<NB, NFB>(acc, wr[r], xr[r]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// K-BLOCKED SUMS (round 3).  One MFMA accumulator is a k-ordered f32 fma chain; over the 1404 / 1024 terms of layers 0 / 1 such a
// chain rounds 1404 / 1024 times against ever larger partial sums and ends ~1.6x further from the exact sum than the blocked
// sums of a CPU GEMM (FX3c: p50 2.7e-5 deg, 0.45 % of the faces beyond 1e-4 deg, against the reference's 1.7e-5 deg / none).  Layers
// 0 and 1 therefore sum in BLOCKS of 128 k: a chain runs over one block (the first from the bias, the others from +0.0), and the
// block sums are added up in block order in a second accumulator set, tot = ((0 + s_0) + s_1) + ...  With that the kernel is
// closer to the exact result than the reference itself in p50, p99 and max (tests/test_gpu_parity.py, FX3c).  Layers 2 and 3 take
// the same form (their second accumulator set costs nothing).  The C oracle's order 2 restates exactly this order; the pre-Tanh
// activations still agree bit for bit.
template <int NB, int NFB>
__device__ __forceinline__ void fold_block(f32x16 (&tot)[NB][NFB], f32x16 (&acc)[NB][NFB]) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) {
      tot[nb][fb] += acc[nb][fb];
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[nb][fb][q] = 0.0f;
    }
}

// kloop_lds with the chains cut every BLK8 steps (of 8 k): `acc` holds the running chain of the current block (the caller
// initialises it: bias or zeros), `tot` receives the block sums.  The weight ring runs through the block boundaries.
template <int NB, int NFB, int K8, int BLK8>
__device__ __forceinline__ void kloop_lds_blocked(f32x16 (&tot)[NB][NFB], f32x16 (&acc)[NB][NFB], const f32x4* __restrict__ w,
                                                  const float* in, int fb_stride) {
  constexpr int R = Ring<NB, NFB>::R, D = R - 1;
  static_assert(K8 % BLK8 == 0 && BLK8 % R == 0, "blocks are whole turns of the ring");
  f32x4 wr[R][NB];
  f32x4 xr[R][NFB];
#pragma unroll
  for (int d = 0; d < D; ++d) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) wr[d][nb] = w[(d * NB + nb) * 64];
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) xr[d][fb] = *reinterpret_cast<const f32x4*>(in + fb * fb_stride + 8 * d);
  }
#pragma unroll 1
  for (int ob = 0; ob < K8 / BLK8; ++ob) {
#pragma unroll 1
    for (int g = 0; g < BLK8 / R; ++g) {
      const int s0 = ob * BLK8 + g * R;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int sp = s0 + r + D;
        const int spx = sp < K8 ? sp : K8 - 1;   // the LDS read stays inside the image
        step_interleaved<NB, NFB>(
            acc, wr[r], xr[r], wr[(r + D) % R], w + (size_t)sp * (NB * 64),
            [&]() {
#pragma unroll
              for (int fb = 0; fb < NFB; ++fb)
                xr[(r + D) % R][fb] = *reinterpret_cast<const f32x4*>(in + fb * fb_stride + 8 * spx);
            },
            [](int) {});
      }
    }
    fold_block<NB, NFB>(tot, acc);
  }
}

// Accumulators -> activation -> LDS image [face][neuron]; `out` points at this lane's
// (face row of block 0, first neuron of the job + 4*h).
template <int NB, int NFB, int ACT>
__device__ __forceinline__ void store_lds(const f32x16 (&acc)[NB][NFB], float* out, int fb_stride) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v;
        v[0] = activate<ACT>(acc[nb][fb][4 * q + 0]);
        v[1] = activate<ACT>(acc[nb][fb][4 * q + 1]);
        v[2] = activate<ACT>(acc[nb][fb][4 * q + 2]);
        v[3] = activate<ACT>(acc[nb][fb][4 * q + 3]);
        *reinterpret_cast<f32x4*>(out + fb * fb_stride + 32 * nb + 8 * q) = v;
      }
}

// Grouped K loop for the heads: NJ independent jobs of NB neuron blocks each (one 32-face block), every
// job with ITS OWN input slice (a different head, or a different column range), walked in lock step
// through one weight ring.  The three jobs a wave owns in a head stage used to run one after the
// other, each paying its own ring fill (an exposed L2 round trip) for 16-32 short steps; together they
// pay it once and put NJ*NB*4 MFMAs behind every step.  Jobs j of a wave are adjacent in the blob:
// job j's fragments start job_stride (in float4) after job j-1's.
template <int NJ, int NB, int K8>
__device__ __forceinline__ void kloop_grouped(f32x16 (&acc)[NJ][NB][1], const f32x4* __restrict__ w0,
                                              size_t job_stride, const float* const (&in)[NJ]) {
  constexpr int NT = NJ * NB;
  constexpr int R = (NT >= 6) ? 4 : 8, D = R - 1;
  f32x4 wr[R][NJ][NB];
  f32x4 xr[R][NJ];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (d < K8) {  // static
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) wr[d][j][nb] = w0[j * job_stride + (d * NB + nb) * 64];
        xr[d][j] = *reinterpret_cast<const f32x4*>(in[j] + 8 * d);
      }
    }
  }
  auto step = [&](int r, int sp, bool prefetch) {
    const int spx = sp < K8 ? sp : K8 - 1;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      if (prefetch) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            if (((j * NB + nb) * 4) / NT == jj)   // static: spread the NT loads over the four gaps
              wr[(r + D) % R][j][nb] = w0[j * job_stride + ((size_t)sp * NB + nb) * 64];
        if (jj == 0) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) xr[(r + D) % R][j] = *reinterpret_cast<const f32x4*>(in[j] + 8 * spx);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          acc[j][nb][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[r][j][nb][jj], xr[r][j][jj], acc[j][nb][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  constexpr int GROUPS = K8 / R, TAIL = K8 % R;
  for (int g = 0; g < GROUPS; ++g) {
#pragma unroll
    for (int r = 0; r < R; ++r) step(r, g * R + r + D, true);
  }
#pragma unroll
  for (int r = 0; r < TAIL; ++r) step(r, 0, false);   // static tail (H0: a single step)
}

struct Ctx {
  const f32x4* blob4;
  const Header* hdr;
  float* lds;
  int lane, f, h, wv;
};

// Bias init + K loop of one job over an LDS image.  face0: first face row (0 or 32) of the job.
template <int NB, int NFB, int STAGE>
__device__ __forceinline__ void job_compute(const Ctx& c, int job, f32x16 (&acc)[NB][NFB],
                                            const float* in_img, int in_stride, int in_col, int face0) {
  static_assert(kStages[STAGE].nb == NB, "job shape");
  load_bias<NB, NFB>(acc, c.blob4 + c.hdr->b_off[STAGE] + job * (NB * 8), c.h);
  const f32x4* w = c.blob4 + c.hdr->w_off[STAGE] + (size_t)job * c.hdr->job_w16[STAGE] + c.lane;
  kloop_lds<NB, NFB, kStages[STAGE].k8>(acc, w, in_img + (face0 + c.f) * in_stride + in_col + 4 * c.h, 32 * in_stride);
}

// The same with the K-blocked sum of layers 0 and 1 (fold_block): chains of 128 k, block sums in `tot`.  Layers 2 and 3 (K = 512,
// 256) hold few accumulators, so their second set is free; they take the blocked form too (FX3c: p50 1.55e-5 -> 1.3e-5 deg).
template <int NB, int NFB, int STAGE>
__device__ __forceinline__ void job_compute_blocked(const Ctx& c, int job, f32x16 (&tot)[NB][NFB],
                                                    const float* in_img, int in_stride, int in_col, int face0) {
  static_assert(kStages[STAGE].nb == NB, "job shape");
  f32x16 acc[NB][NFB];
  load_bias<NB, NFB>(acc, c.blob4 + c.hdr->b_off[STAGE] + job * (NB * 8), c.h);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
      for (int q = 0; q < 16; ++q) tot[nb][fb][q] = 0.0f;
  const f32x4* w = c.blob4 + c.hdr->w_off[STAGE] + (size_t)job * c.hdr->job_w16[STAGE] + c.lane;
  kloop_lds_blocked<NB, NFB, kStages[STAGE].k8, 16>(tot, acc, w, in_img + (face0 + c.f) * in_stride + in_col + 4 * c.h, 32 * in_stride);
}

template <int NB, int NFB, int ACT>
__device__ __forceinline__ void job_store(const Ctx& c, const f32x16 (&acc)[NB][NFB], float* out_img,
                                          int out_stride, int out_col, int face0) {
  store_lds<NB, NFB, ACT>(acc, out_img + (face0 + c.f) * out_stride + out_col + 4 * c.h, 32 * out_stride);
}

// ------------------------------------------------------------------------------------------
// Layer 0, one pass: x[64,F] streamed through LDS slabs of 32 columns; this wave computes the 128
// neurons of job `job` (4 blocks) for both face blocks.
// n / ipd for the IPD normalisation, correctly rounded in f64 from a once-per-row reciprocal (Markstein:
// q = n*y, r = n - q*d exactly by fma, q' = q + r*y with y = RN(1/d)): 3 multiply-adds per element instead
// of a ~35-instruction IEEE division sequence, and the value is rounded to f32 afterwards anyway.
__device__ __forceinline__ double div_ipd(double n, double d, double y) {
  const double q = n * y;
  const double r = fma(-q, d, n);
  return fma(r, y, q);
}

// ipd = ||lm[33] - lm[263]||_2 in f64 (== np.linalg.norm: sqrt of an fma-chained ddot), 1e-6 if exactly 0
__device__ __forceinline__ void ipd_of_row(const float* p, double& ipd, double& rcp) {
  const double dx = (double)p[99] - (double)p[789];
  const double dy = (double)p[100] - (double)p[790];
  const double dz = (double)p[101] - (double)p[791];
  double d = sqrt(fma(dz, dz, fma(dy, dy, dx * dx)));
  if (d == 0.0) d = 1e-6;
  ipd = d;
  rcp = 1.0 / d;
}

struct E0Stager {
  const float *p0, *p1;       // this thread's two rows (srow, srow + 32)
  // scalar members only: arrays inside this struct end up in scratch memory (private stack) on hipcc 7.2,
  // and scratch loads share the vector-memory queue with the weight prefetch
  double ipd0, ipd1, rcp0, rcp1;
  double r0a, r0b, r0c, r1a, r1b, r1c;   // row 0 / row 1 reference coordinates, ROTATED to the column phase
                                         // of the slab being written: element e of the thread's 4 columns
                                         // uses slot e % 3 (a, b, c)
  int srow, scol;
  bool live0, live1;
  unsigned nzbits0, nzbits1;   // OR of the magnitude bits of every staged value of the row
};

template <bool NORM, int NFB>
__device__ __forceinline__ void e0_stager_init(E0Stager& g, const EncArgs& a, int64_t row0, int tid) {
  g.srow = tid >> 3;
  g.scol = (tid & 7) * 4;
  int64_t r0 = row0 + g.srow, r1 = row0 + g.srow + 32;
  g.live0 = r0 < a.B;
  g.live1 = NFB == 2 && r1 < a.B;   // a 32-face tile has no second row set
  r0 = g.live0 ? r0 : a.B - 1;
  r1 = g.live1 ? r1 : a.B - 1;
  g.p0 = a.x + r0 * a.ldx;
  g.p1 = a.x + r1 * a.ldx;
  g.nzbits0 = g.nzbits1 = 0u;
  g.ipd0 = g.ipd1 = 1.0;
  g.rcp0 = g.rcp1 = 1.0;
  g.r0a = g.r0b = g.r0c = g.r1a = g.r1b = g.r1c = 0.0;
  if (NORM) {  // IPD normalisation constants of the two rows (FeatureExtractor.py:38-48,85-86), f64
    ipd_of_row(g.p0, g.ipd0, g.rcp0);
    if (NFB == 2) ipd_of_row(g.p1, g.ipd1, g.rcp1);
  }
}

template <bool VEC4, bool NORM, int NFB>
__device__ __forceinline__ void stage_e0_pass(const Ctx& c, const EncArgs& a, E0Stager& g, int job,
                                              f32x16 (&tot)[4][NFB]) {
  constexpr int NB = 4;
  f32x16 acc[NB][NFB];   // the running chain of the current 128-k block; `tot` collects the block sums (see fold_block)
  float* xs = c.lds + O_XS;
  const int F = a.F;
  const int nslab = (int)c.hdr->k8_e0 / XS_STEPS;   // even: k8_e0 is a multiple of 2*XS_STEPS (pack.cpp)

  // x staging, software-pipelined over THREE LDS slab buffers (slab j lives in buffer j % 3):
  //   top of slab s     issue the global loads of slab s+3 into one of two register sets;
  //   middle of slab s  write slab s+2 (loaded during slab s-1) to LDS, under the MFMAs of steps 2,3;
  //   end of slab s     read the first x fragments of slab s+1 (written during slab s-1, published by
  //                     the barrier that ended slab s-1), then the barrier.
  // So a load has ~1.5 slabs (12k cycles) to arrive, nothing that depends on it sits before MFMAs,
  // and after the barrier the next slab's operands are already in registers.
  // Loads are UNCONDITIONAL (clamped address; the zero padding is a select at write time): an
  // exec-masked load makes the compiler wait vmcnt(0) around it and drain the weight ring.
  auto gload = [&](int s, f32x4 (&st)[2]) {
    s = s < nslab ? s : nslab - 1;
    const int k = s * XS_COLS + g.scol;
    if (VEC4) {
      // beyond F: re-read real columns of the same row; with NORM step back by a multiple of 12 columns so
      // the (x, y, z) phase -- hence the normalised value, hence the all-zero test -- matches column k
      const int kc = k < F ? k : (NORM ? k - 12 * ((k - F + 15) / 12) : F - 4);
      st[0] = *reinterpret_cast<const f32x4*>(g.p0 + kc);
      if (NFB == 2) st[1] = *reinterpret_cast<const f32x4*>(g.p1 + kc);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int kc = k + e < F ? k + e : F - 1;
        st[0][e] = g.p0[kc];
        if (NFB == 2) st[1][e] = g.p1[kc];
      }
    }
  };
  // Columns >= F need no zeroing: their weights are zero in the blob (pack.cpp pads K), the clamped loads
  // return finite values of the same row, and 0 * finite contributes exactly 0.  The "row is all zero"
  // test is an OR of the magnitude bits (one v_and_or per value, no compare chains); clamped duplicates
  // repeat real columns, and the OR is idempotent.
  // The staging of one slab is cut into pieces that each fit the shadow of one MFMA (64 cycles) and are
  // dropped into consecutive gaps of the K steps: 8 x normalise-one-element (NORM only), then the write.
  auto lw_begin = [&](f32x4 (&st)[2]) {
    asm volatile("" : "+v"(st[0]));   // consumers of the loaded values stay where written
    if (NFB == 2) asm volatile("" : "+v"(st[1]));
  };
  // r0a..r1c: the rows' reference coordinates in the order this thread's 4 columns of the CURRENT write
  // slab need them (element e uses slot e % 3); lw_rotate() steps them to the next slab (a slab is 32
  // columns, 32 mod 3 = 2), so no modulo or select sits in the per-element piece.
  auto lw_norm = [&](f32x4 (&st)[2], int row, int e) {   // row, e static
    const int t = e % 3;
    const double r = row ? (t == 0 ? g.r1a : (t == 1 ? g.r1b : g.r1c)) : (t == 0 ? g.r0a : (t == 1 ? g.r0b : g.r0c));
    st[row][e] = (float)div_ipd((double)st[row][e] - r, row ? g.ipd1 : g.ipd0, row ? g.rcp1 : g.rcp0);
  };
  auto lw_rotate = [&]() {   // (a, b, c) <- (c, a, b): column offset +32 == +2 (mod 3)
    const double a0 = g.r0a, b0 = g.r0b, a1 = g.r1a, b1 = g.r1b;
    g.r0a = g.r0c; g.r0b = a0; g.r0c = b0;
    g.r1a = g.r1c; g.r1b = a1; g.r1c = b1;
  };
  auto lw_nz = [&](f32x4 (&st)[2], bool real_slab) {
    // (the clamped extra slabs staged at the end of a pass are never read and do not count)
    const unsigned m = real_slab ? 0x7fffffffu : 0u;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      g.nzbits0 |= __float_as_uint(st[0][e]) & m;
      if (NFB == 2) g.nzbits1 |= __float_as_uint(st[1][e]) & m;
    }
  };
  auto lw_store = [&](int buf_off, f32x4 (&st)[2]) {
    float* d = xs + buf_off + g.srow * S_XS + g.scol;
    *reinterpret_cast<f32x4*>(d) = st[0];
    if (NFB == 2) *reinterpret_cast<f32x4*>(d + 32 * S_XS) = st[1];
  };
  auto lw_finish = [&](int buf_off, f32x4 (&st)[2], bool real_slab) {
    lw_nz(st, real_slab);
    lw_store(buf_off, st);
  };
  auto lwrite = [&](int s, int buf_off, f32x4 (&st)[2]) {   // un-pipelined form (prologue only)
    lw_begin(st);
    if (NORM) {
#pragma unroll
      for (int row = 0; row < NFB; ++row)
#pragma unroll
        for (int e = 0; e < 4; ++e) lw_norm(st, row, e);
      lw_rotate();
    }
    lw_finish(buf_off, st, true);
  };

  load_bias<NB, NFB>(acc, c.blob4 + c.hdr->b_off[ST_E0] + job * (NB * 8), c.h);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
      for (int q = 0; q < 16; ++q) tot[nb][fb][q] = 0.0f;
  const f32x4* w = c.blob4 + c.hdr->w_off[ST_E0] + (size_t)job * c.hdr->job_w16[ST_E0] + c.lane;

  if (NORM) {  // phase of slab 0: this thread's first column is scol, element e is column scol + e;
               // landmark 1 (columns 3,4,5) is the reference point (FeatureExtractor.py:85-86)
    const double x0 = (double)g.p0[3], y0 = (double)g.p0[4], z0 = (double)g.p0[5];
    const double x1 = NFB == 2 ? (double)g.p1[3] : 0.0, y1 = NFB == 2 ? (double)g.p1[4] : 0.0, z1 = NFB == 2 ? (double)g.p1[5] : 0.0;
    const int ph = g.scol % 3;   // coordinate (0 x, 1 y, 2 z) of this thread's first column
    g.r0a = ph == 0 ? x0 : (ph == 1 ? y0 : z0);
    g.r0b = ph == 0 ? y0 : (ph == 1 ? z0 : x0);
    g.r0c = ph == 0 ? z0 : (ph == 1 ? x0 : y0);
    g.r1a = ph == 0 ? x1 : (ph == 1 ? y1 : z1);
    g.r1b = ph == 0 ? y1 : (ph == 1 ? z1 : x1);
    g.r1c = ph == 0 ? z1 : (ph == 1 ? x1 : y1);
  }
  f32x4 setA[2], setB[2];
  gload(0, setA);
  gload(1, setB);
  // weight ring of R0 = 4 slots: K step ks lives in slot ks % 4 and every slab holds exactly 4 steps
  // (the packer pads layer 0's K with zero weights), so the slot of every step is static.
  constexpr int R0 = 4, D0 = R0 - 1;
  static_assert(XS_STEPS == R0, "slab steps == ring slots keeps the slot index static");
  f32x4 wr[R0][NB];
#pragma unroll
  for (int d = 0; d < D0; ++d)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) wr[d][nb] = w[(d * NB + nb) * 64];
  constexpr int SLAB = 64 * S_XS;
  lwrite(0, 0, setA);
  lwrite(1, SLAB, setB);
  gload(2, setA);
  __syncthreads();

  const int lane_off = c.f * S_XS + 4 * c.h;
  f32x4 xr[2][NFB];
#pragma unroll
  for (int fb = 0; fb < NFB; ++fb) xr[0][fb] = *reinterpret_cast<const f32x4*>(xs + lane_off + fb * (32 * S_XS));

  // one slab: `ld` receives the global loads of slab s+3, `wrset` (loaded one slab earlier) is written as
  // slab s+2.  o_cur / o_next / o_wr: LDS offsets of the buffers of slabs s, s+1, s+2 (rotated by the caller,
  // so no modulo-3 arithmetic sits between the MFMAs).
  auto slab = [&](int s, int o_cur, int o_next, int o_wr, f32x4 (&ld)[2], f32x4 (&wrset)[2]) {
    const float* xrow = xs + o_cur + lane_off;
    const float* xnext = xs + o_next + lane_off;
#pragma unroll
    for (int kk = 0; kk < XS_STEPS; ++kk) {
      step_interleaved<NB, NFB>(
          acc, wr[kk % R0], xr[kk & 1], wr[(kk + D0) % R0], w + (size_t)(s * XS_STEPS + kk + D0) * (NB * 64),
          [&]() {
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb)
              xr[(kk + 1) & 1][fb] = (kk + 1 < XS_STEPS)
                                         ? *reinterpret_cast<const f32x4*>(xrow + fb * (32 * S_XS) + 8 * (kk + 1))
                                         : *reinterpret_cast<const f32x4*>(xnext + fb * (32 * S_XS));
          },
          [&](int j) {   // (kk, j) are static: slab s+2's staging, one piece per gap
#ifdef EXP_NO_STAGE
            return;
#endif
            if (kk == 0 && j == 2) gload(s + 3, ld);          // its own gap: address arithmetic + 2 loads
            if (kk == 1 && j == 0) lw_begin(wrset);
            if (NORM && kk == 1) lw_norm(wrset, 0, j);
            if (NORM && NFB == 2 && kk == 2) lw_norm(wrset, 1, j);
            if (NORM && kk == 3 && j == 0) lw_rotate();
            if (kk == 3 && j == 1) lw_nz(wrset, s + 2 < nslab);
            if (kk == 3 && j == 2) lw_store(o_wr, wrset);
          });
    }
#ifndef EXP_NO_BAR
    __syncthreads();
#endif
  };
  // nslab is even (pack.cpp pads K to whole PAIRS of slabs): the two register sets alternate statically
  int o0 = 0, o1 = SLAB, o2 = 2 * SLAB;   // buffers of slabs s, s+1, s+2
  auto slab_pair = [&](int s) {
    slab(s, o0, o1, o2, setB, setA);
    slab(s + 1, o1, o2, o0, setA, setB);
    const int t0 = o0, t1 = o1;          // advance by two slabs: (o0,o1,o2) <- (o2,o0,o1)
    o0 = o2; o1 = t0; o2 = t1;
  };
  // a block of the K-blocked sum is FOUR slabs (128 k); a last block of two slabs where nslab % 4 == 2
  const int nquad = nslab >> 2;
#pragma unroll 1
  for (int qd = 0; qd < nquad; ++qd) {
    slab_pair(4 * qd);
    slab_pair(4 * qd + 2);
    fold_block<NB, NFB>(tot, acc);
  }
  if (nslab & 2) {
    slab_pair(4 * nquad);
    fold_block<NB, NFB>(tot, acc);
  }
}

// ------------------------------------------------------------------------------------------
// DBG = true is the diagnostic build behind nlml_encoder_heads_fwd_debug: it also writes the
// pre-Tanh activations and per-wave s_memtime stamps at the stage boundaries (read their SHARES,
// not their length).  The production instantiations (DBG = false) contain neither.
#define NLML_STAMP(i)                                                                              \
  do {                                                                                             \
    if (DBG && a.stamps && c.lane == 0)                                                            \
      a.stamps[((size_t)blockIdx.x * 4 + wv) * 16 + (i)] = __builtin_amdgcn_s_memtime();         \
  } while (0)

// NFB = face blocks per tile: 2 (64 faces, every weight fragment feeds 8 MFMAs) for large batches, 1 (32 faces)
// for batches too small to give every CU a 64-face tile.
template <bool VEC4, bool NORM, bool DBG, int NFB>
__global__ __launch_bounds__(256, 1) void encoder_heads_f32_kernel(EncArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

  const int tid = threadIdx.x;
  Ctx c;
  c.blob4 = reinterpret_cast<const f32x4*>(a.blob);
  c.hdr = reinterpret_cast<const Header*>(a.blob);
  c.lds = lds;
  c.lane = tid & 63;
  c.f = c.lane & 31;
  c.h = c.lane >> 5;
  c.wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wv = c.wv;
  constexpr int TILE = 32 * NFB;
  const int64_t row0 = (int64_t)blockIdx.x * TILE;
  // Re-evaluation launch: the split-f16 kernel before this one left a NON-FINITE pose for every face whose activations do not fit f16
  // (un-normalised pixel-scale landmarks, the reference's ipd == 0 branch) and re-evaluated such faces itself where a tile had few;
  // a tile with MORE than reeval_over of them comes here -- the whole tile through the f32 matrix cores (this kernel, its own blob
  // image), only the non-finite faces written.  Every other tile ends at once (one 768-byte read).
  // The count is over the split-f16 kernels' 64-face tile whatever this launch's own tiling (a 32-face tile looks at its parent).
  unsigned long long redo = ~0ull;
  if (a.reeval_over >= 0) {
    const int64_t r = (row0 & ~(int64_t)63) + c.lane;
    bool bad = false;
    if (r < a.B) {
      const float p0 = a.out[r * 3 + 0], p1 = a.out[r * 3 + 1], p2 = a.out[r * 3 + 2];
      bad = !(__builtin_isfinite(p0) && __builtin_isfinite(p1) && __builtin_isfinite(p2));
    }
    const unsigned long long parent = __ballot(bad);
    if (__popcll(parent) <= a.reeval_over) return;   // uniform over the workgroup: every wave reads the same 64 poses
    redo = NFB == 2 ? parent : (parent >> (row0 & 32)) & 0xffffffffull;
    if (redo == 0) return;                           // (a 32-face tile whose half of the parent is clean)
  }

  {  // ---- layers 0 and 1 interleaved in two passes over x (see header), both summed in blocks of 128 k (fold_block)
    E0Stager g;
    e0_stager_init<NORM, NFB>(g, a, row0, tid);
    const f32x4* w1 = c.blob4 + c.hdr->w_off[ST_E1] + (size_t)wv * c.hdr->job_w16[ST_E1] + c.lane;
    // layer 1's block sums: neurons 128*wv .. +127, all face blocks.  Layer 0 needs two accumulator sets of its own (chain +
    // block sums), so this set does not exist during pass A's layer 0 and is PARKED IN LDS during pass B's: the h1 half image
    // is dead then (pass A's half has been consumed, pass B's is not written yet), lane-private 16-byte pieces, conflict-free.
    f32x16 acc1[4][NFB];
    float* const park = lds + O_H1H + tid * 4;
    constexpr int PARK_Q = 256 * 4;   // floats between consecutive 16-byte pieces of a lane
    static_assert(32 * PARK_Q <= 64 * S_H1H, "the parked layer-1 sums fit the h1 half image");
    // The loop body is the same for both passes -- fetch the parked sums after layer 0, park them again after layer 1 -- so that
    // the set is dead during layer 0 on EVERY path through the loop (with a reload only in pass B the compiler has to keep 128
    // more registers alive through pass A's layer 0 and spills).  Pass A fetches the zeros parked here.
#pragma unroll
    for (int i = 0; i < 4 * NFB * 4; ++i) *reinterpret_cast<f32x4*>(park + i * PARK_Q) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    NLML_STAMP(0);
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      {  // E0 half: F -> neurons 512*pass + 128*wv .. +127, ReLU
        f32x16 acc0[4][NFB];
        stage_e0_pass<VEC4, NORM, NFB>(c, a, g, pass * 4 + wv, acc0);
        NLML_STAMP(1 + 4 * pass);
        // layer 1's sums back from LDS before the h1 half image is (over)written
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
          for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(park + ((nb * NFB + fb) * 4 + q) * PARK_Q);
              acc1[nb][fb][4 * q + 0] = v[0]; acc1[nb][fb][4 * q + 1] = v[1];
              acc1[nb][fb][4 * q + 2] = v[2]; acc1[nb][fb][4 * q + 3] = v[3];
            }
        __syncthreads();
        job_store<4, NFB, ACT_RELU>(c, acc0, lds + O_H1H, S_H1H, 128 * wv, 0);
      }
      __syncthreads();
      NLML_STAMP(2 + 4 * pass);
      {  // E1 over this K half: k = 512*pass .. +511 = four blocks of 16 steps; the very first chain starts from the bias
        f32x16 ch[4][NFB];
        load_bias<4, NFB>(ch, c.blob4 + c.hdr->b_off[ST_E1] + wv * (4 * 8), c.h);
        if (pass != 0) {
#pragma unroll
          for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
              for (int q = 0; q < 16; ++q) ch[nb][fb][q] = 0.0f;
        }
        kloop_lds_blocked<4, NFB, 64, 16>(acc1, ch, w1 + (size_t)pass * 64 * (4 * 64), lds + O_H1H + c.f * S_H1H + 4 * c.h, 32 * S_H1H);
      }
      NLML_STAMP(3 + 4 * pass);
      __syncthreads();  // all waves done reading this h1 half before it is overwritten
      // park layer 1's sums for the length of the next layer-0 pass (after pass B nothing reads them back: 32 idle LDS writes)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v;
            v[0] = acc1[nb][fb][4 * q + 0]; v[1] = acc1[nb][fb][4 * q + 1];
            v[2] = acc1[nb][fb][4 * q + 2]; v[3] = acc1[nb][fb][4 * q + 3];
            *reinterpret_cast<f32x4*>(park + ((nb * NFB + fb) * 4 + q) * PARK_Q) = v;
          }
      NLML_STAMP(4 + 4 * pass);
    }
    if (a.valid && a.reeval_over < 0) {  // all-zero feature row == "no face" (FeatureExtractor.py:105-106)
      const unsigned long long m0 = __ballot(g.nzbits0 != 0u), m1 = __ballot(g.nzbits1 != 0u);
      if ((tid & 7) == 0) {
        const int sh = c.lane & 56;
        if (g.live0) a.valid[row0 + g.srow] = ((m0 >> sh) & 0xFFull) ? 1 : 0;
        if (g.live1) a.valid[row0 + g.srow + 32] = ((m1 >> sh) & 0xFFull) ? 1 : 0;
      }
    }
    // the pieces parked after pass B are dead, but another wave may still be WRITING its own: H2 overwrites that region, so
    // every wave's park stores must be behind this barrier before any wave stores H2 (write-after-write across waves)
    __syncthreads();
    job_store<4, NFB, ACT_RELU>(c, acc1, lds + O_H2, S_H2, 128 * wv, 0);
  }
  __syncthreads();
  NLML_STAMP(9);
  {  // E2: 512 -> 256, ReLU.  h3 overwrites h2 => barrier between the K loop and the store
    f32x16 acc[2][NFB];
    job_compute_blocked<2, NFB, ST_E2>(c, wv, acc, lds + O_H2, S_H2, 0, 0);
    __syncthreads();
    job_store<2, NFB, ACT_RELU>(c, acc, lds + O_H3, S_H3, 64 * wv, 0);
  }
  __syncthreads();
  NLML_STAMP(10);
  {  // E3: 256 -> 128, ReLU
    f32x16 acc[1][NFB];
    job_compute_blocked<1, NFB, ST_E3>(c, wv, acc, lds + O_H3, S_H3, 0, 0);
    job_store<1, NFB, ACT_RELU>(c, acc, lds + O_H4, S_H4, 32 * wv, 0);
  }
  __syncthreads();
  NLML_STAMP(11);
  if (wv < 2 * NFB) {  // E4: 128 -> 64, Tanh.  Single-face-block jobs: neuron block wv&1, face block wv>>1
    const int nb = wv & 1, face0 = 32 * (wv >> 1);
    f32x16 acc[1][1];
    job_compute<1, 1, ST_E4>(c, nb, acc, lds + O_H4, S_H4, 0, face0);
    if (DBG && a.pre_tanh && row0 + face0 + c.f < a.B) {
#pragma unroll
      for (int q = 0; q < 16; ++q)
        a.pre_tanh[(row0 + face0 + c.f) * 64 + 32 * nb + (q & 3) + 8 * (q >> 2) + 4 * c.h] = acc[0][0][q];
    }
    job_store<1, 1, ACT_TANH>(c, acc, lds + O_H5, S_H5, 32 * nb, face0);
  }
  __syncthreads();
  if (wv < NFB) {  // E5: 64 -> 9 (latent n = 3g+c on row 8g+c, other rows exact zeros); face block wv
    f32x16 acc[1][1];
    job_compute<1, 1, ST_E5>(c, 0, acc, lds + O_H5, S_H5, 0, 32 * wv);
    job_store<1, 1, ACT_NONE>(c, acc, lds + O_LAT, S_LAT, 0, 32 * wv);
  }
  __syncthreads();
  NLML_STAMP(12);
  if (a.latent) {  // optional: the encoder output before the split (Model_Builder.py:58)
    for (int i = tid; i < TILE * NLML_LATENT; i += 256) {
      const int ff = i / NLML_LATENT, n = i % NLML_LATENT;
      if (row0 + ff < a.B && ((redo >> ff) & 1))
        a.latent[(row0 + ff) * NLML_LATENT + n] = lds[O_LAT + ff * S_LAT + 8 * (n / 3) + (n % 3)];
    }
  }
  // ---- heads (yaw, pitch, roll = g 0,1,2), one 32-face block at a time.  A stage's jobs are (head,
  // neuron block) pairs; the jobs a wave owns run together through kloop_grouped.
#pragma unroll 1
  for (int fb = 0; fb < NFB; ++fb) {
    const int face0 = 32 * fb;
    const int lrow = c.f;   // row of the 32-face head images
    {  // H0: 3 -> 128 (K padded to 8 with zeros), ReLU.  12 jobs (g, nb), 3 per wave
      constexpr int ST = ST_H0;
      f32x16 acc[3][1][1];
      const float* in[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wv * 3 + j;
        load_bias<1, 1>(acc[j], c.blob4 + c.hdr->b_off[ST] + job * 8, c.h);
        in[j] = lds + O_LAT + (face0 + c.f) * S_LAT + 8 * (job >> 2) + 4 * c.h;
      }
      kloop_grouped<3, 1, kStages[ST].k8>(acc, c.blob4 + c.hdr->w_off[ST] + (size_t)(wv * 3) * c.hdr->job_w16[ST] + c.lane,
                                          c.hdr->job_w16[ST], in);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wv * 3 + j;
        store_lds<1, 1, ACT_RELU>(acc[j], lds + O_HA + lrow * S_HA + 128 * (job >> 2) + 32 * (job & 3) + 4 * c.h, 0);
      }
    }
    __syncthreads();
    {  // H1: 128 -> 256, ReLU.  12 jobs (g, pair of blocks), 3 per wave
      constexpr int ST = ST_H1;
      f32x16 acc[3][2][1];
      const float* in[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wv * 3 + j;
        load_bias<2, 1>(acc[j], c.blob4 + c.hdr->b_off[ST] + job * 16, c.h);
        in[j] = lds + O_HA + lrow * S_HA + 128 * (job >> 2) + 4 * c.h;
      }
      kloop_grouped<3, 2, kStages[ST].k8>(acc, c.blob4 + c.hdr->w_off[ST] + (size_t)(wv * 3) * c.hdr->job_w16[ST] + c.lane,
                                          c.hdr->job_w16[ST], in);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wv * 3 + j;
        store_lds<2, 1, ACT_RELU>(acc[j], lds + O_HB + lrow * S_HB + 256 * (job >> 2) + 64 * (job & 3) + 4 * c.h, 0);
      }
    }
    __syncthreads();
    {  // H2: 256 -> 128, ReLU.  12 jobs (g, nb), 3 per wave
      constexpr int ST = ST_H2;
      f32x16 acc[3][1][1];
      const float* in[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wv * 3 + j;
        load_bias<1, 1>(acc[j], c.blob4 + c.hdr->b_off[ST] + job * 8, c.h);
        in[j] = lds + O_HB + lrow * S_HB + 256 * (job >> 2) + 4 * c.h;
      }
      kloop_grouped<3, 1, kStages[ST].k8>(acc, c.blob4 + c.hdr->w_off[ST] + (size_t)(wv * 3) * c.hdr->job_w16[ST] + c.lane,
                                          c.hdr->job_w16[ST], in);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int job = wv * 3 + j;
        store_lds<1, 1, ACT_RELU>(acc[j], lds + O_HC + lrow * S_HC + 128 * (job >> 2) + 32 * (job & 3) + 4 * c.h, 0);
      }
    }
    __syncthreads();
    if (wv < 3) {  // H3: 128 -> 64, ReLU.  6 jobs (g, nb): waves 0..2 take the two blocks of head wv
      constexpr int ST = ST_H3;
      f32x16 acc[2][1][1];
      const float* in[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int job = wv * 2 + j;
        load_bias<1, 1>(acc[j], c.blob4 + c.hdr->b_off[ST] + job * 8, c.h);
        in[j] = lds + O_HC + lrow * S_HC + 128 * wv + 4 * c.h;
      }
      kloop_grouped<2, 1, kStages[ST].k8>(acc, c.blob4 + c.hdr->w_off[ST] + (size_t)(wv * 2) * c.hdr->job_w16[ST] + c.lane,
                                          c.hdr->job_w16[ST], in);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        store_lds<1, 1, ACT_RELU>(acc[j], lds + O_HD + lrow * S_HD + 64 * wv + 32 * j + 4 * c.h, 0);
    }
    __syncthreads();
    if (wv < 3) {  // H4: 64 -> 1; neuron on accumulator row 0 = register 0 of lanes 0..31
      f32x16 acc[1][1];
      job_compute<1, 1, ST_H4>(c, wv, acc, lds + O_HD, S_HD, 64 * wv, 0);
      if (c.h == 0 && row0 + face0 + c.f < a.B && ((redo >> (face0 + c.f)) & 1)) a.out[(row0 + face0 + c.f) * 3 + wv] = acc[0][0][0];
    }
    __syncthreads();  // hd / ha regions are reused by the next face block
  }
  NLML_STAMP(13);
}

int launch_encoder_heads_f32(const float* x, int64_t ldx, const float* raw, int normalize,
                             int64_t B, int F, const void* blob, float* out, float* latent,
                             uint8_t* valid, float* pre_tanh, unsigned long long* stamps, void* stream, int reeval_over) {
  if (B == 0) return 0;
  EncArgs a;
  a.reeval_over = reeval_over;
  a.B = B;
  a.F = F;
  a.blob = blob;
  a.out = out;
  a.latent = latent;
  a.valid = valid;
  a.pre_tanh = pre_tanh;
  a.stamps = stamps;
  const bool dbg = pre_tanh || stamps;
  a.norm = 0;
  if (raw) {  // raw landmarks [B,468,3]; without normalisation they ARE the feature rows
    a.x = raw;
    a.ldx = NLML_F_REFERENCE;
    a.norm = normalize ? 1 : 0;
  } else {
    a.x = x;
    a.ldx = ldx;
  }
  const bool vec4 = (F % 4 == 0) && (a.ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0);
  // 64-face tiles halve the weight stream per face (a 64-face tile takes ~1.9x a 32-face tile, measured), but the
  // launch lasts whole ROUNDS of tiles over the 256 CUs (one tile per CU, LDS-bound): pick the tiling whose
  // rounds x tile time is smaller, so small batches and awkward tile counts (257 tiles = 2 rounds) do not pay for it.
  const int64_t rounds64 = ((B + 63) / 64 + 255) / 256, rounds32 = ((B + 31) / 32 + 255) / 256;
  const bool wide = rounds64 * 19 <= rounds32 * 10;
  const int tile = wide ? 64 : 32;
  const dim3 grid((unsigned)((B + tile - 1) / tile)), block(256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define NLML_LAUNCH(V, N, D)                                                                          \
  do {                                                                                                \
    if (wide) hipLaunchKernelGGL((encoder_heads_f32_kernel<V, N, D, 2>), grid, block, 0, st, a);      \
    else hipLaunchKernelGGL((encoder_heads_f32_kernel<V, N, D, 1>), grid, block, 0, st, a);           \
  } while (0)
  if (dbg) {
    if (a.norm || !vec4) return fail(NLML_E_BADARG, "debug build: features input, F % 4 == 0 only");
    hipLaunchKernelGGL((encoder_heads_f32_kernel<true, false, true, 2>), dim3((unsigned)((B + 63) / 64)), block, 0, st, a);
  } else if (a.norm) {
    if (vec4) NLML_LAUNCH(true, true, false); else NLML_LAUNCH(false, true, false);
  } else {
    if (vec4) NLML_LAUNCH(true, false, false); else NLML_LAUNCH(false, false, false);
  }
#undef NLML_LAUNCH
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

}  // namespace nlml
