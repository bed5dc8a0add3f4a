// encoder_heads_f16x2_small.hip -- K2 for SMALL batches (split-f16 parity mode): the big layers as separate launches.
//
// The fused kernel (encoder_heads_f16x2.hip) walks one 64-face tile through the whole network on ONE CU, so a batch of
// 64 faces (a video tick) or 2,000 faces (BASELINE config 4) keeps 1 or 32 of the 256 CUs busy and its latency is the
// time to stream the 9.6 MB of weights through one CU (~0.17 ms).  Here the three big layers (E0, E1, E2: 93 % of the
// weights) are each their own launch over (neuron blocks x face tiles): the weight stream of a layer is spread over up
// to 32 x tiles waves, activations pass between the launches through a caller-provided workspace in MFMA-fragment
// order (so both operand streams are coalesced 1-KiB loads, no LDS), and stream order is the only synchronisation (no
// spin-waits, graph-capturable).  The small tail (E3, E4, E5, heads) is one more launch: a workgroup per tile runs the
// fused kernel's own tail_stages().  Five launches in all: pre-pass, E0, E1, E2, tail.
//
// Arithmetic is the fused kernel's, operation for operation: the same blob (same hi/lo weight pieces and scales), the
// same activation split, per output the same K-ascending sequence of the same three MFMAs -- the results are
// bit-identical to the fused kernel's (tests/test_gpu_parity.py), so everything pinned there holds here.
//
// Workspace: two ping-pong buffers of ntiles x max(k16_e0, 64) K-steps x 4 KiB
//   fragment (tile, step, face block fb, piece, lane) = 8 f16 = activations k = 16*step + 8*(lane>>5) .. +7 of face
//   64*tile + 32*fb + (lane&31).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "../../include/nlml_hpe.h"
#include "abi_internal.h"
#include "encoder_heads_f16x2_dev.h"
#include "layout.h"

namespace nlml {
namespace hxs {

using hx::f32x16;
using hx::f32x4;
using hx::h4;
using hx::h8;
using hx::ACT_NONE;
using hx::ACT_RELU;
using hx::ACT_TANH;
constexpr int STEP_UNITS = 2 * 2 * 64;   // h8 units per (tile, K step): 2 face blocks x 2 pieces x 64 lanes = 4 KiB

__device__ __forceinline__ float activate(int act, float v) {
  if (act == ACT_RELU) return v < 0.0f ? 0.0f : v;
  if (act == ACT_TANH) return tanhf(v);
  return v;
}

__device__ __forceinline__ double div_ipd(double n, double d, double y) {
  const double q = n * y;
  const double r = fma(-q, d, n);
  return fma(r, y, q);
}

// ---- pre-pass: x (f32 rows, or raw landmarks with optional IPD normalisation) -> hi/lo fragments of layer 0's input.
// One wave per face; lane handles 8-column chunks c = lane, lane + 64, ... (chunk c = K step c/2, half c&1).
// VEC4: F, the row stride and the base address allow 16-byte loads (the shipped 1,404-column rows do).
template <bool VEC4>
__global__ __launch_bounds__(256) void prepass_kernel(const float* __restrict__ x, int64_t ldx, int64_t B, int F, int norm,
                                                      int k16, int buf_steps, h8* __restrict__ ws,
                                                      uint8_t* __restrict__ valid) {
  const int lane = threadIdx.x & 63;
  const int64_t face = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t tile = face >> 6;
  const int fi = (int)(face & 63), fb = fi >> 5, f = fi & 31;
  const bool live = face < B;
  const float* p = x + (live ? face : B - 1) * ldx;
  f32x4 q[3][2];   // the trip's chunks as loaded
  auto loadq = [&](int c0) {   // branch-free: an address clamped into the row (the value is zeroed where it lay outside)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int k = 8 * (c0 + 64 * i) + 4 * hh;
        q[i][hh] = *reinterpret_cast<const f32x4*>(p + (k < F ? k : F - 4));
      }
  };
  if (VEC4) loadq(lane);   // before the reference points: in flight while the reciprocal is formed
  double ipd = 1.0, rcp = 1.0, ref0 = 0.0, ref1 = 0.0, ref2 = 0.0;
  if (norm) {   // FeatureExtractor.py:30-66, exactly as K1 and the fused kernels do it
    const double dx = (double)p[99] - (double)p[789], dy = (double)p[100] - (double)p[790], dz = (double)p[101] - (double)p[791];
    ipd = sqrt(fma(dz, dz, fma(dy, dy, dx * dx)));
    if (ipd == 0.0) ipd = 1e-6;
    rcp = 1.0 / ipd;
    ref0 = (double)p[3]; ref1 = (double)p[4]; ref2 = (double)p[5];
  }
  unsigned nz = 0u;
  // Three chunks per lane and trip (layer 0 of the shipped encoder: 176 chunks = one trip): their 24 loads are issued together with the
  // reference-point loads above, so the wave pays ONE memory latency, not one per chunk after the reciprocal's (64 faces: 8.1 -> us).
  for (int c0 = lane; c0 < 2 * k16; c0 += 192) {
    float v[3][8];
    if (VEC4) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int k = 8 * (c0 + 64 * i) + 4 * hh;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[i][4 * hh + e] = k < F ? q[i][hh][e] : 0.0f;
        }
      if (c0 + 192 < 2 * k16) loadq(c0 + 192);
    } else {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = 8 * (c0 + 64 * i) + e;
          v[i][e] = k < F ? p[k] : 0.0f;
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int c = c0 + 64 * i;
      if (c >= 2 * k16) break;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 8 * c + e;
        const int ph = k % 3;
        if (norm && k < F) v[i][e] = (float)div_ipd((double)v[i][e] - (ph == 0 ? ref0 : (ph == 1 ? ref1 : ref2)), ipd, rcp);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(v[i][e]));   // keep the f32 value (see encoder_heads_f16x2.hip)
      h8 hi, lo;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        nz |= __float_as_uint(v[i][e]) & 0x7fffffffu;
        const _Float16 hv = (_Float16)v[i][e];
        hi[e] = hv;
        lo[e] = (_Float16)(v[i][e] - (float)hv);
      }
      if (live) {
        h8* d = ws + ((size_t)tile * buf_steps + (c >> 1)) * STEP_UNITS + (size_t)(fb * 2) * 64 + f + 32 * (c & 1);
        d[0] = hi;
        d[64] = lo;
      }
    }
  }
  const unsigned long long any = __ballot(nz != 0u);
  if (valid && live && lane == 0) valid[face] = any ? 1 : 0;
}

// ---- one layer.  Unit = (tile, job, group of NBW neuron blocks of the job); one wave per unit.
struct LayerArgs {
  const void* blob;
  const h8* xin;
  h8* xout;
  int64_t B;
  int stage, K16, nb_stage, jobs, ntiles, buf_steps, act;
  int fold_step;      // K step after which the small accumulator is folded into the big one mid-way (layer 1 in NLML_MODE_F16X2S: 32; else 0)
  int split_from;     // first K step whose small products go to the small accumulator (SPLIT kernels; layer 1 in NLML_MODE_F16X2: 32)
  int in_step0[12];   // first input K step of job j (its input column / 16)
  int out_col0[12];   // first output column of job j
};

// R = depth of the operand ring.  With few units (one wave per CU) a unit's rate is its own loads in flight, so those
// launches use 64-thread workgroups (the units spread over the CUs instead of sharing one four at a time) and R = 8.
// SPLITK != 0: the small products of a K step accumulate apart, as the fused kernel's step_fine does for this layer in the blob's mode.
// NFB = face blocks per unit: 2 (a weight fragment feeds both face blocks), or 1 for the smallest batches -- a unit is then (tile, job,
// group, face block): twice the waves, each streaming 4 KiB instead of 6 per K step (a wave alone on a CU is bound by its own stream:
// layer 0 at 64 faces 14.0 -> 9.5 us), the weights fetched twice chip-wide.  Faces are MFMA columns: the same bits either way.
template <int NBW, int R, int SPLITK, int NFB = 2>
__global__ __launch_bounds__(256) void layer_kernel(LayerArgs a) {
  constexpr bool SPLIT = SPLITK != 0;   // SPLITK: 0 = single accumulators, 1 = split accumulators, 2 = split from K step a.split_from on
  const int lane = threadIdx.x & 63, f = lane & 31, h = lane >> 5;
  const int groups = a.nb_stage / NBW;                     // units per (tile, job) and face-block choice
  int64_t u = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (u >= (int64_t)a.ntiles * a.jobs * groups * (2 / NFB)) return;    // whole wave leaves together
  const int fb0 = NFB == 1 ? (int)(u & 1) : 0;             // the unit's (first) face block
  if (NFB == 1) u >>= 1;
  const int grp = (int)(u % groups), job = (int)((u / groups) % a.jobs);
  const int64_t tile = u / ((int64_t)groups * a.jobs);
  const int nb0 = grp * NBW;
  const Header* hdr = reinterpret_cast<const Header*>(a.blob);
  const h8* blob8 = reinterpret_cast<const h8*>(a.blob);
  const f32x4* blob4 = reinterpret_cast<const f32x4*>(a.blob);
  const int st = a.stage, NBS = a.nb_stage, K16 = a.K16;

  f32x16 acc[NBW][NFB];
  {  // bias in accumulator-register order (layout.h), scaled like the weights
    const f32x4* b = blob4 + hdr->b_off[st] + (size_t)job * (NBS * 8);
#pragma unroll
    for (int i = 0; i < NBW; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = b[((nb0 + i) * 2 + h) * 4 + q];
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) {
          acc[i][fb][4 * q + 0] = v[0]; acc[i][fb][4 * q + 1] = v[1];
          acc[i][fb][4 * q + 2] = v[2]; acc[i][fb][4 * q + 3] = v[3];
        }
      }
  }
  const h8* w = blob8 + hdr->w_off[st] + (size_t)job * hdr->job_w16[st] + (size_t)nb0 * 128 + lane;   // + step*NBS*128
  const h8* xi = a.xin + ((size_t)tile * a.buf_steps + a.in_step0[job]) * STEP_UNITS + lane;          // + step*256

  constexpr int D = R - 1;
  h8 wr[R][NBW][2], xr[R][NFB][2];
  auto load = [&](int slot, int s) {
    const h8* wp = w + (size_t)s * NBS * 128;
#pragma unroll
    for (int i = 0; i < NBW; ++i)
#pragma unroll
      for (int p = 0; p < 2; ++p) wr[slot][i][p] = wp[(i * 2 + p) * 64];
    const h8* xp = xi + (size_t)s * STEP_UNITS;
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
      for (int p = 0; p < 2; ++p) xr[slot][fb][p] = xp[((fb0 + fb) * 2 + p) * 64];
  };
  constexpr int NBS_ = SPLIT ? NBW : 1;
  f32x16 accS[NBS_][NFB];   // the two small products of every K step; added to acc at the end (layer 1: also at its K midpoint,
#pragma unroll          // where the fused kernel parks layer 1's accumulators in LDS for the length of layer 0's second pass)
  for (int i = 0; i < NBS_; ++i)
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
      for (int q = 0; q < 16; ++q) accS[i][fb][q] = 0.0f;
  auto fold = [&](bool clear) {
    if constexpr (!SPLIT) return;
#pragma unroll
    for (int i = 0; i < NBS_; ++i)
#pragma unroll
      for (int fb = 0; fb < NFB; ++fb) {
        acc[i][fb] += accS[i][fb];
        if (clear) {
#pragma unroll
          for (int q = 0; q < 16; ++q) accS[i][fb][q] = 0.0f;
        }
      }
  };
  const int split_from = a.split_from;
  auto mma3 = [&](int slot, auto split_c) {
    constexpr bool SP = decltype(split_c)::value;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int wp_ = t == 0 ? 1 : 0, xp_ = t == 1 ? 1 : 0;   // (lo,hi), (hi,lo), (hi,hi): the fused kernel's order
#pragma unroll
      for (int i = 0; i < NBW; ++i)
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb)
          // split accumulators, exactly as the fused kernel (encoder_heads_f16x2_dev.h step_fine): small products apart
          if (SP && t < 2) accS[SP ? i : 0][fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[slot][i][wp_], xr[slot][fb][xp_], accS[SP ? i : 0][fb], 0, 0, 0);
          else acc[i][fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[slot][i][wp_], xr[slot][fb][xp_], acc[i][fb], 0, 0, 0);
    }
  };
  // K steps below split_from (wave-uniform) run on the single accumulator, as the fused kernel's does where it has no register for
  // a second set (NLML_MODE_F16X2: layer 1's first K half)
  auto mma = [&](int slot, int step) {
    if constexpr (SPLITK == 2) {
      if (step >= split_from) mma3(slot, std::true_type{});
      else mma3(slot, std::false_type{});
    } else if constexpr (SPLITK == 1) {
      mma3(slot, std::true_type{});
    } else {
      mma3(slot, std::false_type{});
    }
  };
#pragma unroll
  for (int d = 0; d < D; ++d) load(d, d < K16 ? d : K16 - 1);
  const int groups4 = K16 / R;
  // groups [g0, g1) of R steps; sp_c: which accumulators the groups' small products go to (SPLITK == 2: the launcher makes
  // split_from a multiple of R, so the choice is per group and the loop body has no branch)
  auto run_groups = [&](int g0, int g1, auto sp_c) {
    for (int g = g0; g < g1; ++g) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int sp = g * R + r + D;
        load((r + D) % R, sp < K16 ? sp : K16 - 1);
        __builtin_amdgcn_sched_barrier(0);
        mma3(r, sp_c);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  if constexpr (SPLITK == 2) {
    const int gs = split_from / R < groups4 ? split_from / R : groups4;
    run_groups(0, gs, std::false_type{});
    run_groups(gs, groups4, std::true_type{});
  } else if constexpr (SPLITK == 1) {
    // the mid-way fold sits BETWEEN two group loops (fold_step is a multiple of R, launcher): a test per step inside one loop makes
    // every step a basic block of its own, and hipcc then moves both accumulator sets between the register files around each
    // (128-192 v_accvgpr moves per 6 MFMAs: the split layers ran 25 % slower than the plain ones)
    const int gf = a.fold_step > 0 && a.fold_step / R < groups4 ? a.fold_step / R : groups4;
    run_groups(0, gf, std::true_type{});
    if (gf < groups4) {
      fold(true);
      run_groups(gf, groups4, std::true_type{});
    }
  } else {
    run_groups(0, groups4, std::false_type{});
  }
  const int tail = K16 - groups4 * R;   // steps groups4*R + r sit in slot r (loaded D steps earlier, or by the prologue)
#pragma unroll
  for (int r = 0; r < R - 1; ++r)
    if (r < tail) mma(r, groups4 * R + r);

  fold(false);
  const float inv = hdr->inv_scale[st];
  // accumulators * inv -> activation -> hi/lo -> the next layer's input fragments
#pragma unroll
  for (int i = 0; i < NBW; ++i)
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = a.out_col0[job] + 32 * (nb0 + i) + 8 * q + 4 * h;   // this lane's 4 neurons n .. n+3
        h4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = activate(a.act, acc[i][fb][4 * q + e] * inv);
          const _Float16 hv = (_Float16)v;
          hi[e] = hv;
          lo[e] = (_Float16)(v - (float)hv);
        }
        h8* frag = a.xout + ((size_t)tile * a.buf_steps + (n >> 4)) * STEP_UNITS + (size_t)((fb0 + fb) * 2) * 64 + f + 32 * ((n >> 3) & 1);
        _Float16* d = reinterpret_cast<_Float16*>(frag) + (n & 7);
        *reinterpret_cast<h4*>(d) = hi;
        *reinterpret_cast<h4*>(d + 64 * 8) = lo;
      }
}

// ---- the network's tail (E3, E4, E5 and the three heads: 8 of the 11 layers, 7 % of the FLOP) as ONE launch: a
// workgroup per 32-FACE BLOCK copies its half of E2's output fragments into the H3 LDS image and runs the fused
// kernel's tail_stages() on that block (two workgroups per tile: half the sequential stages of the 64-face form).
template <int RESCUE_UP_TO>
__global__ __launch_bounds__(256, 1) void tail_kernel(hx::Args a, const h8* __restrict__ xin, int buf_steps) {
  __shared__ __attribute__((aligned(16))) char lds[hx::LDS_BYTES];
  const int tid = threadIdx.x;
  hx::Ctx c;
  c.blob8 = reinterpret_cast<const h8*>(a.blob);
  c.blob4 = reinterpret_cast<const f32x4*>(a.blob);
  c.hdr = hx::load_hdr(reinterpret_cast<const Header*>(a.blob));
  c.lds = lds;
  c.lane = tid & 63;
  c.f = c.lane & 31;
  c.h = c.lane >> 5;
  c.wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t tile = blockIdx.x >> 1, row0 = tile * TILE_FACES;
  const int fbsel = blockIdx.x & 1;
  if (row0 + 32 * fbsel >= a.B) return;                       // no live face in this block (whole workgroup leaves)
  f32x16 acc3[1][1];
  h8 wr3[hx::ring_slots(1, 1)][1][2];
  hx::tail_pre_e3<true>(c, acc3, wr3);                        // E3's bias and first weights fly while the image is copied
  // fragment (step, fb, piece, lane) -> image row 32*fb + (lane & 31), columns 16*step + 8*(lane >> 5) .. +7 of plane `piece`
  const h8* src = xin + (size_t)tile * buf_steps * STEP_UNITS;
  for (int i = tid; i < 16 * 128; i += 256) {                 // E2's output: 256 columns = 16 K steps, this face block
    const int l = i & 63, piece = (i >> 6) & 1, step = i >> 7;
    const int face = 32 * fbsel + (l & 31), k = 16 * step + 8 * (l >> 5);
    *reinterpret_cast<h8*>(lds + hx::O_H3 + piece * hx::P_H3 + (face * hx::S_H3 + k) * 2) =
        src[(size_t)step * STEP_UNITS + (fbsel * 2 + piece) * 64 + l];
  }
  __syncthreads();
  hx::tail_stages<true, RESCUE_UP_TO>(c, a, row0, acc3, wr3, fbsel);
}

}  // namespace hxs
namespace hx {
// ---- the tail in TWO launches (NLML_MODE_F16X2S): E3, E4, E5 per (tile, face block), then the three heads as SEPARATE workgroups per
// (tile, face block, head) -- the heads are independent chains of five short stages, so at 64 faces six CUs work where two did and the
// tail's 13 sequential stages become 3 + 5 (tail 20.3 -> ~11 us at 64 faces).  The latent image (hi/lo f16, 32 rows x 112 bytes x 2
// planes) passes through the workspace buffer E2 no longer needs.  Per accumulator the MFMAs are the fused kernel's in the fused
// kernel's order (a head stage's jobs one at a time or three in lock step: the same chains), so the bits are the fused kernel's.
// No slow path here: the strict mode's out-of-range faces go to the f32 re-evaluation launch behind the tail.
constexpr int LAT_ROW_BYTES = S_LAT * 2, LAT_BLOCK_BYTES = 32 * LAT_ROW_BYTES;   // one plane of one face block: 3,584 B
static_assert(LAT_ROW_BYTES % 16 == 0 && O_LAT % 16 == 0 && P_LAT % 16 == 0, "16-byte copies of the latent image");

__global__ __launch_bounds__(256, 1) void tail_encoder_kernel(Args a, const h8* __restrict__ xin, int buf_steps, char* __restrict__ latws) {
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  const int tid = threadIdx.x;
  Ctx c;
  c.blob8 = reinterpret_cast<const h8*>(a.blob);
  c.blob4 = reinterpret_cast<const f32x4*>(a.blob);
  c.hdr = load_hdr(reinterpret_cast<const Header*>(a.blob));
  c.lds = lds;
  c.lane = tid & 63;
  c.f = c.lane & 31;
  c.h = c.lane >> 5;
  c.wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wv = c.wv;
  const int64_t tile = blockIdx.x >> 1, row0 = tile * TILE_FACES;
  const int fb = blockIdx.x & 1, face0 = 32 * fb;
  if (row0 + face0 >= a.B) return;                            // no live face in this block (whole workgroup leaves)
  f32x16 acc3[1][1];
  h8 wr3[ring_slots(1, 1)][1][2];
  tail_pre_e3<true>(c, acc3, wr3);                            // E3's bias and first weights fly while the image is copied
  const h8* src = xin + (size_t)tile * buf_steps * hxs::STEP_UNITS;
  for (int i = tid; i < 16 * 128; i += 256) {                 // E2's output: 256 columns = 16 K steps, this face block
    const int l = i & 63, piece = (i >> 6) & 1, step = i >> 7;
    const int face = face0 + (l & 31), k = 16 * step + 8 * (l >> 5);
    *reinterpret_cast<h8*>(lds + O_H3 + piece * P_H3 + (face * S_H3 + k) * 2) = src[(size_t)step * hxs::STEP_UNITS + (fb * 2 + piece) * 64 + l];
  }
  __syncthreads();
  const bool do4 = wv < 2, do5 = wv == 0;                     // E4: neuron block wv (waves 0, 1); E5: wave 0
  f32x16 acc4[1][1], acc5[2][1];
  h8 wr4[6][1][2], wr5[6][2][2];
  // ---- E3: 256 -> 128, ReLU
  job_run<1, 1, ST_E3>(c, wv, acc3, wr3, O_H3, P_H3, S_H3, 0, face0);
  job_store<1, 1, ACT_RELU>(c, acc3, O_H4, P_H4, S_H4, 32 * wv, face0, c.hdr.inv_scale[ST_E3], fetch_hook<2, 1, 1, ST_E4>(c, wv & 1, acc4, wr4));
  __syncthreads();
  // ---- E4: 128 -> 64, Tanh (every wave fetches, also the ones that do not run the stage: encoder_heads_f16x2_dev.h)
  if (do4) job_run<1, 1, ST_E4>(c, wv & 1, acc4, wr4, O_H4, P_H4, S_H4, 0, face0);
  job_pre<2, 1, ST_E5>(c, 0, acc5, wr5);
  if (do4) job_store<1, 1, ACT_TANH>(c, acc4, O_H5, P_H5, S_H5, 32 * (wv & 1), face0, c.hdr.inv_scale[ST_E4]);
  __syncthreads();
  // ---- E5: 64 -> 9, latent n = 3g+c on row 16g+c (2 blocks), other rows exact zeros
  if (do5) {
    job_run<2, 1, ST_E5>(c, 0, acc5, wr5, O_H5, P_H5, S_H5, 0, face0);
    const float inv = c.hdr.inv_scale[ST_E5];
    if (a.latent && row0 + face0 + c.f < a.B) {               // f32 latent straight from the accumulators
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int rowi = 32 * nb + (q & 3) + 8 * (q >> 2) + 4 * c.h, g = rowi >> 4, cc = rowi & 15;
          if (g < 3 && cc < 3) a.latent[(row0 + face0 + c.f) * NLML_LATENT + 3 * g + cc] = acc5[nb][0][q] * inv;
        }
    }
    job_store<2, 1, ACT_NONE, S_LAT>(c, acc5, O_LAT, P_LAT, S_LAT, 0, face0, inv);
  }
  __syncthreads();
  // the face block's rows of the latent image, both planes, to the workspace
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  char* dst = latws + ((size_t)tile * 2 + fb) * (2 * LAT_BLOCK_BYTES);
  for (int i = tid; i < 2 * LAT_BLOCK_BYTES / 16; i += 256) {
    const int piece = i / (LAT_BLOCK_BYTES / 16), o = (i % (LAT_BLOCK_BYTES / 16)) * 16;
    *reinterpret_cast<u4*>(dst + piece * LAT_BLOCK_BYTES + o) = *reinterpret_cast<const u4*>(lds + O_LAT + piece * P_LAT + face0 * LAT_ROW_BYTES + o);
  }
}

__global__ __launch_bounds__(256, 1) void head_kernel(Args a, const char* __restrict__ latws) {
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  const int tid = threadIdx.x;
  Ctx cl;
  cl.blob8 = reinterpret_cast<const h8*>(a.blob);
  cl.blob4 = reinterpret_cast<const f32x4*>(a.blob);
  cl.hdr = load_hdr(reinterpret_cast<const Header*>(a.blob));
  cl.lds = lds;
  cl.lane = tid & 63;
  cl.f = cl.lane & 31;
  cl.h = cl.lane >> 5;
  cl.wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wv = cl.wv;
  const int g = blockIdx.x % 3, fb = (blockIdx.x / 3) & 1, face0 = 32 * fb;
  const int64_t tile = blockIdx.x / 6, row0 = tile * TILE_FACES;
  if (row0 + face0 >= a.B) return;
  const int job = 4 * g + wv;
  f32x16 acc0[1][1], acc1[2][1], acc2[1][1], acc3[1][1], acc4[1][1];
  h8 wr0[6][1][2], wr1[6][2][2], wr2[6][1][2], wr3[6][1][2], wr4[6][1][2];
  job_pre<1, 1, ST_H0>(cl, job, acc0, wr0);                   // H0's operands fly while the latent image is copied in
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const char* src = latws + ((size_t)tile * 2 + fb) * (2 * LAT_BLOCK_BYTES);
  for (int i = tid; i < 2 * LAT_BLOCK_BYTES / 16; i += 256) {
    const int piece = i / (LAT_BLOCK_BYTES / 16), o = (i % (LAT_BLOCK_BYTES / 16)) * 16;
    *reinterpret_cast<u4*>(lds + O_LAT + piece * P_LAT + face0 * LAT_ROW_BYTES + o) = *reinterpret_cast<const u4*>(src + piece * LAT_BLOCK_BYTES + o);
  }
  __syncthreads();
  // H0_g: 3 -> 128 (K padded to 16), ReLU; wave = neuron block
  job_run<1, 1, ST_H0>(cl, job, acc0, wr0, O_LAT, P_LAT, S_LAT, 16 * g, face0);
  job_store<1, 1, ACT_RELU>(cl, acc0, O_GA, P_G128, S_G128, 32 * wv, face0, cl.hdr.inv_scale[ST_H0], fetch_hook<2, 2, 1, ST_H1>(cl, job, acc1, wr1));
  __syncthreads();
  // H1_g: 128 -> 256, ReLU; wave = two neuron blocks
  job_run<2, 1, ST_H1>(cl, job, acc1, wr1, O_GA, P_G128, S_G128, 0, face0);
  job_store<2, 1, ACT_RELU>(cl, acc1, O_GB, P_G256, S_G256, 64 * wv, face0, cl.hdr.inv_scale[ST_H1], fetch_hook<4, 1, 1, ST_H2>(cl, job, acc2, wr2));
  __syncthreads();
  // H2_g: 256 -> 128, ReLU; its output image lies over HA_g (dead since the barrier above)
  job_run<1, 1, ST_H2>(cl, job, acc2, wr2, O_GB, P_G256, S_G256, 0, face0);
  job_store<1, 1, ACT_RELU>(cl, acc2, O_GC, P_G128, S_G128, 32 * wv, face0, cl.hdr.inv_scale[ST_H2], fetch_hook<2, 1, 1, ST_H3>(cl, 2 * g + (wv & 1), acc3, wr3));
  __syncthreads();
  // H3_g: 128 -> 64, ReLU: two jobs, waves 0 and 1 (every wave fetches)
  if (wv < 2) job_run<1, 1, ST_H3>(cl, 2 * g + wv, acc3, wr3, O_GC, P_G128, S_G128, 0, face0);
  job_pre<1, 1, ST_H4>(cl, g, acc4, wr4);
  if (wv < 2) job_store<1, 1, ACT_RELU>(cl, acc3, O_GD, P_G64, S_G64, 32 * wv, face0, cl.hdr.inv_scale[ST_H3]);
  __syncthreads();
  // H4_g: 64 -> 1; neuron on accumulator row 0 = register 0 of lanes 0..31; wave 0
  if (wv == 0) {
    job_run<1, 1, ST_H4>(cl, g, acc4, wr4, O_GD, P_G64, S_G64, 0, face0);
    const int face = face0 + cl.f;
    if (cl.h == 0 && row0 + face < a.B) a.out[(row0 + face) * 3 + g] = acc4[0][0][0] * cl.hdr.inv_scale[ST_H4];
  }
}

}  // namespace hx
namespace hxs {
static int e0_k16(int F) { return (F + 2 * hx::XS_COLS - 1) / (2 * hx::XS_COLS) * (2 * hx::XS_STEPS); }   // as pack.cpp

}  // namespace hxs

size_t small_workspace_bytes(int64_t B, int F) {
  if (B <= 0 || F <= 0) return 0;
  const int64_t ntiles = (B + 63) / 64;
  const int steps = hxs::e0_k16(F) > 64 ? hxs::e0_k16(F) : 64;
  return (size_t)2 * ntiles * steps * hxs::STEP_UNITS * 16;
}

int launch_encoder_heads_f16x2_small(const float* x, int64_t ldx, const float* raw, int normalize, int64_t B, int F,
                                     const void* blob, float* out, float* latent, uint8_t* valid, void* workspace,
                                     size_t ws_bytes, int split, void* stream) {
  using namespace hxs;
  if (B == 0) return 0;
  if (ws_bytes < small_workspace_bytes(B, F) || !workspace) return fail(NLML_E_BADARG, "small-batch path: workspace too small");
  if (reinterpret_cast<uintptr_t>(workspace) & 15) return fail(NLML_E_BADARG, "small-batch path: workspace must be 16-byte aligned");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int ntiles = (int)((B + 63) / 64);
  const int k16 = e0_k16(F);
  const int buf_steps = k16 > 64 ? k16 : 64;
  h8* bufA = reinterpret_cast<h8*>(workspace);
  h8* bufB = bufA + (size_t)ntiles * buf_steps * STEP_UNITS;

  const float* src = raw ? raw : x;
  const int64_t sld = raw ? NLML_F_REFERENCE : ldx;
  const bool vec4 = F % 4 == 0 && F >= 4 && sld % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
  if (vec4)
    hipLaunchKernelGGL(prepass_kernel<true>, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, src, sld, B, F,
                       raw ? (normalize ? 1 : 0) : 0, k16, buf_steps, bufA, valid);
  else
    hipLaunchKernelGGL(prepass_kernel<false>, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, src, sld, B, F,
                       raw ? (normalize ? 1 : 0) : 0, k16, buf_steps, bufA, valid);

  // the three big layers: {stage, K16, blocks per job, jobs}; ReLU; job j covers output columns 32 * blocks * j ..
  struct S { int stage, K16, nb, jobs; };
  const S stages[3] = {{ST_E0, k16, 4, 8}, {ST_E1, 64, 4, 4}, {ST_E2, 32, 2, 4}};
  h8* in = bufA;
  h8* outb = bufB;
  for (int s = 0; s < 3; ++s) {
    LayerArgs a{};
    a.blob = blob; a.xin = in; a.xout = outb; a.B = B; a.stage = stages[s].stage; a.K16 = stages[s].K16;
    a.nb_stage = stages[s].nb; a.jobs = stages[s].jobs; a.ntiles = ntiles; a.buf_steps = buf_steps; a.act = hx::ACT_RELU;
    // which K steps run on split accumulators, exactly as the fused kernel of the blob's mode (encoder_heads_f16x2.hip):
    //   NLML_MODE_F16X2S: all of layers 0, 1, 2, layer 1's small sums added at its K midpoint as well as at its end;
    //   NLML_MODE_F16X2:  layer 1 from its K midpoint on, layer 2; layer 0 never (kernel without the second set)
    const bool e0 = stages[s].stage == ST_E0, e1 = stages[s].stage == ST_E1;
    const bool use_split = split || !e0;
    a.fold_step = (split && e1) ? 32 : 0;     // (like split_from a multiple of the split kernels' ring depths, 4 and 8)
    a.split_from = (!split && e1) ? 32 : 0;   // (a multiple of every ring depth the split kernels are launched with: 4 and 8)
    for (int j = 0; j < a.jobs; ++j) {
      a.in_step0[j] = 0;
      a.out_col0[j] = 32 * a.nb_stage * j;
    }
#ifndef HXS_RS1
#define HXS_RS1 8
#endif
#define NLML_HXS_LAUNCH(NB, R, RS)                                                                   \
  do {                                                                                               \
    if (!use_split) hipLaunchKernelGGL((layer_kernel<NB, R, 0>), grid, block, 0, st, a);             \
    else if (a.split_from == 0) hipLaunchKernelGGL((layer_kernel<NB, RS, 1>), grid, block, 0, st, a); \
    else hipLaunchKernelGGL((layer_kernel<NB, R, 2>), grid, block, 0, st, a);                        \
  } while (0)
    // blocks per wave: as many as still leave enough waves to keep the weight loads of every CU in flight
    int nbw = a.nb_stage;
    if (use_split && nbw > 2) nbw = 2;    // two accumulator sets per block: four blocks per wave would not fit the register file
    constexpr int kMinUnits = 2048;   // measured: 8 waves per CU keep enough loads in flight (256 units: 153 us at B = 2,000; 2,048: 114 us)
    while (nbw > 1 && (int64_t)ntiles * a.jobs * (a.nb_stage / nbw) < kMinUnits) nbw >>= 1;
    const int64_t units = (int64_t)ntiles * a.jobs * (a.nb_stage / nbw);
    static const bool fb_units = [] { const char* e = getenv("NLML_K2_SMALL_FB2"); return !(e && e[0] == '1'); }();   // A/B: =1 keeps both face blocks in a unit
    if (fb_units && nbw == 1 && units <= 256) {   // (wider thresholds, 512 .. 4,096 units, measured: no change)   // at most one wave per CU even so: one unit per (.., face block), see layer_kernel
      const dim3 grid((unsigned)(2 * units)), block(64);
      if (!use_split) hipLaunchKernelGGL((layer_kernel<1, 8, 0, 1>), grid, block, 0, st, a);
      else if (a.split_from == 0) hipLaunchKernelGGL((layer_kernel<1, HXS_RS1, 1, 1>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((layer_kernel<1, 8, 2, 1>), grid, block, 0, st, a);
    } else if (units < 1024) {   // fewer than four waves per CU: one wave per workgroup, deep ring
      const dim3 grid((unsigned)units), block(64);
      if (nbw == 4) hipLaunchKernelGGL((layer_kernel<4, 6, 0>), grid, block, 0, st, a);   // (four blocks per wave: layer 0 without a second set only)
      else if (nbw == 2) NLML_HXS_LAUNCH(2, 8, HXS_RS1);
      else NLML_HXS_LAUNCH(1, 8, HXS_RS1);
    } else {
      const dim3 grid((unsigned)((units + 3) / 4)), block(256);
      if (nbw == 4) hipLaunchKernelGGL((layer_kernel<4, 4, 0>), grid, block, 0, st, a);
      else if (nbw == 2) NLML_HXS_LAUNCH(2, 4, 4);
      else NLML_HXS_LAUNCH(1, 4, 4);
    }
#undef NLML_HXS_LAUNCH
    h8* t = in; in = outb; outb = t;
  }
  {
    hx::Args ta{};
    ta.B = B; ta.F = F; ta.blob = blob; ta.out = out; ta.latent = latent; ta.valid = nullptr;
    ta.x = src; ta.ldx = sld; ta.norm = raw ? (normalize ? 1 : 0) : 0;   // the tail's slow path re-reads the face's input
    if (split) {   // E3..E5, then the three heads as workgroups of their own; the latent image passes through the free buffer
      hipLaunchKernelGGL(hx::tail_encoder_kernel, dim3((unsigned)(2 * ntiles)), dim3(256), 0, st, ta, (const h8*)in, buf_steps, reinterpret_cast<char*>(outb));
      hipLaunchKernelGGL(hx::head_kernel, dim3((unsigned)(6 * ntiles)), dim3(256), 0, st, ta, reinterpret_cast<const char*>(outb));
    } else {   // (the fast mode keeps the one-launch tail: its out-of-range faces are redone inside it)
      hipLaunchKernelGGL(tail_kernel<64>, dim3((unsigned)(2 * ntiles)), dim3(256), 0, st, ta, (const h8*)in, buf_steps);
    }
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
  if (split)   // the strict-fast mode's f32 re-evaluation launch, as behind its fused kernel (encoder_heads_f16x2_w8.hip)
    return launch_encoder_heads_f32(x, ldx, raw, normalize, B, F, static_cast<const char*>(blob) + strict_f32_image_offset(F), out, latent,
                                    nullptr, nullptr, nullptr, stream, STRICT_INKERNEL_RESCUE_MAX);
  return 0;
}

}  // namespace nlml
