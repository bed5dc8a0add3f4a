// encoder_heads_f16x2_wide.hip -- K2 for LARGE batches in the strict-fast mode (NLML_MODE_F16X2S): the three big layers as one launch
// each over 128-FACE tiles, eight waves per workgroup, activations staged through LDS.
//
// Why: the fused eight-wave kernel (encoder_heads_f16x2_w8.hip) walks a 64-face tile through the network on one
// CU and streams the 9.6 MB of weights through that CU once per tile; at the 31-35 B/clk a CU takes in from L2 that stream, not
// the matrix pipe, sets the tile's time (9.6 MB / 33 B/clk = 290 k of a tile's 330 k cycles).  Split accumulators (what makes the
// mode strict) cap a pass at 32,768 outputs per workgroup whatever the wave arrangement, so the only way to halve the bytes per
// face is the tile SHAPE: 128 faces x 256 neurons per pass instead of 64 x 512 -- 24 KB of operands per K step instead of 36 KB
// for the same 12 MFMAs per wave -- and a 128-face tile's layer outputs (512 KB after layer 0) do not fit LDS, so the layers
// hand over through global memory in the layer-per-launch path's fragment order (encoder_heads_f16x2_small.hip) and each layer is
// a launch of its own.
//
//   * workgroup = 512 threads = 8 waves (two per SIMD), tile = 128 faces = 4 MFMA column blocks ("face blocks");
//   * a pass = 256 output neurons = 8 neuron blocks, one per wave, all four face blocks: 64 + 64 accumulator registers (split);
//     layer 0: 4 passes, layer 1: 2 (small sums folded at its K midpoint and its end, like the fused kernel), layer 2: 1;
//   * weights: private to a wave, global -> VGPR through a four-slot ring three K steps ahead (2 KiB per wave and K step);
//   * activations: one K step of the tile = 8 fragments of 1 KiB ((face block, piece) x 64 lanes x 16 B) in LDS, two buffers of four K
//     steps, ONE barrier per four K steps; every wave reads all eight (8 ds_read_b128 per 12 MFMAs), each fetched just in time for the
//     three MFMAs of its face block;
//   * layer 0, pass 0 stages x itself -- f32 rows (or raw landmarks with the IPD normalisation in f64, FeatureExtractor.py:30-66,
//     exactly K1's arithmetic) -> hi/lo f16 -> LDS, one piece per MFMA slot -- and stores the fragments it stages to the scratch
//     buffer, from which passes 1-3 (and nothing else) copy them: the normalisation and the split run once per face;
//   * every other pass copies ready-made fragments global -> VGPR -> LDS, one 1-KiB fragment per wave and K step.
//
// Arithmetic is the fused kernel's, operation for operation: same blob, same split, per accumulator the same K-ascending sequence of
// the same three MFMAs (reference: NLML_HPE_Model_Builder.py:33-53) -- the results are bit-identical to the fused kernel's and to
// the layer-per-launch path's (tests/test_gpu_parity.py).  The tail (E3, E4, E5, heads) is the fused kernel's tail_stages() as a
// launch of its own per 64-face tile; the f32 re-evaluation launch follows as behind every strict-fast path.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/nlml_hpe.h"
#include "abi_internal.h"
#include "encoder_heads_f16x2_dev.h"
#include "layout.h"

namespace nlml {
namespace hxw {

using hx::f32x16;
using hx::f32x4;
using hx::h8;
using hx::ACT_RELU;

constexpr int WFACES = 128;                    // faces per workgroup tile
constexpr int WFB = 4;                         // MFMA column blocks per tile
constexpr int SLOT_BYTES = 1024;               // one fragment: 64 lanes x 16 B
constexpr int STEP_BYTES = 2 * WFB * SLOT_BYTES;   // one K step of the tile: slot = face block * 2 + piece
constexpr int GROUP_STEPS = 4;
constexpr int GROUP_BYTES = GROUP_STEPS * STEP_BYTES;   // 32 KiB
constexpr int LDS_W = 2 * GROUP_BYTES;                  // two buffers
constexpr int STEP_UNITS = 2 * 2 * 64;         // h8 units per (64-face tile, K step) in global memory (encoder_heads_f16x2_small.hip)

typedef __attribute__((address_space(3))) char LdsB;
typedef __attribute__((address_space(3))) h8 LdsH8;
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u2 LdsU2;

struct WArgs {
  const void* blob;
  const float* x;       // layer 0: f32 rows (features, or raw landmarks when norm)
  int64_t ldx, B;
  int F, norm;
  uint8_t* valid;
  const h8* xin;        // input fragments (layer 0: the scratch its pass 0 fills)
  h8* xscr;             // layer 0, pass 0: where the staged fragments go (== xin)
  h8* xout;             // output fragments
  int in_steps;         // K steps per 64-face tile in xin / xout (tile stride)
  int out_steps;
  int k16;              // K steps of this layer
  unsigned long long* stamps;   // -DWIDE_STAMPS diagnostic build only (tools/wide_stamps.py): s_memtime per wave at the pass boundaries
};

#ifdef WIDE_STAMPS
#define WST(i)                                                                                                       \
  do {                                                                                                               \
    if (a.stamps && c.lane == 0) a.stamps[((size_t)blockIdx.x * 8 + c.wv) * 32 + 8 * pass + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define WST(i) do { } while (0)
#endif

__device__ __forceinline__ double div_ipd(double n, double d, double y) {   // == IEEE n / d for these operands (K1)
  const double q = n * y;
  const double r = fma(-q, d, n);
  return fma(r, y, q);
}

// In LDS the 16-byte piece of logical lane (f, h) of a fragment of K step t (mod 4) sits at lane position ((f ^ (4h + t)) + 32h): a
// bijection per fragment, so the MFMA operand reads (ds_read_b128, 16-lane groups) and the copy path's ds_write_b128 (8-lane groups)
// stay conflict-free, and the staging path's 8-byte writes -- four rows x (h, half) per 16-lane group -- do too (plain lane order:
// 2-way conflicts).
__device__ __forceinline__ int swz16(int f, int h, int t) { return ((f ^ (4 * h + t)) + 32 * h) * 16; }

// One K step: 12 MFMAs, face block by face block -- (w_lo, x_hi) and (w_hi, x_lo) into accS, (w_hi, x_hi) into acc, the order
// step_fine gives every accumulator -- and behind a face block's last MFMA the LDS reads of ITS operands for the next step (the x
// operands are single-buffered: 32 registers).  wn: the ring slot for the weights D steps ahead; extra(m): the caller's piece of
// staging / copying work for slot m.
template <typename Extra, typename Pre>
__device__ __forceinline__ void wstep(f32x16 (&acc)[WFB], f32x16 (&accS)[WFB], const h8 (&wc)[2], h8 (&xh)[WFB], h8 (&xl)[WFB], h8 (&wn)[2],
                                      const h8* __restrict__ wp, const LdsB* xnext, Pre pre, Extra extra) {
#pragma unroll
  for (int fb = 0; fb < WFB; ++fb) {
    accS[fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc[1], xh[fb], accS[fb], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (fb == 0) wn[0] = wp[0];
    if (fb == 2) wn[1] = wp[64];
    extra(3 * fb);
    __builtin_amdgcn_sched_barrier(0);
    accS[fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc[0], xl[fb], accS[fb], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (fb == 0) pre();          // (the group's barrier, in its fourth step: before the first read of the other buffer)
    extra(3 * fb + 1);
    __builtin_amdgcn_sched_barrier(0);
    acc[fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc[0], xh[fb], acc[fb], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    xh[fb] = *reinterpret_cast<const LdsH8*>(xnext + (2 * fb) * SLOT_BYTES);
    xl[fb] = *reinterpret_cast<const LdsH8*>(xnext + (2 * fb + 1) * SLOT_BYTES);
    extra(3 * fb + 2);
    __builtin_amdgcn_sched_barrier(0);
  }
}

struct Lane {
  int tid, lane, f, h, wv;
};

// One pass: neuron blocks 8*pass .. 8*pass+7 of STAGE (wave wv takes block 8*pass + wv) for the 128 faces of tile `tile`.
// STAGED: the activations are staged from a.x (layer 0, pass 0); else copied from the fragments at a.xin.
template <int STAGE, bool STAGED, bool NORM>
__device__ __forceinline__ void wide_pass(const WArgs& a, const hx::HdrRegs& hdr, LdsB* lds, const Lane& c, int pass, int64_t tile) {
  constexpr int NBS = hx::kStages[STAGE].nb;                 // neuron blocks per job of this stage
  constexpr int WSTEP = NBS * 2 * 64;                        // h8 units between consecutive K steps of a job's weight stream
  const int K16 = STAGE == ST_E0 ? a.k16 : hx::kStages[STAGE].k8;
  const int ngroups = K16 / GROUP_STEPS;                     // even (launcher)
  const int gb = 8 * pass + c.wv, job = gb / NBS, nb = gb % NBS;
  const h8* blob8 = reinterpret_cast<const h8*>(a.blob);
  const f32x4* blob4 = reinterpret_cast<const f32x4*>(a.blob);

  WST(0);
  f32x16 acc[WFB], accS[WFB];
  {
    f32x16 t[1][WFB];
    hx::load_bias<1, WFB>(t, blob4 + hdr.b_off(STAGE) + job * (NBS * 8) + nb * 8, c.h);
#pragma unroll
    for (int fb = 0; fb < WFB; ++fb) {
      acc[fb] = t[0][fb];
#pragma unroll
      for (int q = 0; q < 16; ++q) accS[fb][q] = 0.0f;
    }
  }
  const h8* w = blob8 + hdr.w_off(STAGE) + (size_t)job * hdr.job_w16(STAGE) + (size_t)nb * 128 + c.lane;   // + step * WSTEP; lo piece at +64
  auto wfrag = [&](int s) { return w + (size_t)(s < K16 ? s : K16 - 1) * WSTEP; };

  // ---- the activation feed.  ONE register set per path, recycled piece by piece: what is held for group g+1 is written to LDS during
  // group g, and each register is re-loaded with its counterpart of group g+2 as soon as it has been consumed -- so every load is
  // issued almost a whole group (~3 k cycles) before its use at a quarter of a group's data in registers.
  // copy path: wave wv moves fragment slot wv of every K step: face block wv >> 1, piece wv & 1 = 64-face tile 2*tile + (wv >> 2), unit wv & 3
  const h8* src = a.xin + ((size_t)(2 * tile + (c.wv >> 2)) * a.in_steps) * STEP_UNITS + (size_t)(c.wv & 3) * 64 + c.lane;   // + step * STEP_UNITS
  h8 cp[GROUP_STEPS];
  // staging path: thread = (row tid >> 2 of the tile, c4 = tid & 3): the four float4 at columns 64 g + 16 i + 4 c4, i = 0..3 (K step i of group g)
  const int srow = c.tid >> 2, c4 = c.tid & 3;
  f32x4 xs[GROUP_STEPS];
  const float* p = nullptr;
  bool live = true;
  double ipd = 1.0, rcp = 1.0, ra = 0.0, rb = 0.0, rc = 0.0;
  unsigned nzbits = 0u;
  // the staged fragments also go to the scratch buffer (plain lane order) for passes 1-3: this thread's 8-byte pieces
  u2* scr = nullptr;
  int st_off = 0;   // this thread's byte offset inside a (K step, face block) pair of LDS slots, without the step's swizzle
  if (STAGED) {
    int64_t r = tile * WFACES + srow;
    live = r < a.B;
    r = live ? r : a.B - 1;
    p = a.x + r * a.ldx;
    if (NORM) {   // exactly K1's arithmetic: the f32 value the reference feeds the network, bit for bit
      const double dx = (double)p[99] - (double)p[789], dy = (double)p[100] - (double)p[790], dz = (double)p[101] - (double)p[791];
      ipd = sqrt(fma(dz, dz, fma(dy, dy, dx * dx)));
      if (ipd == 0.0) ipd = 1e-6;
      rcp = 1.0 / ipd;
      const double x0 = (double)p[3], y0 = (double)p[4], z0 = (double)p[5];
      const int ph = c4 % 3;   // coordinate of column 4 c4 (64 g + 16 i + 4 c4 + e is coordinate (g + i + c4 + e) % 3)
      ra = ph == 0 ? x0 : (ph == 1 ? y0 : z0);
      rb = ph == 0 ? y0 : (ph == 1 ? z0 : x0);
      rc = ph == 0 ? z0 : (ph == 1 ? x0 : y0);
    }
    // column 16 i + 4 c4 + e of a group = K step i, lane half hh = c4 >> 1, elements 4 (c4 & 1) .. + 3 of the lane's 8
    const int hh = c4 >> 1, half = c4 & 1, fb = srow >> 5, fr = srow & 31;
    scr = reinterpret_cast<u2*>(a.xscr + ((size_t)(2 * tile + (fb >> 1)) * a.in_steps) * STEP_UNITS + (size_t)((fb & 1) * 2) * 64 + fr + 32 * hh) + half;
    st_off = (2 * fb) * SLOT_BYTES + 8 * half;
  }
  const int F = a.F;
  auto feed_load = [&](int g, int i) {   // issue the global load of K step i of group g
    const int gg = g < ngroups ? g : ngroups - 1;
    if (STAGED) {
      const int k = 64 * gg + 16 * i + 4 * c4;
      xs[i] = *reinterpret_cast<const f32x4*>(p + (k < F ? k : F - 4));
    } else {
      cp[i] = src[(size_t)(GROUP_STEPS * gg + i) * STEP_UNITS];
    }
  };
  // staging pieces of float4 i of group g, written into buffer `boff`:
  //   piece 0..3: normalise element e (or zero it beyond F); 4, 5: split elements (0,1) / (2,3); 6, 7: the hi / lo 8-byte stores (LDS and scratch)
  unsigned pend_hi[2], pend_lo[2];
  const int fr_ = srow & 31, hh_ = c4 >> 1;
  auto stage_piece = [&](int g, int i, int piece, int boff) {
    f32x4& v = xs[i];
    if (piece < 4) {
      const int e = piece;
      if (piece == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(v[q]));   // the loads have landed: nothing below is hoisted above
      }
      if (NORM) {
        const int t = (i + e) % 3;
        const double rr = t == 0 ? ra : (t == 1 ? rb : rc);
        v[e] = (float)div_ipd((double)v[e] - rr, ipd, rcp);
      }
      const int k = 64 * g + 16 * i + 4 * c4;
      if (k >= F) v[e] = 0.0f;                    // (F % 4 == 0: a float4 lies inside the row or beyond it)
    } else if (piece < 6) {
      const int j = piece - 4;
      nzbits |= (__float_as_uint(v[2 * j]) | __float_as_uint(v[2 * j + 1])) & 0x7fffffffu;
      hx::split2(v[2 * j], v[2 * j + 1], pend_hi[j], pend_lo[j]);
    } else {
      const int pc = piece - 6;
      const u2 val = pc == 0 ? u2{pend_hi[0], pend_hi[1]} : u2{pend_lo[0], pend_lo[1]};
      *reinterpret_cast<LdsU2*>(lds + boff + i * STEP_BYTES + pc * SLOT_BYTES + st_off + swz16(fr_, hh_, i)) = val;
      const int ks = GROUP_STEPS * (g < ngroups ? g : ngroups - 1) + i;
      if (g < ngroups) scr[((size_t)ks * STEP_UNITS + pc * 64) * 2] = val;   // (u2 units: an h8 is two of them)
    }
  };
  auto rotate_refs = [&]() {   // next group: columns + 64 => coordinate + 1
    const double t0 = ra; ra = rb; rb = rc; rc = t0;
  };
  auto copy_write = [&](int i, int boff) {   // fragment (K step i of the group) -> LDS
    *reinterpret_cast<LdsH8*>(lds + boff + i * STEP_BYTES + c.wv * SLOT_BYTES + swz16(c.f, c.h, i)) = cp[i];
  };

  // ---- prologue: group 0 in LDS, group 1 in flight, the weight ring filled
#pragma unroll
  for (int i = 0; i < GROUP_STEPS; ++i) feed_load(0, i);
#ifndef WIDE_D
#define WIDE_D 3
#endif
  constexpr int R = 8, D = WIDE_D;   // ring: the loop body is two groups = 8 K steps, so a step's slot (step % 8) is static; D steps ahead
  static_assert(D >= 1 && D < R, "ring depth");
  h8 wr[R][2];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    wr[d][0] = wfrag(d)[0];
    wr[d][1] = wfrag(d)[64];
  }
  if (STAGED) {
#pragma unroll
    for (int i = 0; i < GROUP_STEPS; ++i) {
#pragma unroll
      for (int pc = 0; pc < 8; ++pc) stage_piece(0, i, pc, 0);
      feed_load(1, i);
    }
    if (NORM) rotate_refs();
  } else {
#pragma unroll
    for (int i = 0; i < GROUP_STEPS; ++i) {
      copy_write(i, 0);
      feed_load(1, i);
    }
  }
  __syncthreads();
  WST(1);

  // reader: lane (f, h) reads its piece of slot j of K step t at  buffer + t * STEP_BYTES + j * SLOT_BYTES + swz16(f, h, t)
  const LdsB* rd[GROUP_STEPS];
#pragma unroll
  for (int t = 0; t < GROUP_STEPS; ++t) rd[t] = lds + t * STEP_BYTES + swz16(c.f, c.h, t);
  h8 xh[WFB], xl[WFB];
#pragma unroll
  for (int fb = 0; fb < WFB; ++fb) {
    xh[fb] = *reinterpret_cast<const LdsH8*>(rd[0] + (2 * fb) * SLOT_BYTES);
    xl[fb] = *reinterpret_cast<const LdsH8*>(rd[0] + (2 * fb + 1) * SLOT_BYTES);
  }

  // group g: reads buffer `cur`, writes group g+1 (held in the register set) into buffer `nxt` during its first steps -- each register
  // re-loaded with group g+2's data right after its last use
  auto group = [&](int g, auto parc, int cur, int nxt) {
    constexpr int PAR = decltype(parc)::value;   // g & 1
#pragma unroll
    for (int t = 0; t < GROUP_STEPS; ++t) {
      const int ks = GROUP_STEPS * g + t;
      const LdsB* xnext = t < 3 ? rd[t + 1] + cur : rd[0] + nxt;
      wstep(acc, accS, wr[(GROUP_STEPS * PAR + t) % R], xh, xl, wr[(GROUP_STEPS * PAR + t + D) % R], wfrag(ks + D), xnext,
            [&]() {
              if (t == 3) __syncthreads();   // group g+1 is complete in LDS; nobody reads group g's buffer any more
            },
            [&](int m) {
              const int s = 12 * t + m;      // slot within the group, 0 .. 47 (the barrier stands between slots 36 and 37)
              if (STAGED) {
                if (s < 32) {
                  stage_piece(g + 1, s / 8, s % 8, nxt);
                  if (s % 8 == 7) feed_load(g + 2, s / 8);
                }
                if (s == 32 && NORM) rotate_refs();
              } else {
                if (t < 2 && (m == 1 || m == 7)) copy_write(2 * t + m / 6, nxt);
                if (t < 2 && (m == 2 || m == 8)) feed_load(g + 2, 2 * t + m / 6);
              }
            });
    }
  };
  auto run_groups = [&](int g0, int g1) {
    for (int g = g0; g < g1; g += 2) {
      group(g, std::integral_constant<int, 0>{}, 0, GROUP_BYTES);
      group(g + 1, std::integral_constant<int, 1>{}, GROUP_BYTES, 0);
    }
  };
  if (STAGE == ST_E1) {   // layer 1: the small sums join the big ones at its K midpoint too (the fused kernel's two K halves); the fold
    run_groups(0, ngroups / 2);   // stands BETWEEN two loops (a test inside one makes every step a basic block, encoder_heads_f16x2_small.hip)
#pragma unroll
    for (int fb = 0; fb < WFB; ++fb) {
      acc[fb] += accS[fb];
#pragma unroll
      for (int q = 0; q < 16; ++q) accS[fb][q] = 0.0f;
    }
    run_groups(ngroups / 2, ngroups);
  } else {
    run_groups(0, ngroups);
  }
#pragma unroll
  for (int fb = 0; fb < WFB; ++fb) acc[fb] += accS[fb];
  WST(2);

  if (STAGED && a.valid) {   // all-zero feature row == "no face" (FeatureExtractor.py:105-106); the 4 lanes of a row are neighbours
    const unsigned long long m = __ballot(nzbits != 0u);
    if (c4 == 0 && live) a.valid[tile * WFACES + srow] = ((m >> (c.lane & 60)) & 0xFull) ? 1 : 0;
  }

  // ---- epilogue: accumulators * inv -> ReLU -> hi/lo -> the next layer's input fragments (the layer-per-launch path's order)
  const float inv = hdr.inv_scale[STAGE];
#pragma unroll
  for (int fb = 0; fb < WFB; ++fb) {
    h8* tile_out = a.xout + ((size_t)(2 * tile + (fb >> 1)) * a.out_steps) * STEP_UNITS + (size_t)((fb & 1) * 2) * 64 + c.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = 32 * gb + 8 * q + 4 * c.h;   // this lane's 4 neurons n .. n+3 = K index of the next layer
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = hx::activate<ACT_RELU>(acc[fb][4 * q + e] * inv);
      unsigned hi[2], lo[2];
      hx::split2(v[0], v[1], hi[0], lo[0]);
      hx::split2(v[2], v[3], hi[1], lo[1]);
      h8* frag = tile_out + (size_t)(n >> 4) * STEP_UNITS + 32 * ((n >> 3) & 1);
      u2* d = reinterpret_cast<u2*>(reinterpret_cast<_Float16*>(frag) + (n & 7));
      d[0] = u2{hi[0], hi[1]};
      d[64 * 2] = u2{lo[0], lo[1]};   // the lo fragment: 64 h8 = 128 u2 further
    }
  }
  WST(3);
}

template <int STAGE, bool NORM>
__global__ __launch_bounds__(512) void wide_layer_kernel(WArgs a) {
  __shared__ __attribute__((aligned(16))) char lds_[LDS_W];
  LdsB* lds = (LdsB*)lds_;
  Lane c;
  c.tid = threadIdx.x;
  c.lane = c.tid & 63;
  c.f = c.lane & 31;
  c.h = c.lane >> 5;
  c.wv = __builtin_amdgcn_readfirstlane(c.tid >> 6);
  const hx::HdrRegs hdr = hx::load_hdr(reinterpret_cast<const Header*>(a.blob));
  const int64_t tile = blockIdx.x;
  constexpr int NPASS = hx::kStages[STAGE].nb * hx::kStages[STAGE].jobs / 8;
  if (STAGE == ST_E0) {
    wide_pass<STAGE, true, NORM>(a, hdr, lds, c, 0, tile);
    __syncthreads();   // (the scratch stores of every wave are complete -- s_waitcnt vmcnt(0) -- before any wave copies them back)
#pragma unroll 1
    for (int pass = 1; pass < NPASS; ++pass) wide_pass<STAGE, false, false>(a, hdr, lds, c, pass, tile);
  } else {
#pragma unroll 1
    for (int pass = 0; pass < NPASS; ++pass) wide_pass<STAGE, false, false>(a, hdr, lds, c, pass, tile);
  }
}

}  // namespace hxw

namespace hx {
// ---- the tail (E3, E4, E5, heads) of one 64-face tile as a launch of its own: E2's output fragments -> the H3 LDS image -> the fused
// kernel's tail_stages()
__global__ __launch_bounds__(256, 1) void tail64_kernel(Args a, const h8* __restrict__ xin, int buf_steps) {
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
  const int64_t tile = blockIdx.x, row0 = tile * TILE_FACES;
  f32x16 acc3[1][2];
  h8 wr3[ring_slots(1, 2)][1][2];
  tail_pre_e3<false>(c, acc3, wr3);                           // E3's bias and first weights fly while the image is copied
  // fragment (step, fb, piece, lane) -> image row 32*fb + (lane & 31), columns 16*step + 8*(lane >> 5) .. +7 of plane `piece`
  const h8* src = xin + (size_t)tile * buf_steps * hxw::STEP_UNITS;
  for (int i = tid; i < 16 * 256; i += 256) {                 // E2's output: 256 columns = 16 K steps
    const int l = i & 63, piece = (i >> 6) & 1, fb = (i >> 7) & 1, step = i >> 8;
    const int face = 32 * fb + (l & 31), k = 16 * step + 8 * (l >> 5);
    *reinterpret_cast<h8*>(lds + O_H3 + piece * P_H3 + (face * S_H3 + k) * 2) = src[(size_t)step * hxw::STEP_UNITS + (fb * 2 + piece) * 64 + l];
  }
  __syncthreads();
  tail_stages<false, STRICT_INKERNEL_RESCUE_MAX>(c, a, row0, acc3, wr3);
}
}  // namespace hx

static int e0_k16(int F) { return (F + 2 * hx::XS_COLS - 1) / (2 * hx::XS_COLS) * (2 * hx::XS_STEPS); }   // as pack.cpp

// The wide path takes: F % 4 == 0 and 16-byte aligned rows (the shipped 1,404-column layout does), an even number of 64-column groups
// in layer 0; everything else goes to the fused kernel.
bool wide_supported(const float* x, int64_t ldx, int F) {
  return F % 4 == 0 && F >= 4 && ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (e0_k16(F) / hxw::GROUP_STEPS) % 2 == 0;
}

size_t wide_workspace_bytes(int64_t B, int F) {
  if (B <= 0 || F <= 0) return 0;
  const int64_t ntiles = (B + hxw::WFACES - 1) / hxw::WFACES * 2;   // 64-face tiles, whole 128-face tiles
  const int steps = e0_k16(F) > 64 ? e0_k16(F) : 64;
  return (size_t)2 * ntiles * steps * hxw::STEP_UNITS * 16;
}

int launch_encoder_heads_f16x2_wide(const float* x, int64_t ldx, const float* raw, int normalize, int64_t B, int F,
                                    const void* blob, float* out, float* latent, uint8_t* valid, void* workspace,
                                    size_t ws_bytes, void* stream) {
  using namespace hxw;
  if (B == 0) return 0;
  if (ws_bytes < wide_workspace_bytes(B, F) || !workspace) return fail(NLML_E_BADARG, "wide path: workspace too small");
  if (reinterpret_cast<uintptr_t>(workspace) & 15) return fail(NLML_E_BADARG, "wide path: workspace must be 16-byte aligned");
  const float* src = raw ? raw : x;
  const int64_t sld = raw ? NLML_F_REFERENCE : ldx;
  if (!wide_supported(src, sld, F)) return fail(NLML_E_BADARG, "wide path: unsupported input layout");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int64_t nt128 = (B + WFACES - 1) / WFACES, nt64 = nt128 * 2;
  const int k16 = e0_k16(F);
  const int buf_steps = k16 > 64 ? k16 : 64;
  h8* bufA = reinterpret_cast<h8*>(workspace);                       // layer 0's split input, then layer 1's output
  h8* bufB = bufA + (size_t)nt64 * buf_steps * STEP_UNITS;           // layer 0's output, then layer 2's

  WArgs a{};
#ifdef WIDE_STAMPS
  if (const char* e = getenv("NLML_WIDE_STAMPS_PTR")) a.stamps = reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0));
  const int stamp_stage = getenv("NLML_WIDE_STAMPS_STAGE") ? atoi(getenv("NLML_WIDE_STAMPS_STAGE")) : 0;
  unsigned long long* const stamp_buf = a.stamps;
#define WIDE_STAMP_SEL(st_) a.stamps = (stamp_stage == (st_)) ? stamp_buf : nullptr
#else
#define WIDE_STAMP_SEL(st_) do { } while (0)
#endif
  a.blob = blob; a.x = src; a.ldx = sld; a.B = B; a.F = F; a.norm = raw ? (normalize ? 1 : 0) : 0; a.valid = valid;
  a.in_steps = buf_steps; a.out_steps = buf_steps;
  const dim3 grid((unsigned)nt128), block(512);
  a.xin = bufA; a.xscr = bufA; a.xout = bufB; a.k16 = k16;
  WIDE_STAMP_SEL(0);
  if (a.norm) hipLaunchKernelGGL((wide_layer_kernel<ST_E0, true>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((wide_layer_kernel<ST_E0, false>), grid, block, 0, st, a);
  a.xin = bufB; a.xscr = nullptr; a.xout = bufA; a.k16 = 64;
  WIDE_STAMP_SEL(1);
  hipLaunchKernelGGL((wide_layer_kernel<ST_E1, false>), grid, block, 0, st, a);
  a.xin = bufA; a.xout = bufB; a.k16 = 32;
  WIDE_STAMP_SEL(2);
  hipLaunchKernelGGL((wide_layer_kernel<ST_E2, false>), grid, block, 0, st, a);
  {
    hx::Args ta{};
    ta.B = B; ta.F = F; ta.blob = blob; ta.out = out; ta.latent = latent; ta.valid = nullptr;
    ta.x = src; ta.ldx = sld; ta.norm = a.norm;
    hipLaunchKernelGGL(hx::tail64_kernel, dim3((unsigned)((B + 63) / 64)), dim3(256), 0, st, ta, (const h8*)bufB, buf_steps);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
  // the strict-fast mode's f32 re-evaluation launch, as behind its fused kernel (encoder_heads_f16x2_w8.hip)
  return launch_encoder_heads_f32(x, ldx, raw, normalize, B, F, static_cast<const char*>(blob) + strict_f32_image_offset(F), out, latent,
                                  nullptr, nullptr, nullptr, stream, STRICT_INKERNEL_RESCUE_MAX);
}

}  // namespace nlml
