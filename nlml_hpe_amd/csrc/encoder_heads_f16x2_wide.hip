// encoder_heads_f16x2_wide.hip -- K2 for LARGE batches in the strict-fast mode (NLML_MODE_F16X2S): the three big layers over 128-FACE
// tiles, eight waves per workgroup, activations staged through LDS, layer outputs handed over through global memory.
//
// Why: the fused eight-wave kernel (encoder_heads_f16x2_w8.hip) walks a 64-face tile through the network on one
// CU and streams the 9.6 MB of weights through that CU once per tile; at the 31-35 B/clk a CU takes in from L2 that stream, not
// the matrix pipe, sets the tile's time (9.6 MB / 33 B/clk = 290 k of a tile's 330 k cycles).  Split accumulators (what makes the
// mode strict) cap a pass at 32,768 outputs per workgroup whatever the wave arrangement, so the only way to cut the bytes per
// face is the tile SHAPE: 128 faces x 256 neurons per pass instead of 64 x 512 -- 24 KB of operands per K step instead of 36 KB
// for the same 12 MFMAs per wave -- and a 128-face tile's layer outputs (512 KB after layer 0) do not fit LDS, so the layers
// hand over through global memory (L2 / Infinity Cache when the same workgroup reads them back).
//
//   * workgroup = 512 threads = 8 waves (two per SIMD), tile = 128 faces = 4 MFMA column blocks ("face blocks");
//   * a pass = 256 output neurons = 8 neuron blocks, one per wave, all four face blocks: 64 + 64 accumulator registers (split);
//     layer 0: 4 passes, layer 1: 2 (small sums folded at its K midpoint and its end, like the fused kernel), layer 2: 1;
//   * weights: private to a wave, global -> VGPR through a ring D K steps ahead (2 KiB per wave and K step) that runs on into the next
//     pass's weights;
//   * activations: one K step of the tile = 8 fragments of 1 KiB ((face block, piece) x 64 lanes x 16 B) in LDS, two buffers of four K
//     steps, ONE barrier per four K steps; every wave reads all eight per K step (8 ds_read_b128 per 12 MFMAs);
//   * handover format ("quad-major" f32): [tile][k / 4][face 0..127][4] -- an accumulator register quad of the producer (4 consecutive
//     neurons of one face) is one 16-byte store, and one 16-byte load per thread and K step feeds the consumer, which splits into
//     hi/lo f16 on the way to LDS (3 VALU instructions per pair, one piece per MFMA slot);
//   * layer 0, pass 0 stages x itself -- f32 rows (or raw landmarks with the IPD normalisation in f64, FeatureExtractor.py:30-66,
//     exactly K1's arithmetic) -- and stores the normalised f32 values quad-major to the scratch buffer, from which passes 1-3 feed like
//     the hidden layers: the normalisation runs once per face;
//   * the activation feed wraps around at a pass's end (the last group stages group 0 again), so consecutive passes of a layer run as
//     ONE pipeline: a pass boundary is the fold, the epilogue stores and the bias -- no drain, no refill.
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

#ifndef WIDE_D
#define WIDE_D 3                               // weight ring: K steps ahead
#endif
#ifndef WIDE_FSETS
#define WIDE_FSETS 1                           // register sets of the activation feed: 1 = loads one group ahead, 2 = two groups ahead
#endif
constexpr int WFACES = 128;                    // faces per workgroup tile
constexpr int WFB = 4;                         // MFMA column blocks per tile
constexpr int SLOT_BYTES = 1024;               // one fragment: 64 lanes x 16 B
constexpr int STEP_BYTES = 2 * WFB * SLOT_BYTES;   // one K step of the tile: slot = face block * 2 + piece
constexpr int GROUP_STEPS = 4;
constexpr int GROUP_BYTES = GROUP_STEPS * STEP_BYTES;   // 32 KiB
constexpr int LDS_W = 2 * GROUP_BYTES;                  // two buffers

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
  f32x4* bufA;          // quad-major: layer 0's normalised input (scratch), then layer 1's output
  f32x4* bufB;          // layer 0's output, then layer 2's
  int k16_e0;           // K steps of layer 0
  int stamp_stage_;     // -DWIDE_STAMPS diagnostic build only (tools/wide_stamps.py): the stage whose passes are stamped ...
  unsigned long long* stamps;   // ... s_memtime per wave at the pass boundaries
};

#ifdef WIDE_STAMPS
#define WST(i)                                                                                                       \
  do {                                                                                                               \
    if (a.stamps && a.stamp_stage_ == STAGE && c.lane == 0)                                                          \
      a.stamps[((size_t)tile * 8 + c.wv) * 32 + 8 * pass + (i)] = __builtin_amdgcn_s_memtime();               \
  } while (0)
#else
#define WST(i) do { } while (0)
#endif

__device__ __forceinline__ double div_ipd(double n, double d, double y) {   // == IEEE n / d for these operands (K1)
  const double q = n * y;
  const double r = fma(-q, d, n);
  return fma(r, y, q);
}

template <int I, int N, typename Fn>
__device__ __forceinline__ void static_for(Fn&& fn) {
  if constexpr (I < N) {
    fn(std::integral_constant<int, I>{});
    static_for<I + 1, N>(fn);
  }
}

struct Lane {
  int tid, lane, f, h, wv;
};

// In LDS the 16-byte piece of logical lane (f, h) of a fragment of K step t (mod 4) sits at lane position ((f ^ (4h + t)) + 32h): a
// bijection per fragment, so the MFMA operand reads (ds_read_b128, 16-lane groups) stay conflict-free, and so do both writers' 8-byte
// stores: the quad-major feed's 16-lane groups are 8 faces x 2 halves (128 contiguous bytes under any XOR below 8), layer 0's x
// staging has four rows x (h, half) per group (plain lane order: 2-way conflicts).
__device__ __forceinline__ int swz16(int f, int h, int t) { return ((f ^ (4 * h + t)) + 32 * h) * 16; }

// quads (16 bytes) of one 128-face tile in the two hand-over buffers: bufA holds layer 0's input (4 * k16 quads per face) or layer 1's
// output (128), bufB layer 0's output (256) or layer 2's (64)
__host__ __device__ inline size_t tile_quads_A(int k16_e0) { return (size_t)(4 * k16_e0 > 128 ? 4 * k16_e0 : 128) * WFACES; }
constexpr size_t TILE_QUADS_B = (size_t)256 * WFACES;

// Number of 256-neuron passes and the quad-major geometry of a stage
template <int STAGE> struct StageGeo {
  static constexpr int NBS = hx::kStages[STAGE].nb;                  // neuron blocks per job
  static constexpr int NPASS = NBS * hx::kStages[STAGE].jobs / 8;
  static constexpr int WSTEP = NBS * 2 * 64;                         // h8 units between consecutive K steps of a job's weight stream
  static constexpr int KQ_OUT = NBS * hx::kStages[STAGE].jobs * 8;   // output neurons / 4
};

// One stage (layer) for the 128 faces of `tile`: all its passes as one pipeline.
// SPLITP: the workgroup runs ONE pass, `pass0` (the grid deals the passes of a tile to workgroups of one XCD: they walk the same input at
// the same time and share it through that XCD's L2); layer 0 then stages x in every pass and writes no scratch.
template <int STAGE, bool NORM, int D, bool SPLITP>
__device__ __forceinline__ void wide_stage(const WArgs& a, LdsB* lds, const Lane& c, int64_t tile, int pass0) {
  using G = StageGeo<STAGE>;
  constexpr int NBS = G::NBS, WSTEP = G::WSTEP, NPASS = G::NPASS;
  constexpr bool XS = STAGE == ST_E0;                        // the stage whose first pass stages x itself
  constexpr int R = 8;                                       // weight ring slots: the loop body is two groups = 8 K steps, so step % 8 is static
  static_assert(D >= 1 && D < R, "ring depth");
  hx::HdrRegs hdr;
  hdr.k8_e0 = (uint32_t)a.k16_e0;
  const int K16 = XS ? a.k16_e0 : hx::kStages[STAGE].k8;
  const int ngroups = K16 / GROUP_STEPS;                     // even (launcher)
  const h8* blob8 = reinterpret_cast<const h8*>(a.blob);
  const f32x4* blob4 = reinterpret_cast<const f32x4*>(a.blob);
  const float inv = reinterpret_cast<const Header*>(a.blob)->inv_scale[STAGE];   // (read once, before any store; first needed by the first epilogue)

  // quad-major buffers: layer 0 reads x / writes bufA (scratch) and bufB; layer 1 bufB -> bufA; layer 2 bufA -> bufB.  A tile's region in
  // a buffer is the same whoever uses it (the largest tenant's size): with a stride per tenant one tile's layer-1 output lay inside another
  // tile's layer-0 scratch, and workgroups are not in step
  const size_t strideA = tile_quads_A(a.k16_e0), strideB = TILE_QUADS_B;
  const f32x4* qin = (STAGE == ST_E1 ? a.bufB + (size_t)tile * strideB : a.bufA + (size_t)tile * strideA);
  f32x4* qout = (STAGE == ST_E1 ? a.bufA + (size_t)tile * strideA : a.bufB + (size_t)tile * strideB);

  // ---- weights: block 8*pass + wv of the stage
  auto wblock = [&](int pass) {
    const int gb = 8 * pass + c.wv, job = gb / NBS, nb = gb % NBS;
    return blob8 + hdr.w_off(STAGE) + (size_t)job * hdr.job_w16(STAGE) + (size_t)nb * 128 + c.lane;   // + step * WSTEP; lo piece at +64
  };
  auto bias_of = [&](int pass) {
    const int gb = 8 * pass + c.wv, job = gb / NBS, nb = gb % NBS;
    return blob4 + hdr.b_off(STAGE) + job * (NBS * 8) + nb * 8 + c.h * 4;                              // 4 quads
  };

  // ---- the activation feed.  ONE register set, recycled piece by piece: what is held for group g+1 is written to LDS during group g,
  // and each register is re-loaded with its counterpart of group g+2 as soon as it has been consumed -- every load is issued almost a
  // whole group (~3 k cycles) before its use with a quarter of a group's data in registers.  Group indices wrap around at the pass's end.
  constexpr int FS = WIDE_FSETS;
  f32x4 fr[FS][GROUP_STEPS];   // group g's data sits in set g % FS
  unsigned pend_hi[2], pend_lo[2];
  // quad-major feed: thread = (half = tid & 1, face = (tid >> 1) & 127, hh = tid >> 8): the quad k = 16 s + 8 hh + 4 half .. + 3 of K step s
  const int q_half = c.tid & 1, q_face = (c.tid >> 1) & 127, q_hh = c.tid >> 8;
  const f32x4* qsrc = qin + (size_t)(2 * q_hh + q_half) * WFACES + q_face;                         // + step * 4 * WFACES
  const int q_wr = (2 * (q_face >> 5)) * SLOT_BYTES + 8 * q_half + 32 * q_hh * 16, q_fx = (q_face & 31) ^ (4 * q_hh);
  // x staging (layer 0, pass 0): thread = (row tid >> 2 of the tile, c4 = tid & 3): the four float4 at columns 64 g + 16 i + 4 c4 (K step i of group g)
  const int srow = c.tid >> 2, c4 = c.tid & 3;
  const int x_wr = (2 * (srow >> 5)) * SLOT_BYTES + 8 * (c4 & 1) + 32 * (c4 >> 1) * 16, x_fx = (srow & 31) ^ (4 * (c4 >> 1));
  const float* p = nullptr;
  f32x4* xscr = nullptr;
  bool live = true;
  double ipd = 1.0, rcp = 1.0, ra = 0.0, rb = 0.0, rc = 0.0;
  unsigned nzbits = 0u;
  const int F = a.F;
  if (XS) {
    int64_t r = tile * WFACES + srow;
    live = r < a.B;
    r = live ? r : a.B - 1;
    p = a.x + r * a.ldx;
    xscr = a.bufA + (size_t)tile * strideA + (size_t)c4 * WFACES + srow;   // + (16 g + 4 i) * WFACES
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
  }
  auto wrapg = [&](int g) { return g >= ngroups ? g - ngroups : g; };
  auto feed_load = [&](int g, int i, auto xfc, auto setc) {   // issue the global load of K step i of group g (wrapping) into set `setc`
    constexpr bool XF = decltype(xfc)::value;
    constexpr int S = decltype(setc)::value % FS;
#ifdef WIDE_ABL_NOFEEDLOAD   // timing-only ablations (wrong results)
    return;
#endif
#ifdef WIDE_ABL_FEEDHOT
    const int gg = 0;
#else
    const int gg = wrapg(g);
#endif
    if (XF) {
      const int k = 64 * gg + 16 * i + 4 * c4;
      fr[S][i] = *reinterpret_cast<const f32x4*>(p + (k < F ? k : F - 4));
    } else {
      fr[S][i] = qsrc[(size_t)(GROUP_STEPS * gg + i) * (4 * WFACES)];
    }
  };
  // feed pieces of K step i of group g, written into the LDS buffer at `boff`:
  //   x staging: 0..3 normalise element e (or zero it beyond F) | 4 store the normalised quad to the scratch | 5, 6 split | 7, 8 the hi / lo LDS stores
  //   quad-major: 5, 6 split | 7, 8 the hi / lo LDS stores
  auto feed_piece = [&](int g, int i, int piece, int boff, auto xfc, auto setc) {
    constexpr bool XF = decltype(xfc)::value;
    f32x4& v = fr[decltype(setc)::value % FS][i];
    if (piece < 5) {
      if (!XF) return;
      const int gg = wrapg(g);
      if (piece == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(v[q]));   // the loads have landed: nothing below is hoisted above
      }
      if (piece < 4) {
        const int e = piece;
        if (NORM) {
          const int t = (i + e) % 3;
          const double rr = t == 0 ? ra : (t == 1 ? rb : rc);
          v[e] = (float)div_ipd((double)v[e] - rr, ipd, rcp);
        }
        if (64 * gg + 16 * i + 4 * c4 >= F) v[e] = 0.0f;          // (F % 4 == 0: a float4 lies inside the row or beyond it)
      } else if (!SPLITP) {
        xscr[(size_t)(16 * gg + 4 * i) * WFACES] = v;
      }
    } else if (piece < 7) {
      const int j = piece - 5;
#ifdef WIDE_ABL_NOFEEDWRITE
      if (j == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(v[q]));
      }
      return;
#endif
      if (XF) nzbits |= (__float_as_uint(v[2 * j]) | __float_as_uint(v[2 * j + 1])) & 0x7fffffffu;
      else if (j == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(v[q]));
      }
      hx::split2(v[2 * j], v[2 * j + 1], pend_hi[j], pend_lo[j]);
    } else {
#ifdef WIDE_ABL_NOFEEDWRITE
      return;
#endif
      const int pc = piece - 7;
      const u2 val = pc == 0 ? u2{pend_hi[0], pend_hi[1]} : u2{pend_lo[0], pend_lo[1]};
      const int off = XF ? x_wr + ((x_fx ^ i) << 4) : q_wr + ((q_fx ^ i) << 4);
      *reinterpret_cast<LdsU2*>(lds + boff + i * STEP_BYTES + pc * SLOT_BYTES + off) = val;
    }
  };
  auto rotate_refs = [&]() {   // next group: columns + 64 => coordinate + 1
    const double t0 = ra; ra = rb; rb = rc; rc = t0;
  };
  // the feed wraps around at the pass's end: group `ngroups` is group 0 again, whose coordinate phase is 0, not ngroups % 3 -- so the
  // references turn (3 - ngroups % 3) % 3 more times there (selects on a uniform condition: no branch in the loop)
  const int wrap_turns = (3 - ngroups % 3) % 3;
  auto rotate_refs_wrap = [&](bool at_wrap) {
    rotate_refs();
    const bool t1 = at_wrap && wrap_turns >= 1, t2 = at_wrap && wrap_turns == 2;
    { const double n0 = t1 ? rb : ra, n1 = t1 ? rc : rb, n2 = t1 ? ra : rc; ra = n0; rb = n1; rc = n2; }
    { const double n0 = t2 ? rb : ra, n1 = t2 ? rc : rb, n2 = t2 ? ra : rc; ra = n0; rb = n1; rc = n2; }
  };

  // ---- MFMA operands: lane (f, h) reads its piece of slot j of K step t at  buffer + t * STEP_BYTES + j * SLOT_BYTES + swz16(f, h, t)
  const LdsB* rd[GROUP_STEPS];
#pragma unroll
  for (int t = 0; t < GROUP_STEPS; ++t) rd[t] = lds + t * STEP_BYTES + swz16(c.f, c.h, t);
  h8 xh[WFB], xl[WFB];
  auto xread = [&](const LdsB* base, int fb) {
    xh[fb] = *reinterpret_cast<const LdsH8*>(base + (2 * fb) * SLOT_BYTES);
    xl[fb] = *reinterpret_cast<const LdsH8*>(base + (2 * fb + 1) * SLOT_BYTES);
  };

  f32x16 acc[WFB], accS[WFB];
  h8 wr[R][2];
  auto init_acc = [&](const f32x4 (&bq)[4]) {
#pragma unroll
    for (int fb = 0; fb < WFB; ++fb)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[fb][4 * q + e] = bq[q][e];
          accS[fb][4 * q + e] = 0.0f;
        }
  };

  // ---- the stage's prologue: group 0 in LDS, group 1 (and 2) in flight, the weight ring's first D steps, bias.  The bias is requested
  // LAST and consumed before the loop, so every load of the prologue has landed when the K loop is entered: hipcc's s_waitcnt pass
  // merges the pending-load state of the loop's two entries (prologue and back edge) to the more conservative one per register, and a
  // prologue that leaves its feed loads as the youngest in flight turned every feed wait of the steady state into vmcnt(4) instead of
  // vmcnt(10) -- a wait for weights requested one K step earlier (stamps: the feed cost 15 % of the loop; 3 % with L2-hot addresses).
  {
    using XF0 = std::integral_constant<bool, XS>;
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
#pragma unroll
    for (int i = 0; i < GROUP_STEPS; ++i) feed_load(0, i, XF0{}, C0{});
    if (FS == 2) {
#pragma unroll
      for (int i = 0; i < GROUP_STEPS; ++i) feed_load(1, i, XF0{}, C1{});
    }
#pragma unroll
    for (int i = 0; i < GROUP_STEPS; ++i) {
#pragma unroll
      for (int pc = 0; pc < 9; ++pc) feed_piece(0, i, pc, 0, XF0{}, C0{});
      feed_load(FS, i, XF0{}, C0{});   // (one set: group 1; two sets: group 2 -- group 1 is already in flight in set 1)
    }
    if (XS && NORM) rotate_refs();
    const h8* w0 = wblock(pass0);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      wr[d][0] = w0[(size_t)d * WSTEP];
      wr[d][1] = w0[(size_t)d * WSTEP + 64];
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[q] = bias_of(pass0)[q];
    init_acc(bq);
    __syncthreads();
#pragma unroll
    for (int fb = 0; fb < WFB; ++fb) xread(rd[0], fb);
  }

  auto do_pass = [&](auto xfc, const int pass, const bool has_next) {
    constexpr bool XF = decltype(xfc)::value;
    using XFC = std::integral_constant<bool, XF>;
    WST(0);
    const h8* wcur = wblock(pass);
    const h8* wnxt = wblock(has_next ? pass + 1 : pass);
    auto wfrag = [&](int s) { return s < K16 ? wcur + (size_t)s * WSTEP : wnxt + (size_t)(s - K16) * WSTEP; };

    // group g: reads buffer `cur`, writes group g+1 (held in the register set) into buffer `nxt` during its first steps -- each register
    // re-loaded with group g+2's data right after its last use.  One K step = 12 MFMAs, face block by face block: (w_lo, x_hi) and (w_hi, x_lo)
    // into accS, (w_hi, x_hi) into acc -- the order step_fine gives every accumulator -- and behind a face block's last MFMA the LDS reads
    // of ITS operands for the next step.
    auto group = [&](int g, auto parc, int cur, int nxt) {
      constexpr int PAR = decltype(parc)::value;   // g & 1
#pragma unroll
      for (int t = 0; t < GROUP_STEPS; ++t) {
        const int ks = GROUP_STEPS * g + t;
        const LdsB* xnext = t < 3 ? rd[t + 1] + cur : rd[0] + nxt;
        const h8 (&wc)[2] = wr[(GROUP_STEPS * PAR + t) % R];
        h8 (&wn)[2] = wr[(GROUP_STEPS * PAR + t + D) % R];
        const h8* wp = wfrag(ks + D);
        auto extra = [&](int m) {
          const int s = 12 * t + m;      // slot within the group, 0 .. 47 (the barrier stands between slots 36 and 37)
#ifdef WIDE_ABL_NOFEED   // timing-only ablations (wrong results): tools/wide_stamps.py
          return;
#endif
          if (s < 36) {
            const int i = s / 9, pc = s % 9;
            using SC = std::integral_constant<int, PAR + 1>;   // group g+1's set
            feed_piece(g + 1, i, pc, nxt, XFC{}, SC{});
            if (pc == 8) feed_load(g + 1 + FS, i, XFC{}, SC{});
          }
          if (s == 36 && XF && NORM) rotate_refs_wrap(g + 2 == ngroups);   // (the references now stand for group g+2)
        };
        if constexpr (!XF) {
#pragma unroll
          for (int fb = 0; fb < WFB; ++fb) {
            accS[fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc[1], xh[fb], accS[fb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#ifndef WIDE_ABL_NOW
            if (fb == 0) wn[0] = wp[0];
            if (fb == 2) wn[1] = wp[64];
#endif
            extra(3 * fb);
            __builtin_amdgcn_sched_barrier(0);
            accS[fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc[0], xl[fb], accS[fb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#ifndef WIDE_ABL_NOBAR
            if (fb == 0 && t == 3) __syncthreads();   // group g+1 is complete in LDS; nobody reads group g's buffer any more
#endif
            extra(3 * fb + 1);
            __builtin_amdgcn_sched_barrier(0);
            acc[fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc[0], xh[fb], acc[fb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#ifndef WIDE_ABL_NOLDS
            xread(xnext, fb);
#endif
            extra(3 * fb + 2);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          // the x-staging pass needs its registers for the staging: the operands of TWO face blocks at a time (xh / xl [fb & 1]), those of
          // face block fb+1 requested behind the first MFMA of face block fb (16 registers instead of 32; three MFMAs of lead)
          const LdsB* xcur = rd[t] + cur;
#pragma unroll
          for (int fb = 0; fb < WFB; ++fb) {
            const int w0 = fb & 1, w1 = (fb + 1) & 1;
            accS[fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc[1], xh[w0], accS[fb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (fb == 0) wn[0] = wp[0];
            if (fb == 2) wn[1] = wp[64];
            if (fb == 3 && t == 3) __syncthreads();   // (before the first read of the other buffer: face block 0 of the next group's first step)
            {
              const LdsB* src = fb < 3 ? xcur + (2 * (fb + 1)) * SLOT_BYTES : xnext;
              xh[w1] = *reinterpret_cast<const LdsH8*>(src);
              xl[w1] = *reinterpret_cast<const LdsH8*>(src + SLOT_BYTES);
            }
            extra(3 * fb);
            __builtin_amdgcn_sched_barrier(0);
            accS[fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc[0], xl[w0], accS[fb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            extra(3 * fb + 1);
            __builtin_amdgcn_sched_barrier(0);
            acc[fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc[0], xh[w0], acc[fb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            extra(3 * fb + 2);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    };
    auto run_groups = [&](int g0, int g1) {
      for (int g = g0; g < g1; g += 2) {
        group(g, std::integral_constant<int, 0>{}, 0, GROUP_BYTES);
        group(g + 1, std::integral_constant<int, 1>{}, GROUP_BYTES, 0);
      }
    };
    WST(1);
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
    WST(2);
    // ---- the pass's end: the next pass's bias is requested before the epilogue's arithmetic; the LDS pipeline and the weight ring have
    // already run on into the next pass (group 0 staged, group 1 in registers, the ring's D steps fetched, step 0's operands read)
    f32x4 bq[4];
    if (has_next) {
#pragma unroll
      for (int q = 0; q < 4; ++q) bq[q] = bias_of(pass + 1)[q];
    }
    if (XF && !SPLITP) {   // from here on the feed is the quad-major one, reading what this pass stored to the scratch: group 1 into the registers
      if (a.valid) {   // all-zero feature row == "no face" (FeatureExtractor.py:105-106); the 4 lanes of a row are neighbours
        const unsigned long long m = __ballot(nzbits != 0u);
        if (c4 == 0 && live) a.valid[tile * WFACES + srow] = ((m >> (c.lane & 60)) & 0xFull) ? 1 : 0;
      }
      __syncthreads();   // every wave's scratch stores are issued (and ordered before the loads below: one CU, one L1)
      if (NPASS > 1) {
#pragma unroll
        for (int fb = 1; fb < WFB; ++fb) xread(rd[0], fb);   // (the staging pass kept two face blocks' operands; the next pass's first step needs all four)
#pragma unroll
        for (int i = 0; i < GROUP_STEPS; ++i) feed_load(1, i, std::false_type{}, std::integral_constant<int, 1>{});
        if (FS == 2) {
#pragma unroll
          for (int i = 0; i < GROUP_STEPS; ++i) feed_load(2, i, std::false_type{}, std::integral_constant<int, 0>{});
        }
      }
    }
    // epilogue: (acc + accS) * inv -> ReLU -> quad-major f32: register quad q of face block fb = neurons 32 gb + 8 q + 4 h .. + 3 of face 32 fb + f
    {
      f32x4* o = qout + (size_t)(8 * (8 * pass + c.wv) + c.h) * WFACES + c.f;
#pragma unroll
      for (int fb = 0; fb < WFB; ++fb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = hx::activate<hx::ACT_RELU>((acc[fb][4 * q + e] + accS[fb][4 * q + e]) * inv);
          o[(size_t)(2 * q) * WFACES + 32 * fb] = v;
        }
    }
    if (has_next) init_acc(bq);
    WST(3);
  };
  if constexpr (SPLITP) {
    do_pass(std::integral_constant<bool, XS>{}, pass0, false);
    if (XS && a.valid && pass0 == 0) {   // all-zero feature row == "no face" (FeatureExtractor.py:105-106); the 4 lanes of a row are neighbours
      const unsigned long long m = __ballot(nzbits != 0u);
      if (c4 == 0 && live) a.valid[tile * WFACES + srow] = ((m >> (c.lane & 60)) & 0xFull) ? 1 : 0;
    }
  } else {
    static_for<0, NPASS>([&](auto passc) {
      constexpr int PASS = decltype(passc)::value;
      do_pass(std::integral_constant<bool, XS && PASS == 0>{}, PASS, PASS + 1 < NPASS);
    });
  }
}

template <int FIRST, int LAST, bool NORM>
__global__ __launch_bounds__(512) void wide_layers_kernel(WArgs a) {
  __shared__ __attribute__((aligned(16))) char lds_[LDS_W];
  LdsB* lds = (LdsB*)lds_;
  Lane c;
  c.tid = threadIdx.x;
  c.lane = c.tid & 63;
  c.f = c.lane & 31;
  c.h = c.lane >> 5;
  c.wv = __builtin_amdgcn_readfirstlane(c.tid >> 6);
  const int64_t tile = blockIdx.x;
  static_for<FIRST, LAST + 1>([&](auto sc) {
    constexpr int STAGE = decltype(sc)::value;
    if (STAGE > FIRST) __syncthreads();   // the previous layer's output of this tile is stored (same workgroup: one CU, one L1) and its LDS reads are over
    wide_stage<STAGE, NORM && STAGE == ST_E0, WIDE_D, false>(a, lds, c, tile, 0);
  });
}

// One layer, one pass per workgroup.  Block b: XCD b & 7 (blocks are dealt round-robin over the XCDs -- observed, not promised: only the
// speed depends on it), j = b >> 3: pass j % NPASS of tile 8 (j / NPASS) + (b & 7) -- the NPASS workgroups of a tile sit on one XCD, start
// together and walk the same input: it is fetched from memory once and found in that L2 by the others.
template <int STAGE, bool NORM>
__global__ __launch_bounds__(512) void wide_pass_kernel(WArgs a, int ntiles) {
  __shared__ __attribute__((aligned(16))) char lds_[LDS_W];
  LdsB* lds = (LdsB*)lds_;
  Lane c;
  c.tid = threadIdx.x;
  c.lane = c.tid & 63;
  c.f = c.lane & 31;
  c.h = c.lane >> 5;
  c.wv = __builtin_amdgcn_readfirstlane(c.tid >> 6);
  constexpr int NPASS = StageGeo<STAGE>::NPASS;
  const int b = blockIdx.x, j = b >> 3;
  const int pass = j % NPASS;
  const int64_t tile = 8 * (int64_t)(j / NPASS) + (b & 7);
  if (tile >= ntiles) return;
  wide_stage<STAGE, NORM, WIDE_D, true>(a, lds, c, tile, pass);
}

}  // namespace hxw

namespace hx {
// ---- the tail (E3, E4, E5, heads) of one 64-face tile as a launch of its own: layer 2's quad-major output -> the H3 LDS image (hi/lo
// planes) -> the fused kernel's tail_stages()
__global__ __launch_bounds__(256, 1) void tail64_kernel(Args a, const f32x4* __restrict__ qin) {
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
  tail_pre_e3<false>(c, acc3, wr3);                           // E3's bias and first weights fly while the image is filled
  // quad (kq, face) of the 128-face tile (tile >> 1), faces 64 (tile & 1) ..: columns 4 kq .. + 3 of image row `face`, both planes
  const f32x4* src = qin + (size_t)(tile >> 1) * hxw::TILE_QUADS_B + 64 * (tile & 1);
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  for (int i = tid; i < 64 * 64; i += 256) {                  // layer 2's output: 256 columns = 64 quads, 64 faces
    const int face = i & 63, kq = i >> 6;
    const f32x4 v = src[(size_t)kq * hxw::WFACES + face];
    unsigned hi[2], lo[2];
    split2(v[0], v[1], hi[0], lo[0]);
    split2(v[2], v[3], hi[1], lo[1]);
    char* d = lds + O_H3 + (face * S_H3 + 4 * kq) * 2;
    *reinterpret_cast<u2*>(d) = u2{hi[0], hi[1]};
    *reinterpret_cast<u2*>(d + P_H3) = u2{lo[0], lo[1]};
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

// per 128-face tile: bufA = max(layer 0's input, layer 1's output) quads, bufB = layer 0's output (layer 2's is smaller)
static size_t wide_bufA_quads(int F) { return hxw::tile_quads_A(e0_k16(F)); }
static size_t wide_bufB_quads() { return hxw::TILE_QUADS_B; }

size_t wide_workspace_bytes(int64_t B, int F) {
  if (B <= 0 || F <= 0) return 0;
  const size_t ntiles = (size_t)((B + hxw::WFACES - 1) / hxw::WFACES);
  return ntiles * (wide_bufA_quads(F) + wide_bufB_quads()) * 16;
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
  const int64_t nt128 = (B + WFACES - 1) / WFACES;

  WArgs a{};
  a.blob = blob; a.x = src; a.ldx = sld; a.B = B; a.F = F; a.norm = raw ? (normalize ? 1 : 0) : 0; a.valid = valid;
  a.bufA = reinterpret_cast<hx::f32x4*>(workspace);
  a.bufB = a.bufA + (size_t)nt128 * wide_bufA_quads(F);
  a.k16_e0 = e0_k16(F);
#ifdef WIDE_STAMPS
  if (const char* e = getenv("NLML_WIDE_STAMPS_PTR")) a.stamps = reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0));
  a.stamp_stage_ = getenv("NLML_WIDE_STAMPS_STAGE") ? atoi(getenv("NLML_WIDE_STAMPS_STAGE")) : 0;
#endif
  const dim3 grid((unsigned)nt128), block(512);
  static const bool fused_layers = [] { const char* e = getenv("NLML_K2_WIDE_SPLIT"); return !(e && e[0] == '1'); }();   // A/B: =1 one launch per layer
  static const bool pass_split = [] { const char* e = getenv("NLML_K2_WIDE_PSPLIT"); return e && e[0] == '1'; }();   // A/B: =1 one pass per workgroup
  if (pass_split) {
    const int nt = (int)nt128, groups = (nt + 7) / 8;
    if (a.norm) hipLaunchKernelGGL((wide_pass_kernel<ST_E0, true>), dim3(groups * 8 * 4), block, 0, st, a, nt);
    else hipLaunchKernelGGL((wide_pass_kernel<ST_E0, false>), dim3(groups * 8 * 4), block, 0, st, a, nt);
    hipLaunchKernelGGL((wide_pass_kernel<ST_E1, false>), dim3(groups * 8 * 2), block, 0, st, a, nt);
    hipLaunchKernelGGL((wide_pass_kernel<ST_E2, false>), dim3(groups * 8), block, 0, st, a, nt);
  } else if (fused_layers) {
    if (a.norm) hipLaunchKernelGGL((wide_layers_kernel<ST_E0, ST_E2, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((wide_layers_kernel<ST_E0, ST_E2, false>), grid, block, 0, st, a);
  } else {
    if (a.norm) hipLaunchKernelGGL((wide_layers_kernel<ST_E0, ST_E0, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((wide_layers_kernel<ST_E0, ST_E0, false>), grid, block, 0, st, a);
    hipLaunchKernelGGL((wide_layers_kernel<ST_E1, ST_E1, false>), grid, block, 0, st, a);
    hipLaunchKernelGGL((wide_layers_kernel<ST_E2, ST_E2, false>), grid, block, 0, st, a);
  }
  {
    hx::Args ta{};
    ta.B = B; ta.F = F; ta.blob = blob; ta.out = out; ta.latent = latent; ta.valid = nullptr;
    ta.x = src; ta.ldx = sld; ta.norm = a.norm;
    hipLaunchKernelGGL(hx::tail64_kernel, dim3((unsigned)((B + 63) / 64)), dim3(256), 0, st, ta, (const hx::f32x4*)a.bufB);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
  // the strict-fast mode's f32 re-evaluation launch, as behind its fused kernel (encoder_heads_f16x2_w8.hip)
  return launch_encoder_heads_f32(x, ldx, raw, normalize, B, F, static_cast<const char*>(blob) + strict_f32_image_offset(F), out, latent,
                                  nullptr, nullptr, nullptr, stream, STRICT_INKERNEL_RESCUE_MAX);
}

}  // namespace nlml
