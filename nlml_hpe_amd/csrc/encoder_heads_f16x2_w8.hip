// encoder_heads_f16x2_w8.hip -- the strict-fast mode (NLML_MODE_F16X2S) with EIGHT waves per workgroup (two per SIMD).
//
// Same network, blob, LDS images, jobs and -- per accumulator -- the same MFMA sequence as encoder_heads_f16x2_kernel<.., SPLIT = true>
// (encoder_heads_f16x2.hip; reference: NLML_HPE_Model_Builder.py:33-53,76-92,115-126), so the results are bit-identical to it and to the
// layer-per-launch path.  What changes is who computes what: a job of the trunk (layer 0: four neuron blocks; layer 1: four; layer 2:
// two) is shared by a PAIR of waves, each taking half of its blocks with both face blocks.  Why:
//   * split accumulators need 2 x (512 neurons x 64 faces) registers per layer-0 pass whatever the wave count; with four waves that is
//     256 of a wave's 512 registers next to a 128-register weight ring, so layer 1's accumulators had to be parked and everything else
//     squeezed (110 spilled registers, 444 B/lane of scratch).  Eight waves hold 128 accumulator registers each; nothing spills.
//   * the L2 -> CU weight stream, not the matrix pipe, bounds layer 0 (9.6 MB per 64-face tile): four waves take in 35-41 B/clk, eight
//     46-51 (tools/probes/l1_stream_probe.hip, profiles/r03_l1_stream_probe.txt) -- more waves asking, more bytes in flight.
//   * two waves per SIMD hide each other's dependent chains: the x staging (f64 normalisation, hi/lo split, LDS writes) and the
//     epilogues of one wave run under the other's MFMAs without hand-placing every link.
// The tail (E3, E4, E5, the three heads; 10 % of the FLOPs, stream-bound short stages) is the four-wave code of
// encoder_heads_f16x2_dev.h run by waves 0-3: waves 4-7 END after layer 2's store.  S_BARRIER waits only for the waves of the
// workgroup that have not terminated (CDNA ISA, S_BARRIER: "If some waves in the threadgroup have already terminated, this waits on
// only the surviving waves"), so the tail's barriers keep working among the four survivors.
#include <hip/hip_runtime.h>

#include "../../include/nlml_hpe.h"
#include "abi_internal.h"
#include "encoder_heads_f16x2_dev.h"
#include "layout.h"

namespace nlml {
namespace hx {

__device__ __forceinline__ double div_ipd_w8(double n, double d, double y) {   // == IEEE n / d for these operands (K1)
  const double q = n * y;
  const double r = fma(-q, d, n);
  return fma(r, y, q);
}

// One pass of layer 0 for 512 threads: x[64,F] f32 -> (optional IPD normalisation in f64, FeatureExtractor.py:30-66) -> hi/lo f16 ->
// three rotating LDS slabs of 32 columns (thread = row tid/8, columns 4*(tid%8) .. +3); this wave computes neuron blocks 2*nbh, 2*nbh+1
// of job 4*pass + jw for both face blocks with split accumulators (step_fine: `acc` takes w_hi*x_hi, `accS` the two small products).
template <bool VEC4, bool NORM>
__device__ __forceinline__ void stage_e0_pass_w8(const Ctx& c, const Args& a, int64_t row0, int tid, int pass, int jw, int nbh,
                                                 f32x16 (&acc)[2][2], f32x16 (&accS)[2][2]) {
  constexpr int NB = 2, NFB = 2;
  constexpr int WSTEP = 4 * 2 * 64;                 // a layer-0 job has four blocks x two pieces per K step
  const int F = a.F;
  const int nslab = (int)c.hdr.k8_e0 / XS_STEPS;   // even (pack.cpp)
  constexpr int SLAB_BYTES = 2 * P_XS;

  const int srow = tid >> 3, scol = (tid & 7) * 4;
  int64_t r = row0 + srow;
  const bool live = r < a.B;
  r = live ? r : a.B - 1;
  const float* p = a.x + r * a.ldx;
  double ipd = 1.0, rcp = 1.0, ra = 0.0, rb = 0.0, rc = 0.0;
  if (NORM) {   // exactly K1's arithmetic: the f32 value the reference feeds the network, bit for bit
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

  struct Set { float v[4]; };
  auto gload = [&](int s, Set& st) {   // columns scol .. scol+3 of slab s
    s = s < nslab ? s : nslab - 1;
    const int k = s * XS_COLS + scol;
    if (VEC4) {
      const int kc = k < F ? k : (NORM ? k - 12 * ((k - F + 15) / 12) : F - 4);   // phase-preserving clamp (zero weights there)
      const f32x4 t = *reinterpret_cast<const f32x4*>(p + kc);
#pragma unroll
      for (int e = 0; e < 4; ++e) st.v[e] = t[e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) st.v[e] = p[k + e < F ? k + e : F - 1];
    }
  };
  auto lw_begin = [&](Set& st) {   // the set's loads must have landed: everything below consumes them
#pragma unroll
    for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(st.v[e]));
  };
  auto lw_norm = [&](Set& st, int q) {   // element q (static): the plain chain (prologue)
    const int t = q % 3;
    const double rr = t == 0 ? ra : (t == 1 ? rb : rc);
    st.v[q] = (float)div_ipd_w8((double)st.v[q] - rr, ipd, rcp);
  };
  double nA = 0.0, nB = 0.0, qA = 0.0, qB = 0.0;   // the pair in flight
  auto lw_norm2 = [&](Set& st, int pair, int link) {   // elements 2*pair, 2*pair+1 (static), link 0..5 of the division's chain
    const int e0 = 2 * pair, e1 = e0 + 1;
    const double r0 = e0 % 3 == 0 ? ra : (e0 % 3 == 1 ? rb : rc), r1 = e1 % 3 == 0 ? ra : (e1 % 3 == 1 ? rb : rc);
    if (link == 0) { nA = (double)st.v[e0]; nB = (double)st.v[e1]; asm volatile("" : "+v"(nA), "+v"(nB)); }
    if (link == 1) { nA = nA - r0; nB = nB - r1; asm volatile("" : "+v"(nA), "+v"(nB)); }
    if (link == 2) { qA = nA * rcp; qB = nB * rcp; asm volatile("" : "+v"(qA), "+v"(qB)); }
    if (link == 3) { nA = fma(-qA, ipd, nA); nB = fma(-qB, ipd, nB); asm volatile("" : "+v"(nA), "+v"(nB)); }
    if (link == 4) { qA = fma(nA, rcp, qA); qB = fma(nB, rcp, qB); asm volatile("" : "+v"(qA), "+v"(qB)); }
    if (link == 5) { st.v[e0] = (float)qA; st.v[e1] = (float)qB; asm volatile("" : "+v"(st.v[e0]), "+v"(st.v[e1])); }
  };
  auto lw_rotate = [&]() {   // next slab: columns + 32 => phase + 2
    const double t0 = rc; rc = rb; rb = ra; ra = t0;
  };
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  unsigned pend_hi[2], pend_lo[2];
  auto lw_split = [&](Set& st, int j, bool real_slab) {   // elements 2j, 2j+1 -> packed hi/lo f16 pairs
    const unsigned m = real_slab ? 0x7fffffffu : 0u;
    nzbits |= (__float_as_uint(st.v[2 * j]) | __float_as_uint(st.v[2 * j + 1])) & m;
    split2(st.v[2 * j], st.v[2 * j + 1], pend_hi[j], pend_lo[j]);
  };
  auto lw_store = [&](int buf_off, int piece) {
    char* d = c.lds + O_XS + buf_off + (srow * S_XS + scol) * 2;
    if (piece == 0) *reinterpret_cast<u2*>(d) = u2{pend_hi[0], pend_hi[1]};
    else *reinterpret_cast<u2*>(d + P_XS) = u2{pend_lo[0], pend_lo[1]};
  };
  auto lwrite = [&](int buf_off, Set& st, bool real_slab) {
    lw_begin(st);
    if (NORM) {
#pragma unroll
      for (int q = 0; q < 4; ++q) lw_norm(st, q);
      lw_rotate();
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) lw_split(st, j, real_slab);
    lw_store(buf_off, 0);
    lw_store(buf_off, 1);
  };

  const int job = 4 * pass + jw;
  load_bias<NB, NFB>(acc, c.blob4 + c.hdr.b_off(ST_E0) + job * (4 * 8) + (2 * nbh) * 8, c.h);
  zero_acc<NB, NFB>(accS);
  const h8* w = c.blob8 + c.hdr.w_off(ST_E0) + (size_t)job * c.hdr.job_w16(ST_E0) + (2 * nbh * 2) * 64 + c.lane;
  auto wfrag = [&](int ks) { return w + (size_t)ks * WSTEP; };   // K step ks of this wave's two blocks

  // TWO staging register sets (4 floats per thread each), one per slab parity: slab s+2 is written to LDS during slab s from
  // set[s & 1], which is refilled at once with the loads of slab s+4 (vmcnt counts in issue order: see encoder_heads_f16x2.hip)
  Set set[2];
  constexpr int R0 = 4, D0 = R0 - 1;   // weight ring: K step ks in slot ks % 4 = 2 * (slab & 1) + step of the slab
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
      for (int pp = 0; pp < 2; ++pp) wr[d][nb][pp] = wfrag(d)[(nb * 2 + pp) * 64];
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
    const bool real = s + 2 < nslab;
#pragma unroll
    for (int kk = 0; kk < XS_STEPS; ++kk) {
      const int slot = 2 * PAR + kk;                        // == K step % 4: two slabs are exactly one turn of the ring
      const int ks = s * XS_STEPS + kk;
      step_fine<NB, NFB>(acc, accS, wr[slot], xr[kk & 1], wr[(slot + D0) % R0], wfrag(ks + D0), true,
              [&](int fb, int pp) {                         // the next K step's x operands
                xr[(kk + 1) & 1][fb][pp] = (kk + 1 < XS_STEPS)
                                               ? *reinterpret_cast<const h8*>(xrow + pp * P_XS + fb * FB + 32 * (kk + 1))
                                               : *reinterpret_cast<const h8*>(xnext + pp * P_XS + fb * FB);
              },
              [&](int m) {   // slab s+2's staging, one piece behind each of the slab's 24 MFMAs
                const int j = 12 * kk + m;
                if (j == 0) lw_begin(set[PAR]);
                if (NORM && j < 12) lw_norm2(set[PAR], j / 6, j % 6);
                if (NORM && j == 12) lw_rotate();
                if (j == 13 || j == 14) lw_split(set[PAR], j - 13, real);
                if (j == 15) lw_store(o2, 0);
                if (j == 16) lw_store(o2, 1);
                if (j == 17) gload(s + 4, set[PAR]);
              });
    }
    __syncthreads();
    const int t0 = o0;   // rotate: (o0, o1, o2) <- (o1, o2, o0)
    o0 = o1; o1 = o2; o2 = t0;
  };
  for (int s = 0; s < nslab; s += 2) {
    slab(s, std::integral_constant<int, 0>{});
    slab(s + 1, std::integral_constant<int, 1>{});
  }
  if (pass == 0 && a.valid) {  // all-zero feature row == "no face" (FeatureExtractor.py:105-106); 8 lanes share a row
    const unsigned long long m = __ballot(nzbits != 0u);
    if ((tid & 7) == 0 && live) a.valid[row0 + srow] = ((m >> (c.lane & 56)) & 0xFFull) ? 1 : 0;
  }
}

template <bool VEC4, bool NORM>
__global__ __launch_bounds__(512) void encoder_heads_f16x2_w8_kernel(Args a) {
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
  c.wv = __builtin_amdgcn_readfirstlane(tid >> 6);       // 0..7
  const int jw = c.wv >> 1, nbh = c.wv & 1;              // the job this wave shares with its partner, and its half of the job's blocks
  const int64_t row0 = (int64_t)blockIdx.x * TILE_FACES;

  {  // E0 (two passes of 512 neurons, split accumulators) interleaved with the two K halves of E1
    constexpr int WSTEP1 = 4 * 2 * 64;
    const h8* w1 = c.blob8 + c.hdr.w_off(ST_E1) + (size_t)jw * c.hdr.job_w16(ST_E1) + (2 * nbh * 2) * 64 + c.lane;
    const float inv0 = c.hdr.inv_scale[ST_E0];
    // Layer 1's accumulators (neurons 128*jw + 64*nbh .. +63, both face blocks) are parked in LDS while a layer-0 pass runs (the h1
    // half image is dead then; lane-private 16-byte pieces, conflict-free) -- the four-wave kernel's structure, at half the registers.
    f32x16 acc1[2][2];
    f32x4* const park = reinterpret_cast<f32x4*>(c.lds + O_H1H) + tid;   // piece i of this lane at park[512 * i]
    static_assert(16 * 512 * 16 <= 2 * P_H1H, "the parked layer-1 accumulators fit the h1 half image");
    load_bias<2, 2>(acc1, c.blob4 + c.hdr.b_off(ST_E1) + jw * (4 * 8) + (2 * nbh) * 8, c.h);
    auto park_acc1 = [&]() {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int fb = 0; fb < 2; ++fb)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            park[512 * ((nb * 2 + fb) * 4 + q)] = f32x4{acc1[nb][fb][4 * q], acc1[nb][fb][4 * q + 1], acc1[nb][fb][4 * q + 2], acc1[nb][fb][4 * q + 3]};
    };
    park_acc1();
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      {
        f32x16 acc0[2][2], acc0s[2][2];
        stage_e0_pass_w8<VEC4, NORM>(c, a, row0, tid, pass, jw, nbh, acc0, acc0s);
        add_acc<2, 2>(acc0, acc0s);
        // layer 1's accumulators back from LDS before the h1 half image is (over)written
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int fb = 0; fb < 2; ++fb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const f32x4 t = park[512 * ((nb * 2 + fb) * 4 + q)];
              acc1[nb][fb][4 * q] = t[0]; acc1[nb][fb][4 * q + 1] = t[1]; acc1[nb][fb][4 * q + 2] = t[2]; acc1[nb][fb][4 * q + 3] = t[3];
            }
        __syncthreads();
        job_store<2, 2, ACT_RELU>(c, acc0, O_H1H, P_H1H, S_H1H, 128 * jw + 64 * nbh, 0, inv0);
      }
      __syncthreads();
      {  // layer 1 over this K half; its small products are added at the end of the half
        f32x16 acc1s[2][2];
        zero_acc<2, 2>(acc1s);
        kloop<2, 2, 32, WSTEP1>(acc1, acc1s, w1 + (size_t)pass * 32 * WSTEP1, c.lds + O_H1H + (c.f * S_H1H + 8 * c.h) * 2, P_H1H,
                                32 * S_H1H * 2);
        add_acc<2, 2>(acc1, acc1s);
      }
      __syncthreads();   // H1H is free again (pass 0: for the parked set and pass 1's store; pass 1: for H2)
      if (pass == 0) park_acc1();
    }
    job_store<2, 2, ACT_RELU>(c, acc1, O_H2, P_H2, S_H2, 128 * jw + 64 * nbh, 0, c.hdr.inv_scale[ST_E1]);
  }
  __syncthreads();
  // E2: 512 -> 256, ReLU, split accumulators; job jw's block nbh.  h3 overwrites h2 => barrier between the K loop and the store
  Ctx ct = c;
  ct.wv = c.wv & 3;                                       // the tail's wave index (waves 4-7 only fetch with it, then end)
  f32x16 acc3[1][2];
  h8 wr3[ring_slots(1, 2)][1][2];
  {
    constexpr int WSTEP2 = 2 * 2 * 64;
    f32x16 acc2[1][2], acc2s[1][2];
    load_bias<1, 2>(acc2, c.blob4 + c.hdr.b_off(ST_E2) + jw * (2 * 8) + nbh * 8, c.h);
    zero_acc<1, 2>(acc2s);
    const h8* w2 = c.blob8 + c.hdr.w_off(ST_E2) + (size_t)jw * c.hdr.job_w16(ST_E2) + (nbh * 2) * 64 + c.lane;
    kloop<1, 2, 32, WSTEP2>(acc2, acc2s, w2, c.lds + O_H2 + (c.f * S_H2 + 8 * c.h) * 2, P_H2, 32 * S_H2 * 2);
    add_acc<1, 2>(acc2, acc2s);
    tail_pre_e3<false>(ct, acc3, wr3);                    // E3's global fetches in front of the store and the barriers
    __syncthreads();
    job_store<1, 2, ACT_RELU>(c, acc2, O_H3, P_H3, S_H3, 64 * jw + 32 * nbh, 0, c.hdr.inv_scale[ST_E2]);
  }
  __syncthreads();
  if (c.wv >= 4) return;   // waves 4-7 end here; the tail's barriers wait only for the surviving waves (see the header)
  tail_stages<false>(ct, a, row0, acc3, wr3);
}

}  // namespace hx

int launch_encoder_heads_f16x2_w8(const float* x, int64_t ldx, const float* raw, int normalize, int64_t B, int F,
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
  const dim3 grid((unsigned)((B + TILE_FACES - 1) / TILE_FACES)), block(512);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a.norm) {
    if (vec4) hipLaunchKernelGGL((hx::encoder_heads_f16x2_w8_kernel<true, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((hx::encoder_heads_f16x2_w8_kernel<false, true>), grid, block, 0, st, a);
  } else {
    if (vec4) hipLaunchKernelGGL((hx::encoder_heads_f16x2_w8_kernel<true, false>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((hx::encoder_heads_f16x2_w8_kernel<false, false>), grid, block, 0, st, a);
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

}  // namespace nlml
