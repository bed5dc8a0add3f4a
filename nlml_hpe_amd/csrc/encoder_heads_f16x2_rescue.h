// encoder_heads_f16x2_rescue.h -- the split-f16 kernels' slow path for faces whose activations leave f16's range.
//
// The f16 pieces of the split-f16 mode hold |v| < 65520.  A face with a larger activation anywhere in the network -- the
// reference's own ipd == 0 -> 1e-6 branch (helpers/FeatureExtractor.py:47-48) makes features of ~1e6; un-normalised
// pixel coordinates do it too -- comes out of the MFMA path as NaN (hi = inf, lo = -inf), never as a wrong number.  The
// reference (f32 ATen) returns a finite pose for such a face, so the tile that produced a non-finite pose re-evaluates
// THAT face here: the same blob (weight = hi + lo pieces, the stage's power-of-two scale undone), the same stages and
// jobs, but f32 activations and a K-ascending f32 fma chain on the vector ALUs -- no range limit, the accuracy class of
// the f32 kernel.  A face whose INPUT holds NaN/Inf is re-evaluated too and stays non-finite, as it does in the reference.
//
// Cost: nothing on the normal path beyond one LDS flag per face and one ballot per tile; a rescued group of up to RG
// faces streams the whole blob (9.6 MB) through its CU once (~0.1-0.2 ms).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/nlml_hpe.h"
#include "layout.h"

namespace nlml {
namespace hx {

constexpr int RESCUE_G = 4;          // faces per group (they share every weight load)
constexpr int RESCUE_W = 1024;       // floats per face in one activation buffer (widest layer output)

struct RescueIn {
  const float* base[RESCUE_G];       // per face: activation vector (LDS) or the face's x row (global, layer 0)
  double ref[RESCUE_G][3], ipd[RESCUE_G], rcp[RESCUE_G];
  int K;                             // true contraction length (values beyond read as 0)
  int mode;                          // 0: LDS f32 vector; 1: global x row; 2: global raw landmarks, IPD-normalised on the fly
};

__device__ __forceinline__ double rescue_div(double n, double d, double y) {   // == IEEE n / d for these operands (K1)
  const double q = n * y;
  const double r = fma(-q, d, n);
  return fma(r, y, q);
}

__device__ __forceinline__ void rescue_load8(const RescueIn& in, int g, int k0, float (&v)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = k0 + e;
    float t = 0.0f;
    if (k < in.K) {
      t = in.base[g][k];
      if (in.mode == 2) {
        const int ph = k % 3;
        const double rr = ph == 0 ? in.ref[g][0] : (ph == 1 ? in.ref[g][1] : in.ref[g][2]);
        t = (float)rescue_div((double)t - rr, in.ipd[g], in.rcp[g]);
      }
    }
    v[e] = t;
  }
}

// One job of a stage for the group: lane (r, h) accumulates neuron row r of each block over its half of every K step,
// the two halves are added, bias (stored scaled, in accumulator-register order) joins, the stage scale is undone.
// emit(nb, r, g, value) receives the pre-activation of block row r.
template <int NB, typename Emit>
__device__ __forceinline__ void rescue_job(const void* blob, int st, int job, int k16, const RescueIn& in, int lane, Emit emit) {
  const Header* hdr = reinterpret_cast<const Header*>(blob);
  const int r = lane & 31, h = lane >> 5;
  const h8* w = reinterpret_cast<const h8*>(blob) + hdr->w_off[st] + (size_t)job * hdr->job_w16[st] + lane;
  float acc[NB][RESCUE_G];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int g = 0; g < RESCUE_G; ++g) acc[nb][g] = 0.0f;
  for (int s = 0; s < k16; ++s) {
    float xv[RESCUE_G][8];
#pragma unroll
    for (int g = 0; g < RESCUE_G; ++g) rescue_load8(in, g, 16 * s + 8 * h, xv[g]);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const h8 hi = w[((size_t)(s * NB + nb) * 2 + 0) * 64], lo = w[((size_t)(s * NB + nb) * 2 + 1) * 64];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float wf = (float)hi[e] + (float)lo[e];
#pragma unroll
        for (int g = 0; g < RESCUE_G; ++g) acc[nb][g] = fmaf(wf, xv[g][e], acc[nb][g]);
      }
    }
  }
  const float inv = hdr->inv_scale[st];
  const float* bias = reinterpret_cast<const float*>(blob) + ((size_t)hdr->b_off[st] + (size_t)job * NB * 8) * 4;
  const int bh = (r >> 2) & 1, bq = (r & 3) + 4 * (r >> 3);     // row r = (q & 3) + 8 * (q >> 2) + 4 * h (layout.h)
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const float b = bias[(nb * 2 + bh) * 16 + bq];
#pragma unroll
    for (int g = 0; g < RESCUE_G; ++g) {
      const float t = acc[nb][g] + __shfl_xor(acc[nb][g], 32, 64);
      if (h == 0) emit(nb, r, g, (t + b) * inv);
    }
  }
}

__device__ __forceinline__ float rescue_relu(float v) { return v < 0.0f ? 0.0f : v; }

// All 256 threads of the workgroup; `mask`: bit f set = face row0 + f of this tile needs the slow path (uniform).
__device__ __forceinline__ void rescue_tile(const float* __restrict__ x, int64_t ldx, int F, int norm,
                                            const void* __restrict__ blob, float* __restrict__ out,
                                            float* __restrict__ latent, char* lds, int64_t row0, unsigned long long mask) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Header* hdr = reinterpret_cast<const Header*>(blob);
  float* bufA = reinterpret_cast<float*>(lds);                    // [RESCUE_G][RESCUE_W]
  float* bufB = bufA + RESCUE_G * RESCUE_W;
  float* lat = bufB + RESCUE_G * RESCUE_W;                        // [RESCUE_G][16]
  __syncthreads();                                                // the tile's LDS images are dead from here on
  while (mask) {
    int64_t row[RESCUE_G];
    int ng = 0;
#pragma unroll
    for (int g = 0; g < RESCUE_G; ++g) {
      if (mask) {
        const int f = __builtin_ctzll(mask);
        mask &= mask - 1;
        row[g] = row0 + f;
        ng = g + 1;
      } else {
        row[g] = row[0];                                          // computed again, not stored
      }
    }
    RescueIn in;
    in.K = F;
    in.mode = norm ? 2 : 1;
#pragma unroll
    for (int g = 0; g < RESCUE_G; ++g) {
      const float* p = x + row[g] * ldx;
      in.base[g] = p;
      in.ipd[g] = 1.0; in.rcp[g] = 1.0; in.ref[g][0] = in.ref[g][1] = in.ref[g][2] = 0.0;
      if (norm) {   // FeatureExtractor.py:30-66, exactly as K1 and the fused staging do it
        const double dx = (double)p[99] - (double)p[789], dy = (double)p[100] - (double)p[790], dz = (double)p[101] - (double)p[791];
        double ipd = sqrt(fma(dz, dz, fma(dy, dy, dx * dx)));
        if (ipd == 0.0) ipd = 1e-6;
        in.ipd[g] = ipd; in.rcp[g] = 1.0 / ipd;
        in.ref[g][0] = (double)p[3]; in.ref[g][1] = (double)p[4]; in.ref[g][2] = (double)p[5];
      }
    }
    // trunk: E0 (x from global) .. E4, ping-pong through bufA / bufB
    for (int job = wv; job < 8; job += 4)
      rescue_job<4>(blob, ST_E0, job, (int)hdr->k8_e0, in, lane,
                    [&](int nb, int r, int g, float v) { bufA[g * RESCUE_W + 128 * job + 32 * nb + r] = rescue_relu(v); });
    __syncthreads();
    in.mode = 0;
    auto from = [&](float* buf, int col, int K) {
#pragma unroll
      for (int g = 0; g < RESCUE_G; ++g) in.base[g] = buf + g * RESCUE_W + col;
      in.K = K;
    };
    from(bufA, 0, 1024);
    rescue_job<4>(blob, ST_E1, wv, 64, in, lane,
                  [&](int nb, int r, int g, float v) { bufB[g * RESCUE_W + 128 * wv + 32 * nb + r] = rescue_relu(v); });
    __syncthreads();
    from(bufB, 0, 512);
    rescue_job<2>(blob, ST_E2, wv, 32, in, lane,
                  [&](int nb, int r, int g, float v) { bufA[g * RESCUE_W + 64 * wv + 32 * nb + r] = rescue_relu(v); });
    __syncthreads();
    from(bufA, 0, 256);
    rescue_job<1>(blob, ST_E3, wv, 16, in, lane,
                  [&](int, int r, int g, float v) { bufB[g * RESCUE_W + 32 * wv + r] = rescue_relu(v); });
    __syncthreads();
    from(bufB, 0, 128);
    if (wv < 2)
      rescue_job<1>(blob, ST_E4, wv, 8, in, lane,
                    [&](int, int r, int g, float v) { bufA[g * RESCUE_W + 32 * wv + r] = tanhf(v); });
    __syncthreads();
    from(bufA, 0, 64);
    if (wv == 0)   // E5: latent n = 3*head + c sits on accumulator row 16*head + c (pack.cpp row_of)
      rescue_job<2>(blob, ST_E5, 0, 4, in, lane, [&](int nb, int r, int g, float v) {
        const int rowi = 32 * nb + r, hd = rowi >> 4, cc = rowi & 15;
        if (hd < 3 && cc < 3) {
          lat[g * 16 + 3 * hd + cc] = v;
          if (latent && g < ng) latent[row[g] * NLML_LATENT + 3 * hd + cc] = v;
        }
      });
    __syncthreads();
    // heads: H0 (K = 3) .. H4; head of a job as in pack.cpp (jobs per head = jobs / 3)
    for (int job = wv; job < 12; job += 4) {
      in.K = 3;
#pragma unroll
      for (int g = 0; g < RESCUE_G; ++g) in.base[g] = lat + g * 16 + 3 * (job >> 2);
      rescue_job<1>(blob, ST_H0, job, 1, in, lane,
                    [&](int, int r, int g, float v) { bufB[g * RESCUE_W + 128 * (job >> 2) + 32 * (job & 3) + r] = rescue_relu(v); });
    }
    __syncthreads();
    for (int job = wv; job < 12; job += 4) {
      from(bufB, 128 * (job >> 2), 128);
      rescue_job<2>(blob, ST_H1, job, 8, in, lane, [&](int nb, int r, int g, float v) {
        bufA[g * RESCUE_W + 256 * (job >> 2) + 64 * (job & 3) + 32 * nb + r] = rescue_relu(v);
      });
    }
    __syncthreads();
    for (int job = wv; job < 12; job += 4) {
      from(bufA, 256 * (job >> 2), 256);
      rescue_job<1>(blob, ST_H2, job, 16, in, lane,
                    [&](int, int r, int g, float v) { bufB[g * RESCUE_W + 128 * (job >> 2) + 32 * (job & 3) + r] = rescue_relu(v); });
    }
    __syncthreads();
    for (int job = wv; job < 6; job += 4) {
      from(bufB, 128 * (job >> 1), 128);
      rescue_job<1>(blob, ST_H3, job, 8, in, lane,
                    [&](int, int r, int g, float v) { bufA[g * RESCUE_W + 64 * (job >> 1) + 32 * (job & 1) + r] = rescue_relu(v); });
    }
    __syncthreads();
    if (wv < 3) {
      from(bufA, 64 * wv, 64);
      rescue_job<1>(blob, ST_H4, wv, 4, in, lane, [&](int, int r, int g, float v) {
        if (r == 0 && g < ng) out[row[g] * 3 + wv] = v;
      });
    }
    __syncthreads();
  }
}

}  // namespace hx
}  // namespace nlml
