// normalize_ipd.hip -- K1: stand-alone IPD landmark normalisation (HBM-bound streaming kernel).
//
// Replaces Read_Landmarks_and_Normalizing_using_IPD (helpers/FeatureExtractor.py:30-66) and the
// f32 cast of its callers (:101): out = f32((f64(v) - f64(lm[1][c])) / ipd), with
// ipd = ||lm[33] - lm[263]||_2 in f64 (np.linalg.norm == sqrt of an fma-chained dot on the
// reference's BLAS; reproduced explicitly below), 1e-6 when exactly 0 (:47-48).
//
// One wave per face: a face is 1404 f32 = 351 float4, read and written as fully coalesced
// 16-B-per-lane accesses (6 wave-instructions each way).  Algorithmic bytes per face:
// 5616 read + 5616 written = 11,232 B.  The f64 subtract + correctly-rounded divide (reciprocal +
// two fma corrections per element) hides under the stream.
#include <hip/hip_runtime.h>

#include "../../include/nlml_hpe.h"
#include "abi_internal.h"

namespace nlml {

// n / d correctly rounded in f64 from y = RN(1/d) (Markstein): q = n*y, r = n - q*d exactly by fma,
// q' = q + r*y.  Same routine as the fused kernel's staging (encoder_heads.hip), so K1 -> K2 and the fused
// path are bit-identical; 3 multiply-adds per element instead of an IEEE division sequence.
__device__ __forceinline__ double div_ipd(double n, double d, double y) {
  const double q = n * y;
  const double r = fma(-q, d, n);
  return fma(r, y, q);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int F4 = NLML_F_REFERENCE / 4;  // 351 float4 per face

__global__ __launch_bounds__(256) void normalize_ipd_kernel(const float* __restrict__ raw, int64_t B,
                                                            int normalize, float* __restrict__ out,
                                                            uint8_t* __restrict__ valid) {
  const int lane = threadIdx.x & 63;
  const int64_t face = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (face >= B) return;  // whole wave leaves together
  const float* p = raw + face * NLML_F_REFERENCE;
  float* o = out + face * NLML_F_REFERENCE;

  // all loads first (6 x 16 B per lane + the 9 reference floats), none of them depends on another:
  // the whole face is in flight before any arithmetic starts
  f32x4 v[6];
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int i = it * 64 + lane;
    v[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + (i < F4 ? i : F4 - 1));
  }
  double rx = 0.0, ry = 0.0, rz = 0.0, ipd = 1.0;
  if (normalize) {
    rx = (double)p[3]; ry = (double)p[4]; rz = (double)p[5];                // nose tip, landmark 1
    const double dx = (double)p[99] - (double)p[789];                        // landmark 33 - 263
    const double dy = (double)p[100] - (double)p[790];
    const double dz = (double)p[101] - (double)p[791];
    ipd = sqrt(fma(dz, dz, fma(dy, dy, dx * dx)));
    if (ipd == 0.0) ipd = 1e-6;
  }
  const double rcp = 1.0 / ipd;
  // element e of float4 i is column 4i+e, coordinate (4i+e) % 3; i = 64*it + lane and 256 % 3 == 1, so the phase
  // of a lane advances by one per iteration: rotate (a, b, c) instead of taking a modulo per element
  const int ph = (4 * lane) % 3;
  double a = ph == 0 ? rx : (ph == 1 ? ry : rz), b = ph == 0 ? ry : (ph == 1 ? rz : rx), c = ph == 0 ? rz : (ph == 1 ? rx : ry);
  unsigned nzbits = 0u;
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int i = it * 64 + lane;
    if (normalize) {
      v[it][0] = (float)div_ipd((double)v[it][0] - a, ipd, rcp);
      v[it][1] = (float)div_ipd((double)v[it][1] - b, ipd, rcp);
      v[it][2] = (float)div_ipd((double)v[it][2] - c, ipd, rcp);
      v[it][3] = (float)div_ipd((double)v[it][3] - a, ipd, rcp);
      const double t = a; a = b; b = c; c = t;     // next iteration: columns + 256 => phase + 1
    }
    if (i < F4) {
      nzbits |= (__float_as_uint(v[it][0]) | __float_as_uint(v[it][1]) | __float_as_uint(v[it][2]) | __float_as_uint(v[it][3])) & 0x7fffffffu;
      __builtin_nontemporal_store(v[it], reinterpret_cast<f32x4*>(o) + i);
    }
  }
  const bool nz = nzbits != 0u;
  if (valid) {
    const unsigned long long m = __ballot(nz);
    if (lane == 0) valid[face] = m ? 1 : 0;
  }
}

int launch_normalize_ipd(const float* raw, int64_t B, int normalize, float* out, uint8_t* valid,
                         void* stream) {
  if (B == 0) return 0;
  const dim3 grid((unsigned)((B + 3) / 4)), block(256);
  hipLaunchKernelGGL(normalize_ipd_kernel, grid, block, 0, reinterpret_cast<hipStream_t>(stream), raw, B,
                     normalize, out, valid);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

}  // namespace nlml
