// artefacts.hip -- the two artefact producers of the reference that are pure tensor algebra.
//
//   cosine table     U[i][j] = a_j * cos(b_j * w_i + c_j) + d_j : the heads' training inputs, built by two nested Python
//                    loops over cosine() in NLML_HPE_MLPHeadsTrainer.py:71-73,179-205 (w_i: f32 radians, (a,b,c,d): rows
//                    of optimized_{yaw,pitch,roll}, f64 arithmetic)
//   mode-5 product   W = core x_5 U_feat, i.e. W[q][m] = sum_r core[q][r] * U_feat[m][r], q over the 135 = 5*3*3*3
//                    leading indices: tl.tensordot(core, transpose(feature_matrix), axes=(4, 0)) in TD_main.py:232-238
//
// Both run once per trained model, so they are plain one-thread-per-output kernels; the product is an r-ascending fmaf
// chain per output (the C oracle's order), f32 like the reference's tensors.
#include <hip/hip_runtime.h>

#include "abi_internal.h"
#include "cr_cos.h"

namespace nlml {

__global__ __launch_bounds__(256) void cosine_table_kernel(const float* __restrict__ w, int64_t n,
                                                           const double* __restrict__ p, int R, double* __restrict__ out) {
#pragma clang fp contract(off)   // without this b*w + c becomes ONE fma and the argument of the cos differs from numpy's in its
                                 // last place (then the cos by up to 256 ulps near its zeros)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * R) return;
  const int64_t row = i / R;
  const int j = (int)(i % R);
  const double* q = p + 4 * j;
  // numpy evaluates b*w, + c, cos, a*, + d as separately rounded f64 operations: no fma contraction here
  // (plain operators under the pragma: the header's __dmul_rn / __dadd_rn are inline functions whose * and + carry THEIR
  // translation unit's contraction licence into this one when inlined)
  const double bw = q[1] * (double)w[row];
  const double arg = bw + q[2];
  const double ac = q[0] * cr_cos(arg);   // correctly rounded cos (cr_cos.h): libm's value in all but ~1e-3 of the cases, and then 1 ulp away
  out[i] = ac + q[3];
}

// 16 x 16 outputs per workgroup, the K range staged through LDS in slices of 64 (both operands are K-contiguous)
__global__ __launch_bounds__(256) void mode5_product_kernel(const float* __restrict__ core, const float* __restrict__ U,
                                                            int Q, int R5, int M, float* __restrict__ W) {
  __shared__ float cs[16][65], us[16][65];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int q0 = blockIdx.y * 16, m0 = blockIdx.x * 16;
  float acc = 0.0f;
  for (int r0 = 0; r0 < R5; r0 += 64) {
    for (int i = threadIdx.x; i < 16 * 64; i += 256) {
      const int row = i >> 6, k = i & 63;
      cs[row][k] = (q0 + row < Q && r0 + k < R5) ? core[(size_t)(q0 + row) * R5 + r0 + k] : 0.0f;
      us[row][k] = (m0 + row < M && r0 + k < R5) ? U[(size_t)(m0 + row) * R5 + r0 + k] : 0.0f;
    }
    __syncthreads();
    const int kmax = R5 - r0 < 64 ? R5 - r0 : 64;
    for (int k = 0; k < kmax; ++k) acc = fmaf(cs[ty][k], us[tx][k], acc);
    __syncthreads();
  }
  if (q0 + ty < Q && m0 + tx < M) W[(size_t)(q0 + ty) * M + m0 + tx] = acc;
}

int launch_cosine_table(const float* angles, int64_t n, const double* cos_params, int R, double* out, void* stream) {
  if (n == 0 || R == 0) return 0;
  const int64_t total = n * R;
  hipLaunchKernelGGL(cosine_table_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), angles, n, cos_params, R, out);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

int launch_mode5_product(const float* core, const float* U, int Q, int R5, int M, float* W, void* stream) {
  if (Q == 0 || M == 0) return 0;
  hipLaunchKernelGGL(mode5_product_kernel, dim3((unsigned)((M + 15) / 16), (unsigned)((Q + 15) / 16)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), core, U, Q, R5, M, W);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

}  // namespace nlml
