// Probe: does v_mfma_f32_32x32x16_f16 keep f16 subnormal inputs on gfx950, and what does it cost?
// Build: hipcc --offload-arch=gfx950 -O3 -w -o exp_libs/mfma_f16_probe tools/probes/mfma_f16_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void denorm(float* out, float aval, float bval) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
  // A[row=lane&31][k = 8*(lane>>5) + i]; B[k][col = lane&31]: put one product at k=0: row r, col c all lanes
  if ((threadIdx.x >> 5) == 0) { a[0] = (_Float16)aval; b[0] = (_Float16)bval; }
  f16v c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void rate(float* out, uint64_t* clk, int iters) {
  f16v acc[8];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  float v[8]; for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
  const uint64_t c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 1) {   // 4 VALU f32 ops per MFMA
        asm volatile("v_add_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(1.0001f));
      }
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
  }
  const uint64_t c1 = clock64();
  float s = 0; for (int i = 0; i < 8; ++i) { for (int j = 0; j < 16; ++j) s += acc[i][j]; s += v[i]; }
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}

template <int MODE> void run(const char* name) {
  const int blocks = 1024, iters = 4000;
  float* out; uint64_t* clk; hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  rate<MODE><<<blocks, 256>>>(out, clk, iters); hipDeviceSynchronize();
  hipEventRecord(e0); rate<MODE><<<blocks, 256>>>(out, clk, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  uint64_t h; hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
  const double flops = (double)blocks * 4 * iters * 8 * 32 * 32 * 16 * 2;
  printf("%-24s %.3f ms  %.0f TFLOP/s  cycles/MFMA (1 wave/SIMD) = %.1f\n", name, ms, flops / ms / 1e9, (double)h / (iters * 8));
}

int main() {
  float* d; hipMalloc(&d, 4); float h;
  const float cases[][2] = {{1.0f, 1.0f}, {3.0e-5f, 1.0f}, {1.0f, 3.0e-5f}, {6.0e-8f, 1.0f}, {3.0e-5f, 3.0e-5f}, {1.0e-3f, 1.0e-3f}};
  for (auto& c : cases) {
    denorm<<<1, 64>>>(d, c[0], c[1]); hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("a=%g (f16 %g) b=%g (f16 %g) -> mfma %.9g   expected %.9g\n", c[0], (float)(_Float16)c[0], c[1], (float)(_Float16)c[1], h,
           (float)(_Float16)c[0] * (float)(_Float16)c[1]);
  }
  run<0>("f16 32x32x16 only");
  run<1>("f16 mfma + 4 f32 valu");
  return 0;
}
