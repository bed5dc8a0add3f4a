// Probe: do f64 vector instructions hide behind v_mfma_f32_32x32x16_f16 on gfx950?  (a) the same wave issues NV f64 fmas (or f32
// fmas) per MFMA; (b) every SIMD runs ONE matrix wave and ONE vector-only wave (blockDim 512: waves 0-3 MFMA, 4-7 f64 fma loop),
// and the matrix waves' rate is compared with running alone.  DESIGN.md section 3 (layer-0 staging) quotes the result.
// Build: hipcc --offload-arch=gfx950 -O3 -w -o exp_libs/mfma_f16_dp_probe tools/probes/mfma_f16_dp_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// KIND 0 none, 1 v_fma_f64, 2 v_fma_f32; ROLE 0: all waves MFMA (+NV vector ops each); ROLE 1: waves 4..7 run only vector ops
template <int KIND, int NV, int ROLE>
__global__ __launch_bounds__(512, 1) void rate(float* out, unsigned long long* clk, int iters) {
  const int wv = threadIdx.x >> 6;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (threadIdx.x % 7 + i)); b[i] = (_Float16)(0.02f * (threadIdx.x % 5 + i)); }
  double ds[8];
  float fs[8];
  for (int i = 0; i < 8; ++i) { ds[i] = 1.0 + threadIdx.x * 1e-3 + i; fs[i] = 1.0f + i; }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (ROLE == 1 && wv >= 4) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int v = 0; v < 4 * NV; ++v) {
        if (KIND == 1) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(ds[v % 8]) : "v"(ds[(v + 1) % 8]));
        if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(fs[v % 8]) : "v"(fs[(v + 1) % 8]));
      }
  } else {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        if (ROLE == 0) {
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            if (KIND == 1) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(ds[v % 8]) : "v"(ds[(v + 1) % 8]));
            if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(fs[v % 8]) : "v"(fs[(v + 1) % 8]));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  for (int i = 0; i < 8; ++i) s += (float)ds[i] + fs[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 8 + wv] = t1 - t0;
}

template <int KIND, int NV, int ROLE>
void run(float* out, unsigned long long* clk, int threads, const char* what) {
  const int iters = 4000;
  rate<KIND, NV, ROLE><<<256, threads>>>(out, clk, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  rate<KIND, NV, ROLE><<<256, threads>>>(out, clk, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8];
  hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
  // memtime ticks at 100 MHz: ticks * 10 ns
  printf("%-58s kernel %7.3f ms = %6.2f ns per MFMA of a matrix wave; wave0 %6.2f ns/MFMA", what, ms, ms * 1e6 / (iters * 4.0),
         h[0] * 10.0 / (iters * 4.0));
  if (threads == 512) printf(", wave4 %6.2f ns per MFMA-slot", h[4] * 10.0 / (iters * 4.0));
  printf("\n");
}

int main() {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 8 * 8);
  run<0, 0, 0>(out, clk, 256, "4 matrix waves (1 per SIMD), MFMA only");
  run<2, 4, 0>(out, clk, 256, "  + 4 v_fma_f32 per MFMA in the same wave");
  run<2, 6, 0>(out, clk, 256, "  + 6 v_fma_f32 per MFMA in the same wave");
  run<1, 1, 0>(out, clk, 256, "  + 1 v_fma_f64 per MFMA in the same wave");
  run<1, 2, 0>(out, clk, 256, "  + 2 v_fma_f64 per MFMA in the same wave");
  run<1, 3, 0>(out, clk, 256, "  + 3 v_fma_f64 per MFMA in the same wave");
  run<1, 2, 1>(out, clk, 512, "4 matrix waves + 4 vector waves: 2 v_fma_f64 per MFMA slot");
  run<1, 3, 1>(out, clk, 512, "4 matrix waves + 4 vector waves: 3 v_fma_f64 per MFMA slot");
  run<2, 6, 1>(out, clk, 512, "4 matrix waves + 4 vector waves: 6 v_fma_f32 per MFMA slot");
  return 0;
}
