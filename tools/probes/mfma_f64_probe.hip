// Microbenchmark: what does v_mfma_f64_16x16x4_f64 sustain on gfx950, and do VALU instructions overlap with it?
// Build: hipcc --offload-arch=gfx950 -O3 -o exp_libs/mfma_f64_probe tools/probes/mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <cstdint>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: MFMA only; 1: + one v_cvt_f64_f32 per MFMA; 2: + four int VALU ops per MFMA; 3: + one v_fma_f64 per MFMA
__global__ __launch_bounds__(512, 2) void probe(double* out, uint64_t* clk, int iters, float seed) {
  f64x4 acc[11];
#pragma unroll
  for (int i = 0; i < 11; ++i) acc[i] = f64x4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3;
  float w[11];
  uint32_t u[11];
  double dd[11];
#pragma unroll
  for (int i = 0; i < 11; ++i) { w[i] = seed + i + threadIdx.x; u[i] = threadIdx.x * 77 + i; dd[i] = seed + i; }
  const uint64_t c0 = clock64(), t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 11; ++i) {
      double b;
      if (MODE == 1) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(b) : "v"(w[i])); }
      else if (MODE == 2) {
        uint32_t x = u[i];
        asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(it));
        u[i] = x; b = dd[i];
      } else if (MODE == 3) { asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(dd[i]) : "v"(a)); b = a; }
      else b = dd[i];
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
  }
  const uint64_t c1 = clock64(), t1 = wall_clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 11; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + u[i] + dd[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[4 * blockIdx.x] = c1 - c0; clk[4 * blockIdx.x + 1] = t1 - t0; clk[4 * blockIdx.x + 2] = t0; clk[4 * blockIdx.x + 3] = t1; }
}

template <int MODE>
void run(const char* name, int blocks) {
  double* out; uint64_t* clk;
  hipMalloc(&out, blocks * 512 * 8); hipMalloc(&clk, blocks * 32);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<blocks, 512>>>(out, clk, iters, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MODE><<<blocks, 512>>>(out, clk, iters, 1.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  uint64_t* h = new uint64_t[4 * blocks]; hipMemcpy(h, clk, 32 * blocks, hipMemcpyDeviceToHost);
  uint64_t tmin = ~0ull, tmax = 0, dmin = ~0ull, dmax = 0, late = 0;
  for (int b = 0; b < blocks; ++b) { tmin = h[4*b+2] < tmin ? h[4*b+2] : tmin; tmax = h[4*b+3] > tmax ? h[4*b+3] : tmax;
    dmin = h[4*b+1] < dmin ? h[4*b+1] : dmin; dmax = h[4*b+1] > dmax ? h[4*b+1] : dmax; }
  for (int b = 0; b < blocks; ++b) late += (h[4*b+2] - tmin) > dmin / 2;
  printf("   span %.1f us  block dur min %.1f max %.1f us  blocks starting late: %llu\n", (tmax - tmin) * 0.01, dmin * 0.01, dmax * 0.01, (unsigned long long)late);
  const double mfmas = (double)blocks * 8 * iters * 11;
  const double flops = mfmas * 16 * 16 * 4 * 2;
  // clock64 = s_memtime (core clock? on gfx9 it is the shader clock counter), wall_clock64 = 100 MHz
  printf("%-28s blocks=%4d  %.3f ms  %.1f TFLOP/s f64  clock64/MFMA-per-SIMD=%.1f  core-clk-est=%.2f GHz (clock64 %llu wall %llu)\n",
         name, blocks, ms, flops / ms / 1e9, (double)h[0] / (iters * 11 * 2), (double)h[0] / ((double)h[1] * 10.0) , (unsigned long long)h[0], (unsigned long long)h[1]);
  hipFree(out); hipFree(clk);
}

int main() {
  for (int blocks : {256, 1024}) {
    run<0>("mfma only", blocks);
    run<1>("mfma + v_cvt_f64_f32", blocks);
    run<2>("mfma + 4 int valu", blocks);
    run<3>("mfma + v_fma_f64", blocks);
  }
  return 0;
}
