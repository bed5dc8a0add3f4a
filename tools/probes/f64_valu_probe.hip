// Probe: issue rate of separately rounded f64 vector operations (v_mul_f64 / v_add_f64) on gfx950 as a function of how many
// INDEPENDENT dependency chains a wave interleaves (CH) and how many waves share a SIMD (workgroup of 256 x WPS threads, one
// workgroup per CU).  The reference-order Tucker pass (csrc/tucker_ref.h) is five dependent operations per (column, evaluation);
// this says how many pairs must advance together to keep the vector ALUs at their issue rate (16 lanes per clock and SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 -w -o exp_libs/f64_valu_probe tools/probes/f64_valu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CH>
__global__ void chains(double* out, double m, int iters) {
  double v[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) v[c] = 1.0 + 1e-9 * (threadIdx.x + c);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int c = 0; c < CH; ++c) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[c]) : "v"(m));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += v[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0);
}

template <int CH>
void run(double* d, int wps) {
  const int iters = 2000;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(chains<CH>, dim3(256), dim3(256 * wps), 0, 0, d, 1.0000001, iters);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(chains<CH>, dim3(256), dim3(256 * wps), 0, 0, d, 1.0000001, iters);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, a, b);
  double ticks;
  hipMemcpy(&ticks, d, 8, hipMemcpyDeviceToHost);
  const double ops = (double)iters * 16 * CH;                         // wave-instructions per wave
  const double tops = ops * wps * 4 * 256 * 64 / (ms * 1e-3) / 1e12;  // lane-operations per second
  printf("chains %2d  waves/SIMD %d : %.3f ms  %.2f T op/s (%.3f of 39.3)  %.2f memtime ticks (100 MHz) per wave-instruction\n", CH, wps, ms,
         tops, tops / 39.3, ticks / ops);
}

int main() {
  double* d;
  hipMalloc(&d, 256 * 1024 * 8);
  for (int wps = 1; wps <= 2; ++wps) {
    run<1>(d, wps); run<2>(d, wps); run<3>(d, wps); run<4>(d, wps); run<6>(d, wps); run<8>(d, wps); run<12>(d, wps);
  }
  return 0;
}
