// Probe 2: which form of separately rounded f64 vector operation issues at what rate on gfx950 -- mul with two vector operands
// (distinct factor registers per chain), mul with a scalar-register operand, add, and the reference-order pass's own pattern
// (8 chains x [mul v, mul v, mul v, mul s, add into a separate accumulator]) -- at 1, 2 and 3 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -w -o exp_libs/f64_valu_probe2 tools/probes/f64_valu_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int CH = 8;

template <int KIND>
__global__ void k(double* out, const double* fin, int iters) {
  double t[CH], f[CH], a[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) { t[c] = 1.0 + 1e-9 * (threadIdx.x + c); f[c] = fin[c] + 1e-12 * threadIdx.x; a[c] = 0.0; }
  const double s0 = fin[8], s1 = fin[9];   // uniform -> scalar registers
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (KIND == 0) {   // mul v,v,v with per-chain factors: 5 rounds
#pragma unroll
        for (int st = 0; st < 5; ++st)
#pragma unroll
          for (int c = 0; c < CH; ++c) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(t[c]) : "v"(f[(c + st) % CH]));
      } else if (KIND == 1) {   // mul v,v,s
#pragma unroll
        for (int st = 0; st < 5; ++st)
#pragma unroll
          for (int c = 0; c < CH; ++c) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(t[c]) : "s"(s0));
      } else if (KIND == 2) {   // add v,v,v
#pragma unroll
        for (int st = 0; st < 5; ++st)
#pragma unroll
          for (int c = 0; c < CH; ++c) asm volatile("v_add_f64 %0, %0, %1" : "+v"(t[c]) : "v"(f[(c + st) % CH]));
      } else {   // the pass's pattern: t = w*u; t*=fy; t*=fp; t*=fr(s); acc = t + acc
#pragma unroll
        for (int c = 0; c < CH; ++c) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(t[c]) : "v"(f[c]), "v"(f[(c + 1) % CH]));
#pragma unroll
        for (int c = 0; c < CH; ++c) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(t[c]) : "v"(f[(c + 2) % CH]));
#pragma unroll
        for (int c = 0; c < CH; ++c) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(t[c]) : "v"(f[(c + 3) % CH]));
#pragma unroll
        for (int c = 0; c < CH; ++c) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(t[c]) : "s"(c & 1 ? s0 : s1));
#pragma unroll
        for (int c = 0; c < CH; ++c) asm volatile("v_add_f64 %0, %1, %0" : "+v"(a[c]) : "v"(t[c]));
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += t[c] + a[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(double* d, const double* f, int wps, const char* name) {
  const int iters = 2000;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * wps), 0, 0, d, f, iters);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256 * wps), 0, 0, d, f, iters);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double ops = (double)iters * 4 * 5 * CH;
  const double tops = ops * wps * 4 * 256 * 64 / (ms * 1e-3) / 1e12;
  printf("%-28s waves/SIMD %d : %.3f ms  %.2f T op/s (%.3f of 39.3)\n", name, wps, ms, tops, tops / 39.3);
}

int main() {
  double *d, *f;
  hipMalloc(&d, 256 * 1024 * 8);
  hipMalloc(&f, 16 * 8);
  double h[16];
  for (int i = 0; i < 16; ++i) h[i] = 1.0 + 1e-7 * i;
  hipMemcpy(f, h, sizeof h, hipMemcpyHostToDevice);
  for (int wps = 1; wps <= 3; ++wps) {
    run<0>(d, f, wps, "mul v,v,v (8 chains)");
    run<1>(d, f, wps, "mul v,v,s");
    run<2>(d, f, wps, "add v,v,v");
    run<3>(d, f, wps, "pass pattern (4 mul + add)");
  }
  return 0;
}
