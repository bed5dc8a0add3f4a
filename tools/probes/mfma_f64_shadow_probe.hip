// Probe: what hides in the shadow of an f64 MFMA on gfx950?  Per v_mfma_f64_16x16x4_f64 (64 cycles) or v_mfma_f64_4x4x4_4b_f64
// (17 cycles) a wave also issues NV independent vector-ALU instructions of one kind: f64 conversions (the DP path), 32-bit integer
// adds, f32 fmas.  If the time per MFMA does not grow, that kind of instruction is free next to the matrix pipe.
// Build: hipcc --offload-arch=gfx950 -O3 -w -o exp_libs/mfma_f64_shadow_probe tools/probes/mfma_f64_shadow_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double f64x4 __attribute__((ext_vector_type(4)));

// KIND 0 none, 1 v_cvt_f64_f32, 2 v_add_u32, 3 v_fma_f32
template <int SHAPE, int KIND, int NV>
__global__ __launch_bounds__(512, 1) void rate(double* out, int iters) {
  double a = threadIdx.x * 1e-3 + 1.0, b = threadIdx.x * 2e-3 + 0.5;
  float fs[8];
  unsigned us[8];
  double ds[8];
  for (int i = 0; i < 8; ++i) { fs[i] = threadIdx.x * 0.5f + i; us[i] = threadIdx.x + i; ds[i] = 0; }
  f64x4 acc16[11];
  double acc4[11];
  for (int i = 0; i < 11; ++i) { acc16[i] = f64x4{0, 0, 0, 0}; acc4[i] = 0; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 11; ++i) {
      if (SHAPE == 16) acc16[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc16[i], 0, 0, 0);
      else acc4[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc4[i], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (KIND == 1) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(ds[v % 8]) : "v"(fs[v % 8]));
        if (KIND == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(us[v % 8]) : "v"(us[(v + 1) % 8]));
        if (KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(fs[v % 8]) : "v"(fs[(v + 1) % 8]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double s = 0;
  for (int i = 0; i < 11; ++i) s += acc16[i][0] + acc16[i][1] + acc16[i][2] + acc16[i][3] + acc4[i];
  for (int i = 0; i < 8; ++i) s += ds[i] + fs[i] + us[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int SHAPE, int KIND, int NV>
void run(double* out, const char* what) {
  const int iters = 2000;
  for (int threads : {256, 512}) {
    rate<SHAPE, KIND, NV><<<256, threads>>>(out, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    rate<SHAPE, KIND, NV><<<256, threads>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e-3 / (iters * 11.0 * (threads / 256));   // seconds per MFMA per SIMD
    printf("shape %2d  %-22s x%d  %d waves/SIMD: %6.2f ns per MFMA per SIMD (= %5.1f cycles at 2.4 GHz)\n", SHAPE, what, NV, threads / 256,
           per * 1e9, per * 2.4e9);
  }
}

int main() {
  double* out; hipMalloc(&out, 256 * 512 * 8);
  run<16, 0, 0>(out, "alone");
  run<16, 1, 1>(out, "+ v_cvt_f64_f32");
  run<16, 1, 2>(out, "+ v_cvt_f64_f32");
  run<16, 2, 4>(out, "+ v_add_u32");
  run<16, 2, 8>(out, "+ v_add_u32");
  run<16, 2, 12>(out, "+ v_add_u32");
  run<16, 3, 8>(out, "+ v_fma_f32");
  run<4, 0, 0>(out, "alone");
  run<4, 1, 1>(out, "+ v_cvt_f64_f32");
  run<4, 2, 2>(out, "+ v_add_u32");
  run<4, 2, 4>(out, "+ v_add_u32");
  run<4, 2, 6>(out, "+ v_add_u32");
  run<4, 3, 4>(out, "+ v_fma_f32");
  return 0;
}
