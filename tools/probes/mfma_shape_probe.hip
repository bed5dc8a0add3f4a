// Probe: sustained (power-limited) f16 MFMA rate of the two gfx950 shapes on random operands, whole chip.
// Build: hipcc --offload-arch=gfx950 -O3 -w -o exp_libs/mfma_shape_probe tools/probes/mfma_shape_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

__device__ inline uint32_t rnd(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }
__device__ inline h8 rand_h8(uint32_t& s, float scale) {
  h8 v;
  for (int i = 0; i < 8; ++i) v[i] = (_Float16)(((int)(rnd(s) >> 8) - (1 << 23)) * (scale / (1 << 23)));
  return v;
}

// SHAPE 0: 32x32x16, 8 accumulators (4 A x 2 B operands).  SHAPE 1: 16x16x32, 32 accumulators (8 A x 4 B): the same MACs per
// operand byte held in registers as the K2 tile (128 neurons x 64 faces per wave).
template <int SHAPE>
__global__ __launch_bounds__(256, 1) void rate(float* out, int iters) {
  uint32_t s = threadIdx.x * 7919u + blockIdx.x * 104729u + 1u;
  float sum = 0.f;
  if (SHAPE == 0) {
    f16v acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
    h8 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = rand_h8(s, 1.0f);
    for (int j = 0; j < 2; ++j) b[j] = rand_h8(s, 1.0f);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 3; ++rep)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
      asm volatile("" : "+v"(a[0]));
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 16; ++k) sum += acc[i][j][k];
  } else {
    f4v acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int k = 0; k < 4; ++k) acc[i][j][k] = 0.f;
    h8 a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = rand_h8(s, 1.0f);
    for (int j = 0; j < 4; ++j) b[j] = rand_h8(s, 1.0f);
    for (int it = 0; it < iters; ++it) {   // per iteration: 3 * 8 * 4 = 96 MFMAs of 8192 MACs = the 24 MFMAs of 16384 x 2 (K = 32)
#pragma unroll
      for (int rep = 0; rep < 3; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
      asm volatile("" : "+v"(a[0]));
    }
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int k = 0; k < 4; ++k) sum += acc[i][j][k];
  }
  out[blockIdx.x * 256 + threadIdx.x] = sum;
}

template <int SHAPE> void run(const char* name, int launches) {
  const int blocks = 1024;
  const int iters = SHAPE == 0 ? 4000 : 2000;     // same MACs per launch
  float* out; hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) rate<SHAPE><<<blocks, 256>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < launches; ++i) rate<SHAPE><<<blocks, 256>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double macs_per_it = SHAPE == 0 ? 24.0 * 32 * 32 * 16 : 96.0 * 16 * 16 * 32;
  const double flops = (double)launches * blocks * 4 * iters * macs_per_it * 2;
  printf("%-28s %8.2f ms / %d launches  %.0f TFLOP/s sustained\n", name, ms, launches, flops / ms / 1e9);
  fflush(stdout);
  hipFree(out);
}

int main() {
  for (int r = 0; r < 2; ++r) {
    run<0>("f16 32x32x16 (8 acc tiles)", 400);
    run<1>("f16 16x16x32 (32 acc tiles)", 400);
  }
  return 0;
}
