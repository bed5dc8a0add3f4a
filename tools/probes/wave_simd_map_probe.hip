// wave_simd_map_probe.hip -- which SIMD does wave w of a 768-thread workgroup run on?  (tucker_ref.h's balanced pass assumes w & 3.)
// build: hipcc --offload-arch=gfx950 -O2 -o exp_libs/wave_simd_map_probe tools/probes/wave_simd_map_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(768) void probe(unsigned* out) {
  unsigned hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 12 + (threadIdx.x >> 6)] = hwid;
}
int main() {
  const int G = 512;
  unsigned* d;
  hipMalloc(&d, G * 12 * 4);
  hipLaunchKernelGGL(probe, dim3(G), dim3(768), 0, 0, d);
  std::vector<unsigned> h(G * 12);
  hipMemcpy(h.data(), d, G * 12 * 4, hipMemcpyDeviceToHost);
  int ok = 0;
  for (int g = 0; g < G; ++g) {
    bool rr = true;
    for (int w = 0; w < 12; ++w) rr &= ((h[g * 12 + w] >> 4) & 3) == ((h[g * 12] >> 4) + w) % 4 % 4;
    ok += rr;
  }
  // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
  for (int g = 0; g < 4; ++g) {
    printf("wg %d: simd of waves 0..11:", g);
    for (int w = 0; w < 12; ++w) printf(" %u", (h[g * 12 + w] >> 4) & 3);
    printf("   cu %u se %u\n", (h[g * 12] >> 8) & 15, (h[g * 12] >> 13) & 7);
  }
  int pat = 0;
  for (int g = 0; g < G; ++g) {
    bool p = true;
    for (int w = 0; w < 12; ++w) p &= ((h[g * 12 + w] >> 4) & 3) == (unsigned)(w & 3);
    pat += p;
  }
  printf("workgroups with simd(w) == w & 3: %d of %d; round-robin from any start: %d\n", pat, G, ok);
  return 0;
}
