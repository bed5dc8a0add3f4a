// Probe: the 3-instruction hi/lo split (v_cvt_pk_f16_f32 + v_fma_mixlo/mixhi_f16) against the plain C form, bit for bit,
// over normal, subnormal-lo, tiny, huge, inf and NaN inputs.
// Build: hipcc --offload-arch=gfx950 -O3 -w -o exp_libs/split_probe tools/probes/split_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

__device__ inline void split2_asm(float v0, float v1, unsigned& hi, unsigned& lo) {
  asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
               "v_fma_mixlo_f16 %1, -%0, 1.0, %2 op_sel_hi:[1,0,0]\n\t"
               "v_fma_mixhi_f16 %1, -%0, 1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
               : "=&v"(hi), "=&v"(lo) : "v"(v0), "v"(v1));
}

__global__ void k(const float* in, unsigned* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n / 2) return;
  float v0 = in[2 * i], v1 = in[2 * i + 1];
  asm volatile("" : "+v"(v0), "+v"(v1));
  h2 hi, lo;
  hi[0] = (_Float16)v0; hi[1] = (_Float16)v1;
  lo[0] = (_Float16)(v0 - (float)hi[0]);
  lo[1] = (_Float16)(v1 - (float)hi[1]);
  unsigned ahi, alo;
  split2_asm(v0, v1, ahi, alo);
  out[4 * i + 0] = __builtin_bit_cast(unsigned, hi);
  out[4 * i + 1] = __builtin_bit_cast(unsigned, lo);
  out[4 * i + 2] = ahi;
  out[4 * i + 3] = alo;
}

int main() {
  const int n = 1 << 22;
  std::vector<float> h(n);
  uint32_t s = 12345u;
  for (int i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    uint32_t bits = s;
    if ((i & 7) == 0) bits = (s & 0x807fffffu) | ((100u + (s >> 23) % 60u) << 23);   // exponents around f16's range edges
    memcpy(&h[i], &bits, 4);
  }
  const float special[] = {0.f, -0.f, 65504.f, 65519.9f, 65520.f, 7e4f, -7e4f, 6.1e-5f, 5.96e-8f, 2.98e-8f, 1e-10f, 1.f / 3.f,
                           __builtin_inff(), -__builtin_inff(), __builtin_nanf(""), -__builtin_nanf("")};
  for (int i = 0; i < 16; ++i) h[i] = special[i];
  float* d; unsigned* o;
  hipMalloc(&d, n * 4); hipMalloc(&o, n * 2 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 2 / 256, 256>>>(d, o, n);
  std::vector<unsigned> r(n * 2);
  hipMemcpy(r.data(), o, n * 2 * 4, hipMemcpyDeviceToHost);
  long bad_hi = 0, bad_lo = 0, nan_diff = 0;
  for (int i = 0; i < n / 2; ++i) {
    if (r[4 * i] != r[4 * i + 2]) ++bad_hi;
    if (r[4 * i + 1] != r[4 * i + 3]) {
      // NaN payload/sign differences are not value differences
      auto isnan16 = [](unsigned x) { return (x & 0x7c00u) == 0x7c00u && (x & 0x3ffu); };
      const unsigned a = r[4 * i + 1], b = r[4 * i + 3];
      const bool lo_ok = ((a & 0xffffu) == (b & 0xffffu)) || (isnan16(a & 0xffffu) && isnan16(b & 0xffffu));
      const bool hi_ok = ((a >> 16) == (b >> 16)) || (isnan16(a >> 16) && isnan16(b >> 16));
      if (lo_ok && hi_ok) ++nan_diff; else { if (bad_lo < 8) printf("lo differs: in %g %g  C %08x asm %08x\n", h[2 * i], h[2 * i + 1], a, b); ++bad_lo; }
    }
  }
  printf("pairs %d: hi mismatches %ld, lo mismatches %ld (NaN-encoding-only differences %ld)\n", n / 2, bad_hi, bad_lo, nan_diff);
  return (bad_hi || bad_lo) ? 1 : 0;
}
