// Probe: how fast does ONE CU take in a stream that always misses its L1, and does a cache policy change it?  Like K2's weight
// stream: 256 workgroups (one per CU, 4 waves) all read the SAME 9.6 MB buffer front to back (L2 hits after the first toucher),
// 16 bytes per lane per load, with the cache-policy bits of a raw buffer load: 0 (default), sc0, sc1, sc0+sc1, nt.
// Build: hipcc --offload-arch=gfx950 -O3 -w -o exp_libs/l1_stream_probe tools/probes/l1_stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// SHIFT: every workgroup starts SHIFT * blockIdx bytes further into the buffer (and wraps): the CUs then read different addresses
// at any one time instead of walking the buffer in lock step
template <int AUX, int DEPTH>
__global__ __launch_bounds__(256, 1) void stream_shift(const float* buf, uint32_t bytes, int reps, float* out, uint32_t shift) {
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, bytes, 0x00020000);
  const uint32_t tid = threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  const uint32_t step = 256 * 16;
  const uint32_t nsteps = bytes / step;
  uint32_t s0 = (blockIdx.x * (shift / step)) % nsteps;
  for (int r = 0; r < reps; ++r) {
    for (uint32_t i = 0; i + DEPTH <= nsteps; i += DEPTH) {
      f32x4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        uint32_t st = s0 + i + d;
        st = st >= nsteps ? st - nsteps : st;
        const i32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, st * step + tid * 16, 0, AUX);
        v[d] = __builtin_bit_cast(f32x4, t);
      }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc += v[d];
    }
  }
  out[blockIdx.x * 256 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int AUX, int DEPTH>
__global__ __launch_bounds__(256, 1) void stream(const float* buf, uint32_t bytes, int reps, float* out) {
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, bytes, 0x00020000);
  const uint32_t tid = threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  const uint32_t step = 256 * 16;   // bytes per workgroup-wide load
  for (int r = 0; r < reps; ++r) {
    for (uint32_t off = tid * 16; off + DEPTH * step <= bytes; off += DEPTH * step) {
      f32x4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const i32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + d * step, 0, AUX);
        v[d] = __builtin_bit_cast(f32x4, t);
      }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc += v[d];
    }
  }
  out[blockIdx.x * 256 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int AUX, int DEPTH>
void run(const float* buf, uint32_t bytes, float* out, const char* what) {
  const int reps = 8;
  stream<AUX, DEPTH><<<256, 256>>>(buf, bytes, 2, out);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  stream<AUX, DEPTH><<<256, 256>>>(buf, bytes, reps, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bps = (double)bytes * reps / (ms * 1e-3);
  printf("%-22s depth %2d: %7.3f ms  %6.1f GB/s per CU = %5.1f B/clk at 2.4 GHz (%5.1f at 1.9)\n", what, DEPTH, ms, bps / 1e9, bps / 2.4e9, bps / 1.9e9);
}

template <int DEPTH>
void run_shift(const float* buf, uint32_t bytes, float* out, uint32_t shift) {
  const int reps = 8;
  stream_shift<0, DEPTH><<<256, 256>>>(buf, bytes, 2, out, shift);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  stream_shift<0, DEPTH><<<256, 256>>>(buf, bytes, reps, out, shift);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bps = (double)bytes * reps / (ms * 1e-3);
  printf("start shifted by %7u B per workgroup, depth %2d: %7.3f ms  %6.1f GB/s per CU = %5.1f B/clk at 2.4 GHz\n", shift, DEPTH, ms, bps / 1e9, bps / 2.4e9);
}

// SHAPE sweep: the same bytes per workgroup-wide step through other load shapes: THREADS per workgroup (4 or 16 waves) and BYTES per lane
template <int THREADS, int BYTES, int DEPTH>
__global__ __launch_bounds__(THREADS, 1) void stream_shape(const float* buf, uint32_t bytes, int reps, float* out) {
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, bytes, 0x00020000);
  const uint32_t tid = threadIdx.x;
  float acc = 0;
  const uint32_t step = THREADS * BYTES;
  for (int r = 0; r < reps; ++r) {
    for (uint32_t off = tid * BYTES; off + DEPTH * step <= bytes; off += DEPTH * step) {
      if constexpr (BYTES == 16) {
        i32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + d * step, 0, 0);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += __builtin_bit_cast(float, v[d][0] ^ v[d][1] ^ v[d][2] ^ v[d][3]);
      } else if constexpr (BYTES == 8) {
        typedef int i32x2 __attribute__((ext_vector_type(2)));
        i32x2 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, off + d * step, 0, 0);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += __builtin_bit_cast(float, v[d][0] ^ v[d][1]);
      } else {
        int v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off + d * step, 0, 0);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += __builtin_bit_cast(float, v[d]);
      }
    }
  }
  out[blockIdx.x * THREADS + tid] = acc;
}
template <int THREADS, int BYTES, int DEPTH>
void run_shape(const float* buf, uint32_t bytes, float* out, int grid) {
  const int reps = 8;
  stream_shape<THREADS, BYTES, DEPTH><<<grid, THREADS>>>(buf, bytes, 2, out);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  stream_shape<THREADS, BYTES, DEPTH><<<grid, THREADS>>>(buf, bytes, reps, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bps = (double)bytes * reps / (ms * 1e-3);
  printf("%3d workgroups x %4d threads, %2d B per lane and load, %2d loads in flight per lane: %6.1f GB/s per CU = %5.1f B/clk at 2.4 GHz\n", grid, THREADS,
         BYTES, DEPTH, bps / 1e9, bps / 2.4e9);
}

// GRID sweep: the same stream with fewer workgroups (= fewer CUs streaming at once): is the ceiling the CU's intake or the L2's output?
template <int DEPTH>
void run_grid(const float* buf, uint32_t bytes, float* out, int grid) {
  const int reps = 8;
  stream<0, DEPTH><<<grid, 256>>>(buf, bytes, 2, out);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  stream<0, DEPTH><<<grid, 256>>>(buf, bytes, reps, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bps = (double)bytes * reps / (ms * 1e-3);
  printf("%3d workgroups (one per CU), depth %2d: %7.3f ms  %6.1f GB/s per CU = %5.1f B/clk at 2.4 GHz; all together %6.2f TB/s\n", grid, DEPTH, ms,
         bps / 1e9, bps / 2.4e9, bps * grid / 1e12);
}

int main() {
  const uint32_t bytes = 9600 * 1024;
  float *buf, *out;
  hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes); hipMalloc(&out, 256 * 1024 * 4);
  run<0, 8>(buf, bytes, out, "default");
  run<0, 16>(buf, bytes, out, "default");
  run<1, 8>(buf, bytes, out, "sc0");
  run<16, 8>(buf, bytes, out, "sc1");
  run<16, 16>(buf, bytes, out, "sc1");
  run<17, 8>(buf, bytes, out, "sc0 sc1");
  run<2, 8>(buf, bytes, out, "nt");
  run<18, 8>(buf, bytes, out, "sc1 nt");
  for (uint32_t shift : {0u, 4096u, 36864u, 299008u, 1200128u}) run_shift<16>(buf, bytes, out, shift);
  for (int grid : {1, 8, 16, 32, 64, 128, 192, 256}) run_grid<16>(buf, bytes, out, grid);
  for (int grid : {8, 256}) run_grid<32>(buf, bytes, out, grid);
  for (int grid : {8, 256}) {
    run_shape<256, 16, 32>(buf, bytes, out, grid);
    run_shape<1024, 16, 8>(buf, bytes, out, grid);
    run_shape<1024, 16, 16>(buf, bytes, out, grid);
    run_shape<512, 16, 16>(buf, bytes, out, grid);
    run_shape<256, 8, 32>(buf, bytes, out, grid);
    run_shape<1024, 8, 16>(buf, bytes, out, grid);
    run_shape<1024, 4, 16>(buf, bytes, out, grid);
  }
  return 0;
}
