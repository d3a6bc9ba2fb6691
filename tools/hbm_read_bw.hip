// tools/hbm_read_bw.hip -- read-only HBM streaming ceiling on this box (calibration for the
// roofline denominator discussion in DESIGN.md).  Reads a buffer once per launch with
// 16 B/lane loads, 8 or 16 in flight per lane, plain or non-temporal.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <bool NT, int U>
__global__ __launch_bounds__(256) void k_read(const f32x4_t* __restrict__ p, size_t n_per_block, float* out) {
  const f32x4_t* b = p + (size_t)blockIdx.x * n_per_block + threadIdx.x;
  float acc = 0.f;
  for (size_t i = 0; i < n_per_block; i += 256 * U) {
    f32x4_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(b + i + u * 256) : b[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  if (acc == 123.456f) out[0] = acc;
}
template <bool NT, int U>
double run(const f32x4_t* d, size_t bytes, int blocks, float* out) {
  size_t n_per_block = bytes / 16 / blocks;
  n_per_block -= n_per_block % (256 * U);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_read<NT, U>), dim3(blocks), dim3(256), 0, 0, d, n_per_block, out);
  (void)hipEventRecord(e0, 0);
  const int it = 10;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL((k_read<NT, U>), dim3(blocks), dim3(256), 0, 0, d, n_per_block, out);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return (double)n_per_block * 16 * blocks * it / (ms * 1e-3) / 1e9;
}
int main() {
  const size_t bytes = (size_t)4 << 30;
  f32x4_t* d; float* out;
  if (hipMalloc(&d, bytes) != hipSuccess) return 1;
  (void)hipMalloc(&out, 4);
  (void)hipMemset(d, 1, bytes);
  for (int blocks : {512, 1024, 2048, 4096, 8192}) {
    printf("blocks %5d: plain u8 %7.0f  nt u8 %7.0f  plain u16 %7.0f  nt u16 %7.0f GB/s\n", blocks, run<false, 8>(d, bytes, blocks, out),
           run<true, 8>(d, bytes, blocks, out), run<false, 16>(d, bytes, blocks, out), run<true, 16>(d, bytes, blocks, out));
  }
  return 0;
}
