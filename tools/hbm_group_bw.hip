// tools/hbm_group_bw.hip -- would a "tile group" layout stream faster than the per-tile store?  Same bytes as
// hbm_chunk_bw (128 slots x one 4 KiB plane of six per tile), but the planes of one (slot, segment) of ALL tiles lie
// next to each other: plane(j, k, t) = ((j * 6 + k) * n_tiles + t) * 4 KiB, so workgroups running side by side read
// adjacent 4 KiB pieces.  Build: make tools
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int U, bool GROUPED>
__global__ __launch_bounds__(256) void k(const f32x4_t* __restrict__ p, int n_slots, size_t n_tiles, int seg, float* out) {
  const size_t t = blockIdx.x;
  float acc = 0.f;
  for (int j = 0; j < n_slots; j += U) {
    f32x4_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t k6 = (size_t)((seg + j + u) % 2 + 2);
      const size_t plane = GROUPED ? ((size_t)(j + u) * 6 + k6) * n_tiles + t : (t * n_slots + (size_t)(j + u)) * 6 + k6;
      v[u] = __builtin_nontemporal_load(p + plane * 256 + threadIdx.x);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  if (acc == 123.456f) out[0] = acc;
}
int main() {
  const int n_slots = 128;
  const size_t n_tiles = 3906, bytes = n_tiles * n_slots * 6 * 4096;
  f32x4_t* d; float* out;
  if (hipMalloc(&d, bytes) != hipSuccess) return 1;
  (void)hipMalloc(&out, 4);
  (void)hipMemset(d, 1, bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int grouped = 0; grouped < 2; ++grouped)
    for (int rep = 0; rep < 2; ++rep) {
      auto launch = [&](int it) {
        if (grouped) hipLaunchKernelGGL((k<8, true>), dim3(n_tiles), dim3(256), 0, 0, d, n_slots, n_tiles, it, out);
        else hipLaunchKernelGGL((k<8, false>), dim3(n_tiles), dim3(256), 0, 0, d, n_slots, n_tiles, it, out);
      };
      for (int it = 0; it < 3; ++it) launch(it);
      (void)hipEventRecord(e0, 0);
      const int iters = 20;
      for (int it = 0; it < iters; ++it) launch(it);
      (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      printf("%s: %7.0f GB/s (%.1f us/launch)\n", grouped ? "planes of one (slot, segment) of all tiles adjacent" : "per-tile store (as shipped)            ",
             (double)n_tiles * n_slots * 4096 * iters / (ms * 1e-3) / 1e9, ms * 1e3 / iters);
    }
  return 0;
}
