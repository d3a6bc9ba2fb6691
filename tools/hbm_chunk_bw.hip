// tools/hbm_chunk_bw.hip -- read bandwidth when every workgroup walks CHUNK-byte pieces at a
// stride of 6 x CHUNK (the access shape of the segment-major tile store: one active plane out
// of six per slot).  Calibrates how much of the linear-stream ceiling the layout can reach.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4_t __attribute__((ext_vector_type(4)));
// one workgroup = 256 lanes; a "tile" = n_slots chunks of LANES_PER_CHUNK*16 bytes, chunk stride 6x
template <int U>
__global__ __launch_bounds__(256) void k_chunk(const f32x4_t* __restrict__ p, int n_slots, size_t chunk_f4, size_t tile_f4,
                                               int seg, float* out) {
  // chunk_f4 = float4 per chunk (multiple of 256); lanes cover a chunk in chunk_f4/256 passes
  const f32x4_t* base = p + (size_t)blockIdx.x * tile_f4 + threadIdx.x;
  float acc = 0.f;
  const int passes = (int)(chunk_f4 / 256);
  for (int ps = 0; ps < passes; ++ps) {
    for (int j = 0; j < n_slots; j += U) {
      f32x4_t v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(base + ((size_t)(j + u) * 6 + ((seg + j + u) % 2 + 2)) * chunk_f4 + ps * 256);
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
  }
  if (acc == 123.456f) out[0] = acc;
}
int main() {
  const size_t bytes = (size_t)12 << 30;
  f32x4_t* d; float* out;
  if (hipMalloc(&d, bytes) != hipSuccess) return 1;
  (void)hipMalloc(&out, 4);
  (void)hipMemset(d, 1, bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int n_slots = 128;
  for (size_t chunk : {1024, 4096, 16384, 65536}) {
    const size_t chunk_f4 = chunk / 16;  // for 1 KB chunks only 64 lanes useful: handled by using 256-lane = 4 KB min
    if (chunk_f4 < 256) continue;
    const size_t tile_f4 = (size_t)n_slots * 6 * chunk_f4;
    const int blocks = (int)(bytes / 16 / tile_f4);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k_chunk<8>), dim3(blocks), dim3(256), 0, 0, d, n_slots, chunk_f4, tile_f4, it, out);
    (void)hipEventRecord(e0, 0);
    const int iters = 10;
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL((k_chunk<8>), dim3(blocks), dim3(256), 0, 0, d, n_slots, chunk_f4, tile_f4, it, out);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double read = (double)blocks * n_slots * chunk * iters;
    printf("chunk %6zu B, %5d blocks: %7.0f GB/s (%.1f us/launch)\n", chunk, blocks, read / (ms * 1e-3) / 1e9, ms * 1e3 / iters);
  }
  return 0;
}
