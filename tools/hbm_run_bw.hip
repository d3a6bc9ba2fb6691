// tools/hbm_run_bw.hip -- read bandwidth for the brick store's access shape: responses are consecutive 16-byte
// quads inside a RUN of `run_bytes` (one parameter's events in a brick), and a step reads one of six segment
// planes of every run, i.e. run_bytes contiguous, then a jump of 6 x run_bytes.  A pure stream (no LDS, no
// barriers, 8 loads in flight per lane, 8 waves per SIMD): the ceiling this granularity allows, to hold the brick
// kernel's measured traffic rate against.  Build: hipcc --offload-arch=gfx950 -O3 -o build/hbm_run_bw tools/hbm_run_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int U>
__global__ __launch_bounds__(1024, 8) void k_runs(const f32x4_t* __restrict__ p, size_t quads_per_run, size_t n_resp_per_block, int seg, float* out) {
  const size_t r0 = (size_t)blockIdx.x * n_resp_per_block + threadIdx.x;
  float acc = 0.f;
  for (size_t r = r0; r < (size_t)(blockIdx.x + 1) * n_resp_per_block; r += (size_t)U * 1024) {
    f32x4_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // (a batch may reach past the workgroup's share when that is not a multiple of U x 1024: clamped to the share's end)
      const size_t end = (size_t)(blockIdx.x + 1) * n_resp_per_block - 1;
      const size_t rj = r + (size_t)u * 1024, ri = rj < end ? rj : end, run = ri / quads_per_run, q = ri % quads_per_run;
      v[u] = __builtin_nontemporal_load(p + (run * 6 + (size_t)((seg + run) % 3 + 2)) * quads_per_run + q);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
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
  for (size_t run_bytes : {128, 256, 512, 1024, 2048, 4096}) {
    const size_t qpr = run_bytes / 16, n_resp = 32768;              // 32 768 responses per workgroup, like a config-3 brick
    const int blocks = (int)(bytes / 16 / 6 / n_resp);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k_runs<8>), dim3(blocks), dim3(1024), 0, 0, d, qpr, n_resp, it, out);
    (void)hipEventRecord(e0, 0);
    const int iters = 10;
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL((k_runs<8>), dim3(blocks), dim3(1024), 0, 0, d, qpr, n_resp, it, out);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double read = (double)blocks * n_resp * 16 * iters;
    printf("run %5zu B, %5d blocks: %7.0f GB/s (%.1f us/launch)\n", run_bytes, blocks, read / (ms * 1e-3) / 1e9, ms * 1e3 / iters);
  }
  // second table: the same stream at a run of 1 KiB, for workgroups that live as long as a brick (16 384 responses = 37 us)
  // or shorter / longer, and with 5 loads in flight per lane (the brick kernel's depth) instead of 8
  for (size_t n_resp : {10240, 15360, 30720, 61440})  // multiples of 5 x 1024 (and close to multiples of 8 x 1024: the last batch is clamped)
    for (int U : {5, 8}) {
      const size_t qpr = 64;
      const int blocks = (int)(bytes / 16 / 6 / n_resp);
      auto launch = [&](int it) {
        if (U == 5) hipLaunchKernelGGL((k_runs<5>), dim3(blocks), dim3(1024), 0, 0, d, qpr, n_resp, it, out);
        else hipLaunchKernelGGL((k_runs<8>), dim3(blocks), dim3(1024), 0, 0, d, qpr, n_resp, it, out);
      };
      for (int it = 0; it < 2; ++it) launch(it);
      (void)hipEventRecord(e0, 0);
      const int iters = 10;
      for (int it = 0; it < iters; ++it) launch(it);
      (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      const double read = (double)blocks * n_resp * 16 * iters;
      printf("run 1024 B, %6zu responses per workgroup, %d in flight, %5d blocks: %7.0f GB/s (%.1f us/launch)\n", n_resp, U, blocks, read / (ms * 1e-3) / 1e9, ms * 1e3 / iters);
    }
  return 0;
}
