// tools/sector_fetch.hip -- what does a partially read 128-B line cost on MI355X?
// Decides the alignment of the brick store's segment planes (mach3_amd/csrc/m3hip_layout.cpp): a step reads ONE plane of
// a run, so a plane that ends inside a line drags the rest of that line in if the memory side fetches whole lines.
// Each kernel touches the first `USED` bytes of every 128-B line of a 2 GiB buffer with 16-B non-temporal loads (lanes
// packed: consecutive lanes read consecutive used quads).  Prints time per pass and the rate of USED bytes; run under
// `rocprofv3 --pmc FETCH_SIZE --kernel-trace` as well and compare the counter (x2, MI355X_MICROARCH.md) with the used bytes.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int USED>  // bytes used per 128-B line: 16, 32, 64, 96 or 128
__global__ __launch_bounds__(256) void k_partial(const f32x4_t* __restrict__ p, size_t n_lines, float* out) {
  constexpr int Q = USED / 16;  // quads per line
  float acc = 0.f;
  const size_t n = n_lines * Q;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t line = i / Q, q = i % Q;
    const f32x4_t v = __builtin_nontemporal_load(p + line * 8 + q);
    acc += v.x;
  }
  if (acc == 123.456f) out[0] = acc;
}
template <int USED>
static void run(const f32x4_t* d, size_t n_lines, float* out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256 * 8;
  hipLaunchKernelGGL((k_partial<USED>), dim3(blocks), dim3(256), 0, 0, d, n_lines, out);
  (void)hipEventRecord(e0, 0);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_partial<USED>), dim3(blocks), dim3(256), 0, 0, d, n_lines, out);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double used = double(n_lines) * USED, lines = double(n_lines) * 128;
  printf("used %3d B of every 128-B line: %.3f ms  used-bytes rate %.2f TB/s  whole-lines rate %.2f TB/s\n", USED, ms,
         used / ms * 1e-9, lines / ms * 1e-9);
}
int main() {
  const size_t bytes = (size_t)2 << 30;
  void* d; float* out;
  if (hipMalloc(&d, bytes) != hipSuccess) return 1;
  (void)hipMalloc((void**)&out, 8);
  (void)hipMemset(d, 0, bytes);
  const size_t n_lines = bytes / 128;
  run<128>((const f32x4_t*)d, n_lines, out);
  run<96>((const f32x4_t*)d, n_lines, out);
  run<64>((const f32x4_t*)d, n_lines, out);
  run<32>((const f32x4_t*)d, n_lines, out);
  run<16>((const f32x4_t*)d, n_lines, out);
  (void)hipDeviceSynchronize();
  return 0;
}
