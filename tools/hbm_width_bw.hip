// tools/hbm_width_bw.hip -- does the width of a lane's load limit a read-only stream on MI355X?
// The brick kernel reads TSpline3 quads with 16-B loads (1 KiB per wavefront instruction) and jagged TF1 rows with 8-B
// loads (512 B per instruction).  Each wavefront here streams a contiguous region of its own (like a wave's TF1 rows)
// with W-byte non-temporal loads, IN_FLIGHT of them issued back to back, 1024-lane workgroups, two per CU.
// Prints the rate per width; the same bytes in flight per lane are compared (16 x 8 B against 8 x 16 B).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
template <class T, int IN_FLIGHT>
__global__ __launch_bounds__(1024, 8) void k_stream(const T* __restrict__ p, size_t per_wave_elems, float* out) {
  const size_t wave = ((size_t)blockIdx.x * 1024 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const T* base = p + wave * per_wave_elems;
  float acc = 0.f;
  for (size_t i = 0; i < per_wave_elems; i += (size_t)64 * IN_FLIGHT) {
    T v[IN_FLIGHT];
#pragma unroll
    for (int k = 0; k < IN_FLIGHT; ++k) v[k] = __builtin_nontemporal_load(base + i + (size_t)k * 64 + lane);
#pragma unroll
    for (int k = 0; k < IN_FLIGHT; ++k) acc += ((const float*)&v[k])[0];
  }
  if (acc == 123.456f) out[0] = acc;
}
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <class T, int IN_FLIGHT>
static void run(const void* d, size_t bytes, float* out, const char* what) {
  const int blocks = 512 * 8;  // 8 rounds of the 512 resident workgroups
  const size_t waves = (size_t)blocks * 16;
  size_t per_wave = bytes / waves / sizeof(T);
  per_wave -= per_wave % ((size_t)64 * IN_FLIGHT);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k_stream<T, IN_FLIGHT>), dim3(blocks), dim3(1024), 0, 0, (const T*)d, per_wave, out);
  (void)hipEventRecord(e0, 0);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_stream<T, IN_FLIGHT>), dim3(blocks), dim3(1024), 0, 0, (const T*)d, per_wave, out);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double b = double(per_wave) * waves * sizeof(T);
  printf("%-34s %.3f ms  %.2f TB/s\n", what, ms, b / ms * 1e-9);
}
int main() {
  const size_t bytes = (size_t)4 << 30;
  void* d; float* out;
  if (hipMalloc(&d, bytes) != hipSuccess) return 1;
  (void)hipMalloc((void**)&out, 8);
  (void)hipMemset(d, 0, bytes);
  run<f32x4_t, 8>(d, bytes, out, "16 B per lane, 8 in flight");
  run<f32x2_t, 16>(d, bytes, out, "8 B per lane, 16 in flight");
  run<f32x2_t, 8>(d, bytes, out, "8 B per lane, 8 in flight");
  run<float, 16>(d, bytes, out, "4 B per lane, 16 in flight");
  run<f32x4_t, 5>(d, bytes, out, "16 B per lane, 5 in flight");
  run<f32x4_t, 8>(d, bytes, out, "16 B per lane, 8 in flight (again)");
  (void)hipDeviceSynchronize();
  return 0;
}
