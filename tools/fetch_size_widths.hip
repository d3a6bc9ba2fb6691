// tools/fetch_size_widths.hip -- calibration of rocprofv3's FETCH_SIZE for load widths other than 16 B per lane
// (MI355X_MICROARCH.md calibrates only the 16-B case: x2).  Each kernel streams the same 1 GiB once with lane-contiguous
// loads of 2, 4, 8 or 16 bytes; run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and compare the counter with 2^30.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
template <class T>
__global__ __launch_bounds__(256) void k_stream(const T* __restrict__ p, size_t n, unsigned long long* out) {
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const T v = __builtin_nontemporal_load(p + i);
    const unsigned char* b = reinterpret_cast<const unsigned char*>(&v);
    acc += b[0];
  }
  if (acc == 0x123456789abcdefull) out[0] = acc;
}
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
int main() {
  const size_t bytes = (size_t)1 << 30;
  void* d; unsigned long long* out;
  if (hipMalloc(&d, bytes) != hipSuccess) return 1;
  (void)hipMalloc(&out, 8);
  (void)hipMemset(d, 1, bytes);
  const int blocks = 4096;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((k_stream<uint16_t>), dim3(blocks), dim3(256), 0, 0, (const uint16_t*)d, bytes / 2, out);
    hipLaunchKernelGGL((k_stream<uint32_t>), dim3(blocks), dim3(256), 0, 0, (const uint32_t*)d, bytes / 4, out);
    hipLaunchKernelGGL((k_stream<f32x2_t>), dim3(blocks), dim3(256), 0, 0, (const f32x2_t*)d, bytes / 8, out);
    hipLaunchKernelGGL((k_stream<f32x4_t>), dim3(blocks), dim3(256), 0, 0, (const f32x4_t*)d, bytes / 16, out);
  }
  (void)hipDeviceSynchronize();
  printf("done: each kernel read %zu bytes\n", bytes);
  return 0;
}
