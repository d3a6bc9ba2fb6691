#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
struct Args { uint32_t* out; uint32_t n; uint32_t stride; };
template <int K>
__device__ __forceinline__ uint32_t body(const Args& a) {
  const uint32_t lane = threadIdx.x;
  const uint32_t b = blockIdx.x;
  a.out[b * 1024 + lane] = b * K + lane;
  const uint32_t next = b + a.stride;
  return next < a.n ? next : 0xFFFFFFFFu;
}
#define JUMP(next, NAME)                                                                     \
  if ((next) != 0xFFFFFFFFu) {                                                               \
    auto kp = __builtin_amdgcn_kernarg_segment_ptr();                                        \
    const uint32_t l_ = threadIdx.x;                                                         \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n"                                           \
                 "s_mov_b64 s[0:1], %0\n"                                                    \
                 "s_mov_b32 s2, %1\n"                                                        \
                 "v_mov_b32 v0, %2\n"                                                        \
                 "s_branch " NAME "\n" ::"s"(kp), "s"(next), "v"(l_)                         \
                 : "s0", "s1", "s2", "v0", "memory");                                        \
  }
extern "C" __global__ __launch_bounds__(1024, 8) void kern3(Args a) {
  const uint32_t next = body<3>(a);
  JUMP(next, "kern3");
}
int main() {
  uint32_t* d; const uint32_t n = 100;
  (void)hipMalloc(&d, n * 1024 * 4); (void)hipMemset(d, 0xff, n * 1024 * 4);
  Args a{d, n, 7};
  hipLaunchKernelGGL(kern3, dim3(7), dim3(1024), 0, 0, a);
  uint32_t* h = new uint32_t[n * 1024];
  (void)hipMemcpy(h, d, n * 1024 * 4, hipMemcpyDeviceToHost);
  int bad = 0; for (uint32_t i = 0; i < n * 1024; ++i) bad += h[i] != (i / 1024) * 3 + (i % 1024);
  printf("bad %d\n", bad); return bad != 0;
}
