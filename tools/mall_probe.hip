// tools/mall_probe.hip -- does the 256 MB Infinity Cache (MALL) keep a buffer that is re-read every "step" while a much larger
// stream passes through, and does a non-temporal stream leave it alone?  The brick kernel re-reads ~70 KB of metadata per brick
// every step (680 MB on config 3; the TF1 codes alone 176 MB) beside 4 GB of coefficients that are read once.
//   A: `small_mb` MB, read with plain loads (the candidate resident);  B: 6 GB, read with plain or non-temporal loads
// Prints the rate of reading A (i) right after reading A, (ii) after streaming B with plain loads, (iii) after streaming B with
// non-temporal loads.  Build: hipcc --offload-arch=gfx950 -O3 -o build/mall_probe tools/mall_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(1024, 8) void k_read(const f32x4_t* __restrict__ p, size_t n, float* out) {
  float acc = 0.f;
  const size_t stride = (size_t)gridDim.x * 1024;
  for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += 8 * stride) {
    f32x4_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t j = i + u * stride < n ? i + u * stride : n - 1;
      v[u] = NT ? __builtin_nontemporal_load(p + j) : p[j];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u].x + v[u].w;
  }
  if (acc == 123.456f) out[0] = acc;
}
static float timed(hipEvent_t e0, hipEvent_t e1, void (*fn)(void*), void* ctx) {
  (void)hipEventRecord(e0, 0);
  fn(ctx);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
struct Ctx { const f32x4_t* p; size_t n; float* out; bool nt; };
static void launch(void* c_) {
  Ctx* c = (Ctx*)c_;
  if (c->nt) hipLaunchKernelGGL(k_read<true>, dim3(512), dim3(1024), 0, 0, c->p, c->n, c->out);
  else hipLaunchKernelGGL(k_read<false>, dim3(512), dim3(1024), 0, 0, c->p, c->n, c->out);
}
int main(int argc, char** argv) {
  const size_t big = (size_t)6 << 30;
  f32x4_t *A, *B; float* out;
  if (hipMalloc(&A, (size_t)512 << 20) != hipSuccess || hipMalloc(&B, big) != hipSuccess) return 1;
  (void)hipMalloc(&out, 4);
  (void)hipMemset(A, 1, (size_t)512 << 20);
  (void)hipMemset(B, 1, big);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (size_t small_mb : {16, 64, 128, 192, 384}) {
    Ctx a{A, (small_mb << 20) / 16, out, false}, a_nt{A, (small_mb << 20) / 16, out, true}, b{B, big / 16, out, false}, b_nt{B, big / 16, out, true};
    double r[4] = {0, 0, 0, 0};
    const int reps = 5;
    for (int k = 0; k < reps; ++k) {
      launch(&a); timed(e0, e1, launch, &a);
      r[0] += (double)(small_mb << 20) / (timed(e0, e1, launch, &a) * 1e-3) / 1e12;      // A after A
      timed(e0, e1, launch, &b);
      r[1] += (double)(small_mb << 20) / (timed(e0, e1, launch, &a) * 1e-3) / 1e12;      // A after B (plain)
      launch(&a); launch(&a);
      const float tb = timed(e0, e1, launch, &b_nt);
      r[2] += (double)(small_mb << 20) / (timed(e0, e1, launch, &a) * 1e-3) / 1e12;      // A after B (non-temporal)
      r[3] += (double)big / (tb * 1e-3) / 1e12;
    }
    printf("A = %4zu MB: read A after A %.2f TB/s | after 6 GB of B, plain loads %.2f TB/s | after 6 GB of B, non-temporal loads %.2f TB/s   (B non-temporal streams at %.2f TB/s)\n",
           small_mb, r[0] / reps, r[1] / reps, r[2] / reps, r[3] / reps);
  }
  return 0;
}
