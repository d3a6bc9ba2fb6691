// tools/step_floor.hip -- what does the LAUNCH SCHEME of a small sample's step cost on this runtime?
//
// The engine's step is three launches on one stream (k_prep: one workgroup, parameters in over PCIe, per-step tables, histogram
// reset; the reweight kernel: tiles x 256 lanes; k_llh: one workgroup, bin likelihood, result + ticket out over PCIe) and a
// host spin on the ticket.  On BASELINE configs[0] (1e5 events x 32 splines: an 11 us reweight kernel) the step takes 32 us.
// This program times stand-in kernels of exactly those shapes (391 workgroups streaming 51 MB, 32 flush replicas of 200 bins,
// parameters and results through mapped host memory) under four schemes:
//   1  three launches + host spin                         (what the engine does)
//   2  the same three kernels as ONE hipGraphLaunch       (arguments fixed; the step number travels in a device word)
//   3  two launches: the prep work folded into every workgroup's prologue of the reweight kernel (table in LDS)
//   4  one launch: as 3, and the workgroup that finishes LAST (atomic counter behind a device-scope fence) runs the likelihood
//   5  the three launches of step s + 1 ENQUEUED WHILE STEP s RUNS: its prep kernel waits (one lane polls a doorbell word in mapped
//      host memory, with a time-out) until the host has written the step's parameters -- launch latency leaves the critical path
//   6  3 and 5 together: two launches enqueued ahead, every workgroup of the reweight kernel waits for the doorbell itself
//   7  as 4 with the hand-off protocol of MI355X_MICROARCH.md (barrier-counter / fanin rows): every wave waits for its own flush adds
//      (s_waitcnt vmcnt(0)), workgroup barrier, then LANE 0 ALONE issues the agent-scope release fence and ONE relaxed returning add;
//      the workgroup whose add came last acquires once and runs the likelihood.  (4 fences from all 256 lanes of every workgroup.)
//   8  two launches: k_prep as in 1, and the reweight kernel ends in 7's last-arriver likelihood -- the one launch it saves, alone
// Results are checked against each other (the same sum every scheme).  Build: make tools; run: build/step_floor [steps=3000]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                   \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } \
  } while (0)

constexpr int kPar = 48, kBins = 200, kRep = 32, kTiles = 391, kLanes = 256;
constexpr size_t kQuadsPerLane = 32;  // 391 x 256 x 32 x 16 B = 51 MB per step
typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct Args {
  const double* params_host;  // mapped pinned: [kPar] + the step number behind them
  float* table;               // [kPar] per-step table (device)
  double* hist;               // [kRep][kBins]
  const f32x4_t* store;       // the tiles' coefficient planes
  const double* data;         // [kBins]
  double* out_host;           // mapped pinned: [0] = likelihood, [1] = ticket (as a double's bits)
  unsigned long long* step;   // device word: the step number (graph scheme), bumped by the prep kernel
  unsigned int* arrived;      // device word: workgroups through (one-launch scheme)
  unsigned long long ticket;  // direct schemes: this step's number
  const unsigned long long* door;  // mapped pinned: the doorbell of scheme 5 (nullptr: no waiting)
};

__device__ void prep_table(const Args& a, float* table, int tid, int nt) {
  for (int i = tid; i < kPar; i += nt) {
    const double v = a.params_host[i];  // over PCIe
    // a stand-in for the segment search: a few dependent steps
    int seg = 0, hi = 6;
    while (hi - seg > 1) {
      const int mid = (seg + hi) / 2;
      if (v > -3.0 + mid) seg = mid; else hi = mid;
    }
    table[i] = (float)(v - (-3.0 + seg));
  }
}

__global__ __launch_bounds__(kLanes) void k_prep(Args a) {
  if (a.door) {  // scheme 5: enqueued ahead of its parameters
    if (threadIdx.x == 0) {
      const unsigned long long t0 = wall_clock64();  // 100 MHz
      while (__hip_atomic_load(a.door, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < a.ticket && wall_clock64() - t0 < 200000000ull) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
  }
  for (int b = threadIdx.x; b < kRep * kBins; b += kLanes) a.hist[b] = 0.0;
  prep_table(a, a.table, threadIdx.x, kLanes);
  if (a.step && threadIdx.x == 0) *a.step += 1;
}

__global__ __launch_bounds__(kLanes) void k_prep_table_only(Args a) { prep_table(a, a.table, threadIdx.x, kLanes); }

template <bool OWN_PREP, int LAST_LLH>
__global__ __launch_bounds__(kLanes) void k_reweight(Args a);

__device__ double bin_llh(double d, double m) { return d > 0 && m > 0 ? m - d + d * log(d / m) : m; }

// the likelihood of one workgroup: fold the replicas, per-bin statistic, tree; optionally reset the replicas for the next step
template <bool RESET>
__device__ void llh_block(const Args& a, unsigned long long ticket) {
  __shared__ double part[kLanes / 64];
  double s = 0.0;
  for (int b = threadIdx.x; b < kBins; b += kLanes) {
    double m = 0.0;
    for (int r = 0; r < kRep; ++r) {
      m += a.hist[r * kBins + b];
      if (RESET) a.hist[r * kBins + b] = 0.0;
    }
    s += bin_llh(a.data[b], m);
  }
  for (int off = 32; off; off >>= 1) s += __shfl_down(s, off);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < kLanes / 64; ++w) t += part[w];
    a.out_host[0] = t;
    __threadfence_system();
    ((volatile unsigned long long*)a.out_host)[1] = ticket;
  }
}

template <bool RESET>
__global__ __launch_bounds__(kLanes) void k_llh(Args a) { llh_block<RESET>(a, a.step ? *a.step : a.ticket); }

template <bool OWN_PREP, int LAST_LLH>
__global__ __launch_bounds__(kLanes) void k_reweight(Args a) {
  __shared__ float s_table[kPar];
  __shared__ double s_hist[kBins];
  __shared__ bool s_last;
  if (OWN_PREP && a.door) {  // scheme 6: every workgroup waits for the doorbell itself
    if (threadIdx.x == 0) {
      const unsigned long long t0 = wall_clock64();
      while (__hip_atomic_load(a.door, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < a.ticket && wall_clock64() - t0 < 200000000ull) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
  }
  for (int b = threadIdx.x; b < kBins; b += kLanes) s_hist[b] = 0.0;
  if (OWN_PREP) prep_table(a, s_table, threadIdx.x, kLanes);  // every workgroup fetches the parameters itself (PCIe) and builds its table
  else
    for (int i = threadIdx.x; i < kPar; i += kLanes) s_table[i] = a.table[i];
  __syncthreads();
  const f32x4_t* p = a.store + ((size_t)blockIdx.x * kQuadsPerLane) * kLanes + threadIdx.x;
  float w = 1.0f;
#pragma unroll 8
  for (size_t j = 0; j < kQuadsPerLane; ++j) {
    const f32x4_t c = __builtin_nontemporal_load(p + j * kLanes);
    const float dx = s_table[j % kPar];
    w *= fmaf(dx, fmaf(dx, fmaf(dx, c.w, c.z), c.y), c.x);
  }
  atomicAdd(&s_hist[(blockIdx.x * kLanes + threadIdx.x) % kBins], (double)w);
  __syncthreads();
  const int rep = blockIdx.x % kRep;
  for (int b = threadIdx.x; b < kBins; b += kLanes)
    if (s_hist[b] != 0.0) atomicAdd(&a.hist[rep * kBins + b], s_hist[b]);
  if (LAST_LLH == 2) {  // the guide's protocol
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's flush adds have completed ...
    __syncthreads();                                    // ... and so have every other wave's of this workgroup
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the compiler may drop the wait behind the write-back: MI355X_MICROARCH.md, compiler hazard)
      s_last = __hip_atomic_fetch_add(a.arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    __syncthreads();
    if (s_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      if (threadIdx.x == 0) __hip_atomic_store(a.arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      llh_block<true>(a, a.ticket);
    }
  } else if (LAST_LLH) {
    __threadfence();  // this workgroup's adds are visible device-wide before it announces itself
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(a.arrived, 1u) == gridDim.x - 1;
    __syncthreads();
    if (s_last) {
      __threadfence();
      if (threadIdx.x == 0) *a.arrived = 0u;
      llh_block<true>(a, a.ticket);  // ... and leaves the replicas empty for the next step
    }
  }
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const int steps = argc > 1 ? std::atoi(argv[1]) : 3000;
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  Args a{};
  double *h_par, *h_out;
  CK(hipHostMalloc((void**)&h_par, sizeof(double) * (kPar + 1), hipHostMallocMapped));
  CK(hipHostMalloc((void**)&h_out, sizeof(double) * 2, hipHostMallocMapped));
  CK(hipHostGetDevicePointer((void**)&a.params_host, h_par, 0));
  CK(hipHostGetDevicePointer((void**)&a.out_host, h_out, 0));
  CK(hipMalloc((void**)&a.table, sizeof(float) * kPar));
  CK(hipMalloc((void**)&a.hist, sizeof(double) * kRep * kBins));
  CK(hipMemset(a.hist, 0, sizeof(double) * kRep * kBins));
  const size_t n_quads = (size_t)kTiles * kLanes * kQuadsPerLane;
  f32x4_t* store;
  CK(hipMalloc((void**)&store, n_quads * 16));
  {
    std::vector<float> h(n_quads * 4);
    for (size_t i = 0; i < n_quads; ++i) { h[4 * i] = 1.0f; h[4 * i + 1] = 1e-3f * float(i % 7); h[4 * i + 2] = 1e-4f; h[4 * i + 3] = 1e-5f; }
    CK(hipMemcpy(store, h.data(), n_quads * 16, hipMemcpyHostToDevice));
  }
  a.store = store;
  double* d_data;
  CK(hipMalloc((void**)&d_data, sizeof(double) * kBins));
  {
    std::vector<double> d(kBins, 500.0);
    CK(hipMemcpy(d_data, d.data(), sizeof(double) * kBins, hipMemcpyHostToDevice));
  }
  a.data = d_data;
  unsigned long long* d_step;
  unsigned int* d_arr;
  CK(hipMalloc((void**)&d_step, 8));
  CK(hipMalloc((void**)&d_arr, 4));
  CK(hipMemset(d_step, 0, 8));
  CK(hipMemset(d_arr, 0, 4));
  unsigned long long* h_door;
  CK(hipHostMalloc((void**)&h_door, 64, hipHostMallocMapped));
  *h_door = 0;
  const unsigned long long* d_door;
  CK(hipHostGetDevicePointer((void**)&d_door, h_door, 0));
  volatile unsigned long long* flag = (volatile unsigned long long*)h_out + 1;
  *flag = 0;
  unsigned long long ticket = 0;
  auto set_params = [&](unsigned long long s) {
    for (int i = 0; i < kPar; ++i) h_par[i] = 0.3 * std::sin(0.37 * double(i) + 0.01 * double(s % 64));
  };
  auto wait = [&](unsigned long long t) {
    unsigned long long spins = 0;
    (void)hipStreamQuery(st);
    while (*flag != t)
      if ((++spins & 0xFF) == 0) (void)hipStreamQuery(st);
    (void)hipStreamQuery(st);
  };
  // ---- scheme 2's graph: captured once, arguments fixed (the step number is the device word k_prep bumps)
  Args ag = a;
  ag.step = d_step;
  hipGraph_t graph;
  hipGraphExec_t gexec;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(k_prep, dim3(1), dim3(kLanes), 0, st, ag);
  hipLaunchKernelGGL((k_reweight<false, 0>), dim3(kTiles), dim3(kLanes), 0, st, ag);
  hipLaunchKernelGGL(k_llh<false>, dim3(1), dim3(kLanes), 0, st, ag);
  CK(hipStreamEndCapture(st, &graph));
  CK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));

  double ref[64];
  bool have_ref = false;
  unsigned long long graph_steps = 0;
  auto run = [&](int scheme, const char* name) {
    CK(hipStreamSynchronize(st));
    CK(hipMemset(a.hist, 0, sizeof(double) * kRep * kBins));
    unsigned long long bad = 0;
    double t0 = 0;
    const int warm = 300;
    auto enqueue_armed = [&](unsigned long long t) {
      Args x = a;
      x.ticket = t;
      x.door = d_door;
      if (scheme == 5) {
        hipLaunchKernelGGL(k_prep, dim3(1), dim3(kLanes), 0, st, x);
        hipLaunchKernelGGL((k_reweight<false, 0>), dim3(kTiles), dim3(kLanes), 0, st, x);
        hipLaunchKernelGGL(k_llh<false>, dim3(1), dim3(kLanes), 0, st, x);
      } else {
        hipLaunchKernelGGL((k_reweight<true, 0>), dim3(kTiles), dim3(kLanes), 0, st, x);
        hipLaunchKernelGGL(k_llh<true>, dim3(1), dim3(kLanes), 0, st, x);
      }
    };
    if (scheme == 5 || scheme == 6) enqueue_armed(ticket + 1 + (1ull << 40));
    for (int s = 0; s < warm + steps; ++s) {
      if (s == warm) t0 = now_s();
      set_params((unsigned long long)s);
      Args x = a;
      unsigned long long want;
      if (scheme == 5 || scheme == 6) {
        want = ++ticket + (1ull << 40);
        __atomic_store_n(h_door, want, __ATOMIC_RELEASE);  // the parameters are in place: go
        enqueue_armed(want + 1);                           // the next step's launches, behind this step's kernels
      } else if (scheme == 2) {
        want = ++graph_steps;
        CK(hipGraphLaunch(gexec, st));
      } else {
        want = x.ticket = ++ticket + (1ull << 40);
        if (scheme == 1) {
          hipLaunchKernelGGL(k_prep, dim3(1), dim3(kLanes), 0, st, x);
          hipLaunchKernelGGL((k_reweight<false, 0>), dim3(kTiles), dim3(kLanes), 0, st, x);
          hipLaunchKernelGGL(k_llh<false>, dim3(1), dim3(kLanes), 0, st, x);
        } else if (scheme == 3) {
          hipLaunchKernelGGL((k_reweight<true, 0>), dim3(kTiles), dim3(kLanes), 0, st, x);
          hipLaunchKernelGGL(k_llh<true>, dim3(1), dim3(kLanes), 0, st, x);  // (no k_prep to empty the replicas: the likelihood kernel leaves them empty)
        } else if (scheme == 4) {
          x.arrived = d_arr;
          hipLaunchKernelGGL((k_reweight<true, 1>), dim3(kTiles), dim3(kLanes), 0, st, x);
        } else if (scheme == 7) {
          x.arrived = d_arr;
          hipLaunchKernelGGL((k_reweight<true, 2>), dim3(kTiles), dim3(kLanes), 0, st, x);
        } else {  // 8
          x.arrived = d_arr;
          hipLaunchKernelGGL(k_prep_table_only, dim3(1), dim3(kLanes), 0, st, x);  // (the last arriver leaves the replicas empty)
          hipLaunchKernelGGL((k_reweight<false, 2>), dim3(kTiles), dim3(kLanes), 0, st, x);
        }
      }
      wait(want);
      const double v = h_out[0];
      if (!have_ref) ref[s % 64] = v;
      else if (s >= 64 && std::fabs(v - ref[s % 64]) > 1e-9 * std::fabs(ref[s % 64])) ++bad;
      if (!have_ref && s == 127) have_ref = true;
    }
    const double el = now_s() - t0;
    if (scheme == 5 || scheme == 6) {  // the step that was enqueued last never gets parameters of its own: let it run on the last ones
      __atomic_store_n(h_door, ++ticket + (1ull << 40), __ATOMIC_RELEASE);
      wait(ticket + (1ull << 40));
    }
    std::printf("scheme %d  %-58s %7.2f us/step%s\n", scheme, name, 1e6 * el / steps, bad ? "   (RESULTS DIFFER)" : "");
    std::fflush(stdout);
  };
  // device time of the middle kernel alone, for scale
  {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_reweight<false, 0>), dim3(kTiles), dim3(kLanes), 0, st, a);
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((k_reweight<false, 0>), dim3(kTiles), dim3(kLanes), 0, st, a);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("the stand-in reweight kernel back to back: %.2f us per launch (%.0f GB/s)\n", 1e3 * ms / 200, double(n_quads) * 16 / (ms / 200 * 1e-3) / 1e9);
  }
  for (int rep = 0; rep < 2; ++rep) {
    run(1, "three launches + host spin (the engine's step)");
    run(2, "the three kernels as one hipGraphLaunch");
    run(3, "two launches: prep inside the reweight kernel's prologue");
    run(4, "one launch: ... and the likelihood by the last workgroup");
    run(5, "three launches enqueued a step ahead, prep waits for a doorbell");
    run(6, "two launches enqueued a step ahead, every workgroup waits");
    run(7, "one launch, likelihood by the last workgroup, lane-0 release");
    run(8, "two launches: prep, reweight + last-workgroup likelihood");
  }
  return 0;
}
