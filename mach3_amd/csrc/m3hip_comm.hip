// m3hip_comm.hip -- one sample over several GPUs (new; the reference is one device, Manager/gpuUtils.cu:71): the in-library RCCL
// communicator and the one-shot exchange through IPC-mapped fine-grained buffers.
#include "m3hip_ctx.h"

extern "C" {

/* ---------------------------------------------------------------- multi-GPU */

int m3hip_comm_available(void) {
  std::string err;
  if (!g_rccl.load(&err)) { g_create_err = err; return 0; }
  return 1;
}

int m3hip_comm_get_unique_id(void* out_id) try {
  if (!out_id) return M3HIP_EINVAL;
  std::string err;
  if (!g_rccl.load(&err)) return fail(nullptr, M3HIP_ECOMM, err);
  ncclUniqueId id;
  const ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) return fail(nullptr, M3HIP_ECOMM, "ncclGetUniqueId: " + g_rccl.why(r));
  memcpy(out_id, &id, sizeof(id));
  return M3HIP_OK;
} M3_API_CATCH(nullptr)

int m3hip_comm_init(m3hip_handle h, const void* unique_id, int rank, int n_ranks) try {
  if (!h || !unique_id) return M3HIP_EINVAL;
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(h, M3HIP_EINVAL, "comm_init: bad rank / n_ranks");
  std::string err;
  if (!g_rccl.load(&err)) return fail(h, M3HIP_ECOMM, err);
  HIPCHK(h, hipSetDevice(h->device));
  m3hip_comm_destroy(h);
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  ncclComm_t comm = nullptr;
  const ncclResult_t r = g_rccl.CommInitRank(&comm, n_ranks, id, rank);
  if (r != ncclSuccess) return fail(h, M3HIP_ECOMM, "ncclCommInitRank: " + g_rccl.why(r));
  h->comm = comm;
  h->rank = rank;
  h->n_ranks = n_ranks;
  { const char* f = getenv("M3HIP_COMM_VALIDATE"); if (f && f[0] == '1') h->comm_validate = true; }  // ORed into a setting made before the init
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_comm_set_validate(m3hip_handle h, int on) try {
  if (!h) return M3HIP_EINVAL;
  h->comm_validate = on != 0;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_comm_last_validation(m3hip_handle h, double* max_rel_diff, uint64_t* n_checked) try {
  if (!h) return M3HIP_EINVAL;
  if (max_rel_diff) *max_rel_diff = h->validate_max_rel;
  if (n_checked) *n_checked = h->validate_count;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_comm_destroy(m3hip_handle h) try {
  if (!h) return M3HIP_EINVAL;
  if (h->comm) {
    (void)hipStreamSynchronize(h->stream);
    g_rccl.CommDestroy((ncclComm_t)h->comm);
    h->comm = nullptr;
  }
  dev_free(h->d_gather); h->d_gather = nullptr;
  h->gather_cap = 0;
  h->rank = 0;
  h->n_ranks = 1;
  return M3HIP_OK;
} M3_API_CATCH(h)

/* ---------------------------------------------------------------- one-shot exchange */

int m3hip_xch_destroy(m3hip_handle h) try {
  if (!h) return M3HIP_EINVAL;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (int r = 0; r < int(h->xch_peer.size()); ++r)
    if (h->xch_peer[r] && !h->xch_peer_local[size_t(r)]) (void)hipIpcCloseMemHandle(h->xch_peer[r]);
  h->xch_peer.clear();
  h->xch_peer_local.clear();
  if (h->xch_own) {
    std::lock_guard<std::mutex> lk(g_xch_mu);
    for (auto it = g_xch_local.begin(); it != g_xch_local.end();)
      it = it->second.first == (void*)h->xch_own ? g_xch_local.erase(it) : std::next(it);
  }
  if (h->xch_own) (void)hipFree(h->xch_own);
  h->xch_own = nullptr;
  if (h->d_xch_targets) (void)hipFree(h->d_xch_targets);
  h->d_xch_targets = nullptr;
  if (h->h_xch_status) (void)hipHostFree(h->h_xch_status);
  h->h_xch_status = h->d_xch_status = nullptr;
  if (h->d_xch_pushed) (void)hipFree(h->d_xch_pushed);
  h->d_xch_pushed = nullptr;
  h->xch_active = false;
  h->xch_ranks = h->xch_rank = h->xch_slot = 0;
  h->xch_step = 0;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_xch_set_timeout(m3hip_handle h, double seconds) try {
  // the spin runs ON THE DEVICE: a time-out beyond a minute would outlast every host-side limit around it (the likelihood wait
  // gives up after 60 s, bench.py's watchdog after 180 s) and leave a kernel spinning behind a caller that has already left
  if (!h || !(seconds > 0.0) || seconds > 50.0) return fail(h, M3HIP_EINVAL, "xch_set_timeout: between 0 and 50 s (below the host-side 60 s wait)");
  h->xch_timeout_s = seconds;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_xch_create(m3hip_handle h, int rank, int n_ranks, void* handle_out) try {
  if (!h || !handle_out) return M3HIP_EINVAL;
  if (n_ranks < 1 || n_ranks > 64 || rank < 0 || rank >= n_ranks) return fail(h, M3HIP_EINVAL, "xch_create: bad rank / n_ranks");
  if (h->n_bins == 0) return fail(h, M3HIP_ESTATE, "xch_create before set_binning (the slots are sized by the bin space)");
  static_assert(sizeof(hipIpcMemHandle_t) <= M3HIP_XCH_HANDLE_BYTES, "IPC handle size");
  HIPCHK(h, hipSetDevice(h->device));
  m3hip_xch_destroy(h);
  h->xch_ranks = n_ranks;
  h->xch_rank = rank;
  if (const char* f = getenv("M3HIP_XCH_TIMEOUT_S")) {
    const double v = atof(f);
    if (v > 0.0 && v <= 50.0) h->xch_timeout_s = v;
  }
  h->xch_slot = 2 * h->n_bins + 2;
  const size_t bytes = sizeof(double) * 2 * size_t(n_ranks) * size_t(h->xch_slot);
  // fine-grained: a peer's stores must become visible INSIDE a running kernel (coarse-grained memory only promises
  // visibility at kernel boundaries)
  hipError_t e = hipExtMallocWithFlags((void**)&h->xch_own, bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) { h->xch_own = nullptr; return fail(h, M3HIP_EHIP, std::string("xch_create: fine-grained allocation: ") + hipGetErrorString(e)); }
  HIPCHK(h, hipMemset(h->xch_own, 0, bytes));  // tickets start at 0, steps at 1
  HIPCHK(h, hipDeviceSynchronize());
  hipIpcMemHandle_t ipc;
  e = hipIpcGetMemHandle(&ipc, h->xch_own);
  if (e != hipSuccess) {
    const std::string why = std::string("xch_create: hipIpcGetMemHandle: ") + hipGetErrorString(e);
    m3hip_xch_destroy(h);
    return fail(h, M3HIP_ECOMM, why);
  }
  memset(handle_out, 0, M3HIP_XCH_HANDLE_BYTES);
  memcpy(handle_out, &ipc, sizeof(ipc));
  {
    std::lock_guard<std::mutex> lk(g_xch_mu);
    g_xch_local[std::string(static_cast<const char*>(handle_out), M3HIP_XCH_HANDLE_BYTES)] = {h->xch_own, h->device};
  }
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_xch_open(m3hip_handle h, const void* handles) try {
  if (!h || !handles) return M3HIP_EINVAL;
  if (!h->xch_own) return fail(h, M3HIP_ESTATE, "xch_open before xch_create");
  HIPCHK(h, hipSetDevice(h->device));
  const int N = h->xch_ranks;
  h->xch_peer.assign(size_t(N), nullptr);
  h->xch_peer_local.assign(size_t(N), 0);
  for (int r = 0; r < N; ++r) {
    if (r == h->xch_rank) { h->xch_peer[r] = h->xch_own; h->xch_peer_local[r] = 1; continue; }
    const char* hb = static_cast<const char*>(handles) + size_t(r) * M3HIP_XCH_HANDLE_BYTES;
    void* local = nullptr;
    int local_dev = -1;
    {  // a peer in this process: its buffer as it is
      std::lock_guard<std::mutex> lk(g_xch_mu);
      auto it = g_xch_local.find(std::string(hb, M3HIP_XCH_HANDLE_BYTES));
      if (it != g_xch_local.end()) { local = it->second.first; local_dev = it->second.second; }
    }
    if (local) {
      if (local_dev != h->device) {
        // ... on ANOTHER device of this process: this device's kernels store into it, so it must be mapped here.  The IPC route
        // gets that from hipIpcMemLazyEnablePeerAccess; a raw pointer needs peer access enabled by hand, or the first push faults
        int can = 0;
        const hipError_t ec = hipDeviceCanAccessPeer(&can, h->device, local_dev);
        if (ec != hipSuccess || !can) {
          const std::string why = "xch_open: device " + std::to_string(h->device) + " has no peer access to device " + std::to_string(local_dev) + " (rank " + std::to_string(r) + ")";
          m3hip_xch_destroy(h);
          return fail(h, M3HIP_ECOMM, why);
        }
        const hipError_t ep = hipDeviceEnablePeerAccess(local_dev, 0);  // (current device = h->device)
        if (ep != hipSuccess && ep != hipErrorPeerAccessAlreadyEnabled) {
          const std::string why = std::string("xch_open: hipDeviceEnablePeerAccess: ") + hipGetErrorString(ep);
          m3hip_xch_destroy(h);
          return fail(h, M3HIP_ECOMM, why);
        }
        (void)hipGetLastError();
      }
      h->xch_peer[r] = local;
      h->xch_peer_local[r] = 1;
      continue;
    }
    hipIpcMemHandle_t ipc;
    memcpy(&ipc, hb, sizeof(ipc));
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, ipc, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      const std::string why = "xch_open: hipIpcOpenMemHandle of rank " + std::to_string(r) + ": " + hipGetErrorString(e);
      m3hip_xch_destroy(h);
      return fail(h, M3HIP_ECOMM, why);
    }
    h->xch_peer[r] = p;
  }
  // slot [parity][this rank] of every rank's buffer
  std::vector<double*> tg(size_t(2) * N);
  for (int par = 0; par < 2; ++par)
    for (int r = 0; r < N; ++r)
      tg[size_t(par) * N + r] = static_cast<double*>(h->xch_peer[r]) + (size_t(par) * N + size_t(h->xch_rank)) * size_t(h->xch_slot);
  HIPCHK(h, hipMalloc((void**)&h->d_xch_targets, sizeof(double*) * tg.size()));
  HIPCHK(h, hipMemcpy(h->d_xch_targets, tg.data(), sizeof(double*) * tg.size(), hipMemcpyHostToDevice));
  HIPCHK(h, hipHostMalloc((void**)&h->h_xch_status, sizeof(uint64_t), hipHostMallocMapped));
  *h->h_xch_status = 0;
  HIPCHK(h, hipHostGetDevicePointer((void**)&h->d_xch_status, h->h_xch_status, 0));
  HIPCHK(h, hipMalloc((void**)&h->d_xch_pushed, sizeof(unsigned long long)));
  HIPCHK(h, hipMemset(h->d_xch_pushed, 0, sizeof(unsigned long long)));
  h->xch_step = 0;
  h->xch_active = true;
  return M3HIP_OK;
} M3_API_CATCH(h)

}  // extern "C"
