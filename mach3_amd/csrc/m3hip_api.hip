// m3hip_api.hip -- libm3hip's core: handle lifetime, pinned memory, introspection, timing, test hooks, and the helpers the
// other translation units share (m3hip_ctx.h has the map).  Implements the part of include/m3hip.h named there.
#include "m3hip_ctx.h"

Rccl g_rccl;
std::string g_create_err;
std::mutex g_xch_mu;
std::map<std::string, std::pair<void*, int>> g_xch_local;

int fail(m3hip_handle h, int code, const std::string& msg) {
  if (h) h->err = msg;
  else g_create_err = msg;
  return code;
}

int api_exception(m3hip_handle h) noexcept {
  try {
    try {
      throw;
    } catch (const std::bad_alloc&) {
      return fail(h, M3HIP_ENOMEM, "host allocation failed (std::bad_alloc)");
    } catch (const std::exception& e) {
      return fail(h, M3HIP_EINVAL, std::string("unexpected exception: ") + e.what());
    } catch (...) {
      return fail(h, M3HIP_EINVAL, "unexpected exception");
    }
  } catch (...) {  // the message itself could not be stored
    return M3HIP_ENOMEM;
  }
}

void dev_free(void* p) {
  if (p) (void)hipFree(p);
}

// Low-latency wait for the stream to drain.  hipStreamSynchronize can fall into a timed sleep
// (observed: ~1 ms wake-up granularity once hipEvents are in flight on the stream), which is
// several step times at this path's ~0.4 ms/step; polling hipStreamQuery keeps the wait at
// the kernel's own duration.  After 5 s of polling it degrades to the blocking call.
hipError_t wait_stream(hipStream_t s) {
  const double t0 = now_s();
  for (;;) {
    hipError_t e = hipStreamQuery(s);
    if (e != hipErrorNotReady) return e;
    if (now_s() - t0 > 5.0) return hipStreamSynchronize(s);
  }
}

void collect_timing(m3hip_handle h) {
  if (h->pending >= 0) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) {
      h->acc_ms[h->pending] += ms;
      h->cnt[h->pending]++;
    }
    if (h->gap_pending && h->gap_recorded && hipEventElapsedTime(&ms, h->ev1, h->ev2) == hipSuccess && ms >= 0.f) {
      h->acc_ms[2] += ms;
      h->cnt[2]++;
    }
    h->gap_pending = h->gap_recorded = false;
    h->pending = -1;
  }
}

int ensure_param_capacity(m3hip_handle h, int n_par) {
  if (n_par <= h->par_cap) return M3HIP_OK;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (int i = 0; i < kRing; ++i) {
    if (h->h_params[i]) (void)hipHostFree(h->h_params[i]);
    HIPCHK(h, hipHostMalloc((void**)&h->h_params[i], sizeof(double) * n_par, hipHostMallocMapped));
    HIPCHK(h, hipHostGetDevicePointer((void**)&h->d_params_src[i], h->h_params[i], 0));
    h->ring_ticket[i] = 0;
  }
  dev_free(h->d_params);
  dev_free(h->d_params_f32);
  h->d_params_f32 = nullptr;
  int rc = dev_alloc(h, &h->d_params, size_t(n_par) + 1);
  if (rc) return rc;
  if ((rc = dev_alloc(h, &h->d_params_f32, size_t(n_par) + 1))) return rc;
  h->par_cap = n_par;
  return M3HIP_OK;
}

extern "C" {

const char* m3hip_version(void) { return "mach3_amd/m3hip 0.1 (gfx950)"; }

const char* m3hip_last_error(m3hip_handle h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int m3hip_create(int device_id, m3hip_handle* out) try {
  if (!out) return fail(nullptr, M3HIP_EINVAL, "out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(nullptr, M3HIP_EHIP, std::string("no HIP device available (") + hipGetErrorString(e) +
                                         "); this engine has no CPU fallback");
  if (device_id < 0 || device_id >= n) return fail(nullptr, M3HIP_EINVAL, "device_id out of range");
  {  // the library's code objects and its LDS budget (two 1024-lane workgroups of up to 66 KB per CU) are gfx950's
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) != 0)
      return fail(nullptr, M3HIP_EHIP, std::string("device ") + std::to_string(device_id) + " is " + prop.gcnArchName +
                                           ": this library is built for gfx950 (MI355X) only; this engine has no CPU fallback");
  }
  m3hip_ctx* h = new m3hip_ctx();
  h->device = device_id;
  (void)hipDeviceGetAttribute(&h->n_cu, hipDeviceAttributeMultiprocessorCount, device_id);
  if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipEventCreate(&h->ev0)) != hipSuccess || (e = hipEventCreate(&h->ev1)) != hipSuccess || (e = hipEventCreate(&h->ev2)) != hipSuccess) {
    std::string m = hipGetErrorString(e);
    delete h;
    return fail(nullptr, M3HIP_EHIP, "device initialisation failed: " + m);
  }
  h->stream = h->own_stream;
  *out = h;
  return M3HIP_OK;
} M3_API_CATCH(nullptr)

int m3hip_destroy(m3hip_handle h) try {
  if (!h) return M3HIP_OK;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  m3hip_comm_destroy(h);
  m3hip_xch_destroy(h);
  free_monolith(h);
  dev_free(h->d_xfer);
  for (int i = 0; i < kRing; ++i) {
    if (h->h_params[i]) (void)hipHostFree(h->h_params[i]);
  }
  dev_free(h->d_params);
  dev_free(h->d_params_f32);
  dev_free(h->d_osc_table);
  dev_free(h->d_osc_table_d);
  for (int i = 0; i < 2; ++i)
    if (h->h_osc_stage[i]) (void)hipHostFree(h->h_osc_stage[i]);
  dev_free(h->d_bin_samples);
  dev_free(h->d_bin_axes);
  dev_free(h->d_edges);
  dev_free(h->d_grid_off);
  dev_free(h->d_grid_list);
  dev_free(h->d_boxes);
  dev_free(h->d_shifts);
  dev_free(h->d_cuts);
  dev_free(h->d_sample_off);
  dev_free(h->d_data);
  dev_free(h->d_hist);
  dev_free(h->d_llh_terms);
  dev_free(h->d_dbg_clock);
  for (auto& p : h->pars) par_free(h, p);
  h->pars.clear();
  if (h->par_stream) (void)hipStreamDestroy(h->par_stream);
  if (h->h_out) (void)hipHostFree(h->h_out);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->ev2) (void)hipEventDestroy(h->ev2);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_set_stream(m3hip_handle h, void* hip_stream) try {
  if (!h) return M3HIP_EINVAL;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_device_count(int* out) try {
  if (!out) return M3HIP_EINVAL;
  *out = 0;
  int n = 0;
  const hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(nullptr, M3HIP_EHIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  *out = n;
  return M3HIP_OK;
} M3_API_CATCH(nullptr)

int m3hip_mem_info(m3hip_handle h, size_t* free_bytes, size_t* total_bytes) try {
  if (!h) return M3HIP_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemGetInfo(free_bytes, total_bytes));
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_alloc_pinned(m3hip_handle h, void** ptr, size_t bytes) try {
  if (!ptr) return M3HIP_EINVAL;
  // h may be NULL: the reference allocates its pinned segment / value arrays before its GPU object exists
  // (SplineMonolith.cpp:82-83 calls InitGPU_Segments / InitGPU_Vals through a null gpu_spline_handler)
  if (h) HIPCHK(h, hipSetDevice(h->device));
  const hipError_t e = hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault);
  if (e != hipSuccess) return fail(h, e == hipErrorOutOfMemory ? M3HIP_ENOMEM : M3HIP_EHIP, std::string("alloc_pinned: ") + hipGetErrorString(e));
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_free_pinned(m3hip_handle h, void* ptr) try {
  if (ptr) {
    const hipError_t e = hipHostFree(ptr);
    if (e != hipSuccess) return fail(h, M3HIP_EHIP, std::string("free_pinned: ") + hipGetErrorString(e));
  }
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_device_synchronize(void) try {  // SynchroniseSplines(): cudaDeviceSynchronize (gpuSplineUtils.cu:71-73)
  const hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return fail(nullptr, M3HIP_EHIP, std::string("device_synchronize: ") + hipGetErrorString(e));
  return M3HIP_OK;
} M3_API_CATCH(nullptr)

/* ---------------------------------------------------------------- test hooks */

int m3hip_debug_set_knobs(m3hip_handle h, int unroll, int nt, int lds_pad) try {
  if (!h) return M3HIP_EINVAL;
  h->knobs.unroll = unroll;
  h->knobs.nt = nt;
  h->knobs.lds_pad = lds_pad;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_debug_set_hist_replicas(m3hip_handle h, int n) try {
  if (!h || n < 1 || n > 64) return M3HIP_EINVAL;
  if (h->n_bins != 0) return fail(h, M3HIP_ESTATE, "set_hist_replicas must precede set_binning");
  h->hist_rep = h->hist_rep_cfg = n;
  h->hist_rep_forced = true;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_debug_host_times(m3hip_handle h, double* out6) try {
  if (!h || !out6) return M3HIP_EINVAL;
  for (int i = 0; i < 6; ++i) out6[i] = h->host_t[i];
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_debug_device_span(m3hip_handle h, int enable, double* last_us) try {
  if (!h) return M3HIP_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  if (enable && !h->d_dbg_clock) {
    HIPCHK(h, hipMalloc((void**)&h->d_dbg_clock, 16));
    HIPCHK(h, hipMemset(h->d_dbg_clock, 0, 16));
  }
  if (last_us && h->d_dbg_clock) {
    uint64_t c[2];
    HIPCHK(h, hipMemcpy(c, h->d_dbg_clock, 16, hipMemcpyDeviceToHost));
    last_us[0] = double(int64_t(c[1] - c[0])) * 0.01;  // wall_clock64 ticks at 100 MHz
    last_us[1] = double(c[0]) * 0.01;                  // device clock (us) at k_prep start
    last_us[2] = double(c[1]) * 0.01;                  // ... at k_llh end
    last_us[3] = h->host_abs[0] * 1e6;                 // host clock (us) at the k_prep launch call
    last_us[4] = h->host_abs[1] * 1e6;                 // ... when the completion flag was seen
  }
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_debug_set_bins(m3hip_handle h, const double* mc, const double* w2, int n_bins) try {
  if (!h) return M3HIP_EINVAL;
  if (h->n_bins == 0 || n_bins != h->n_bins || !mc || !w2) return fail(h, M3HIP_EINVAL, "debug_set_bins: n_bins mismatch");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemset(h->d_hist, 0, sizeof(double) * n_bins * 2 * h->hist_rep));
  HIPCHK(h, hipMemcpy(h->d_hist, mc, sizeof(double) * n_bins, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_hist + size_t(h->hist_rep) * n_bins, w2, sizeof(double) * n_bins, hipMemcpyHostToDevice));
  h->w2_spread = false;
  return M3HIP_OK;
} M3_API_CATCH(h)

/* ---------------------------------------------------------------- introspection */

int m3hip_get_info(m3hip_handle h, m3hip_info* out) try {
  if (!h || !out) return M3HIP_EINVAL;
  memset(out, 0, sizeof(*out));
  out->n_events = h->E;
  out->n_splines = h->S;
  out->n_tf1 = h->T;
  out->n_tiles_uniform = h->n_uniform;
  out->n_tiles_brick = h->n_brick;
  out->coeff_bytes = h->coeff_bytes;
  out->alg_bytes_eval = 16 * h->S + 8 * h->T + 4 * h->E;
  out->alg_bytes_fused = 16 * h->S + 8 * h->T + 8 * h->E;
  out->n_params = h->P;
  out->max_knots = h->K;
  out->n_bins = h->n_bins;
  out->n_samples = h->n_samples;
  out->n_tiles_padded = h->n_padded;
  out->pad_responses = h->pad_responses;
  out->n_grouped_chunks = h->n_grouped_chunks;
  out->brick_groups = uint64_t(h->brick_groups);
  out->cached_stream_bytes = h->last_cached_bytes;
  out->tf1_codes_around = h->last_nt_codes;
  if (h->osc_lines_stale && h->d_osc && h->E) {  // an upload the engine re-ordered (m3hip_monolith_order_key): counted in ITS order, from the device
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, wait_stream(h->stream));
    std::vector<int32_t> osc(size_t(h->E)), ln(1024);
    HIPCHK(h, hipMemcpy(osc.data(), h->d_osc, sizeof(int32_t) * h->E, hipMemcpyDeviceToHost));
    h->osc_line_sum = h->osc_line_blocks = 0;
    for (uint64_t b0 = 0; b0 + 1024 <= h->E; b0 += 1024) {
      size_t m = 0;
      for (uint64_t i = b0; i < b0 + 1024; ++i)
        if (osc[i] >= 0) ln[m++] = osc[i] >> 5;
      std::sort(ln.begin(), ln.begin() + m);
      h->osc_line_sum += uint64_t(std::unique(ln.begin(), ln.begin() + m) - ln.begin());
      h->osc_line_blocks += 1;
    }
    h->osc_lines_stale = false;
  }
  out->osc_lines_per_kevent = h->osc_line_blocks ? double(h->osc_line_sum) / double(h->osc_line_blocks) : 0.0;
  out->n_keyed_chunks = h->n_keyed_chunks;
  out->n_key_refused = h->n_key_refused;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_timing_enable(m3hip_handle h, int enable) try {
  if (!h) return M3HIP_EINVAL;
  h->timing = enable > 0 ? enable : 0;
  h->timing_ctr = 0;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_timing_read(m3hip_handle h, int which, double* avg_ms, uint64_t* n_launches) try {
  if (!h || which < 0 || which > 2) return M3HIP_EINVAL;
  if (avg_ms) *avg_ms = h->cnt[which] ? h->acc_ms[which] / double(h->cnt[which]) : 0.0;
  if (n_launches) *n_launches = h->cnt[which];
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_timing_reset(m3hip_handle h) try {
  if (!h) return M3HIP_EINVAL;
  h->acc_ms[0] = h->acc_ms[1] = h->acc_ms[2] = 0;
  h->cnt[0] = h->cnt[1] = h->cnt[2] = 0;
  return M3HIP_OK;
} M3_API_CATCH(h)

}  // extern "C"
