// m3hip_api.hip -- implementation of the C ABI declared in include/m3hip.h.
// Host orchestration only: device memory, the per-step launch sequence, RCCL, timing.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/m3hip.h"
#include "m3hip_kernels.h"

// RCCL: types and enum values only -- the library itself is loaded with dlopen.  A build host without the RCCL development
// headers (single-GPU installs) gets the handful of declarations the loader needs, written out from RCCL's public ABI.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId*);
ncclResult_t ncclCommInitRank(ncclComm_t*, int, ncclUniqueId, int);
ncclResult_t ncclAllReduce(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
ncclResult_t ncclAllGather(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
ncclResult_t ncclCommDestroy(ncclComm_t);
const char* ncclGetErrorString(ncclResult_t);
}
#endif
#include "m3hip_layout.h"

using namespace m3hip;

namespace {

constexpr int kRing = 4;  // parameter staging slots in flight
constexpr int kOscStaticSteps = 2;        // reweights an oscillation table must have been left alone before it counts as pre-staged
constexpr int kInlineResolveMax = 8192;  // unique slots resolved inside k_prep; beyond that k_resolve runs

struct Chunk {
  void* d_spl = nullptr;
  void* d_tf1 = nullptr;
  uint32_t* d_words = nullptr;  // brick metadata streams
};

// RCCL is loaded on demand so that single-GPU users never touch it (no link-time dependency); the header supplies the
// types and enum values, so the calls through the dlsym'd pointers have RCCL's own signatures.
struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool load(std::string* err) {
    if (lib) return true;
    lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) { *err = std::string("cannot load librccl: ") + dlerror(); return false; }
    GetUniqueId = (decltype(GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    CommInitRank = (decltype(CommInitRank))dlsym(lib, "ncclCommInitRank");
    AllReduce = (decltype(AllReduce))dlsym(lib, "ncclAllReduce");
    AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
    CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
    GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!GetUniqueId || !CommInitRank || !AllReduce || !AllGather || !CommDestroy) {
      *err = "librccl lacks nccl* symbols";
      dlclose(lib);
      lib = nullptr;
      GetUniqueId = nullptr; CommInitRank = nullptr; AllReduce = nullptr; AllGather = nullptr; CommDestroy = nullptr; GetErrorString = nullptr;
      return false;
    }
    return true;
  }
  std::string why(ncclResult_t r) const { return GetErrorString ? GetErrorString(r) : "error"; }
};
static_assert(sizeof(ncclUniqueId) == M3HIP_UNIQUE_ID_BYTES, "m3hip.h documents the unique id as 128 bytes");
Rccl g_rccl;
std::string g_create_err;

// One-shot exchange buffers created in THIS process, by the bytes of their IPC handle: a rank whose peer lives in the same
// process (one process driving several engines / devices) takes the peer's pointer from here -- hipIpcOpenMemHandle refuses
// a handle of the opening process's own.
std::mutex g_xch_mu;
std::map<std::string, void*> g_xch_local;

}  // namespace

// which events of an upload get bricks of three quarters of the size (described at ShortBrickPlan::begin)
struct ShortBrickPlan {
  bool decided = false;   // begin() has run for this upload
  uint64_t B = 0;         // events of a full brick
  uint64_t zone0 = 0;     // first event (global) that may be converted
  uint64_t left = 0;      // conversions (three full bricks -> four short ones) still to place
  void begin(uint64_t expect_events, uint64_t first_chunk_events, int G, int n_cu);
  void take(uint64_t chunk_first, uint64_t chunk_events, uint64_t* from, uint64_t* to);
};

struct m3hip_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  std::string err;

  // monolith
  int P = 0, K = 0;
  std::vector<int16_t> h_npts;
  float* d_coeff_x = nullptr;
  int16_t* d_npts = nullptr;
  bool mono_open = false, mono_done = false;
  uint64_t E = 0, S = 0, T = 0, coeff_bytes = 0, E_pad = 0;
  uint64_t n_uniform = 0, n_brick = 0, n_padded = 0, pad_responses = 0;
  std::vector<Chunk> chunks;
  std::vector<TileDev> h_tiles;
  TileDev* d_tiles = nullptr;
  std::vector<BrickDev> h_bricks;
  BrickDev* d_bricks = nullptr;
  // unique slot lists (flattened) and their per-step resolved form
  std::map<std::vector<int32_t>, uint32_t> list_index;
  std::vector<int32_t> u_param, u_plane0;
  std::vector<uint32_t> tile_res_off;
  std::vector<uint32_t> brick_res_off;  // per brick: first resolved slot of its chunk's TF1 dictionary
  int32_t* d_u_param = nullptr;
  int32_t* d_u_plane0 = nullptr;
  RSlot* d_resolved = nullptr;
  int32_t n_res = 0;

  // per-step parameter state
  PTab* d_ptab = nullptr;
  float* d_pval = nullptr;
  int16_t* d_seg = nullptr;
  int16_t* d_currseg = nullptr;
  int32_t* d_parmap = nullptr;
  bool parmap_set = false;
  double* h_params[kRing] = {nullptr, nullptr, nullptr, nullptr};  // pinned, device-mapped
  double* d_params_src[kRing] = {nullptr, nullptr, nullptr, nullptr};
  uint64_t ring_ticket[kRing] = {0, 0, 0, 0};  // step that last used the slot
  int ring_pos = 0;
  uint64_t ticket = 0;       // steps submitted
  uint64_t ticket_done = 0;  // steps known to have completed
  uint64_t llh_ticket = 0;
  uint64_t* d_dbg_clock = nullptr;  // 2 x uint64 device words (diagnostics)
  double dbg_dev_us = 0;            // device-side k_prep start -> k_llh end of the last step
  int par_cap = 0;
  double* d_params = nullptr;  // [par_cap + 1]
  float* d_params_f32 = nullptr;  // [par_cap + 1]: float(value), the cast CalcWeightTotal makes per event (k_prep writes both)
  float* h_vals = nullptr;     // pinned staging for the vals/segments path
  int16_t* h_segs = nullptr;
  float* d_vals_in = nullptr;
  int16_t* d_segs_in = nullptr;

  // events
  int32_t* d_bin = nullptr;
  int32_t* d_osc = nullptr;
  int16_t* d_nidx = nullptr;
  uint64_t* d_evw = nullptr;      // packed {bin, osc, norm indices} per event (ReweightArgs::evw)
  bool evw_dirty = true;          // d_bin / d_osc / d_nidx changed since d_evw was packed
  bool evw_ok = false;            // the sample fits the packed word's fields
  bool evw_weight = false;        // d_evw as packed: the words carry the oscillation WEIGHT (ReweightArgs::evw_weight), not the index
  char sw_packed = '1', sw_rebin = 'p';  // M3HIP_PACKED_EVENTS / M3HIP_REBIN as they stood at monolith_begin (experiments, tests): the step path reads no environment
  bool bins_in_word = false;      // the last re-binning pass wrote the words' bin field only: d_bin is stale (bins_from_word)
  uint32_t evw_norm_slots = 0;    // norm slots the word carries: M3_EVW_NORM_SLOTS, or 0 when the sample's norm indices do not fit its codes
  int64_t min_norm_idx = INT64_MAX, max_norm_idx_evw = -1;  // over the slots the packed word holds (an event's first M3_EVW_NORM_SLOTS)
  uint32_t M = 0;
  // largest indices any uploaded event carries (checked against the table / vector sizes at reweight: the kernels
  // index with them unguarded)
  int64_t max_osc_idx = -1, max_norm_idx = -1, max_parmap = -1, max_bin_idx = -1;
  // experiment weights behind the splines (m3hip_upload_event_weights / _weight_index / m3hip_set_weight_table)
  struct Extra {
    float* d_col = nullptr;
    int32_t* d_idx = nullptr;
    float* d_tab = nullptr;
    uint64_t n_tab = 0;
    int64_t max_idx = -1;
    void* h_stage[2] = {nullptr, nullptr};
    uint64_t stage_ticket[2] = {0, 0};
    int stage_pos = 0;
  } extra[M3_MAX_EXTRA];
  // deterministic fill (m3hip_set_deterministic): one row of bins per workgroup + the level-1 rows of the ordered fold
  bool deterministic = false;
  double* d_partial = nullptr;
  double* d_partial_l1 = nullptr;
  size_t partial_rows = 0, partial_bins = 0;
  ExtraSlot* d_extras = nullptr;
  int n_extra = 0;  // highest slot in use + 1
  bool extras_dirty = false;
  float* d_spline_w = nullptr;
  float* d_event_w = nullptr;
  float* d_osc_table = nullptr;
  uint64_t n_osc = 0;
  bool osc_valid_f32 = false, osc_valid_f64 = false;  // which precision of the table is current on the device
  // per-event oscillation weights for the brick kernel, expanded from a large table once it has stopped changing
  float* d_oscw = nullptr;
  bool oscw_valid = false;
  int osc_static_steps = 0;       // reweights since the table or the events' oscillation indices last changed
  int oscw_mode = -1;             // M3HIP_OSC_COLUMN: 0 never, 1 as soon as possible whatever the table's size, -1 the rule
  void* h_osc_stage[2] = {nullptr, nullptr};           // pinned staging of the per-step table upload
  uint64_t osc_stage_ticket[2] = {0, 0};
  int osc_stage_pos = 0;

  // binned spline handler (optional)
  int b_M = 0, b_U = 0, b_C = 0;
  int32_t* d_b_cidx = nullptr;
  int32_t* d_b_usv = nullptr;
  int32_t* d_b_uniq = nullptr;
  float* d_b_xcoeff = nullptr;
  float4* d_b_many = nullptr;
  float* d_b_w = nullptr;
  // M3::float_t = double build of the sample side (m3hip_binned_upload_f64)
  bool f64 = false;
  double* d_b_xcoeff_d = nullptr;
  double* d_b_many_d = nullptr;
  double* d_b_w_d = nullptr;
  double* d_coeff_x_d = nullptr;
  double* d_osc_table_d = nullptr;
  double* d_event_w_d = nullptr;
  int32_t* d_bwidx = nullptr;
  uint32_t Kb = 0;
  // shared-list form of the same information (chosen at the first m3hip_binned_upload_events when neighbouring
  // events mostly carry identical lists)
  int bl_mode = -1;  // -1 undecided, 0 ELL, 1 shared lists
  std::map<std::vector<int32_t>, int32_t> bl_map;
  std::vector<int32_t> bl_off, bl_idx;
  int32_t *d_bl_id = nullptr, *d_bl_off = nullptr, *d_bl_idx = nullptr;
  std::vector<int32_t> h_bl_id;   // host copy of the list ids (the processing order is derived from it)
  uint32_t* d_perm = nullptr;     // processing order: lane position -> event (binned-only spaces: grouped by pointer
                                  // list; monoliths: chunks whose events were grouped by parameter list at upload)
  bool perm_dirty = false;
  std::vector<std::pair<uint64_t, std::vector<uint32_t>>> grouped;  // (first event, order) of every grouped chunk
  bool perm_monolith = false;     // some chunk was grouped
  int brick_groups = 0;           // G of this monolith's bricks (m3hip_layout.h); 0 until the first chunk with ragged events fixes it
  uint64_t expect_events = 0;     // m3hip_monolith_expect_events: the events this device will hold (0: unknown); taken by the next monolith_begin
  uint64_t upload_expect = 0;     // ... the hint of the upload in progress
  int n_cu = 0;                   // compute units of the device
  ShortBrickPlan short_bricks;    // of the upload in progress
  uint64_t re_tables = 0, re_book = 0, re_codes = 0;  // bytes of the bricks' small re-read streams (stage tables, TF1 bookkeeping, TF1 codes)
  uint64_t last_cached_bytes = 0, last_nt_codes = 0;  // what the last fused reweight planned (m3hip_get_info)
  int nt_force = -1;              // M3HIP_NT_MASK = 0 | 4: the TF1 codes cached / around the Infinity Cache instead of the engine's choice (experiments)
  double cached_mb = 200.0;       // M3HIP_CACHED_MB: what the engine plans to keep in the Infinity Cache from step to step
  uint64_t n_grouped_chunks = 0;

  // device-side binning (optional): axes, nominal kinematics, shifts
  BinSample* d_bin_samples = nullptr;
  BinAxis* d_bin_axes = nullptr;
  double* d_edges = nullptr;
  int32_t* d_grid_off = nullptr;
  int32_t* d_grid_list = nullptr;
  double* d_boxes = nullptr;
  std::vector<int> h_sample_ndim;
  // m3hip_binning_begin .. _end builder state
  bool binning_open = false;
  std::vector<BinSample> bb_samples;
  std::vector<BinAxis> bb_axes;
  std::vector<double> bb_edges, bb_boxes;
  std::vector<int32_t> bb_grid_off, bb_grid_list, bb_nbins;
  int32_t* d_sample_id = nullptr;
  double* d_kin = nullptr;
  int32_t* d_nom_bin = nullptr;
  int kin_stride = 0;
  bool bins_computed = false;
  std::vector<KinShift> h_shifts;
  KinShift* d_shifts = nullptr;
  std::vector<KinCut> h_cuts;
  KinCut* d_cuts = nullptr;

  // bins
  int n_bins = 0, n_samples = 0;
  int32_t* d_sample_off = nullptr;
  double* d_data = nullptr;
  double* d_hist = nullptr;  // [2][hist_rep][n_bins]
  double* d_llh_terms = nullptr;  // split form of k_llh (many bins): per-bin terms [n_bins]
  int hist_rep = 32;      // flush replicas of the bin vector in use (chosen at set_binning, settled at the first reweight)
  int hist_rep_cfg = 64;  // ... upper limit
  int hist_rep_big = 32;  // ... for samples of many workgroups (allocated for), see set_binning
  bool hist_rep_locked = false;  // the first reweight has settled it: w2's replicas sit behind mc's, the layout must not move
  bool hist_rep_forced = false;  // m3hip_debug_set_hist_replicas: use exactly that many
  int teststat = M3HIP_kPoisson;
  bool update_w2 = false, first_w2 = true;
  bool w2_spread = false;  // w2 currently lives spread over the flush replicas (not yet folded into replica 0)
  double* h_out = nullptr;  // pinned, device-mapped: [1 + n_samples], then one uint64 ticket word
  double* d_out = nullptr;
  bool want_event_w = false;

  // comm
  void* comm = nullptr;
  // validator of the exchange: every rank's partial vector is gathered, copied to the host and summed there
  // in rank order; the all-reduced vector must agree to <= 1e-15 relative per bin (M3HIP_COMM_VALIDATE=1)
  bool comm_validate = false;
  double* d_gather = nullptr;
  size_t gather_cap = 0;
  std::vector<double> h_gather, h_reduced;
  double validate_max_rel = 0;
  uint64_t validate_count = 0;
  int rank = 0, n_ranks = 1;
  // one-shot exchange (m3hip_xch_*): own buffer xch[2][N][slot] in fine-grained device memory, the peers' buffers mapped
  // through IPC, the slots this rank pushes into, the step counter and the time-out word
  double* xch_own = nullptr;
  std::vector<void*> xch_peer;     // [N] base of every rank's buffer as mapped here (own: xch_own)
  std::vector<char> xch_peer_local;  // [N] 1: the buffer belongs to an engine of this process (nothing to close)
  double** d_xch_targets = nullptr;  // [2][N] device array: slot [parity][rank] of every rank's buffer
  int xch_ranks = 0, xch_rank = 0, xch_slot = 0;  // slot: doubles per slot (2 * n_bins + 2)
  bool xch_active = false;
  uint64_t xch_step = 0;
  // parameter-handler side of a step on the device (m3hip_par_*): prior and correlated throw
  struct ParDev {
    int n = 0;
    double* d_inv = nullptr;    // [n][n] inverse prior covariance (lower triangle read)
    double* d_chol = nullptr;   // [n][n] Cholesky factor of the throw matrix (lower triangular)
    double* d_rows = nullptr;   // [2n] per-row results
    double* d_in = nullptr;     // [2n] the step's [d | z] on board
    double* h_in[2] = {nullptr, nullptr};  // pinned, device-mapped: [d | z] of two steps in flight at most
    double* d_in_host[2] = {nullptr, nullptr};
    double* h_out = nullptr;    // pinned, device-mapped: [prior | corr_throw[n] | ticket]
    double* d_out = nullptr;
    uint64_t ticket = 0;        // launches so far
    uint64_t collected = 0;     // ... whose results have been waited for
    bool have_inv = false, have_chol = false;
    bool live = false;          // handed out by m3hip_par_create
  };
  std::vector<ParDev> pars;     // one per attached parameter handler (m3hip_par_create)
  hipStream_t par_stream = nullptr;  // its own stream: two latency-bound launches that need not queue behind the reweight kernel
  double xch_timeout_s = 2.0;     // every ticket spin of k_xch gives up after this long (m3hip_xch_set_timeout, M3HIP_XCH_TIMEOUT_S)
  uint64_t* h_xch_status = nullptr;  // mapped pinned host word (0: fine; else the step whose tickets did not arrive)
  uint64_t* d_xch_status = nullptr;
  unsigned long long* d_xch_pushed = nullptr;  // this rank's finished push blocks (n_ranks per step)

  // timing
  int timing = 0;           // 0: off, N: bracket every N-th launch with hipEvents
  uint64_t timing_ctr = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;  // reweight dispatch start / end; marker in front of k_llh (exchange gap)
  bool gap_pending = false;       // this step's reweight launch was timed and an exchange followed it
  bool gap_recorded = false;      // ... and ev2 has been recorded for it (m3hip_get_likelihood)
  int pending = -1;
  double acc_ms[3] = {0, 0, 0};   // [0] fused reweight kernel, [1] seam-2 evaluation, [2] reweight end -> k_llh ready (fold + all-reduce + unpark)
  uint64_t cnt[3] = {0, 0, 0};
  ReweightKnobs knobs;
  double host_t[6] = {0, 0, 0, 0, 0, 0};  // host-side seconds of the last step's phases (debug)
  double host_abs[2] = {0, 0};             // absolute host time (s): k_prep launch call, flag seen
};

namespace {

inline double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int fail(m3hip_handle h, int code, const std::string& msg) {
  if (h) h->err = msg;
  else g_create_err = msg;
  return code;
}

#define HIPCHK(h, expr)                                                                              \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess)                                                                            \
      return fail(h, e_ == hipErrorOutOfMemory ? M3HIP_ENOMEM : M3HIP_EHIP,                          \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                                \
  } while (0)

// No C++ exception may leave an extern "C" entry point (the callers are C, ctypes, cgo ...): every entry point is a
// function-try-block whose handler lands here and turns what was thrown into an error code + message.
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
#define M3_API_CATCH(h) catch (...) { return api_exception(h); }

template <class T>
int dev_alloc(m3hip_handle h, T** p, size_t n) {
  *p = nullptr;
  if (n == 0) n = 1;
  HIPCHK(h, hipMalloc((void**)p, n * sizeof(T)));
  return M3HIP_OK;
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

void par_free(m3hip_handle h, m3hip_ctx::ParDev& p) {  // buffers only: the slot stays live
  if (h->par_stream) (void)hipStreamSynchronize(h->par_stream);
  else if (h->stream) (void)hipStreamSynchronize(h->stream);
  dev_free(p.d_inv); dev_free(p.d_chol); dev_free(p.d_rows); dev_free(p.d_in);
  for (int i = 0; i < 2; ++i)
    if (p.h_in[i]) (void)hipHostFree(p.h_in[i]);
  if (p.h_out) (void)hipHostFree(p.h_out);
  const bool live = p.live;
  p = m3hip_ctx::ParDev();
  p.live = live;
}

// The one-shot exchange's verdict on the last step (the stream is idle up to that step's k_xch).  The status word holds
// step << 1 | kind: kind 0 = a ticket did not arrive within the time-out, 1 = a slot already carried a LATER step (this
// rank fell more than a whole step behind its peers and their data of this step is gone).  Either way the bins on this
// rank are its own partial sums and the GROUP has diverged: M3HIP_ECOMM from the exchange is fatal for every rank.
int xch_check(m3hip_handle h) {
  if (!h->xch_active || !h->h_xch_status || !*h->h_xch_status) return M3HIP_OK;
  const uint64_t st = *h->h_xch_status;
  *h->h_xch_status = 0;
  if (st & 1u)
    return fail(h, M3HIP_ECOMM, "one-shot exchange: a peer's slot already holds a later step than " + std::to_string(st >> 1) +
                                    " (this rank fell a whole step behind; the group has diverged)");
  return fail(h, M3HIP_ECOMM, "one-shot exchange: a peer's bin vector of step " + std::to_string(st >> 1) + " did not arrive within " +
                                  std::to_string(h->xch_timeout_s) + " s (the group has diverged)");
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

void free_monolith(m3hip_handle h) {
  for (auto& c : h->chunks) {
    dev_free(c.d_spl);
    dev_free(c.d_tf1);
    dev_free(c.d_words);
  }
  h->chunks.clear();
  h->h_bricks.clear();
  dev_free(h->d_bricks); h->d_bricks = nullptr;
  h->h_tiles.clear();
  dev_free(h->d_tiles); h->d_tiles = nullptr;
  dev_free(h->d_u_param); h->d_u_param = nullptr;
  dev_free(h->d_u_plane0); h->d_u_plane0 = nullptr;
  dev_free(h->d_resolved); h->d_resolved = nullptr;
  h->list_index.clear(); h->u_param.clear(); h->u_plane0.clear(); h->tile_res_off.clear(); h->brick_res_off.clear();
  h->n_res = 0;
  dev_free(h->d_coeff_x); h->d_coeff_x = nullptr;
  dev_free(h->d_npts); h->d_npts = nullptr;
  dev_free(h->d_ptab); h->d_ptab = nullptr;
  dev_free(h->d_pval); h->d_pval = nullptr;
  dev_free(h->d_seg); h->d_seg = nullptr;
  dev_free(h->d_currseg); h->d_currseg = nullptr;
  dev_free(h->d_parmap); h->d_parmap = nullptr;
  dev_free(h->d_vals_in); h->d_vals_in = nullptr;
  dev_free(h->d_segs_in); h->d_segs_in = nullptr;
  dev_free(h->d_bin); h->d_bin = nullptr;
  dev_free(h->d_osc); h->d_osc = nullptr;
  dev_free(h->d_nidx); h->d_nidx = nullptr;
  dev_free(h->d_oscw); h->d_oscw = nullptr;
  h->oscw_valid = false;
  h->osc_static_steps = 0;
  dev_free(h->d_evw); h->d_evw = nullptr;
  h->evw_dirty = true;
  h->evw_ok = false;
  h->evw_weight = false;
  h->bins_in_word = false;
  h->min_norm_idx = INT64_MAX;
  h->max_norm_idx_evw = -1;
  for (auto& x : h->extra) {
    dev_free(x.d_col); dev_free(x.d_idx); dev_free(x.d_tab);
    for (int i = 0; i < 2; ++i)
      if (x.h_stage[i]) (void)hipHostFree(x.h_stage[i]);
    x = m3hip_ctx::Extra();
  }
  dev_free(h->d_extras); h->d_extras = nullptr;
  dev_free(h->d_partial); dev_free(h->d_partial_l1);
  h->d_partial = h->d_partial_l1 = nullptr;
  h->partial_rows = h->partial_bins = 0;
  h->n_extra = 0;
  h->extras_dirty = false;
  dev_free(h->d_spline_w); h->d_spline_w = nullptr;
  dev_free(h->d_b_cidx); h->d_b_cidx = nullptr;
  dev_free(h->d_b_usv); h->d_b_usv = nullptr;
  dev_free(h->d_b_uniq); h->d_b_uniq = nullptr;
  dev_free(h->d_b_xcoeff); h->d_b_xcoeff = nullptr;
  dev_free(h->d_b_many); h->d_b_many = nullptr;
  dev_free(h->d_b_w); h->d_b_w = nullptr;
  dev_free(h->d_b_xcoeff_d); h->d_b_xcoeff_d = nullptr;
  dev_free(h->d_b_many_d); h->d_b_many_d = nullptr;
  dev_free(h->d_b_w_d); h->d_b_w_d = nullptr;
  dev_free(h->d_coeff_x_d); h->d_coeff_x_d = nullptr;
  dev_free(h->d_event_w_d); h->d_event_w_d = nullptr;
  h->f64 = false;
  dev_free(h->d_bwidx); h->d_bwidx = nullptr;
  dev_free(h->d_bl_id); h->d_bl_id = nullptr;
  dev_free(h->d_bl_off); h->d_bl_off = nullptr;
  dev_free(h->d_bl_idx); h->d_bl_idx = nullptr;
  dev_free(h->d_perm); h->d_perm = nullptr;
  h->grouped.clear();
  h->perm_monolith = false;
  h->n_grouped_chunks = 0;
  h->brick_groups = 0;
  h->h_bl_id.clear();
  h->perm_dirty = false;
  h->bl_mode = -1;
  h->bl_map.clear(); h->bl_off.clear(); h->bl_idx.clear();
  h->b_M = h->b_U = h->b_C = 0;
  h->Kb = 0;
  dev_free(h->d_sample_id); h->d_sample_id = nullptr;
  dev_free(h->d_kin); h->d_kin = nullptr;
  dev_free(h->d_nom_bin); h->d_nom_bin = nullptr;
  h->kin_stride = 0;
  h->bins_computed = false;
  dev_free(h->d_event_w); h->d_event_w = nullptr;
  if (h->h_vals) { (void)hipHostFree(h->h_vals); h->h_vals = nullptr; }
  if (h->h_segs) { (void)hipHostFree(h->h_segs); h->h_segs = nullptr; }
  h->M = 0;
  h->max_osc_idx = h->max_norm_idx = h->max_parmap = h->max_bin_idx = -1;
  h->E = h->S = h->T = h->coeff_bytes = h->E_pad = 0;
  h->re_tables = h->re_book = h->re_codes = 0;
  h->last_cached_bytes = h->last_nt_codes = 0;
  h->n_uniform = h->n_brick = h->n_padded = h->pad_responses = 0;
  h->mono_open = h->mono_done = false;
  h->parmap_set = false;
}

BinArgs make_bin_args(m3hip_handle h) {
  BinArgs b;
  memset(&b, 0, sizeof(b));
  b.samples = h->d_bin_samples;
  b.axes = h->d_bin_axes;
  b.edges = h->d_edges;
  b.grid_off = h->d_grid_off;
  b.grid_list = h->d_grid_list;
  b.boxes = h->d_boxes;
  b.sample_id = h->d_sample_id;
  b.kin = h->d_kin;
  b.nom_bin = h->d_nom_bin;
  b.bin = h->d_bin;
  b.shifts = h->d_shifts;
  b.n_shifts = int32_t(h->h_shifts.size());
  b.cuts = h->d_cuts;
  b.n_cuts = int32_t(h->h_cuts.size());
  b.params = h->d_params;
  b.kin_stride = h->kin_stride;
  b.n_events = h->E;
  return b;
}

// d_bin up to date again after re-binning passes that wrote the packed words only
int bins_from_word(m3hip_handle h) {
  if (!h->bins_in_word) return M3HIP_OK;
  h->bins_in_word = false;
  if (!h->d_evw || !h->d_bin) return M3HIP_OK;
  launch_unpack_bins(h->d_evw, h->d_bin, h->E, h->evw_weight ? kEvwBinMaskWeight : kEvwBinMask, h->stream);
  HIPCHK(h, hipGetLastError());
  return M3HIP_OK;
}

// whether the events' sample side fits the brick kernel's packed 8-byte word (ReweightArgs::evw)
bool evw_fits(m3hip_handle h) {
  bool ok = !h->f64 && !h->h_bricks.empty() && h->max_bin_idx < (1 << 20) - 2 && h->n_bins < (1 << 20) - 2 && h->max_osc_idx < (1 << 24) - 2 && h->d_bin && h->d_osc;
  return ok && h->sw_packed != '0';
}

// What stays in the Infinity Cache from step to step.  The brick kernel's large streams -- coefficients, TF1 rows, response
// words -- are loaded non-temporally and never allocate there, so what does is what a step re-reads: the stages' tables (a
// stage's idle waves re-read them), the sample words, slab 0's TF1 bookkeeping and the TF1 codes.  When all four fit the budget
// (200 MB of the 256: a table survives a non-temporal stream up to about 192 MB, tools/mall_probe.hip) they all keep their cached
// loads; when they do not, the codes -- the largest -- go around too (ReweightArgs::nt_codes: the kernel's NTC instantiation),
// unless the other three alone are beyond twice the budget: such a sample keeps everything cached (nothing would stay, and cached
// loads at least share lines inside a step).  Measured (kernel time, one box per group) on one GPU's share of config 5 (the three
// are 200.5 MB, the codes 125 MB): codes around 459.6 us, cached 474.0; on config 3:
// 1e7 events -- tables + sample words + bookkeeping cached (160 MB), codes (176 MB) around: 690.5 us; codes cached alone 695.0;
// sample words + bookkeeping only 701; everything cached, large streams included (round 3's way) 712; everything around 721.5.
// 5e6 events -- all four cached (168 MB): 341.0; codes alone 354.4; round 3's way 362.9.
// M3HIP_NT_MASK = 0 | 4 forces the choice (experiments), M3HIP_CACHED_MB the budget.
static uint32_t choose_nt_codes(m3hip_handle h, bool word_valid) {
  if (h->nt_force >= 0) return (uint32_t(h->nt_force) & 4u) ? 1u : 0u;
  const double budget = h->cached_mb * 1e6;
  const double small3 = double(h->re_tables) + (word_valid ? 8.0 * double(h->E) : 0.0) + double(h->re_book);
  return small3 <= 2.0 * budget && small3 + double(h->re_codes) > budget ? 1u : 0u;
}

ReweightArgs make_rw_args(m3hip_handle h) {
  ReweightArgs a;
  memset(&a, 0, sizeof(a));
  a.tiles = h->d_tiles;
  a.bricks = h->d_bricks;
  a.brick_groups = uint32_t(h->brick_groups > 0 ? h->brick_groups : 1);
  a.ptab = h->d_ptab;
  a.pval = h->d_pval;
  a.n_params = h->P;
  a.spline_w = h->d_spline_w;
  a.event_w = h->d_event_w;
  a.bin = h->d_bin;
  a.osc = h->d_osc;
  a.nidx = h->d_nidx;
  a.params = h->d_params;
  a.osc_table = h->d_osc_table;
  a.extras = h->d_extras;
  a.n_extra = uint32_t(h->n_extra);
  a.bwidx = h->d_bwidx;
  a.bl_id = h->bl_mode == 1 ? h->d_bl_id : nullptr;
  a.perm = (h->perm_monolith || (h->bl_mode == 1 && h->S + h->T == 0)) ? h->d_perm : nullptr;
  a.bl_off = h->d_bl_off;
  a.bl_idx = h->d_bl_idx;
  a.bw = h->d_b_w;
  a.bw_d = h->d_b_w_d;
  a.osc_table_d = h->d_osc_table_d;
  a.event_w_d = h->d_event_w_d;
  a.n_bw_slots = h->Kb;
  a.hist = h->d_hist;
  a.hist_rep = h->hist_rep;
  a.e_pad = h->E_pad;
  a.n_norm_slots = h->M;
  a.n_bins = h->n_bins;
  a.fill_w2 = h->first_w2 ? 1 : 0;
  a.hist_lds = sizeof(double) * (a.fill_w2 ? 2 : 1) * size_t(h->n_bins) <= size_t(M3_HIST_LDS_MAX_BYTES) ? 1 : 0;
  a.tile_base = 0;
  return a;
}

// Validator, part 1 (before the all-reduce): all-gather of the ranks' partial vectors, to the host.
int comm_validate_begin(m3hip_handle h, size_t n_msg) {
  const size_t need = n_msg * size_t(h->n_ranks);
  if (h->gather_cap < need) {
    dev_free(h->d_gather); h->d_gather = nullptr;
    const int rc = dev_alloc(h, &h->d_gather, need);
    if (rc) return rc;
    h->gather_cap = need;
  }
  const ncclResult_t r = g_rccl.AllGather(h->d_hist, h->d_gather, n_msg, ncclDouble, (ncclComm_t)h->comm, h->stream);
  if (r != ncclSuccess) return fail(h, M3HIP_ECOMM, "ncclAllGather (validator): " + g_rccl.why(r));
  h->h_gather.resize(need);
  HIPCHK(h, hipMemcpyAsync(h->h_gather.data(), h->d_gather, sizeof(double) * need, hipMemcpyDeviceToHost, h->stream));
  return M3HIP_OK;
}
// Part 2 (after it): the reduced vector to the host, against the host's own sum in rank order.
int comm_validate_end(m3hip_handle h, size_t n_msg) {
  h->h_reduced.resize(n_msg);
  HIPCHK(h, hipMemcpyAsync(h->h_reduced.data(), h->d_hist, sizeof(double) * n_msg, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  double worst = 0;
  for (size_t i = 0; i < n_msg; ++i) {
    long double sum = 0;
    for (int r = 0; r < h->n_ranks; ++r) sum += h->h_gather[size_t(r) * n_msg + i];
    const double ref = double(sum), got = h->h_reduced[i];
    const double rel = std::fabs(got - ref) / std::max(std::fabs(ref), 1e-300);
    if (!(got == ref)) worst = std::max(worst, rel);
  }
  h->validate_max_rel = std::max(h->validate_max_rel, worst);
  h->validate_count++;
  // n - 1 additions, each within half an ulp of its partial sum: the reduced value may differ from the correctly rounded
  // one by about n ulps whatever order RCCL adds in
  if (worst > 1.2e-16 * 2 * std::max(2, h->n_ranks))
    return fail(h, M3HIP_ECOMM, "all-reduce validator: bin vector differs from the host-side sum of the ranks' partial vectors by " +
                                    std::to_string(worst) + " relative");
  return M3HIP_OK;
}

}  // namespace

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

/* ---------------------------------------------------------------- monolith upload */

int m3hip_monolith_begin(m3hip_handle h, int n_params, int max_knots, const float* coeff_x, const int16_t* npts) try {
  if (!h) return M3HIP_EINVAL;
  if (n_params <= 0 || n_params > 32767 || max_knots < 1 || !coeff_x || !npts)
    return fail(h, M3HIP_EINVAL, "monolith_begin: bad n_params / max_knots / NULL arrays");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  free_monolith(h);
  { const char* f = getenv("M3HIP_PACKED_EVENTS"); h->sw_packed = f && f[0] ? f[0] : '1'; }
  { const char* f = getenv("M3HIP_REBIN"); h->sw_rebin = f && f[0] ? f[0] : 'p'; }
  { const char* f = getenv("M3HIP_OSC_COLUMN"); h->oscw_mode = f && f[0] ? atoi(f) : -1; }
  h->upload_expect = h->expect_events;  // the hint belongs to this upload, whether it completes or not
  h->short_bricks = ShortBrickPlan();
  { const char* f = getenv("M3HIP_NT_MASK"); h->nt_force = f && f[0] ? atoi(f) : -1; }
  { const char* f = getenv("M3HIP_CACHED_MB"); h->cached_mb = f && f[0] ? atof(f) : 200.0; }
  h->expect_events = 0;
  h->P = n_params;
  h->K = max_knots;
  h->h_npts.assign(npts, npts + n_params);
  int rc;
  if ((rc = dev_alloc(h, &h->d_coeff_x, size_t(n_params) * max_knots))) return rc;
  if ((rc = dev_alloc(h, &h->d_npts, size_t(n_params)))) return rc;
  if ((rc = dev_alloc(h, &h->d_ptab, size_t(n_params)))) return rc;
  if ((rc = dev_alloc(h, &h->d_pval, size_t(n_params)))) return rc;
  if ((rc = dev_alloc(h, &h->d_seg, size_t(n_params)))) return rc;
  if ((rc = dev_alloc(h, &h->d_currseg, size_t(n_params)))) return rc;
  if ((rc = dev_alloc(h, &h->d_parmap, size_t(n_params)))) return rc;
  if ((rc = dev_alloc(h, &h->d_vals_in, size_t(n_params)))) return rc;
  if ((rc = dev_alloc(h, &h->d_segs_in, size_t(n_params)))) return rc;
  HIPCHK(h, hipHostMalloc((void**)&h->h_vals, sizeof(float) * n_params, hipHostMallocDefault));
  HIPCHK(h, hipHostMalloc((void**)&h->h_segs, sizeof(int16_t) * n_params, hipHostMallocDefault));
  HIPCHK(h, hipMemcpy(h->d_coeff_x, coeff_x, sizeof(float) * n_params * max_knots, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_npts, npts, sizeof(int16_t) * n_params, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemset(h->d_currseg, 0, sizeof(int16_t) * n_params));  // FastSplineInfo(): CurrSegment = 0
  HIPCHK(h, hipMemset(h->d_seg, 0, sizeof(int16_t) * n_params));      // SplineMonolith.cpp:91
  std::vector<int32_t> ident(n_params);
  for (int i = 0; i < n_params; ++i) ident[i] = i;
  HIPCHK(h, hipMemcpy(h->d_parmap, ident.data(), sizeof(int32_t) * n_params, hipMemcpyHostToDevice));
  h->mono_open = true;
  return M3HIP_OK;
} M3_API_CATCH(h)

// G of an engine's bricks (m3hip_layout.h: a brick holds up to G x 1024 events).  Larger bricks spread a workgroup's fixed
// phases over more events and lengthen the parameter runs, but leave fewer workgroups: the device wants several rounds
// of its 512 resident ones.  M3HIP_BRICK_GROUPS = 1 | 2 | 4 overrides; without m3hip_monolith_expect_events the total is
// unknown and G stays 1.
static int choose_brick_groups(const ChunkInput& in, uint64_t expect_events) {
  if (const char* f = getenv("M3HIP_BRICK_GROUPS")) {
    const int g = atoi(f);
    if (g == 1 || g == 2 || g == 4) return g;
  }
  if (expect_events == 0 || in.n_events == 0) return 1;
  uint64_t resp = 0;
  for (uint64_t e = 0; e < in.n_events; ++e) resp += uint64_t(in.npe[2 * e]) + in.npe_tf1[2 * e];
  const double per_event = double(resp) / double(in.n_events);
  // measured (round 3, scripts/gexp.py, one GPU's share of config 5 -- 16 responses per event, 1.25e7 events): bricks of 1024 /
  // 2048 / 4096 events stream at 0.542 / 0.566 / 0.586 of the HBM peak (runs four times as long, a quarter of the per-brick
  // phases); at 32 responses per event (config 3) 0.672 / 0.664 / 0.655: what larger bricks save there is lost in the longer
  // tail of the launch and the second slab's exposed first round trip
  int g = per_event <= 24.0 ? 4 : 1;
  while (g > 1 && expect_events / (uint64_t(M3_GROUP) * g) < 2048) g /= 2;  // at least four rounds of the 512 resident workgroups
  return g;
}

// Equal shares of a brick launch.  The device hands workgroups to its XCDs in turn and each compute unit works on two at a time,
// so a launch of nb equal bricks lasts about as long as the compute unit with the most of them: ceil(nb / n_cu) bricks' worth.
// When the upload announced its size (m3hip_monolith_expect_events) and the last, partly filled round would idle more than
// 3 % of the launch, the count is brought to the next multiple of n_cu by cutting bricks near the END of the sample at three
// quarters of the size -- four where three would be, so that a converted stretch ends where the full bricks' boundaries are
// and chunk borders cost nothing; last, so that the greedy dispatch finishes on the small ones.  Measured (round 4, config-3
// events, kernel time without / with): 5e6 events = 19.07 bricks per unit 360 / 357 us (four samples of 800 bins: 373.5 /
// 361.8), 2.5e6 = 9.5 per unit 185.3 / 183.4; NOT applied where it measured level or worse: 1e7 events = 38.15 per unit
// (693.3 / 692.8: the short bricks' fixed costs eat what the even finish gains), 1.25e6 = 4.8 per unit (86.6 / 88.3: with so
// few rounds the units stay in step and a fifth of the bricks short only costs).  begin(): the rule, from the announced size
// and the first chunk's (the stretch is started early by one conversion per chunk border expected inside it, because a
// chunk's share of it is rounded down to whole conversions); take(): a chunk's share.  M3HIP_SHORT_BRICKS=0 switches it off,
// =2 applies it wherever bricks can be gained (tests).
void ShortBrickPlan::begin(uint64_t expect_events, uint64_t first_chunk_events, int G, int n_cu) {
  *this = ShortBrickPlan();
  decided = true;
  bool force = false;
  if (const char* f = getenv("M3HIP_SHORT_BRICKS")) {
    if (f[0] == '0') return;
    force = f[0] == '2';
  }
  if (expect_events == 0 || n_cu <= 0 || first_chunk_events == 0) return;
  B = uint64_t(M3_GROUP) * uint64_t(G);
  // bricks are cut chunk by chunk: every chunk of the first one's size ends in a partial brick of its own when that size is no
  // multiple of B (G > 1 with chunks that are multiples of 1024 only)
  const uint64_t c = first_chunk_events, cu = uint64_t(n_cu);
  const uint64_t nb = expect_events / c * ((c + B - 1) / B) + (expect_events % c + B - 1) / B;
  if (nb % cu == 0) return;
  if (!force && (nb / cu < 8 || (cu - nb % cu) * 100 < 3 * nb)) return;  // few rounds, or less than 3 % of the launch to gain
  const uint64_t extra = (nb + cu - 1) / cu * cu - nb;              // bricks to gain: one per conversion of 3 B events
  const uint64_t slack = (3 * B * extra + first_chunk_events - 1) / first_chunk_events + 1;
  const uint64_t full_end = expect_events / B * B;                    // (the events behind it are the sample's last, partial brick)
  if (3 * B * (extra + slack) * 4 > full_end) return;
  zone0 = full_end - 3 * B * (extra + slack);
  left = extra;
}
void ShortBrickPlan::take(uint64_t chunk_first, uint64_t chunk_events, uint64_t* from, uint64_t* to) {
  *from = *to = ~0ull;
  if (left == 0) return;
  // bricks are cut chunk-locally (a chunk's first brick starts at its first event): the stretch starts on the chunk's own
  // next brick boundary at or behind zone0, wherever the chunk borders fall against the global multiples of B
  const uint64_t b = chunk_first + chunk_events;
  const uint64_t a = chunk_first + (std::max(zone0, chunk_first) - chunk_first + B - 1) / B * B;
  if (a >= b) return;
  const uint64_t n = std::min(left, (b - a) / (3 * B));
  if (n == 0) return;
  *from = a - chunk_first;
  *to = *from + 3 * B * n;
  left -= n;
}

int m3hip_monolith_expect_events(m3hip_handle h, uint64_t n_events) try {
  if (!h) return M3HIP_EINVAL;
  h->expect_events = n_events;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_monolith_append(m3hip_handle h, uint64_t n_events, const float* coeff_many, uint64_t total_nknots,
                          const unsigned int* nknots_arr, const int16_t* param_no, uint64_t n_splines,
                          const float* coeff_tf1, const int16_t* param_no_tf1, uint64_t n_tf1,
                          const unsigned int* n_param_per_event, const unsigned int* n_param_per_event_tf1) try {
  if (!h) return M3HIP_EINVAL;
  if (!h->mono_open) return fail(h, M3HIP_ESTATE, "monolith_append before monolith_begin");
  if (h->E % M3_GROUP != 0) return fail(h, M3HIP_EINVAL, "only the last chunk may hold a non-multiple of 1024 events");
  if (h->E + n_events > 0xFFFFF000ull) return fail(h, M3HIP_EINVAL, "more than 2^32 events on one device");
  if (n_events == 0) return M3HIP_OK;
  if ((n_splines && (!coeff_many || !nknots_arr || !param_no)) || (n_tf1 && (!coeff_tf1 || !param_no_tf1)))
    return fail(h, M3HIP_EINVAL, "monolith_append: NULL coefficient arrays");
  HIPCHK(h, hipSetDevice(h->device));
  ChunkInput in;
  in.n_events = n_events;
  in.coeff_many = coeff_many;
  in.total_nknots = total_nknots;
  in.nknots_arr = nknots_arr;
  in.param_no = param_no;
  in.n_splines = n_splines;
  in.coeff_tf1 = coeff_tf1;
  in.param_no_tf1 = param_no_tf1;
  in.n_tf1 = n_tf1;
  in.npe = n_param_per_event;
  in.npe_tf1 = n_param_per_event_tf1;
  ChunkPlan plan;
  std::string err;
  { const char* f = getenv("M3HIP_PAD_MIN_FILL"); g_pad_min_fill = f ? atof(f) : kPadMinFillDefault; }  // experiments: > 1 disables the identity padding
  { const char* f = getenv("M3HIP_TF1_SORT"); g_tf1_sort = f ? atoi(f) : 2; }
  { const char* f = getenv("M3HIP_PAD_RUNS"); g_pad_runs = f ? atoi(f) : 2; }
  // G of the bricks, fixed by the first chunk (all bricks of an engine share one kernel instantiation)
  if (h->brick_groups == 0) h->brick_groups = choose_brick_groups(in, h->upload_expect);
  const int G = h->brick_groups;
  if (!h->short_bricks.decided) h->short_bricks.begin(h->upload_expect, n_events, G, h->n_cu);
  uint64_t short_from, short_to;  // chunk-local
  h->short_bricks.take(h->E, n_events, &short_from, &short_to);
  if (!plan_chunk(h->P, h->K, h->h_npts.data(), in, &plan, &err, G, short_from, short_to)) return fail(h, M3HIP_EINVAL, "monolith_append: " + err);

  // Grouping.  When a sizeable part of the chunk would end up in bricks, try the chunk with its events grouped by
  // parameter list (stable sort): events of one interaction mode then sit together and form uniform (or padded)
  // tiles.  Only the small {count,start} arrays are permuted -- they point into the flat response arrays, which
  // stay as they are -- and the kernels map a lane position back to its event through the permutation, so every
  // per-event array and every result keeps the caller's event order.
  std::vector<uint32_t> order;
  std::vector<unsigned int> npe_g, npe_tf1_g;
  {
    uint64_t ragged = 0;
    for (const BrickHost& bh : plan.bricks) ragged += bh.n_events;
    const char* f = getenv("M3HIP_GROUP_EVENTS");
    const bool enabled = !(f && f[0] == '0');
    if (enabled && ragged * 10 > n_events) {
      std::vector<uint32_t> ord(n_events);
      for (uint64_t e = 0; e < n_events; ++e) ord[e] = uint32_t(e);
      auto less = [&](uint32_t x, uint32_t y) {  // lexicographic on (spline parameters, TF1 parameters)
        const unsigned int cx = n_param_per_event[2 * x], sx = n_param_per_event[2 * x + 1];
        const unsigned int cy = n_param_per_event[2 * y], sy = n_param_per_event[2 * y + 1];
        if (std::lexicographical_compare(param_no + sx, param_no + sx + cx, param_no + sy, param_no + sy + cy)) return true;
        if (std::lexicographical_compare(param_no + sy, param_no + sy + cy, param_no + sx, param_no + sx + cx)) return false;
        const unsigned int tx = n_param_per_event_tf1[2 * x], ux = n_param_per_event_tf1[2 * x + 1];
        const unsigned int ty = n_param_per_event_tf1[2 * y], uy = n_param_per_event_tf1[2 * y + 1];
        return std::lexicographical_compare(param_no_tf1 + ux, param_no_tf1 + ux + tx, param_no_tf1 + uy, param_no_tf1 + uy + ty);
      };
      std::stable_sort(ord.begin(), ord.end(), less);
      npe_g.resize(2 * n_events);
      npe_tf1_g.resize(2 * n_events);
      for (uint64_t pos = 0; pos < n_events; ++pos) {
        const uint32_t e = ord[pos];
        npe_g[2 * pos] = n_param_per_event[2 * e];
        npe_g[2 * pos + 1] = n_param_per_event[2 * e + 1];
        npe_tf1_g[2 * pos] = n_param_per_event_tf1[2 * e];
        npe_tf1_g[2 * pos + 1] = n_param_per_event_tf1[2 * e + 1];
      }
      ChunkInput in_g = in;
      in_g.npe = npe_g.data();
      in_g.npe_tf1 = npe_tf1_g.data();
      ChunkPlan plan_g;
      std::string err_g;
      if (plan_chunk(h->P, h->K, h->h_npts.data(), in_g, &plan_g, &err_g, G, short_from, short_to)) {
        uint64_t ragged_g = 0;
        for (const BrickHost& bh : plan_g.bricks) ragged_g += bh.n_events;
        if (ragged_g * 2 <= ragged) {  // at least halves what goes through bricks
          in = in_g;
          plan = std::move(plan_g);
          order = std::move(ord);
        }
      }
    }
  }
  // host staging (pinned) -> device
  struct Pinned {  // released on every way out of this function
    void* p = nullptr;
    ~Pinned() { if (p) (void)hipHostFree(p); }
  } st_spl, st_tf1, st_meta;
  const size_t b_spl = size_t(plan.spl_float4) * 16, b_tf1 = size_t(plan.tf1_float2) * 8, b_meta = size_t(plan.meta_int32) * 4;
  HIPCHK(h, hipHostMalloc(&st_spl.p, b_spl ? b_spl : 16, hipHostMallocDefault));
  HIPCHK(h, hipHostMalloc(&st_tf1.p, b_tf1 ? b_tf1 : 16, hipHostMallocDefault));
  HIPCHK(h, hipHostMalloc(&st_meta.p, b_meta ? b_meta : 16, hipHostMallocDefault));
  float *s_spl = static_cast<float*>(st_spl.p), *s_tf1 = static_cast<float*>(st_tf1.p);
  int32_t* s_meta = static_cast<int32_t*>(st_meta.p);
  fill_chunk(h->P, h->h_npts.data(), in, plan, s_spl, s_tf1, s_meta);
  const std::vector<int32_t> h_meta(s_meta, s_meta + plan.meta_int32);

  Chunk c;
  // + slack: idle lanes of the brick kernel read (and drop) what lies behind the list they index (m3hip_layout.h)
  hipError_t e = hipMalloc(&c.d_spl, b_spl + M3_STORE_SLACK_BYTES);
  if (e == hipSuccess) e = hipMalloc(&c.d_tf1, b_tf1 + M3_STORE_SLACK_BYTES);
  const size_t b_words = plan.brick_words.size() * sizeof(uint32_t);
  if (e == hipSuccess && b_words) e = hipMalloc((void**)&c.d_words, b_words + M3_STORE_SLACK_BYTES);
  if (e == hipSuccess && b_words) e = hipMemcpy(c.d_words, plan.brick_words.data(), b_words, hipMemcpyHostToDevice);
  if (e == hipSuccess && b_spl) e = hipMemcpy(c.d_spl, s_spl, b_spl, hipMemcpyHostToDevice);
  if (e == hipSuccess && b_tf1) e = hipMemcpy(c.d_tf1, s_tf1, b_tf1, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    dev_free(c.d_spl); dev_free(c.d_tf1); dev_free(c.d_words);
    return fail(h, e == hipErrorOutOfMemory ? M3HIP_ENOMEM : M3HIP_EHIP, std::string("monolith_append upload: ") + hipGetErrorString(e));
  }
  if (!order.empty()) {  // only now: a chunk that failed to upload leaves no trace
    h->perm_monolith = true;
    h->n_grouped_chunks++;
    h->grouped.emplace_back(h->E, std::move(order));
  }
  h->chunks.push_back(c);
  for (const TileHost& th : plan.tiles) {
    TileDev td;
    td.spl = reinterpret_cast<const float4*>(c.d_spl) + th.spl_off;
    td.tf1 = reinterpret_cast<const float2*>(c.d_tf1) + th.tf1_off;
    td.res = nullptr;  // set in monolith_end once the unique slot lists are known
    td.n_spl_slots = th.n_spl_slots;
    td.n_tf1_slots = th.n_tf1_slots;
    td.n_events = th.n_events;
    td.first_event = uint32_t(h->E + th.first_event);
    h->h_tiles.push_back(td);
    // unique ordered parameter lists: tiles that share one also share its resolved slots
    std::vector<int32_t> key(h_meta.begin() + th.meta_off, h_meta.begin() + th.meta_off + 2 * th.n_spl_slots + th.n_tf1_slots);
    key.push_back(int32_t(th.n_spl_slots));
    auto it = h->list_index.find(key);
    uint32_t off;
    if (it == h->list_index.end()) {
      off = uint32_t(h->u_param.size());
      for (uint32_t j = 0; j < th.n_spl_slots; ++j) {
        h->u_param.push_back(key[2 * j]);
        h->u_plane0.push_back(key[2 * j + 1]);
      }
      for (uint32_t j = 0; j < th.n_tf1_slots; ++j) {
        h->u_param.push_back(key[2 * th.n_spl_slots + j]);
        h->u_plane0.push_back(-1);
      }
      h->list_index.emplace(std::move(key), off);
    } else {
      off = it->second;
    }
    h->tile_res_off.push_back(off);
  }
  // The bricks' TF1 codes step through the chunk's dictionary of TF1 parameters; the parameter VALUE of every entry is
  // resolved once per step next to the uniform tiles' slot lists (k_prep / k_resolve: RSlot::f), so a brick fetches
  // its table with one load per lane.  Chunks with the same dictionary share the resolved entries.
  uint32_t dict_res = 0;
  if (!plan.bricks.empty() && !plan.tf1_dict.empty()) {
    std::vector<int32_t> key(plan.tf1_dict.begin(), plan.tf1_dict.end());
    key.push_back(-2);  // cannot collide with a tile's list (those end in their TSpline3 slot count >= 0)
    auto it = h->list_index.find(key);
    if (it == h->list_index.end()) {
      dict_res = uint32_t(h->u_param.size());
      for (uint16_t p : plan.tf1_dict) {
        h->u_param.push_back(int32_t(p));
        h->u_plane0.push_back(-1);
      }
      h->list_index.emplace(std::move(key), dict_res);
    } else {
      dict_res = it->second;
    }
  }
  for (const BrickHost& bh : plan.bricks) {
    h->re_tables += bh.bytes_tables;
    h->re_book += bh.bytes_book;
    h->re_codes += bh.bytes_codes;
    h->brick_res_off.push_back(dict_res);
    BrickDev bd;
    bd.spl = reinterpret_cast<const float4*>(c.d_spl) + bh.spl_off;
    bd.tf1 = reinterpret_cast<const float2*>(c.d_tf1) + bh.tf1_off;
    bd.words = c.d_words + bh.words_off;
    bd.tf1_x = nullptr;  // set in monolith_end once the resolved slots are allocated
    bd.n_events = bh.n_events;
    bd.first_event = uint32_t(h->E + bh.first_event);
    const uint32_t* bw = plan.brick_words.data() + bh.words_off;
    bd.n_stages = bw[0];
    bd.off_tf1 = bw[1];
    for (int i = 0; i < 4; ++i) bd.st0[i] = bd.n_stages > 0 ? bw[M3_BRICK_HDR0 + i] : 0u;
    for (int i = 0; i < 4; ++i) bd.st1[i] = bd.n_stages > 1 ? bw[M3_BRICK_HDR0 + M3_STAGE_HDR + i] : 0u;
    bd.n_dict = uint32_t(plan.tf1_dict.size());
    bd.spl_last_off = bh.n_spl_f4 ? uint32_t((bh.n_spl_f4 - 1) * 16) : 0u;
    bd.pad_[0] = bh.groups;
    bd.pad_[1] = bh.tf1_sorted ? 1u : 0u;
    h->h_bricks.push_back(bd);
  }
  h->n_uniform += plan.n_uniform;
  h->n_brick += plan.n_brick;
  h->n_padded += plan.n_padded;
  h->pad_responses += plan.pad_spl + plan.pad_tf1;
  h->E += n_events;
  // the responses of THIS chunk's events: the arrays handed over may be those of a larger sample (the {count,start}
  // pairs index them), as when the C++ facade uploads one sample in slices
  for (uint64_t e = 0; e < n_events; ++e) {
    h->S += n_param_per_event[2 * e];
    h->T += n_param_per_event_tf1[2 * e];
  }
  h->coeff_bytes += b_spl + b_tf1;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_monolith_end(m3hip_handle h) try {
  if (!h) return M3HIP_EINVAL;
  if (!h->mono_open) return fail(h, M3HIP_ESTATE, "monolith_end before monolith_begin");
  HIPCHK(h, hipSetDevice(h->device));
  const uint64_t n_tiles = h->h_tiles.size();
  h->E_pad = ((h->E + M3_GROUP - 1) / M3_GROUP) * M3_GROUP;
  if (h->E_pad == 0) h->E_pad = M3_GROUP;
  int rc;
  h->n_res = int32_t(h->u_param.size());
  if ((rc = dev_alloc(h, &h->d_u_param, size_t(h->n_res)))) return rc;
  if ((rc = dev_alloc(h, &h->d_u_plane0, size_t(h->n_res)))) return rc;
  if ((rc = dev_alloc(h, &h->d_resolved, size_t(h->n_res)))) return rc;
  if (h->n_res) {
    HIPCHK(h, hipMemcpy(h->d_u_param, h->u_param.data(), sizeof(int32_t) * h->n_res, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_u_plane0, h->u_plane0.data(), sizeof(int32_t) * h->n_res, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemset(h->d_resolved, 0, sizeof(RSlot) * h->n_res));
  }
  for (uint64_t t = 0; t < n_tiles; ++t) h->h_tiles[t].res = h->d_resolved + h->tile_res_off[t];
  for (size_t b = 0; b < h->h_bricks.size(); ++b) h->h_bricks[b].tf1_x = h->d_resolved + h->brick_res_off[b];
  h->upload_expect = 0;
  h->list_index.clear();
  if ((rc = dev_alloc(h, &h->d_bricks, h->h_bricks.size()))) return rc;
  if (!h->h_bricks.empty())
    HIPCHK(h, hipMemcpy(h->d_bricks, h->h_bricks.data(), sizeof(BrickDev) * h->h_bricks.size(), hipMemcpyHostToDevice));
  if ((rc = dev_alloc(h, &h->d_tiles, size_t(n_tiles)))) return rc;
  if (n_tiles) HIPCHK(h, hipMemcpy(h->d_tiles, h->h_tiles.data(), sizeof(TileDev) * n_tiles, hipMemcpyHostToDevice));
  if ((rc = dev_alloc(h, &h->d_spline_w, size_t(h->E_pad)))) return rc;
  if ((rc = dev_alloc(h, &h->d_bin, size_t(h->E_pad)))) return rc;
  if ((rc = dev_alloc(h, &h->d_osc, size_t(h->E_pad)))) return rc;
  HIPCHK(h, hipMemset(h->d_bin, 0xFF, sizeof(int32_t) * h->E_pad));  // -1: under/overflow
  HIPCHK(h, hipMemset(h->d_osc, 0xFF, sizeof(int32_t) * h->E_pad));  // -1: no oscillation pointer
  HIPCHK(h, hipMemset(h->d_spline_w, 0, sizeof(float) * h->E_pad));
  if (h->perm_monolith) {  // lane position -> event, identity over the padding and over the chunks that were not grouped
    std::vector<uint32_t> perm(size_t(h->E_pad));
    for (size_t i = 0; i < perm.size(); ++i) perm[i] = uint32_t(i);
    for (const auto& g : h->grouped)
      for (size_t pos = 0; pos < g.second.size(); ++pos) perm[size_t(g.first) + pos] = uint32_t(g.first + g.second[pos]);
    if ((rc = dev_alloc(h, &h->d_perm, size_t(h->E_pad)))) return rc;
    HIPCHK(h, hipMemcpy(h->d_perm, perm.data(), sizeof(uint32_t) * h->E_pad, hipMemcpyHostToDevice));
  }
  h->grouped.clear();
  h->grouped.shrink_to_fit();
  h->mono_open = false;
  h->mono_done = true;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_upload_monolith(m3hip_handle h, uint64_t n_events, int n_params, int max_knots, const float* coeff_x,
                          const int16_t* npts, const float* coeff_many, uint64_t total_nknots,
                          const unsigned int* nknots_arr, const int16_t* param_no, uint64_t n_splines,
                          const float* coeff_tf1, const int16_t* param_no_tf1, uint64_t n_tf1,
                          const unsigned int* n_param_per_event, const unsigned int* n_param_per_event_tf1) try {
  if (h && h->expect_events == 0) h->expect_events = n_events;  // the one-call upload knows its size
  int rc = m3hip_monolith_begin(h, n_params, max_knots, coeff_x, npts);
  if (rc) return rc;
  rc = m3hip_monolith_append(h, n_events, coeff_many, total_nknots, nknots_arr, param_no, n_splines, coeff_tf1,
                             param_no_tf1, n_tf1, n_param_per_event, n_param_per_event_tf1);
  if (rc) return rc;
  return m3hip_monolith_end(h);
} M3_API_CATCH(h)

/* ---------------------------------------------------------------- seam 2 per step */

int m3hip_eval_splines(m3hip_handle h, const float* vals, const int16_t* segment, float* cpu_total_weights) try {
  if (!h) return M3HIP_EINVAL;
  if (!h->mono_done) return fail(h, M3HIP_ESTATE, "eval_splines before the monolith was uploaded");
  if (!vals || !segment) return fail(h, M3HIP_EINVAL, "eval_splines: NULL vals / segment");
  HIPCHK(h, hipSetDevice(h->device));
  for (int i = 0; i < h->P; ++i) {
    const int np = h->h_npts[i];
    if (np > 1 && (segment[i] < 0 || segment[i] > np - 2))
      return fail(h, M3HIP_EINVAL, "eval_splines: segment out of range for parameter " + std::to_string(i));
  }
  // Both staging buffers are reused every call: the previous copies must have drained.
  HIPCHK(h, hipStreamSynchronize(h->stream));
  collect_timing(h);
  memcpy(h->h_vals, vals, sizeof(float) * h->P);
  memcpy(h->h_segs, segment, sizeof(int16_t) * h->P);
  HIPCHK(h, hipMemcpyAsync(h->d_vals_in, h->h_vals, sizeof(float) * h->P, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->d_segs_in, h->h_segs, sizeof(int16_t) * h->P, hipMemcpyHostToDevice, h->stream));
  PrepArgs p;
  memset(&p, 0, sizeof(p));
  p.coeff_x = h->d_coeff_x;
  p.npts = h->d_npts;
  p.curr_seg = h->d_currseg;
  p.n_params = h->P;
  p.max_knots = h->K;
  p.ptab = h->d_ptab;
  p.pval = h->d_pval;
  p.seg_out = h->d_seg;
  p.vals_in = h->d_vals_in;
  p.seg_in = h->d_segs_in;
  p.u_param = h->d_u_param;
  p.u_plane0 = h->d_u_plane0;
  p.resolved = h->d_resolved;
  p.n_res = h->n_res <= kInlineResolveMax ? h->n_res : 0;
  launch_prep(p, h->stream);
  if (h->n_res > kInlineResolveMax) launch_resolve(p, h->n_res, h->stream);
  ReweightArgs a = make_rw_args(h);
  if (h->timing) HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  launch_reweight_uniform(a, uint32_t(h->h_tiles.size()), kModeEval, h->knobs, h->stream);
  launch_reweight_brick(a, uint32_t(h->h_bricks.size()), kModeEval, h->knobs, h->stream);
  if (h->timing) {
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    h->pending = 1;
  }
  HIPCHK(h, hipGetLastError());
  if (cpu_total_weights)  // gpuSplineUtils.cu:505 cudaMemcpyAsync(cpu_total_weights, gpu_total_weights, ...)
    HIPCHK(h, hipMemcpyAsync(cpu_total_weights, h->d_spline_w, sizeof(float) * h->E, hipMemcpyDeviceToHost, h->stream));
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_sync(m3hip_handle h) try {
  if (!h) return M3HIP_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, wait_stream(h->stream));
  collect_timing(h);
  return M3HIP_OK;
} M3_API_CATCH(h)

/* ---------------------------------------------------------------- seam 3 setup */

int m3hip_set_spline_param_map(m3hip_handle h, const int32_t* global_index, int n_params) try {
  if (!h) return M3HIP_EINVAL;
  if (!h->mono_open && !h->mono_done) return fail(h, M3HIP_ESTATE, "set_spline_param_map before monolith_begin");
  if (!global_index || n_params != h->P) return fail(h, M3HIP_EINVAL, "set_spline_param_map: n_params mismatch");
  int64_t mx = -1;
  for (int i = 0; i < n_params; ++i) {
    if (global_index[i] < 0) return fail(h, M3HIP_EINVAL, "set_spline_param_map: negative index");
    mx = std::max<int64_t>(mx, global_index[i]);
  }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(h->d_parmap, global_index, sizeof(int32_t) * n_params, hipMemcpyHostToDevice));
  h->parmap_set = true;
  h->max_parmap = mx;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_set_binning(m3hip_handle h, int n_samples, const int32_t* sample_bin_offset) try {
  if (!h) return M3HIP_EINVAL;
  if (n_samples <= 0 || !sample_bin_offset || sample_bin_offset[0] != 0) return fail(h, M3HIP_EINVAL, "set_binning: bad offsets");
  for (int s = 0; s < n_samples; ++s)
    if (sample_bin_offset[s + 1] < sample_bin_offset[s]) return fail(h, M3HIP_EINVAL, "set_binning: offsets must not decrease");
  const int nb = sample_bin_offset[n_samples];
  if (nb <= 0) return fail(h, M3HIP_EINVAL, "set_binning: no bins");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  dev_free(h->d_sample_off); dev_free(h->d_data); dev_free(h->d_hist); dev_free(h->d_llh_terms);
  h->d_sample_off = nullptr; h->d_data = nullptr; h->d_hist = nullptr; h->d_llh_terms = nullptr;
  // The replicas spread the adds of thousands of workgroups over more addresses than one bin vector has: 64
  // copies at 200 bins = 12 800 addresses (round 1 settled on 32; round 3's replica sweep on config 3: 1 / 8 / 32 / 64
  // replicas 850 / 744 / 727.5 / 722 us; config 2 -1 % kernel and step at 64 with the bin likelihood's fold 32 loads deep).  More bins spread the adds by
  // themselves, while the per-step reset (k_prep) and the fold (k_llh) are linear in replicas x bins (118 us of a
  // 132 us step at 7 200 bins with 32 replicas): keep the number of addresses, not the number of replicas.
  // Twice the replicas pay for samples of thousands of workgroups only (config 2: -1 % kernel and step); a small sample's step
  // is reset + fold (1e5 events: 28 -> 31 us with 64).  Allocated for the larger number, settled at the first reweight.
  if (!h->hist_rep_forced) {
    h->hist_rep = std::max(1, std::min(std::min(h->hist_rep_cfg, 32), 6400 / nb));
    h->hist_rep_big = std::max(h->hist_rep, std::min(h->hist_rep_cfg, 12800 / nb));
  } else {
    h->hist_rep_big = h->hist_rep;
  }
  h->hist_rep_locked = false;
  int rc;
  if ((rc = dev_alloc(h, &h->d_sample_off, size_t(n_samples) + 1))) return rc;
  if ((rc = dev_alloc(h, &h->d_data, size_t(nb)))) return rc;
  if ((rc = dev_alloc(h, &h->d_hist, size_t(nb) * 2 * h->hist_rep_big))) return rc;
  if (nb > M3_LLH_SPLIT_BINS && (rc = dev_alloc(h, &h->d_llh_terms, size_t(nb)))) return rc;
  HIPCHK(h, hipMemcpy(h->d_sample_off, sample_bin_offset, sizeof(int32_t) * (n_samples + 1), hipMemcpyHostToDevice));
  HIPCHK(h, hipMemset(h->d_data, 0, sizeof(double) * nb));
  HIPCHK(h, hipMemset(h->d_hist, 0, sizeof(double) * nb * 2 * h->hist_rep_big));
  if (h->h_out) { (void)hipHostFree(h->h_out); h->h_out = nullptr; }
  HIPCHK(h, hipHostMalloc((void**)&h->h_out, sizeof(double) * (2 + n_samples), hipHostMallocMapped));
  memset(h->h_out, 0, sizeof(double) * (2 + n_samples));
  HIPCHK(h, hipHostGetDevicePointer((void**)&h->d_out, h->h_out, 0));
  h->n_bins = nb;
  h->n_samples = n_samples;
  h->first_w2 = true;
  h->w2_spread = false;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_upload_events(m3hip_handle h, uint64_t event_offset, uint64_t n, const int32_t* bin_idx,
                        const int32_t* osc_idx, const uint64_t* norm_off, const int16_t* norm_idx) try {
  if (!h) return M3HIP_EINVAL;
  if (!h->mono_done) return fail(h, M3HIP_ESTATE, "upload_events before monolith_end");
  if (event_offset + n > h->E) return fail(h, M3HIP_EINVAL, "upload_events: range exceeds the monolith's events");
  if (n == 0) return M3HIP_OK;
  if (!bin_idx) return fail(h, M3HIP_EINVAL, "upload_events: bin_idx is NULL");
  HIPCHK(h, hipSetDevice(h->device));
  if (int rc_b = bins_from_word(h)) return rc_b;  // the other events' bins must be in d_bin before a slice of it is replaced
  HIPCHK(h, hipStreamSynchronize(h->stream));
  int64_t mx = -1;
  for (uint64_t i = 0; i < n; ++i) {
    if (bin_idx[i] < -1 || (h->n_bins > 0 && bin_idx[i] >= h->n_bins)) return fail(h, M3HIP_EINVAL, "upload_events: bin index out of range");
    mx = std::max<int64_t>(mx, bin_idx[i]);
  }
  h->max_bin_idx = std::max(h->max_bin_idx, mx);  // re-checked by m3hip_reweight: the bin space may be (re)defined later
  h->evw_dirty = true;
  HIPCHK(h, hipMemcpy(h->d_bin + event_offset, bin_idx, sizeof(int32_t) * n, hipMemcpyHostToDevice));
  if (osc_idx) {
    h->oscw_valid = false;
    h->osc_static_steps = 0;
    for (uint64_t i = 0; i < n; ++i) {
      // -1 = no oscillation weight; anything below would run into the packed sample word's neighbouring fields (k_pack_events)
      if (osc_idx[i] < -1) return fail(h, M3HIP_EINVAL, "upload_events: oscillation index below -1");
      h->max_osc_idx = std::max<int64_t>(h->max_osc_idx, osc_idx[i]);
    }
    HIPCHK(h, hipMemcpy(h->d_osc + event_offset, osc_idx, sizeof(int32_t) * n, hipMemcpyHostToDevice));
  }
  if (norm_off) {
    if (norm_off[n] > norm_off[0] && !norm_idx) return fail(h, M3HIP_EINVAL, "upload_events: norm_idx is NULL");
    uint32_t mx = 0;
    for (uint64_t i = 0; i < n; ++i) {
      const uint64_t c = norm_off[i + 1] - norm_off[i];
      if (c > 4096) return fail(h, M3HIP_EINVAL, "upload_events: implausible norm list length");
      if (c > mx) mx = uint32_t(c);
    }
    if (mx > h->M) {  // grow the ELL planes, keep what is there
      int16_t* nn = nullptr;
      int rc = dev_alloc(h, &nn, size_t(mx) * h->E_pad);
      if (rc) return rc;
      HIPCHK(h, hipMemset(nn, 0xFF, sizeof(int16_t) * size_t(mx) * h->E_pad));  // -1: empty slot
      if (h->M) HIPCHK(h, hipMemcpy(nn, h->d_nidx, sizeof(int16_t) * size_t(h->M) * h->E_pad, hipMemcpyDeviceToDevice));
      dev_free(h->d_nidx);
      h->d_nidx = nn;
      h->M = mx;
    }
    if (h->M) {
      std::vector<int16_t> plane(n);
      for (uint32_t m = 0; m < h->M; ++m) {
        for (uint64_t i = 0; i < n; ++i) {
          const uint64_t c = norm_off[i + 1] - norm_off[i];
          plane[i] = (m < c) ? norm_idx[norm_off[i] + m] : int16_t(-1);
          if (m < c && plane[i] < 0) return fail(h, M3HIP_EINVAL, "upload_events: negative norm parameter index");
          h->max_norm_idx = std::max<int64_t>(h->max_norm_idx, plane[i]);
          if (m < c && m < uint32_t(M3_EVW_NORM_SLOTS)) {
            h->min_norm_idx = std::min<int64_t>(h->min_norm_idx, plane[i]);
            h->max_norm_idx_evw = std::max<int64_t>(h->max_norm_idx_evw, plane[i]);
          }
        }
        HIPCHK(h, hipMemcpy(h->d_nidx + size_t(m) * h->E_pad + event_offset, plane.data(), sizeof(int16_t) * n, hipMemcpyHostToDevice));
      }
    }
  }
  return M3HIP_OK;
} M3_API_CATCH(h)

// shared part of m3hip_binned_upload / _f64: validation, the empty event space, the index tables
static int binned_upload_common(m3hip_handle h, uint64_t n_events, int n_params, int max_knots, const float* xpts, const int16_t* npts,
                                int monolith_size, const int32_t* coeffindexvec, const int32_t* uniquesplinevec, int n_unique,
                                const int32_t* uniquecoeffindices, int n_coeff, const void* xcoeff, const void* manycoeff) {
  if (n_events == 0 || monolith_size <= 0 || n_unique < 0 || n_coeff < 0 || !coeffindexvec || !uniquesplinevec ||
      (n_unique && (!uniquecoeffindices || !xcoeff || !manycoeff)))
    return fail(h, M3HIP_EINVAL, "binned_upload: empty / NULL arguments");
  for (int u = 0; u < n_unique; ++u) {
    const int i = uniquecoeffindices[u];
    if (i < 0 || i >= monolith_size) return fail(h, M3HIP_EINVAL, "binned_upload: uniquecoeffindices entry out of range");
    const int p = uniquesplinevec[i];
    if (p < 0 || p >= n_params) return fail(h, M3HIP_EINVAL, "binned_upload: unique spline parameter out of range");
    const int np = npts[p];
    if (np < 2 || coeffindexvec[i] < 0 || coeffindexvec[i] + np > n_coeff)
      return fail(h, M3HIP_EINVAL, "binned_upload: bin-spline " + std::to_string(i) + " exceeds the coefficient arrays");
  }
  // an event space without event-by-event responses: every event is an empty uniform slot list
  int rc = m3hip_monolith_begin(h, n_params, max_knots, xpts, npts);
  if (rc) return rc;
  const uint64_t step = 1u << 20;
  std::vector<unsigned int> zeros(2 * size_t(std::min<uint64_t>(n_events, step)), 0u);
  for (uint64_t e0 = 0; e0 < n_events; e0 += step) {
    const uint64_t n = std::min<uint64_t>(step, n_events - e0);
    rc = m3hip_monolith_append(h, n, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, 0, zeros.data(), zeros.data());
    if (rc) return rc;
  }
  if ((rc = m3hip_monolith_end(h))) return rc;
  if ((rc = dev_alloc(h, &h->d_b_cidx, size_t(monolith_size)))) return rc;
  if ((rc = dev_alloc(h, &h->d_b_usv, size_t(monolith_size)))) return rc;
  if ((rc = dev_alloc(h, &h->d_b_uniq, size_t(n_unique)))) return rc;
  HIPCHK(h, hipMemcpy(h->d_b_cidx, coeffindexvec, sizeof(int32_t) * monolith_size, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_b_usv, uniquesplinevec, sizeof(int32_t) * monolith_size, hipMemcpyHostToDevice));
  if (n_unique) HIPCHK(h, hipMemcpy(h->d_b_uniq, uniquecoeffindices, sizeof(int32_t) * n_unique, hipMemcpyHostToDevice));
  h->b_M = monolith_size;
  h->b_U = n_unique;
  h->b_C = n_coeff;
  return M3HIP_OK;
}

int m3hip_binned_upload(m3hip_handle h, uint64_t n_events, int n_params, int max_knots, const float* xpts, const int16_t* npts,
                        int monolith_size, const int32_t* coeffindexvec, const int32_t* uniquesplinevec, int n_unique,
                        const int32_t* uniquecoeffindices, int n_coeff, const float* xcoeff, const float* manycoeff) try {
  if (!h) return M3HIP_EINVAL;
  int rc = binned_upload_common(h, n_events, n_params, max_knots, xpts, npts, monolith_size, coeffindexvec, uniquesplinevec, n_unique,
                                uniquecoeffindices, n_coeff, xcoeff, manycoeff);
  if (rc) return rc;
  if ((rc = dev_alloc(h, &h->d_b_xcoeff, size_t(n_coeff)))) return rc;
  if ((rc = dev_alloc(h, &h->d_b_many, size_t(n_coeff)))) return rc;
  if ((rc = dev_alloc(h, &h->d_b_w, size_t(monolith_size)))) return rc;
  if (n_coeff) {
    HIPCHK(h, hipMemcpy(h->d_b_xcoeff, xcoeff, sizeof(float) * n_coeff, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_b_many, manycoeff, sizeof(float) * 4 * size_t(n_coeff), hipMemcpyHostToDevice));
  }
  std::vector<float> ones(size_t(monolith_size), 1.0f);  // flat bin-splines keep weight 1 (BinnedSplineHandler.cpp:236)
  HIPCHK(h, hipMemcpy(h->d_b_w, ones.data(), sizeof(float) * monolith_size, hipMemcpyHostToDevice));
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_binned_upload_f64(m3hip_handle h, uint64_t n_events, int n_params, int max_knots, const double* xpts, const int16_t* npts,
                            int monolith_size, const int32_t* coeffindexvec, const int32_t* uniquesplinevec, int n_unique,
                            const int32_t* uniquecoeffindices, int n_coeff, const double* xcoeff, const double* manycoeff) try {
  if (!h) return M3HIP_EINVAL;
  if (!xpts || !npts || n_params <= 0 || max_knots <= 0) return fail(h, M3HIP_EINVAL, "binned_upload_f64: NULL / empty knot table");
  // the float knot table only feeds dx of event-by-event splines, of which this build has none
  std::vector<float> xf(size_t(n_params) * max_knots);
  for (size_t i = 0; i < xf.size(); ++i) xf[i] = float(xpts[i]);
  int rc = binned_upload_common(h, n_events, n_params, max_knots, xf.data(), npts, monolith_size, coeffindexvec, uniquesplinevec, n_unique,
                                uniquecoeffindices, n_coeff, xcoeff, manycoeff);
  if (rc) return rc;
  if ((rc = dev_alloc(h, &h->d_coeff_x_d, xf.size()))) return rc;
  if ((rc = dev_alloc(h, &h->d_b_xcoeff_d, size_t(n_coeff)))) return rc;
  if ((rc = dev_alloc(h, &h->d_b_many_d, 4 * size_t(n_coeff)))) return rc;
  if ((rc = dev_alloc(h, &h->d_b_w_d, size_t(monolith_size)))) return rc;
  HIPCHK(h, hipMemcpy(h->d_coeff_x_d, xpts, sizeof(double) * xf.size(), hipMemcpyHostToDevice));
  if (n_coeff) {
    HIPCHK(h, hipMemcpy(h->d_b_xcoeff_d, xcoeff, sizeof(double) * n_coeff, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_b_many_d, manycoeff, sizeof(double) * 4 * size_t(n_coeff), hipMemcpyHostToDevice));
  }
  std::vector<double> ones(size_t(monolith_size), 1.0);
  HIPCHK(h, hipMemcpy(h->d_b_w_d, ones.data(), sizeof(double) * monolith_size, hipMemcpyHostToDevice));
  h->f64 = true;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_binned_upload_events(m3hip_handle h, uint64_t event_offset, uint64_t n, const uint64_t* w_off, const int32_t* w_idx) try {
  if (!h) return M3HIP_EINVAL;
  if (h->b_M == 0) return fail(h, M3HIP_ESTATE, "binned_upload_events before binned_upload");
  if (event_offset + n > h->E || !w_off) return fail(h, M3HIP_EINVAL, "binned_upload_events: bad range / NULL");
  if (n == 0) return M3HIP_OK;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, wait_stream(h->stream));
  uint32_t mx = 0;
  for (uint64_t i = 0; i < n; ++i) {
    const uint64_t c = w_off[i + 1] - w_off[i];
    if (c > 1024) return fail(h, M3HIP_EINVAL, "binned_upload_events: implausible number of bin-splines for one event");
    if (c > mx) mx = uint32_t(c);
    for (uint64_t k = w_off[i]; k < w_off[i + 1]; ++k)
      if (w_idx[k] < 0 || w_idx[k] >= h->b_M) return fail(h, M3HIP_EINVAL, "binned_upload_events: weight index out of range");
  }
  if (h->bl_mode < 0) {
    // Events of one spline bin carry the same pointer list: when there are few distinct lists, each is stored once
    // and an event keeps a list id (4 B instead of 4 B per weight); the events are then PROCESSED grouped by list
    // (a permutation of the lanes, see ReweightArgs::perm), so that a wave gathers the same weights whatever order
    // the MC was registered in.  With (nearly) one list per event the ELL form stays.  M3HIP_BINNED_LISTS=ell|shared
    // overrides the choice.
    // (decided on the first upload: when its events carry at most one distinct list per 8 events)
    std::map<std::vector<int32_t>, int> seen;
    std::vector<int32_t> key;
    for (uint64_t i = 0; i < n && seen.size() * 8 <= n; ++i) {
      key.assign(w_idx + w_off[i], w_idx + w_off[i + 1]);
      seen.emplace(key, 0);
    }
    h->bl_mode = (n > 1 && seen.size() * 8 <= n) ? 1 : 0;
    if (const char* f = getenv("M3HIP_BINNED_LISTS")) h->bl_mode = (f[0] == 's') ? 1 : 0;
  }
  if (h->bl_mode == 1) {
    if (!h->d_bl_id) {
      int rc = dev_alloc(h, &h->d_bl_id, size_t(h->E_pad));
      if (rc) return rc;
      HIPCHK(h, hipMemset(h->d_bl_id, 0xFF, sizeof(int32_t) * h->E_pad));
      h->bl_off.assign(1, 0);
    }
    std::vector<int32_t> ids(n);
    std::vector<int32_t> key;
    for (uint64_t i = 0; i < n; ++i) {
      key.assign(w_idx + w_off[i], w_idx + w_off[i + 1]);
      if (key.empty()) { ids[i] = -1; continue; }
      auto it = h->bl_map.find(key);
      if (it == h->bl_map.end()) {
        it = h->bl_map.emplace(key, int32_t(h->bl_off.size()) - 1).first;
        h->bl_idx.insert(h->bl_idx.end(), key.begin(), key.end());
        h->bl_off.push_back(int32_t(h->bl_idx.size()));
      }
      ids[i] = it->second;
    }
    dev_free(h->d_bl_off);
    dev_free(h->d_bl_idx);
    h->d_bl_off = h->d_bl_idx = nullptr;
    int rc;
    if ((rc = dev_alloc(h, &h->d_bl_off, h->bl_off.size()))) return rc;
    if ((rc = dev_alloc(h, &h->d_bl_idx, h->bl_idx.size()))) return rc;
    HIPCHK(h, hipMemcpy(h->d_bl_off, h->bl_off.data(), sizeof(int32_t) * h->bl_off.size(), hipMemcpyHostToDevice));
    if (!h->bl_idx.empty()) HIPCHK(h, hipMemcpy(h->d_bl_idx, h->bl_idx.data(), sizeof(int32_t) * h->bl_idx.size(), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_bl_id + event_offset, ids.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
    if (h->h_bl_id.size() != h->E_pad) h->h_bl_id.assign(size_t(h->E_pad), -1);
    std::copy(ids.begin(), ids.end(), h->h_bl_id.begin() + event_offset);
    h->perm_dirty = true;
    return M3HIP_OK;
  }
  if (mx > h->Kb) {
    int32_t* nn = nullptr;
    int rc = dev_alloc(h, &nn, size_t(mx) * h->E_pad);
    if (rc) return rc;
    HIPCHK(h, hipMemset(nn, 0xFF, sizeof(int32_t) * size_t(mx) * h->E_pad));
    if (h->Kb) HIPCHK(h, hipMemcpy(nn, h->d_bwidx, sizeof(int32_t) * size_t(h->Kb) * h->E_pad, hipMemcpyDeviceToDevice));
    dev_free(h->d_bwidx);
    h->d_bwidx = nn;
    h->Kb = mx;
  }
  std::vector<int32_t> plane(n);
  for (uint32_t k = 0; k < h->Kb; ++k) {
    for (uint64_t i = 0; i < n; ++i) plane[i] = (k < w_off[i + 1] - w_off[i]) ? w_idx[w_off[i] + k] : -1;
    HIPCHK(h, hipMemcpy(h->d_bwidx + size_t(k) * h->E_pad + event_offset, plane.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
  }
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_get_binned_weights_f64(m3hip_handle h, double* out, int monolith_size) try {
  if (!h || !out) return M3HIP_EINVAL;
  if (monolith_size != h->b_M || h->b_M == 0) return fail(h, M3HIP_EINVAL, "get_binned_weights_f64: size mismatch");
  if (!h->f64) return fail(h, M3HIP_ESTATE, "get_binned_weights_f64: the bin-splines were uploaded as float");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, wait_stream(h->stream));
  HIPCHK(h, hipMemcpy(out, h->d_b_w_d, sizeof(double) * monolith_size, hipMemcpyDeviceToHost));
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_get_binned_weights(m3hip_handle h, float* out, int monolith_size) try {
  if (!h || !out) return M3HIP_EINVAL;
  if (monolith_size != h->b_M || h->b_M == 0) return fail(h, M3HIP_EINVAL, "get_binned_weights: size mismatch");
  if (h->f64) return fail(h, M3HIP_ESTATE, "get_binned_weights: the bin-splines were uploaded as double, use _f64");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, wait_stream(h->stream));
  HIPCHK(h, hipMemcpy(out, h->d_b_w, sizeof(float) * monolith_size, hipMemcpyDeviceToHost));
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_binning_begin(m3hip_handle h) try {
  if (!h) return M3HIP_EINVAL;
  h->bb_samples.clear(); h->bb_axes.clear(); h->bb_edges.clear(); h->bb_boxes.clear();
  h->bb_grid_off.clear(); h->bb_grid_list.clear(); h->bb_nbins.clear();
  h->binning_open = true;
  return M3HIP_OK;
} M3_API_CATCH(h)

// one axis of a sample: edges appended to the shared table, stride = product of the earlier axes' bins
// (SampleBinningInfo::InitialiseStrides, SampleStructs.h:648-659)
static void push_axis(m3hip_handle h, const double* e, int ne, int& stride) {
  BinAxis a;
  a.edge_off = int32_t(h->bb_edges.size());
  a.n_bins = ne - 1;
  a.stride = stride;
  a.pad_ = 0;
  h->bb_axes.push_back(a);
  h->bb_edges.insert(h->bb_edges.end(), e, e + ne);
  stride *= ne - 1;
}

int m3hip_binning_add_uniform(m3hip_handle h, int n_dim, const int32_t* n_edges, const double* edges) try {
  if (!h) return M3HIP_EINVAL;
  if (!h->binning_open) return fail(h, M3HIP_ESTATE, "binning_add_uniform outside binning_begin/end");
  if (!n_edges || !edges) return fail(h, M3HIP_EINVAL, "binning_add_uniform: NULL arguments");
  if (n_dim < 1 || n_dim > M3_MAX_DIM) return fail(h, M3HIP_EINVAL, "binning_add_uniform: 1..4 dimensions per sample");
  int e_off = 0;
  for (int d = 0; d < n_dim; ++d) {
    const int ne = n_edges[d];
    if (ne < 2) return fail(h, M3HIP_EINVAL, "binning_add_uniform: an axis needs at least two edges");
    for (int i = 1; i < ne; ++i)
      if (edges[e_off + i] < edges[e_off + i - 1]) return fail(h, M3HIP_EINVAL, "binning_add_uniform: bin edges must be increasing");  // SampleStructs.h:259-263
    e_off += ne;
  }
  BinSample bs;
  bs.n_dim = n_dim;
  bs.axis_off = int32_t(h->bb_axes.size());
  bs.global_offset = 0;
  bs.uniform = 1;
  bs.map_off = bs.box_off = 0;
  int stride = 1;
  e_off = 0;
  for (int d = 0; d < n_dim; ++d) {
    push_axis(h, edges + e_off, n_edges[d], stride);
    e_off += n_edges[d];
  }
  h->bb_samples.push_back(bs);
  h->bb_nbins.push_back(stride);
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_binning_add_nonuniform(m3hip_handle h, int n_boxes, int n_dim, const double* extents) try {
  if (!h) return M3HIP_EINVAL;
  if (!h->binning_open) return fail(h, M3HIP_ESTATE, "binning_add_nonuniform outside binning_begin/end");
  if (!extents) return fail(h, M3HIP_EINVAL, "binning_add_nonuniform: NULL arguments");
  if (n_dim <= 1 || n_dim > M3_MAX_DIM || n_boxes <= 0)  // SampleStructs.h:471-474: 1-D non-uniform binning is refused
    return fail(h, M3HIP_EINVAL, "binning_add_nonuniform: needs 2..4 dimensions and at least one bin");
  for (int64_t i = 0; i < int64_t(n_boxes) * n_dim; ++i)
    if (!(extents[2 * i] < extents[2 * i + 1])) return fail(h, M3HIP_EINVAL, "binning_add_nonuniform: a bin extent needs lo < hi");
  BinSample bs;
  bs.n_dim = n_dim;
  bs.axis_off = int32_t(h->bb_axes.size());
  bs.global_offset = 0;
  bs.uniform = 0;
  bs.map_off = int32_t(h->bb_grid_off.size());
  bs.box_off = int32_t(h->bb_boxes.size() / (2 * size_t(n_dim)));
  if (h->bb_boxes.size() % (2 * size_t(n_dim)) != 0) {  // keep box_off exact when samples of different dimension mix
    h->bb_boxes.resize((h->bb_boxes.size() / (2 * size_t(n_dim)) + 1) * 2 * size_t(n_dim), 0.0);
    bs.box_off = int32_t(h->bb_boxes.size() / (2 * size_t(n_dim)));
  }
  // the mega-bin grid: 10 equal bins per axis over the boxes' bounding range (SampleStructs.h:499-517)
  constexpr int BinsPerDimension = 10;
  int stride = 1;
  std::vector<std::vector<double>> grid(static_cast<size_t>(n_dim), std::vector<double>(BinsPerDimension + 1));
  for (int d = 0; d < n_dim; ++d) {
    double MinVal = std::numeric_limits<double>::max(), MaxVal = std::numeric_limits<double>::lowest();
    for (int b = 0; b < n_boxes; ++b) {
      MinVal = std::min(MinVal, extents[(size_t(b) * n_dim + d) * 2]);
      MaxVal = std::max(MaxVal, extents[(size_t(b) * n_dim + d) * 2 + 1]);
    }
    const double BinWidth = (MaxVal - MinVal) / double(BinsPerDimension);
    for (int i = 0; i <= BinsPerDimension; ++i) grid[size_t(d)][size_t(i)] = MinVal + double(i) * BinWidth;
    push_axis(h, grid[size_t(d)].data(), BinsPerDimension + 1, stride);
  }
  // BinGridMapping (SampleStructs.h:397-462): boxes overlapping each mega-bin, in box order
  const int n_cells = stride;
  std::vector<int> mi(static_cast<size_t>(n_dim));
  for (int cell = 0; cell < n_cells; ++cell) {
    int rem = cell;
    for (int d = 0; d < n_dim; ++d) { mi[size_t(d)] = rem % BinsPerDimension; rem /= BinsPerDimension; }  // first axis fastest
    h->bb_grid_off.push_back(int32_t(h->bb_grid_list.size()));
    for (int b = 0; b < n_boxes; ++b) {
      bool overlap = true;
      for (int d = 0; d < n_dim && overlap; ++d) {
        const double a_lo = extents[(size_t(b) * n_dim + d) * 2], a_hi = extents[(size_t(b) * n_dim + d) * 2 + 1];
        const double b_lo = grid[size_t(d)][size_t(mi[size_t(d)])], b_hi = grid[size_t(d)][size_t(mi[size_t(d)]) + 1];
        if (!(a_hi > b_lo && a_lo < b_hi)) overlap = false;
      }
      if (overlap) h->bb_grid_list.push_back(b);
    }
  }
  h->bb_grid_off.push_back(int32_t(h->bb_grid_list.size()));
  h->bb_boxes.insert(h->bb_boxes.end(), extents, extents + size_t(n_boxes) * n_dim * 2);
  h->bb_samples.push_back(bs);
  h->bb_nbins.push_back(n_boxes);
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_binning_end(m3hip_handle h) try {
  if (!h) return M3HIP_EINVAL;
  if (!h->binning_open) return fail(h, M3HIP_ESTATE, "binning_end without binning_begin");
  const int n_samples = int(h->bb_samples.size());
  if (n_samples == 0) return fail(h, M3HIP_EINVAL, "binning_end: no samples added");
  h->binning_open = false;
  std::vector<int32_t> off(size_t(n_samples) + 1, 0);  // BinningHandler::SetGlobalBinNumbers (BinningHandler.cpp:341-355)
  for (int s = 0; s < n_samples; ++s) {
    h->bb_samples[size_t(s)].global_offset = off[size_t(s)];
    off[size_t(s) + 1] = off[size_t(s)] + h->bb_nbins[size_t(s)];
  }
  int rc = m3hip_set_binning(h, n_samples, off.data());
  if (rc) return rc;
  dev_free(h->d_bin_samples); dev_free(h->d_bin_axes); dev_free(h->d_edges);
  dev_free(h->d_grid_off); dev_free(h->d_grid_list); dev_free(h->d_boxes);
  h->d_bin_samples = nullptr; h->d_bin_axes = nullptr; h->d_edges = nullptr;
  h->d_grid_off = nullptr; h->d_grid_list = nullptr; h->d_boxes = nullptr;
  auto up = [&](auto** dptr, const auto& v) -> int {
    if (v.empty()) return M3HIP_OK;
    int r = dev_alloc(h, dptr, v.size());
    if (r) return r;
    if (hipMemcpy(*dptr, v.data(), sizeof(v[0]) * v.size(), hipMemcpyHostToDevice) != hipSuccess) return fail(h, M3HIP_EHIP, "binning_end: upload failed");
    return M3HIP_OK;
  };
  if ((rc = up(&h->d_bin_samples, h->bb_samples))) return rc;
  if ((rc = up(&h->d_bin_axes, h->bb_axes))) return rc;
  if ((rc = up(&h->d_edges, h->bb_edges))) return rc;
  if ((rc = up(&h->d_grid_off, h->bb_grid_off))) return rc;
  if ((rc = up(&h->d_grid_list, h->bb_grid_list))) return rc;
  if ((rc = up(&h->d_boxes, h->bb_boxes))) return rc;
  h->h_sample_ndim.clear();
  for (const BinSample& s : h->bb_samples) h->h_sample_ndim.push_back(s.n_dim);
  h->bins_computed = false;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_set_binning_edges(m3hip_handle h, int n_samples, const int32_t* n_dim, const int32_t* n_edges, const double* edges) try {
  if (!h) return M3HIP_EINVAL;
  if (n_samples <= 0 || !n_dim || !n_edges || !edges) return fail(h, M3HIP_EINVAL, "set_binning_edges: NULL / empty arguments");
  int rc = m3hip_binning_begin(h);
  int k = 0;
  size_t e_off = 0;
  for (int s = 0; s < n_samples && !rc; ++s) {
    if (n_dim[s] < 1 || n_dim[s] > M3_MAX_DIM) { rc = fail(h, M3HIP_EINVAL, "set_binning_edges: 1..4 dimensions per sample"); break; }
    rc = m3hip_binning_add_uniform(h, n_dim[s], n_edges + k, edges + e_off);
    for (int d = 0; d < n_dim[s]; ++d) e_off += size_t(std::max(n_edges[k + d], 0));
    k += n_dim[s];
  }
  if (rc) { h->binning_open = false; return rc; }
  return m3hip_binning_end(h);
} M3_API_CATCH(h)

int m3hip_upload_kinematics(m3hip_handle h, uint64_t event_offset, uint64_t n, const int32_t* sample_id, const double* kin, int kin_stride) try {
  if (!h) return M3HIP_EINVAL;
  if (!h->mono_done) return fail(h, M3HIP_ESTATE, "upload_kinematics before monolith_end");
  if (h->h_sample_ndim.empty()) return fail(h, M3HIP_ESTATE, "upload_kinematics before set_binning_edges");
  if (event_offset + n > h->E) return fail(h, M3HIP_EINVAL, "upload_kinematics: range exceeds the monolith's events");
  if (kin_stride < 1 || kin_stride > M3_MAX_DIM || (h->kin_stride && kin_stride != h->kin_stride)) return fail(h, M3HIP_EINVAL, "upload_kinematics: bad kin_stride");
  if (n == 0) return M3HIP_OK;
  if (!kin) return fail(h, M3HIP_EINVAL, "upload_kinematics: kin is NULL");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, wait_stream(h->stream));
  int rc;
  if (!h->d_kin) {
    if ((rc = dev_alloc(h, &h->d_sample_id, size_t(h->E_pad)))) return rc;
    if ((rc = dev_alloc(h, &h->d_kin, size_t(h->E_pad) * kin_stride))) return rc;
    if ((rc = dev_alloc(h, &h->d_nom_bin, size_t(h->E_pad) * kin_stride))) return rc;
    HIPCHK(h, hipMemset(h->d_sample_id, 0, sizeof(int32_t) * h->E_pad));
    HIPCHK(h, hipMemset(h->d_kin, 0, sizeof(double) * h->E_pad * kin_stride));
    HIPCHK(h, hipMemset(h->d_nom_bin, 0xFF, sizeof(int32_t) * h->E_pad * kin_stride));
    h->kin_stride = kin_stride;
  }
  if (sample_id) {
    for (uint64_t i = 0; i < n; ++i) {
      if (sample_id[i] < 0 || sample_id[i] >= int(h->h_sample_ndim.size())) return fail(h, M3HIP_EINVAL, "upload_kinematics: sample id out of range");
      if (h->h_sample_ndim[sample_id[i]] > kin_stride) return fail(h, M3HIP_EINVAL, "upload_kinematics: kin_stride smaller than the sample's dimension");
    }
    HIPCHK(h, hipMemcpy(h->d_sample_id + event_offset, sample_id, sizeof(int32_t) * n, hipMemcpyHostToDevice));
  }
  HIPCHK(h, hipMemcpy(h->d_kin + event_offset * kin_stride, kin, sizeof(double) * n * kin_stride, hipMemcpyHostToDevice));
  h->bins_computed = false;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_compute_bins(m3hip_handle h) try {
  if (!h) return M3HIP_EINVAL;
  if (!h->d_kin || !h->d_bin_samples) return fail(h, M3HIP_ESTATE, "compute_bins needs set_binning_edges and upload_kinematics");
  for (const KinCut& c : h->h_cuts)
    if (c.column >= h->kin_stride) return fail(h, M3HIP_EINVAL, "compute_bins: a selection cut names a column beyond kin_stride");
  HIPCHK(h, hipSetDevice(h->device));
  BinArgs b = make_bin_args(h);
  b.compute_nominal = 1;
  b.n_shifts = 0;
  launch_find_bins(b, h->stream);
  h->evw_dirty = true;  // d_bin is rewritten
  h->bins_in_word = false;
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, wait_stream(h->stream));
  h->bins_computed = true;
  h->max_bin_idx = -1;  // every event's bin now comes from the device-side search over the current bin space
  return M3HIP_OK;
} M3_API_CATCH(h)

static int add_shift(m3hip_handle h, int sample, int dim, int kind, int aux, int param_index) {
  if (dim < 0 || dim >= M3_MAX_DIM || kind < 1 || kind > 3 || aux < 0 || aux >= M3_MAX_DIM || param_index < 0 || sample < -1)
    return fail(h, M3HIP_EINVAL, "add_kin_shift: bad arguments");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, wait_stream(h->stream));
  KinShift k;
  k.sample = sample; k.dim = dim; k.kind = kind | (aux << 8); k.param = param_index;
  h->h_shifts.push_back(k);
  dev_free(h->d_shifts);
  h->d_shifts = nullptr;
  int rc = dev_alloc(h, &h->d_shifts, h->h_shifts.size());
  if (rc) return rc;
  HIPCHK(h, hipMemcpy(h->d_shifts, h->h_shifts.data(), sizeof(KinShift) * h->h_shifts.size(), hipMemcpyHostToDevice));
  return M3HIP_OK;
}

int m3hip_add_kin_shift(m3hip_handle h, int sample, int dim, int kind, int param_index) try {
  if (!h) return M3HIP_EINVAL;
  if (kind != 1 && kind != 2) return fail(h, M3HIP_EINVAL, "add_kin_shift: kind must be 1 (scale) or 2 (offset)");
  return add_shift(h, sample, dim, kind, 0, param_index);
} M3_API_CATCH(h)

int m3hip_add_kin_shift_aux(m3hip_handle h, int sample, int dim, int aux_column, int param_index) try {
  if (!h) return M3HIP_EINVAL;
  return add_shift(h, sample, dim, 3, aux_column, param_index);
} M3_API_CATCH(h)

int m3hip_add_kin_cut(m3hip_handle h, int sample, int column, double lower, double upper) try {
  if (!h) return M3HIP_EINVAL;
  if (column < 0 || column >= M3_MAX_DIM || sample < -1 || !(lower <= upper)) return fail(h, M3HIP_EINVAL, "add_kin_cut: bad arguments");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, wait_stream(h->stream));
  KinCut c;
  c.sample = sample; c.column = column; c.lo = lower; c.hi = upper;
  h->h_cuts.push_back(c);
  dev_free(h->d_cuts);
  h->d_cuts = nullptr;
  int rc = dev_alloc(h, &h->d_cuts, h->h_cuts.size());
  if (rc) return rc;
  HIPCHK(h, hipMemcpy(h->d_cuts, h->h_cuts.data(), sizeof(KinCut) * h->h_cuts.size(), hipMemcpyHostToDevice));
  if (h->bins_computed) return m3hip_compute_bins(h);  // the selection changes which events fill
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_clear_kin_shifts(m3hip_handle h) try {
  if (!h) return M3HIP_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, wait_stream(h->stream));
  h->h_shifts.clear();
  if (h->bins_computed) return m3hip_compute_bins(h);  // back to the nominal bins
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_get_event_bins(m3hip_handle h, int32_t* out, uint64_t event_offset, uint64_t n) try {
  if (!h || !out) return M3HIP_EINVAL;
  if (!h->mono_done || event_offset + n > h->E) return fail(h, M3HIP_EINVAL, "get_event_bins: bad range");
  HIPCHK(h, hipSetDevice(h->device));
  if (int rc_b = bins_from_word(h)) return rc_b;
  HIPCHK(h, wait_stream(h->stream));
  HIPCHK(h, hipMemcpy(out, h->d_bin + event_offset, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
  return M3HIP_OK;
} M3_API_CATCH(h)

static int osc_table_resize(m3hip_handle h, uint64_t n) {
  if (n == h->n_osc) return M3HIP_OK;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  dev_free(h->d_osc_table);
  dev_free(h->d_osc_table_d);
  h->d_osc_table = nullptr;
  h->d_osc_table_d = nullptr;
  for (int i = 0; i < 2; ++i) {
    if (h->h_osc_stage[i]) (void)hipHostFree(h->h_osc_stage[i]);
    h->h_osc_stage[i] = nullptr;
    h->osc_stage_ticket[i] = 0;
  }
  h->n_osc = 0;
  h->osc_valid_f32 = h->osc_valid_f64 = false;
  int rc;
  if ((rc = dev_alloc(h, &h->d_osc_table, size_t(n)))) return rc;
  if ((rc = dev_alloc(h, &h->d_osc_table_d, size_t(n)))) return rc;
  for (int i = 0; i < 2; ++i) HIPCHK(h, hipHostMalloc(&h->h_osc_stage[i], sizeof(double) * n, hipHostMallocDefault));
  h->n_osc = n;
  return M3HIP_OK;
}

// Per-step upload (OscillationHandler::Evaluate runs inside every Reweight, SampleHandlerFD.cpp:319-324): the
// table is copied into a pinned staging slot and goes to the device with an asynchronous copy on the engine's
// stream, no synchronisation.  Only the precision handed over is uploaded; the other one is derived on the
// device when a reweight needs it.
static int osc_upload(m3hip_handle h, const void* table, uint64_t n, size_t elem, void* dst) {
  const int slot = h->osc_stage_pos;
  h->osc_stage_pos ^= 1;
  if (h->osc_stage_ticket[slot] > h->ticket_done) {  // the copy that last used this slot may still be in flight
    HIPCHK(h, wait_stream(h->stream));
    h->ticket_done = h->ticket;
  }
  memcpy(h->h_osc_stage[slot], table, elem * n);
  HIPCHK(h, hipMemcpyAsync(dst, h->h_osc_stage[slot], elem * n, hipMemcpyHostToDevice, h->stream));
  h->osc_stage_ticket[slot] = h->ticket + 1;  // completes no later than the next step's kernels
  return M3HIP_OK;
}

int m3hip_set_osc_table(m3hip_handle h, const float* table, uint64_t n) try {
  if (!h) return M3HIP_EINVAL;
  if (!table || n == 0) return fail(h, M3HIP_EINVAL, "set_osc_table: empty table");
  HIPCHK(h, hipSetDevice(h->device));
  int rc = osc_table_resize(h, n);
  if (rc) return rc;
  if ((rc = osc_upload(h, table, n, sizeof(float), h->d_osc_table))) return rc;
  h->osc_valid_f32 = true;
  h->osc_valid_f64 = false;
  h->oscw_valid = false;
  h->osc_static_steps = 0;
  if (h->evw_weight) h->evw_dirty = true;  // the packed words carry weights of the table that was just replaced
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_set_osc_table_f64(m3hip_handle h, const double* table, uint64_t n) try {
  if (!h) return M3HIP_EINVAL;
  if (!table || n == 0) return fail(h, M3HIP_EINVAL, "set_osc_table_f64: empty table");
  HIPCHK(h, hipSetDevice(h->device));
  int rc = osc_table_resize(h, n);
  if (rc) return rc;
  if ((rc = osc_upload(h, table, n, sizeof(double), h->d_osc_table_d))) return rc;
  h->osc_valid_f64 = true;
  h->osc_valid_f32 = false;
  h->oscw_valid = false;
  h->osc_static_steps = 0;
  if (h->evw_weight) h->evw_dirty = true;
  return M3HIP_OK;
} M3_API_CATCH(h)

/* ---------------------------------------------------------------- experiment weights behind the splines */

static int extra_slot_check(m3hip_handle h, int slot, uint64_t event_offset, uint64_t n, const char* who) {
  if (!h->mono_done) return fail(h, M3HIP_ESTATE, std::string(who) + " before monolith_end");
  if (slot < 0 || slot >= M3_MAX_EXTRA) return fail(h, M3HIP_EINVAL, std::string(who) + ": slot out of range (0.." + std::to_string(M3_MAX_EXTRA - 1) + ")");
  if (event_offset + n > h->E) return fail(h, M3HIP_EINVAL, std::string(who) + ": range exceeds the monolith's events");
  return M3HIP_OK;
}

int m3hip_upload_event_weights(m3hip_handle h, int slot, uint64_t event_offset, uint64_t n, const float* w) try {
  if (!h) return M3HIP_EINVAL;
  int rc = extra_slot_check(h, slot, event_offset, n, "upload_event_weights");
  if (rc) return rc;
  if (n == 0) return M3HIP_OK;
  if (!w) return fail(h, M3HIP_EINVAL, "upload_event_weights: w is NULL");
  auto& x = h->extra[slot];
  if (x.d_idx) return fail(h, M3HIP_ESTATE, "upload_event_weights: the slot already holds table indices");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (!x.d_col) {
    if ((rc = dev_alloc(h, &x.d_col, size_t(h->E_pad)))) return rc;
    std::vector<float> ones(size_t(h->E_pad), 1.0f);  // events never uploaded: no such pointer, i.e. a factor of exactly 1
    HIPCHK(h, hipMemcpy(x.d_col, ones.data(), sizeof(float) * h->E_pad, hipMemcpyHostToDevice));
  }
  HIPCHK(h, hipMemcpy(x.d_col + event_offset, w, sizeof(float) * n, hipMemcpyHostToDevice));
  h->n_extra = std::max(h->n_extra, slot + 1);
  h->extras_dirty = true;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_upload_weight_index(m3hip_handle h, int slot, uint64_t event_offset, uint64_t n, const int32_t* idx) try {
  if (!h) return M3HIP_EINVAL;
  int rc = extra_slot_check(h, slot, event_offset, n, "upload_weight_index");
  if (rc) return rc;
  if (n == 0) return M3HIP_OK;
  if (!idx) return fail(h, M3HIP_EINVAL, "upload_weight_index: idx is NULL");
  auto& x = h->extra[slot];
  if (x.d_col) return fail(h, M3HIP_ESTATE, "upload_weight_index: the slot already holds per-event values");
  for (uint64_t i = 0; i < n; ++i) {
    if (idx[i] < -1) return fail(h, M3HIP_EINVAL, "upload_weight_index: index below -1");
    x.max_idx = std::max<int64_t>(x.max_idx, idx[i]);
  }
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (!x.d_idx) {
    if ((rc = dev_alloc(h, &x.d_idx, size_t(h->E_pad)))) return rc;
    HIPCHK(h, hipMemset(x.d_idx, 0xFF, sizeof(int32_t) * h->E_pad));  // -1: no pointer
  }
  HIPCHK(h, hipMemcpy(x.d_idx + event_offset, idx, sizeof(int32_t) * n, hipMemcpyHostToDevice));
  h->n_extra = std::max(h->n_extra, slot + 1);
  h->extras_dirty = true;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_set_weight_table(m3hip_handle h, int slot, const float* table, uint64_t n) try {
  if (!h) return M3HIP_EINVAL;
  if (slot < 0 || slot >= M3_MAX_EXTRA) return fail(h, M3HIP_EINVAL, "set_weight_table: slot out of range");
  if (!table || n == 0) return fail(h, M3HIP_EINVAL, "set_weight_table: empty table");
  auto& x = h->extra[slot];
  if (x.d_col) return fail(h, M3HIP_ESTATE, "set_weight_table: the slot holds per-event values");
  HIPCHK(h, hipSetDevice(h->device));
  int rc;
  if (n != x.n_tab) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    dev_free(x.d_tab); x.d_tab = nullptr;
    x.n_tab = 0;  // the size follows the buffers: an allocation that fails below leaves an empty table, not a stale size
    h->extras_dirty = true;
    for (int i = 0; i < 2; ++i) {
      if (x.h_stage[i]) (void)hipHostFree(x.h_stage[i]);
      x.h_stage[i] = nullptr;
      x.stage_ticket[i] = 0;
    }
    if ((rc = dev_alloc(h, &x.d_tab, size_t(n)))) return rc;
    for (int i = 0; i < 2; ++i) HIPCHK(h, hipHostMalloc(&x.h_stage[i], sizeof(float) * n, hipHostMallocDefault));
    x.n_tab = n;
    h->extras_dirty = true;
  }
  // per-step path, as for the oscillation weights: pinned staging slot, asynchronous copy on the engine's stream
  const int st = x.stage_pos;
  x.stage_pos ^= 1;
  if (x.stage_ticket[st] > h->ticket_done) {
    HIPCHK(h, wait_stream(h->stream));
    h->ticket_done = h->ticket;
  }
  memcpy(x.h_stage[st], table, sizeof(float) * n);
  HIPCHK(h, hipMemcpyAsync(x.d_tab, x.h_stage[st], sizeof(float) * n, hipMemcpyHostToDevice, h->stream));
  x.stage_ticket[st] = h->ticket + 1;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_set_data(m3hip_handle h, const double* data, int n_bins) try {
  if (!h) return M3HIP_EINVAL;
  if (h->n_bins == 0) return fail(h, M3HIP_ESTATE, "set_data before set_binning");
  if (!data || n_bins != h->n_bins) return fail(h, M3HIP_EINVAL, "set_data: n_bins mismatch");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(h->d_data, data, sizeof(double) * n_bins, hipMemcpyHostToDevice));
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_set_deterministic(m3hip_handle h, int on) try {
  if (!h) return M3HIP_EINVAL;
  h->deterministic = on != 0;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_set_test_statistic(m3hip_handle h, int kind, int update_w2) try {
  if (!h) return M3HIP_EINVAL;
  if (kind < 0 || kind > 4) return fail(h, M3HIP_EINVAL, "set_test_statistic: unknown kind");
  h->teststat = kind;
  h->update_w2 = update_w2 != 0;
  return M3HIP_OK;
} M3_API_CATCH(h)

/* ---------------------------------------------------------------- seam 3 per step */

int m3hip_reweight(m3hip_handle h, const double* params, int n_par) try {
  if (!h) return M3HIP_EINVAL;
  if (!h->mono_done) return fail(h, M3HIP_ESTATE, "reweight before the monolith was uploaded");
  if (h->n_bins == 0) return fail(h, M3HIP_ESTATE, "reweight before set_binning");
  if (!params || n_par <= 0) return fail(h, M3HIP_EINVAL, "reweight: empty parameter vector");
  if (!h->hist_rep_locked) {  // flush replicas: the larger number for samples of thousands of workgroups (set_binning)
    if (!h->hist_rep_forced && h->h_tiles.size() + h->h_bricks.size() >= 2048) h->hist_rep = h->hist_rep_big;
    h->hist_rep_locked = true;
  }
  if (!h->parmap_set && n_par < h->P) return fail(h, M3HIP_EINVAL, "reweight: parameter vector shorter than the spline parameters");
  if (h->parmap_set && h->max_parmap >= n_par) return fail(h, M3HIP_EINVAL, "reweight: the spline parameter map points past the parameter vector");
  if (h->max_norm_idx >= n_par) return fail(h, M3HIP_EINVAL, "reweight: an event's norm parameter index points past the parameter vector");
  if (h->max_osc_idx >= int64_t(h->n_osc))
    return fail(h, M3HIP_EINVAL, "reweight: an event's oscillation index (" + std::to_string(h->max_osc_idx) + ") points past the oscillation table (" +
                                     std::to_string(h->n_osc) + " entries; m3hip_set_osc_table)");
  for (const KinShift& k : h->h_shifts) {
    if (k.param >= n_par) return fail(h, M3HIP_EINVAL, "reweight: a kinematic shift points past the parameter vector");
    if (k.dim >= h->kin_stride || (k.kind >> 8) >= h->kin_stride) return fail(h, M3HIP_EINVAL, "reweight: a kinematic shift names a column beyond kin_stride");
  }
  // every precondition is checked BEFORE a ring slot is taken and k_prep empties the bin vectors: an error return must not
  // leave a reset histogram behind for a later get_likelihood / get_bins to read as the result of a step
  if (h->max_bin_idx >= int64_t(h->n_bins))
    return fail(h, M3HIP_EINVAL, "reweight: an event's bin index (" + std::to_string(h->max_bin_idx) + ") lies past the bin space (" +
                                     std::to_string(h->n_bins) + " bins; the binning was redefined after m3hip_upload_events)");
  if (!h->h_shifts.empty() && !h->bins_computed) return fail(h, M3HIP_ESTATE, "reweight: kinematic shifts need m3hip_compute_bins first");
  for (int sl = 0; sl < h->n_extra; ++sl) {
    const auto& x = h->extra[sl];
    if (x.d_idx && x.max_idx >= int64_t(x.n_tab))
      return fail(h, M3HIP_EINVAL, "reweight: weight slot " + std::to_string(sl) + ": an event's index (" + std::to_string(x.max_idx) +
                                       ") points past its table (" + std::to_string(x.n_tab) + " entries; m3hip_set_weight_table)");
  }
  if (h->f64 && h->n_extra) return fail(h, M3HIP_ESTATE, "reweight: experiment weight slots are float (the low-memory build of the sample side)");
  if (h->deterministic) {
    if (sizeof(double) * 2 * size_t(h->n_bins) > size_t(M3_HIST_LDS_MAX_BYTES))
      return fail(h, M3HIP_ESTATE, "reweight: the deterministic fill needs the workgroup-private bin vector (at most " +
                                       std::to_string(M3_HIST_LDS_MAX_BYTES / 16) + " bins)");
    if (h->f64) return fail(h, M3HIP_ESTATE, "reweight: the deterministic fill is implemented for the float build of the sample side");
    const size_t rows = h->h_tiles.size() + h->h_bricks.size();
    if (rows != h->partial_rows || size_t(h->n_bins) != h->partial_bins) {
      HIPCHK(h, hipSetDevice(h->device));
      HIPCHK(h, hipStreamSynchronize(h->stream));
      dev_free(h->d_partial); dev_free(h->d_partial_l1);
      h->d_partial = h->d_partial_l1 = nullptr;
      h->partial_rows = h->partial_bins = 0;
      int rc_;
      if ((rc_ = dev_alloc(h, &h->d_partial, std::max<size_t>(1, rows) * 2 * size_t(h->n_bins)))) return rc_;
      if ((rc_ = dev_alloc(h, &h->d_partial_l1, ((rows + 31) / 32 + 1) * 2 * size_t(h->n_bins)))) return rc_;
      h->partial_rows = rows;
      h->partial_bins = size_t(h->n_bins);
    }
  }
  if (h->f64 && !h->h_bricks.empty()) return fail(h, M3HIP_ESTATE, "reweight: the float_t = double build has no event-by-event splines");
  const double t_a = now_s();
  HIPCHK(h, hipSetDevice(h->device));
  int rc = ensure_param_capacity(h, n_par);
  if (rc) return rc;
  if (h->extras_dirty) {  // the device-side slot descriptors follow the uploads (set-up time)
    ExtraSlot hs[M3_MAX_EXTRA];
    for (int sl = 0; sl < M3_MAX_EXTRA; ++sl) hs[sl] = ExtraSlot{h->extra[sl].d_col, h->extra[sl].d_idx, h->extra[sl].d_tab, 0};
    if (!h->d_extras && (rc = dev_alloc(h, &h->d_extras, size_t(M3_MAX_EXTRA)))) return rc;
    HIPCHK(h, hipMemcpyAsync(h->d_extras, hs, sizeof(hs), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));  // `hs` is on this stack frame
    h->extras_dirty = false;
  }
  collect_timing(h);
  // stage the proposed values; the slot is free once the k_prep that read it has run
  const int slot = h->ring_pos;
  h->ring_pos = (h->ring_pos + 1) % kRing;
  if (h->ring_ticket[slot] > h->ticket_done) {  // the k_prep that read this slot may still be pending
    HIPCHK(h, wait_stream(h->stream));
    h->ticket_done = h->ticket;
  }
  h->ring_ticket[slot] = ++h->ticket;
  memcpy(h->h_params[slot], params, sizeof(double) * n_par);

  PrepArgs p;
  memset(&p, 0, sizeof(p));
  p.params_src = h->d_params_src[slot];
  p.params_dev = h->d_params;
  p.params_f32 = h->d_params_f32;
  p.n_par = n_par;
  p.par_map = h->d_parmap;
  p.coeff_x = h->d_coeff_x;
  p.coeff_x_d = h->d_coeff_x_d;
  p.npts = h->d_npts;
  p.curr_seg = h->d_currseg;
  p.n_params = h->P;
  p.max_knots = h->K;
  p.ptab = h->d_ptab;
  p.pval = h->d_pval;
  p.seg_out = h->d_seg;
  p.hist = h->d_hist;
  p.hist_rep = h->hist_rep;
  p.n_bins = h->n_bins;
  p.reset_hist = 1;
  p.reset_w2 = h->first_w2 ? 1 : 0;
  p.dbg_clock = h->d_dbg_clock;
  p.u_param = h->d_u_param;
  p.u_plane0 = h->d_u_plane0;
  p.resolved = h->d_resolved;
  p.n_res = h->n_res <= kInlineResolveMax ? h->n_res : 0;
  const double t_b = now_s();
  h->host_abs[0] = t_b;
  launch_prep(p, h->stream);
  if (h->n_res > kInlineResolveMax) launch_resolve(p, h->n_res, h->stream);
  if (h->b_U > 0) {  // BinnedSplineHandler::CalcSplineWeights
    BinnedArgs ba;
    ba.uniquecoeffindices = h->d_b_uniq;
    ba.uniquesplinevec = h->d_b_usv;
    ba.coeffindexvec = h->d_b_cidx;
    ba.xcoeff = h->d_b_xcoeff;
    ba.manycoeff = h->d_b_many;
    ba.ptab = h->d_ptab;
    ba.pval = h->d_pval;
    ba.weightvec = h->d_b_w;
    ba.n_unique = h->b_U;
    ba.xcoeff_d = h->d_b_xcoeff_d;
    ba.manycoeff_d = h->d_b_many_d;
    ba.weightvec_d = h->d_b_w_d;
    ba.params = h->d_params;
    ba.par_map = h->d_parmap;
    launch_binned_weights(ba, h->stream);
  }
  // The brick kernel reads an event's bin, oscillation index and norm indices as one packed word when they fit its fields:
  // (re)built here when one of the three arrays changed (set-up time).
  const bool packed_fit = evw_fits(h);
  // A pre-staged oscillation table that has been left alone for a few steps -- a fit that moves oscillation parameters re-uploads
  // it every step and never gets here -- is folded INTO the packed words: they then carry the event's weight instead of its index
  // (ReweightArgs::evw_weight; needs a bin space below 2^14), the brick kernel looks nothing up, and a table too large to stay in
  // L2 (BASELINE configs[4]: 204 000 entries) needs no 4-byte column beside the word.  Table or indices touched again: back to
  // the index form at the next reweight (one packing pass each way).  M3HIP_OSC_COLUMN = 0 never, 1 at once.
  const bool osc_static_counts = !h->f64 && !h->h_bricks.empty() && h->n_osc && h->d_osc && h->oscw_mode != 0;
  if (osc_static_counts) ++h->osc_static_steps;
  const bool osc_is_static = osc_static_counts && h->osc_valid_f32 && (h->oscw_mode == 1 || h->osc_static_steps > kOscStaticSteps);
  // (tables that stay in L1 / L2 keep the index form: config 3's 16 KB table measured level either way, 700.4 / 701.9 us)
  const bool osc_large = sizeof(float) * h->n_osc > 64 * 1024;
  const bool want_weight = osc_is_static && (osc_large || h->oscw_mode == 1) && packed_fit && h->n_bins < (1 << 14) - 2 && h->max_bin_idx < (1 << 14) - 2;
  if (want_weight != h->evw_weight) h->evw_dirty = true;
  if (!h->f64 && !h->h_bricks.empty() && h->evw_dirty) {
    h->evw_ok = packed_fit;
    // norm parameters spread over more than 62 indices do not fit the word's 6-bit codes: the word then carries bin and
    // oscillation index only, and the kernel's tail reads every norm slot from the index planes
    h->evw_norm_slots = (h->max_norm_idx_evw < 0 || h->max_norm_idx_evw - h->min_norm_idx <= 62) ? uint32_t(M3_EVW_NORM_SLOTS) : 0u;
    if (h->sw_packed == 'b') h->evw_norm_slots = 0;  // M3HIP_PACKED_EVENTS=b: bin and oscillation index only (tests)
    if (int rc_b = bins_from_word(h)) return rc_b;  // (bins that live in the words only -- in the form they were packed in -- into d_bin before the words are rebuilt)
    h->evw_weight = false;
    if (h->evw_ok) {
      if (!h->d_evw && (rc = dev_alloc(h, &h->d_evw, size_t(h->E_pad)))) return rc;
      launch_pack_events(h->d_bin, h->d_osc, h->d_nidx, std::min(h->M, h->evw_norm_slots), h->E_pad,
                         int32_t(h->max_norm_idx_evw < 0 ? 0 : h->min_norm_idx), h->d_evw, want_weight ? h->d_osc_table : nullptr, h->stream);
      h->evw_weight = want_weight;
    }
    h->evw_dirty = false;
  }
  const bool word_valid = !h->f64 && !h->h_bricks.empty() && h->evw_ok && packed_fit && h->d_evw;
  // Functional shifts: every event is re-binned from its shifted kinematics (ApplyShifts + FindBin).  Uniform tiles do it in
  // their kernel's epilogue.  Samples with bricks (and the float_t = double variant) by a pass of its own in front of the
  // reweight kernel: with the binning tables in LDS when they fit (uniform samples, <= 2048 edges) and, when the packed words
  // exist, straight into their bin field -- 8 B of kinematics + the word, read and written, per event; k_rebin_pack /
  // k_find_bins from global tables otherwise.  Bricks of 1024 events with the packed word can also re-bin in the brick
  // kernel's own tail (no scratch since round 3's register work): measured slower than the LDS pass (the cascade's look-ups
  // come from L2 there), so only M3HIP_REBIN=tail selects it.
  const bool dyn_any = !h->h_shifts.empty();
  bool nonuniform = false;
  for (const BinSample& bs : h->bb_samples) nonuniform = nonuniform || !bs.uniform;
  const bool lds_ok = rebin_tables_fit_lds(int(h->bb_samples.size()), int(h->bb_axes.size()), int(h->bb_edges.size()), int(h->h_shifts.size()),
                                           int(h->h_cuts.size()), nonuniform);
  const bool want_tail = h->sw_rebin == 't';  // M3HIP_REBIN = tail | pass | global (experiments / tests)
  const bool use_lds = lds_ok && h->sw_rebin != 'g';
  const bool brick_tail = want_tail && brick_can_rebin(uint32_t(h->brick_groups), word_valid);
  const bool dyn_pass = dyn_any && (h->f64 || (!h->h_bricks.empty() && !brick_tail));
  const bool dyn_bins = dyn_any && !dyn_pass;  // fused
  if (dyn_pass) {
    BinArgs b = make_bin_args(h);
    b.compute_nominal = 0;
    b.word_bin_mask = h->evw_weight ? kEvwBinMaskWeight : kEvwBinMask;
    if (word_valid && h->h_tiles.empty()) {  // bricks only: nothing but the words needs the bins
      if (use_lds) {
        launch_rebin_lds(b, h->d_evw, int(h->bb_samples.size()), int(h->bb_axes.size()), int(h->bb_edges.size()), h->stream);
        h->bins_in_word = true;  // d_bin is stale until somebody asks for it (bins_from_word)
      } else {
        launch_rebin_pack(b, h->d_evw, h->stream);
      }
    } else {
      if (use_lds) launch_rebin_lds(b, nullptr, int(h->bb_samples.size()), int(h->bb_axes.size()), int(h->bb_edges.size()), h->stream);
      else launch_find_bins(b, h->stream);
      if (word_valid)  // (mixed tiles and bricks: the words follow the bins; rare)
        launch_pack_events(h->d_bin, h->d_osc, h->d_nidx, std::min(h->M, h->evw_norm_slots), h->E_pad,
                           int32_t(h->max_norm_idx_evw < 0 ? 0 : h->min_norm_idx), h->d_evw, h->evw_weight ? h->d_osc_table : nullptr, h->stream);
    }
  }
  const double t_c = now_s();

  if (h->n_osc) {  // the oscillation weights in the precision this build of the sample side multiplies in
    if (h->f64 && !h->osc_valid_f64) {
      launch_osc_convert(h->d_osc_table, nullptr, h->d_osc_table_d, h->n_osc, true, h->stream);
      h->osc_valid_f64 = true;
    } else if (!h->f64 && !h->osc_valid_f32) {
      launch_osc_convert(nullptr, h->d_osc_table_d, h->d_osc_table, h->n_osc, false, h->stream);
      h->osc_valid_f32 = true;
    }
  }
  // Engines whose brick kernel reads the UNPACKED sample side (a sample that does not fit the word's fields, or a bin space of
  // 2^14 and more): a large static table is gathered ONCE into a per-event column the kernel streams (4 B per event) -- it does not
  // stay in L2 beside the streams, and every event's gather from it then goes to memory (round 4, one GPU's share of config 5:
  // 5.5 % of the kernel).  With the packed word the weight travels in the word itself (above).
  if (osc_static_counts && !(word_valid && h->evw_weight)) {
    const bool large = sizeof(float) * h->n_osc > 64 * 1024;
    if (!h->oscw_valid && ((h->oscw_mode == 1) || (large && osc_is_static))) {
      if (!h->d_oscw && (rc = dev_alloc(h, &h->d_oscw, size_t(h->E_pad)))) return rc;
      launch_osc_expand(h->d_osc, h->d_osc_table, h->d_oscw, h->E_pad, h->stream);
      h->oscw_valid = true;
    }
  }
  if (h->perm_dirty) {  // processing order of a binned-only event space: stable sort of the events by pointer list
    h->perm_dirty = false;
    if (h->bl_mode == 1 && h->S + h->T == 0) {
      std::vector<uint32_t> perm(size_t(h->E_pad));
      for (size_t i = 0; i < perm.size(); ++i) perm[i] = uint32_t(i);
      const std::vector<int32_t>& id = h->h_bl_id;
      std::stable_sort(perm.begin(), perm.begin() + h->E, [&](uint32_t x, uint32_t y) {
        const uint32_t kx = uint32_t(id[x]), ky = uint32_t(id[y]);  // -1 (no list) sorts last
        return kx < ky;
      });
      if (!h->d_perm && (rc = dev_alloc(h, &h->d_perm, perm.size()))) return rc;
      HIPCHK(h, wait_stream(h->stream));
      HIPCHK(h, hipMemcpy(h->d_perm, perm.data(), sizeof(uint32_t) * perm.size(), hipMemcpyHostToDevice));
    }
  }
  ReweightArgs a = make_rw_args(h);
  a.n_par_total = n_par;
  a.params_f32 = h->d_params_f32;
  if (dyn_bins && !h->f64) {
    a.dyn = make_bin_args(h);
    a.dyn.compute_nominal = 0;
  }
  if (h->oscw_valid && !h->f64 && !(word_valid && h->evw_weight)) a.oscw = h->d_oscw;
  a.nt_codes = choose_nt_codes(h, word_valid);
  h->last_nt_codes = a.nt_codes;
  h->last_cached_bytes = h->h_bricks.empty() ? 0 : h->re_tables + (word_valid ? 8 * h->E : 0) + h->re_book + (a.nt_codes ? 0 : h->re_codes);
  if (word_valid) {
    a.evw = h->d_evw;
    a.evw_weight = h->evw_weight ? 1u : 0u;
    a.evw_norm_base = int32_t(h->max_norm_idx_evw < 0 ? 0 : h->min_norm_idx);
    a.evw_norm_slots = h->evw_norm_slots;
  }
  if (a.fill_w2) h->w2_spread = true;
  int mode = kModeFill;
  if (h->want_event_w) {
    if (!h->d_event_w) {
      if ((rc = dev_alloc(h, &h->d_event_w, size_t(h->E_pad)))) return rc;
      a.event_w = h->d_event_w;
    }
    if (h->f64 && !h->d_event_w_d) {
      if ((rc = dev_alloc(h, &h->d_event_w_d, size_t(h->E_pad)))) return rc;
      a.event_w_d = h->d_event_w_d;
    }
    mode |= kModeEventW | kModeEval;
  }
  h->gap_pending = h->gap_recorded = false;  // a gap belongs to the step whose reweight set it
  // sampled launches carry two events that receive the dispatch's own start / end timestamps
  const bool timed = h->timing > 0 && (h->timing_ctr++ % uint64_t(h->timing)) == 0;
  const uint32_t n_tiles = uint32_t(h->h_tiles.size()), n_bricks = uint32_t(h->h_bricks.size());
  ReweightKnobs ku = h->knobs, kb = h->knobs;  // with both kinds of tile: start on the first launch, stop on the second
  if (timed) {
    ku.ev_start = n_tiles ? h->ev0 : nullptr;
    ku.ev_stop = (n_tiles && !n_bricks) ? h->ev1 : nullptr;
    kb.ev_start = n_tiles ? nullptr : h->ev0;
    kb.ev_stop = n_bricks ? h->ev1 : nullptr;
  }
  if (h->deterministic) {
    a.hist_lds = 1;  // sized for fill_w2 in the check above
    a.partial = h->d_partial;
    a.partial_row0 = n_tiles;
  }
  if (h->f64) {
    launch_reweight_uniform_f64(a, n_tiles, mode, ku, h->stream);
  } else {
    launch_reweight_uniform(a, n_tiles, mode, ku, h->stream, dyn_bins);
    launch_reweight_brick(a, n_bricks, mode, kb, h->stream, dyn_bins);
  }
  if (h->deterministic && (n_tiles || n_bricks)) {  // ordered sum of the workgroups' rows -> replica 0
    launch_fold_partials(h->d_partial, int(n_tiles + n_bricks), h->n_bins, a.fill_w2, h->d_partial_l1, h->d_hist, h->hist_rep, h->stream);
    if (a.fill_w2) h->w2_spread = false;  // w2 sits folded in its replica 0
  }
  if (timed && (n_tiles || n_bricks)) h->pending = 0;
  HIPCHK(h, hipGetLastError());
  if (h->xch_active) {
    // one-shot exchange: fold, push this rank's [mc | w2] into every rank's buffer, pull the rank-ordered sum of the N slots
    if (h->n_bins * 2 + 2 != h->xch_slot) return fail(h, M3HIP_ESTATE, "reweight: the bin space changed after m3hip_xch_create");
    const bool with_w2 = h->first_w2;
    const bool parked = with_w2 && h->hist_rep > 1;
    if (h->hist_rep > 1) launch_fold_hist(h->d_hist, h->hist_rep, h->n_bins, with_w2 ? 1 : 0, parked ? 1 : 0, h->stream);
    if (with_w2) h->w2_spread = false;
    // the message always has the slot's layout [mc | w2]; w2's half is only meaningful (and only used) while it accumulates
    const int n_msg = h->n_bins * (with_w2 ? 2 : 1);
    const uint64_t step = ++h->xch_step;
    launch_xch(h->d_hist, h->d_hist, n_msg, h->d_xch_targets + size_t(step & 1) * h->xch_ranks, h->xch_ranks,
               h->xch_own + size_t(step & 1) * h->xch_ranks * h->xch_slot, h->xch_slot, step, h->d_xch_status, h->d_xch_pushed,
               uint64_t(h->xch_timeout_s * 1e8), h->stream);
    if (parked) launch_unpark_w2(h->d_hist, h->hist_rep, h->n_bins, h->stream);
    h->gap_pending = timed && (n_tiles || n_bricks);
  } else if (h->comm) {  // also with one rank: the communicator is exercised exactly as with N
    // Fold the flush replicas first: the message is n_bins doubles (1.6 KB at 200 bins).  While w2 is still being
    // accumulated it travels in the SAME all-reduce: the fold parks its sum right behind mc's (in mc's second replica,
    // which the fold has just emptied), one call moves [mc | w2], and k_unpark_w2 puts it back where k_llh looks for it.
    const bool with_w2 = h->first_w2;
    const bool parked = with_w2 && h->hist_rep > 1;
    if (h->hist_rep > 1) launch_fold_hist(h->d_hist, h->hist_rep, h->n_bins, with_w2 ? 1 : 0, parked ? 1 : 0, h->stream);  // one replica: nothing to fold
    if (with_w2) h->w2_spread = false;
    const size_t n_msg = size_t(h->n_bins) * (with_w2 ? 2 : 1);  // one replica: w2 sits behind mc as it is
    // whatever fails below, w2 goes back to where k_llh and m3hip_get_bins look for it before the error is returned
    int rc_comm = h->comm_validate ? comm_validate_begin(h, n_msg) : M3HIP_OK;
    if (rc_comm == M3HIP_OK) {
      ncclResult_t r = g_rccl.AllReduce(h->d_hist, h->d_hist, n_msg, ncclDouble, ncclSum, (ncclComm_t)h->comm, h->stream);
      if (r != ncclSuccess) rc_comm = fail(h, M3HIP_ECOMM, "ncclAllReduce: " + g_rccl.why(r));
    }
    if (rc_comm == M3HIP_OK && h->comm_validate) rc_comm = comm_validate_end(h, n_msg);
    if (parked) launch_unpark_w2(h->d_hist, h->hist_rep, h->n_bins, h->stream);
    if (rc_comm != M3HIP_OK) return rc_comm;
    h->gap_pending = timed && (n_tiles || n_bricks);  // m3hip_get_likelihood drops a marker in front of k_llh
  }
  if (!h->update_w2) h->first_w2 = false;  // SampleHandlerFD.cpp:342
  const double t_d = now_s();
  h->host_t[0] = t_b - t_a;
  h->host_t[1] = t_c - t_b;
  h->host_t[2] = t_d - t_c;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_get_likelihood(m3hip_handle h, double* llh_total, double* per_sample) try {
  if (!h) return M3HIP_EINVAL;
  if (h->n_bins == 0) return fail(h, M3HIP_ESTATE, "get_likelihood before set_binning");
  HIPCHK(h, hipSetDevice(h->device));
  LlhArgs l;
  l.data = h->d_data;
  l.hist = h->d_hist;
  l.hist_rep = h->hist_rep;
  l.n_bins = h->n_bins;
  l.n_samples = h->n_samples;
  l.sample_off = h->d_sample_off;
  l.kind = h->teststat;
  // W2 that was frozen by the last reweight is folded in place once; while UpdateW2 keeps accumulating it is summed on the fly
  l.w2_mode = !h->w2_spread ? kW2Replica0 : (h->update_w2 ? kW2Fold : kW2FoldInPlace);
  l.out = h->d_out;
  l.terms = h->d_llh_terms;
  // completion is signalled through host memory: k_llh writes the results, then this ticket
  volatile uint64_t* flag = reinterpret_cast<volatile uint64_t*>(h->h_out + 1 + h->n_samples);
  l.done_flag = reinterpret_cast<volatile uint64_t*>(h->d_out + 1 + h->n_samples);
  l.ticket = ++h->llh_ticket;
  l.dbg_clock = h->d_dbg_clock;
  const double t_a = now_s();
  if (h->gap_pending) h->gap_recorded = hipEventRecord(h->ev2, h->stream) == hipSuccess;  // sampled steps with an exchange: reweight end -> here
  launch_llh(l, h->stream);
  if (l.w2_mode == kW2FoldInPlace) h->w2_spread = false;
  HIPCHK(h, hipGetLastError());
  const double t_b = now_s();
  {
    // The host-memory flag gives the completion with PCIe-write latency; the periodic
    // hipStreamQuery keeps the runtime submitting / retiring commands (with flag polling alone
    // steps stalled intermittently for 0.4-0.9 ms) and surfaces device errors.
    const double t0 = now_s();
    uint64_t spins = 0;
    (void)hipStreamQuery(h->stream);
    while (*flag != l.ticket) {
      if ((++spins & 0xFFu) == 0) {
        hipError_t e = hipStreamQuery(h->stream);
        if (e != hipSuccess && e != hipErrorNotReady) return fail(h, M3HIP_EHIP, std::string("get_likelihood: ") + hipGetErrorString(e));
        if (e == hipSuccess && *flag != l.ticket && now_s() - t0 > 5.0) return fail(h, M3HIP_EHIP, "get_likelihood: kernel finished without publishing its result");
        if (now_s() - t0 > 60.0) return fail(h, M3HIP_EHIP, "get_likelihood: timed out waiting for the device");
      }
    }
  }
  // Let the runtime retire the finished commands (non-blocking).  Without any query/sync call the
  // HIP command list only grows and a later launch stalls for ~1 ms when it has to be trimmed.
  (void)hipStreamQuery(h->stream);
  h->ticket_done = h->ticket;  // everything submitted before k_llh has completed (stream order)
  if (int rc_x = xch_check(h)) return rc_x;
  const double t_c = now_s();
  h->host_abs[1] = t_c;
  collect_timing(h);
  h->host_t[3] = t_b - t_a;
  h->host_t[4] = t_c - t_b;
  h->host_t[5] = now_s() - t_c;
  if (llh_total) *llh_total = h->h_out[0];
  if (per_sample)
    for (int s = 0; s < h->n_samples; ++s) per_sample[s] = h->h_out[1 + s];
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_step(m3hip_handle h, const double* params, int n_par, double* llh_total, double* per_sample) try {
  int rc = m3hip_reweight(h, params, n_par);
  if (rc) return rc;
  return m3hip_get_likelihood(h, llh_total, per_sample);
} M3_API_CATCH(h)

int m3hip_get_bins(m3hip_handle h, double* mc, double* w2, int n_bins) try {
  if (!h) return M3HIP_EINVAL;
  if (h->n_bins == 0 || n_bins != h->n_bins) return fail(h, M3HIP_EINVAL, "get_bins: n_bins mismatch");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  collect_timing(h);
  if (int rc_x = xch_check(h)) return rc_x;  // a timed-out exchange leaves this rank's PARTIAL bins in d_hist
  // replicas are folded in replica order, exactly as k_llh does
  std::vector<double> tmp(size_t(n_bins) * 2 * h->hist_rep);
  HIPCHK(h, hipMemcpy(tmp.data(), h->d_hist, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost));
  for (int i = 0; i < n_bins; ++i) {
    double m = tmp[i], v = tmp[size_t(h->hist_rep) * n_bins + i];
    for (int r = 1; r < h->hist_rep; ++r) {
      m += tmp[size_t(r) * n_bins + i];
      v += tmp[size_t(h->hist_rep + r) * n_bins + i];
    }
    if (mc) mc[i] = m;
    if (w2) w2[i] = v;
  }
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_get_spline_weights(m3hip_handle h, float* out, uint64_t event_offset, uint64_t n) try {
  if (!h || !out) return M3HIP_EINVAL;
  if (!h->mono_done || event_offset + n > h->E) return fail(h, M3HIP_EINVAL, "get_spline_weights: bad range");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  collect_timing(h);
  HIPCHK(h, hipMemcpy(out, h->d_spline_w + event_offset, sizeof(float) * n, hipMemcpyDeviceToHost));
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_record_event_weights(m3hip_handle h, int enable) try {
  if (!h) return M3HIP_EINVAL;
  h->want_event_w = enable != 0;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_get_event_weights(m3hip_handle h, float* out, uint64_t event_offset, uint64_t n) try {
  if (!h || !out) return M3HIP_EINVAL;
  if (!h->mono_done || event_offset + n > h->E) return fail(h, M3HIP_EINVAL, "get_event_weights: bad range");
  if (!h->d_event_w) return fail(h, M3HIP_ESTATE, "get_event_weights: call m3hip_record_event_weights(h, 1) before the reweight");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  collect_timing(h);
  HIPCHK(h, hipMemcpy(out, h->d_event_w + event_offset, sizeof(float) * n, hipMemcpyDeviceToHost));
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_get_event_weights_f64(m3hip_handle h, double* out, uint64_t event_offset, uint64_t n) try {
  if (!h || !out) return M3HIP_EINVAL;
  if (!h->mono_done || event_offset + n > h->E) return fail(h, M3HIP_EINVAL, "get_event_weights_f64: bad range");
  if (!h->d_event_w_d) return fail(h, M3HIP_ESTATE, "get_event_weights_f64: needs m3hip_binned_upload_f64 and m3hip_record_event_weights(h, 1) before the reweight");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  collect_timing(h);
  HIPCHK(h, hipMemcpy(out, h->d_event_w_d + event_offset, sizeof(double) * n, hipMemcpyDeviceToHost));
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_get_segments(m3hip_handle h, int16_t* segments, float* vals, int n_params) try {
  if (!h) return M3HIP_EINVAL;
  if (!h->mono_done || n_params != h->P) return fail(h, M3HIP_EINVAL, "get_segments: n_params mismatch");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (segments) HIPCHK(h, hipMemcpy(segments, h->d_seg, sizeof(int16_t) * n_params, hipMemcpyDeviceToHost));
  if (vals) HIPCHK(h, hipMemcpy(vals, h->d_pval, sizeof(float) * n_params, hipMemcpyDeviceToHost));
  return M3HIP_OK;
} M3_API_CATCH(h)

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
      it = it->second == (void*)h->xch_own ? g_xch_local.erase(it) : std::next(it);
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
    g_xch_local[std::string(static_cast<const char*>(handle_out), M3HIP_XCH_HANDLE_BYTES)] = h->xch_own;
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
    {  // a peer in this process: its buffer as it is (another device of this process: peer access through the runtime's default)
      std::lock_guard<std::mutex> lk(g_xch_mu);
      auto it = g_xch_local.find(std::string(hb, M3HIP_XCH_HANDLE_BYTES));
      if (it != g_xch_local.end()) { h->xch_peer[r] = it->second; h->xch_peer_local[r] = 1; continue; }
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

/* ---------------------------------------------------------------- parameter handler on the device */

static m3hip_ctx::ParDev* par_slot(m3hip_handle h, int slot, const char* who) {
  if (slot < 0 || slot >= int(h->pars.size()) || !h->pars[size_t(slot)].live) {
    fail(h, M3HIP_EINVAL, std::string(who) + ": no such parameter-handler slot (m3hip_par_create)");
    return nullptr;
  }
  return &h->pars[size_t(slot)];
}

static int par_wait(m3hip_handle h, m3hip_ctx::ParDev& p) {  // every launched step of this slot has published its results
  if (p.collected == p.ticket) return M3HIP_OK;
  volatile uint64_t* flag = reinterpret_cast<volatile uint64_t*>(p.h_out + 1 + p.n);
  hipStream_t st = h->par_stream ? h->par_stream : h->stream;
  const double t0 = now_s();
  uint64_t spins = 0;
  (void)hipStreamQuery(st);
  while (*flag != p.ticket) {
    if ((++spins & 0xFFu) == 0) {
      hipError_t e = hipStreamQuery(st);
      if (e != hipSuccess && e != hipErrorNotReady) return fail(h, M3HIP_EHIP, std::string("par_result: ") + hipGetErrorString(e));
      // the stream has drained and the ticket is not there: the kernels that would publish it never ran
      if (e == hipSuccess && *flag != p.ticket && now_s() - t0 > 2.0) return fail(h, M3HIP_EHIP, "par_result: the stream is idle but the step's results were never published");
      if (now_s() - t0 > 60.0) return fail(h, M3HIP_EHIP, "par_result: timed out waiting for the device");
    }
  }
  (void)hipStreamQuery(st);
  p.collected = p.ticket;
  return M3HIP_OK;
}

int m3hip_par_create(m3hip_handle h, int* slot_out) try {
  if (!h || !slot_out) return M3HIP_EINVAL;
  for (size_t i = 0; i < h->pars.size(); ++i)
    if (!h->pars[i].live) {
      h->pars[i] = m3hip_ctx::ParDev();
      h->pars[i].live = true;
      *slot_out = int(i);
      return M3HIP_OK;
    }
  if (h->pars.size() >= size_t(M3HIP_MAX_PAR_SLOTS)) return fail(h, M3HIP_ESTATE, "par_create: all " + std::to_string(M3HIP_MAX_PAR_SLOTS) + " parameter-handler slots are in use");
  h->pars.emplace_back();
  h->pars.back().live = true;
  *slot_out = int(h->pars.size()) - 1;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_par_destroy(m3hip_handle h, int slot) try {
  if (!h) return M3HIP_EINVAL;
  m3hip_ctx::ParDev* p = par_slot(h, slot, "par_destroy");
  if (!p) return M3HIP_EINVAL;
  (void)hipSetDevice(h->device);
  par_free(h, *p);
  p->live = false;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_par_upload(m3hip_handle h, int slot, int which, const double* m, int n) try {
  if (!h) return M3HIP_EINVAL;
  m3hip_ctx::ParDev* pp = par_slot(h, slot, "par_upload");
  if (!pp) return M3HIP_EINVAL;
  if ((which != 0 && which != 1) || !m || n < 1 || n > 16384) return fail(h, M3HIP_EINVAL, "par_upload: which is 0 (inverse covariance) or 1 (Cholesky factor), 1 <= n <= 16384");
  HIPCHK(h, hipSetDevice(h->device));
  auto& p = *pp;
  int rc;
  if (p.n != n) {  // (re)size everything
    par_free(h, p);
    if (!h->par_stream) {
      const char* f = getenv("M3HIP_PAR_STREAM");
      if (!(f && f[0] == '0')) HIPCHK(h, hipStreamCreateWithFlags(&h->par_stream, hipStreamNonBlocking));
    }
    if ((rc = dev_alloc(h, &p.d_inv, size_t(n) * n))) return rc;
    if ((rc = dev_alloc(h, &p.d_chol, size_t(n) * n))) return rc;
    if ((rc = dev_alloc(h, &p.d_rows, size_t(2) * n))) return rc;
    if ((rc = dev_alloc(h, &p.d_in, size_t(2) * n))) return rc;
    for (int i = 0; i < 2; ++i) {
      HIPCHK(h, hipHostMalloc((void**)&p.h_in[i], sizeof(double) * 2 * n, hipHostMallocMapped));
      HIPCHK(h, hipHostGetDevicePointer((void**)&p.d_in_host[i], p.h_in[i], 0));
    }
    HIPCHK(h, hipHostMalloc((void**)&p.h_out, sizeof(double) * (n + 2), hipHostMallocMapped));
    memset(p.h_out, 0, sizeof(double) * (n + 2));
    HIPCHK(h, hipHostGetDevicePointer((void**)&p.d_out, p.h_out, 0));
    p.n = n;
  }
  if ((rc = par_wait(h, p))) return rc;  // a step in flight still reads the old matrix
  HIPCHK(h, hipMemcpy(which == 0 ? p.d_inv : p.d_chol, m, sizeof(double) * size_t(n) * n, hipMemcpyHostToDevice));
  (which == 0 ? p.have_inv : p.have_chol) = true;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_par_launch(m3hip_handle h, int slot, const double* d, const double* z, int n) try {
  if (!h) return M3HIP_EINVAL;
  m3hip_ctx::ParDev* pp = par_slot(h, slot, "par_launch");
  if (!pp) return M3HIP_EINVAL;
  auto& p = *pp;
  if (!p.have_inv || !p.have_chol) return fail(h, M3HIP_ESTATE, "par_launch before both matrices were uploaded (m3hip_par_upload)");
  if (!d || !z || n != p.n) return fail(h, M3HIP_EINVAL, "par_launch: NULL vectors or n differs from the uploaded matrices");
  HIPCHK(h, hipSetDevice(h->device));
  int rc;
  if (p.ticket > p.collected && (rc = par_wait(h, p))) return rc;  // results nobody collected: the output block is about to be rewritten
  const int in = int(p.ticket & 1);
  memcpy(p.h_in[in], d, sizeof(double) * n);
  memcpy(p.h_in[in] + n, z, sizeof(double) * n);
  // the ticket counts a step only once its launches went out: a failed launch must not leave every later wait spinning
  // for a ticket no kernel will ever publish
  const uint64_t ticket = p.ticket + 1;
  launch_par_step(p.d_inv, p.d_chol, p.d_in_host[in], p.d_in, n, p.d_rows, p.d_out, reinterpret_cast<volatile uint64_t*>(p.d_out + 1 + n), ticket,
                  h->par_stream ? h->par_stream : h->stream);
  HIPCHK(h, hipGetLastError());
  p.ticket = ticket;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_par_result(m3hip_handle h, int slot, double* prior, double* corr_throw, int n) try {
  if (!h) return M3HIP_EINVAL;
  m3hip_ctx::ParDev* pp = par_slot(h, slot, "par_result");
  if (!pp) return M3HIP_EINVAL;
  auto& p = *pp;
  if (p.ticket == 0) return fail(h, M3HIP_ESTATE, "par_result before par_launch");
  if (n != p.n) return fail(h, M3HIP_EINVAL, "par_result: n differs from the uploaded matrices");
  int rc = par_wait(h, p);
  if (rc) return rc;
  if (prior) *prior = p.h_out[0];
  if (corr_throw) memcpy(corr_throw, p.h_out + 1, sizeof(double) * n);
  return M3HIP_OK;
} M3_API_CATCH(h)

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

/* ---------------------------------------------------------------- host-only layout probe */

int m3hip_debug_layout(int n_params, int max_knots, const int16_t* npts, uint64_t n_events, const float* coeff_many,
                       uint64_t total_nknots, const unsigned int* nknots_arr, const int16_t* param_no,
                       uint64_t n_splines, const float* coeff_tf1, const int16_t* param_no_tf1, uint64_t n_tf1,
                       const unsigned int* n_param_per_event, const unsigned int* n_param_per_event_tf1,
                       uint64_t* sizes, float* spl, float* tf1, int32_t* meta, uint64_t* tile_desc,
                       uint64_t* brick_desc, uint32_t* brick_words) try {
  if (!npts || !sizes) return fail(nullptr, M3HIP_EINVAL, "debug_layout: NULL arguments");
  ChunkInput in;
  in.n_events = n_events;
  in.coeff_many = coeff_many;
  in.total_nknots = total_nknots;
  in.nknots_arr = nknots_arr;
  in.param_no = param_no;
  in.n_splines = n_splines;
  in.coeff_tf1 = coeff_tf1;
  in.param_no_tf1 = param_no_tf1;
  in.n_tf1 = n_tf1;
  in.npe = n_param_per_event;
  in.npe_tf1 = n_param_per_event_tf1;
  ChunkPlan plan;
  std::string err;
  { const char* f = getenv("M3HIP_PAD_MIN_FILL"); g_pad_min_fill = f ? atof(f) : kPadMinFillDefault; }
  { const char* f = getenv("M3HIP_TF1_SORT"); g_tf1_sort = f ? atoi(f) : 2; }
  { const char* f = getenv("M3HIP_PAD_RUNS"); g_pad_runs = f ? atoi(f) : 2; }
  // M3HIP_DEBUG_SHORT_FROM / _TO: the stretch of events whose bricks are cut short (tests/test_layout.py; an engine gets them from its ShortBrickPlan)
  uint64_t short_from = ~0ull, short_to = ~0ull;
  if (const char* f = getenv("M3HIP_DEBUG_SHORT_FROM")) if (f[0]) short_from = strtoull(f, nullptr, 10);
  if (const char* f = getenv("M3HIP_DEBUG_SHORT_TO")) if (f[0]) short_to = strtoull(f, nullptr, 10);
  if (!plan_chunk(n_params, max_knots, npts, in, &plan, &err, choose_brick_groups(in, 0), short_from, short_to)) return fail(nullptr, M3HIP_EINVAL, "debug_layout: " + err);
  sizes[0] = plan.tiles.size();
  sizes[1] = plan.spl_float4;
  sizes[2] = plan.tf1_float2;
  sizes[3] = plan.meta_int32;
  sizes[4] = plan.bricks.size();
  sizes[5] = plan.brick_words.size();
  sizes[6] = plan.dict_off;          // first word of the bricks' TF1 dictionary inside brick_words
  sizes[7] = plan.tf1_dict.size();
  if (!spl || !tf1 || !meta || !tile_desc || !brick_desc || !brick_words) return M3HIP_OK;  // plan only
  fill_chunk(n_params, npts, in, plan, spl, tf1, meta);
  for (size_t t = 0; t < plan.tiles.size(); ++t) {
    const TileHost& th = plan.tiles[t];
    uint64_t* d = tile_desc + 8 * t;
    d[0] = th.spl_off; d[1] = th.tf1_off; d[2] = th.meta_off; d[3] = th.n_spl_slots;
    d[4] = th.n_tf1_slots; d[5] = th.n_events; d[6] = th.first_event; d[7] = th.n_planes;
  }
  for (size_t b = 0; b < plan.bricks.size(); ++b) {
    const BrickHost& bh = plan.bricks[b];
    uint64_t* d = brick_desc + 8 * b;
    d[0] = bh.spl_off; d[1] = bh.tf1_off; d[2] = bh.words_off; d[3] = bh.n_spl_f4;
    d[4] = bh.n_tf1_f2; d[5] = bh.n_events; d[6] = bh.first_event; d[7] = bh.n_words;
  }
  if (!plan.brick_words.empty()) memcpy(brick_words, plan.brick_words.data(), plan.brick_words.size() * sizeof(uint32_t));
  return M3HIP_OK;
} M3_API_CATCH(nullptr)

int m3hip_debug_short_brick_plan(uint64_t expect_events, uint64_t chunk_events, int brick_groups, int n_cu, uint64_t* n_bricks,
                                 uint64_t* n_short) try {
  if (!n_bricks || !n_short || chunk_events == 0 || chunk_events % M3_GROUP != 0 || (brick_groups != 1 && brick_groups != 2 && brick_groups != 4))
    return fail(nullptr, M3HIP_EINVAL, "debug_short_brick_plan: bad arguments");
  ShortBrickPlan sp;
  const uint64_t B = uint64_t(M3_GROUP) * uint64_t(brick_groups);
  *n_bricks = *n_short = 0;
  for (uint64_t e0 = 0; e0 < expect_events; e0 += chunk_events) {
    const uint64_t n = std::min(chunk_events, expect_events - e0);
    if (!sp.decided) sp.begin(expect_events, n, brick_groups, n_cu);
    uint64_t from, to;
    sp.take(e0, n, &from, &to);
    if (from == ~0ull) { *n_bricks += (n + B - 1) / B; continue; }  // what plan_chunk cuts: full bricks, short ones in [from, to)
    const uint64_t shorts = (to - from) / (3 * B / 4);
    *n_short += shorts;
    *n_bricks += from / B + shorts + (n - to + B - 1) / B;
  }
  return M3HIP_OK;
} M3_API_CATCH(nullptr)

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
