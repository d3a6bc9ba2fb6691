// m3hip_ctx.h -- internal header of libm3hip's host side: the engine's context (struct m3hip_ctx behind m3hip_handle), the
// RCCL loader, and the helpers the translation units share.  Not installed; include/m3hip.h is the interface.
//   m3hip_api.hip     lifetime, pinned memory, introspection, timing, test hooks; the shared helpers' definitions
//   m3hip_upload.hip  the one-time upload: monolith begin / append / end (re-layout, grouping, short bricks), seam 2's evaluation,
//                     binned splines, the host-only layout probes
//   m3hip_sample.hip  the sample side's set-up: parameter map, bin space, per-event arrays, device binning, shifts and cuts,
//                     oscillation tables, experiment weights, data, test statistic
//   m3hip_step.hip    the per-step path: m3hip_reweight, m3hip_get_likelihood, m3hip_step, results
//   m3hip_comm.hip    one sample over several GPUs: the RCCL communicator and the one-shot exchange
//   m3hip_par.hip     the parameter handler's prior and correlated throw on the device
#pragma once
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

constexpr int kRing = 4;  // parameter staging slots in flight

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
extern Rccl g_rccl;
extern std::string g_create_err;
// One-shot exchange buffers created in THIS process, by the bytes of their IPC handle: a rank whose peer lives in the same
// process (one process driving several engines / devices) takes the peer's pointer from here -- hipIpcOpenMemHandle refuses
// a handle of the opening process's own.
extern std::mutex g_xch_mu;
extern std::map<std::string, std::pair<void*, int>> g_xch_local;  // IPC handle -> {buffer, owning device} of the engines of THIS process

// which events of an upload get bricks of three quarters of the size (described at ShortBrickPlan::begin)
struct ShortBrickPlan {
  bool decided = false;   // begin() has run for this upload
  uint64_t B = 0;         // events of a full brick
  uint64_t zone0 = 0;     // first event (global) that may be converted
  uint64_t left = 0;      // conversions (three full bricks -> four short ones) still to place
  void begin(uint64_t expect_events, uint64_t first_chunk_events, int G, int n_cu);
  void take(uint64_t chunk_first, uint64_t chunk_events, uint64_t* from, uint64_t* to);
};

// keyed chunks held back until enough events are together (m3hip_upload.hip, hold_back_keyed): a chunk in the reference's flat
// layout, owned by the engine
struct KeyedChunks {
  uint64_t n_events = 0;
  std::vector<float> coeff_many, coeff_tf1;
  std::vector<unsigned int> nknots_arr, npe, npe_tf1;
  std::vector<int16_t> param_no, param_no_tf1;
  std::vector<int32_t> key;
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
  // m3hip_set_osc_table_device(_f64): the current table is the CALLER's device array, read where a GPU calculator left it (no
  // copy; ordered by the handle's stream).  Null: the engine's own copy (d_osc_table / d_osc_table_d) is the current one.
  const float* d_osc_ext = nullptr;
  const double* d_osc_ext_d = nullptr;
  uint64_t osc_line_sum = 0, osc_line_blocks = 0;  // distinct table lines per 1024 consecutive events, summed over the uploaded blocks (m3hip_info)
  const float* osc_f32() const { return d_osc_ext ? d_osc_ext : d_osc_table; }
  const double* osc_f64() const { return d_osc_ext_d ? d_osc_ext_d : d_osc_table_d; }
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
  // the handler's tables in the engine's layout (BinnedArgs, m3hip_kernels.h): non-flat bin-splines in blocks, sorted by parameter
  // inside a block, coefficients and knots segment-major per block
  int b_block = 0, b_nseg = 0;
  int16_t* d_b_par = nullptr;
  int32_t* d_b_target = nullptr;
  float* d_b_xk = nullptr;
  float4* d_b_many = nullptr;
  float* d_b_w = nullptr;
  // M3::float_t = double build of the sample side (m3hip_binned_upload_f64)
  bool f64 = false;
  double* d_b_xk_d = nullptr;
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
  // the event side of a binned-only, shared-list sample in processing order (PosArgs; k_pack_positions), for k_reweight_lists
  int32_t *d_pos_bin = nullptr, *d_pos_osc = nullptr;
  uint32_t* d_pos_loff = nullptr;
  uint16_t* d_pos_llen = nullptr;
  bool pos_dirty = true;          // a per-event array (bin, osc, list id), the lists or the processing order changed since they were packed
  int sw_lists = -1;              // M3HIP_LISTS_KERNEL = 0: keep such samples on the uniform-tile kernel (tests, A/B)
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
  double cached_mb = 200.0;       // what the engine plans to keep in the Infinity Cache from step to step (of its 256 MB)
  uint64_t n_grouped_chunks = 0;
  // m3hip_monolith_order_key: chunks the engine laid out in the order of a caller-supplied key (stable sort inside the chunk).
  // EVERYTHING on the device is in that physical order -- the kernels know nothing of it; the API translates: per-event uploads
  // are scattered and per-event results gathered through c2p (caller's event number -> physical event).  Empty: identity.
  std::vector<int32_t> next_key;  // key of the next m3hip_monolith_append (which takes it)
  KeyedChunks keyed_acc;          // keyed chunks held back until 2^20 events are together
  std::vector<std::pair<uint64_t, std::vector<uint32_t>>> keyed;  // (first event, physical position -> caller's event, chunk-local) of every keyed chunk, until monolith_end
  std::vector<uint32_t> h_c2p;
  uint32_t* d_c2p = nullptr;
  uint64_t n_keyed_chunks = 0, n_key_refused = 0;
  void* d_xfer = nullptr;         // device staging of the translated transfers
  size_t xfer_cap = 0;
  bool osc_lines_stale = false;   // osc_line_sum / _blocks are to be re-counted from d_osc (physical order) by m3hip_get_info

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

// ---- helpers shared by the translation units (defined in m3hip_api.hip unless noted)
inline double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
int fail(m3hip_handle h, int code, const std::string& msg);

#define HIPCHK(h, expr)                                                                              \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess)                                                                            \
      return fail(h, e_ == hipErrorOutOfMemory ? M3HIP_ENOMEM : M3HIP_EHIP,                          \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                                \
  } while (0)

// No C++ exception may leave an extern "C" entry point (the callers are C, ctypes, cgo ...): every entry point is a
// function-try-block whose handler lands here and turns what was thrown into an error code + message.
int api_exception(m3hip_handle h) noexcept;
#define M3_API_CATCH(h) catch (...) { return api_exception(h); }

template <class T>
int dev_alloc(m3hip_handle h, T** p, size_t n) {
  *p = nullptr;
  if (n == 0) n = 1;
  HIPCHK(h, hipMalloc((void**)p, n * sizeof(T)));
  return M3HIP_OK;
}
void dev_free(void* p);
hipError_t wait_stream(hipStream_t s);
void collect_timing(m3hip_handle h);
int ensure_param_capacity(m3hip_handle h, int n_par);
void free_monolith(m3hip_handle h);                     // m3hip_upload.hip
void par_free(m3hip_handle h, m3hip_ctx::ParDev& p);    // m3hip_par.hip
BinArgs make_bin_args(m3hip_handle h);                  // m3hip_sample.hip
int bins_from_word(m3hip_handle h);                     // m3hip_sample.hip
// Per-event transfers in the CALLER's event order (m3hip_sample.hip): elements [off, off + n) of an array over the events, rows of
// `width` elements of `elem` bytes (2, 4 or 8).  Plain copies unless m3hip_monolith_order_key re-ordered a chunk (m3hip_ctx::h_c2p).
int put_events(m3hip_handle h, void* dst_base, uint64_t off, const void* src, uint64_t n, size_t elem, uint32_t width = 1);
int get_events(m3hip_handle h, void* out, const void* src_base, uint64_t off, uint64_t n, size_t elem);
int gather_events_async(m3hip_handle h, const void* src_base, uint64_t n, size_t elem);  // all n events, caller's order -> m3hip_ctx::d_xfer, on the handle's stream (c2p non-empty)
bool evw_fits(m3hip_handle h);                          // m3hip_sample.hip
