// m3hip_step.hip -- the per-step path of include/m3hip.h: seam 2's m3hip_eval_splines, m3hip_reweight (parameter staging, k_prep, re-binning passes, the
// reweight kernels, the exchange), m3hip_get_likelihood, m3hip_step, and the results a caller reads back.
#include "m3hip_ctx.h"

namespace {

constexpr int kOscStaticSteps = 2;        // reweights an oscillation table must have been left alone before it counts as pre-staged
constexpr int kInlineResolveMax = 8192;  // unique slots resolved inside k_prep; beyond that k_resolve runs

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
// M3HIP_NT_MASK = 0 | 4 forces the choice (tests).
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
  a.osc_table = h->osc_f32();
  a.extras = h->d_extras;
  a.n_extra = uint32_t(h->n_extra);
  a.bwidx = h->d_bwidx;
  a.bl_id = h->bl_mode == 1 ? h->d_bl_id : nullptr;
  a.perm = (h->perm_monolith || (h->bl_mode == 1 && h->S + h->T == 0)) ? h->d_perm : nullptr;
  a.bl_off = h->d_bl_off;
  a.bl_idx = h->d_bl_idx;
  a.bw = h->d_b_w;
  a.bw_d = h->d_b_w_d;
  a.osc_table_d = h->osc_f64();
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
  if (cpu_total_weights) {  // gpuSplineUtils.cu:505 cudaMemcpyAsync(cpu_total_weights, gpu_total_weights, ...)
    const float* src = h->d_spline_w;
    if (!h->h_c2p.empty()) {  // an upload the engine re-ordered (m3hip_monolith_order_key): back into the caller's event order first
      if (int rc_g = gather_events_async(h, h->d_spline_w, h->E, sizeof(float))) return rc_g;
      src = static_cast<const float*>(h->d_xfer);
    }
    HIPCHK(h, hipMemcpyAsync(cpu_total_weights, src, sizeof(float) * h->E, hipMemcpyDeviceToHost, h->stream));
  }
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_sync(m3hip_handle h) try {
  if (!h) return M3HIP_EINVAL;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, wait_stream(h->stream));
  collect_timing(h);
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
    memset(&ba, 0, sizeof(ba));
    ba.par = h->d_b_par;
    ba.target = h->d_b_target;
    ba.many = h->d_b_many;
    ba.xk = h->d_b_xk;
    ba.many_d = h->d_b_many_d;
    ba.xk_d = h->d_b_xk_d;
    ba.ptab = h->d_ptab;
    ba.pval = h->d_pval;
    ba.weightvec = h->d_b_w;
    ba.weightvec_d = h->d_b_w_d;
    ba.params = h->d_params;
    ba.par_map = h->d_parmap;
    ba.n_unique = h->b_U;
    ba.block = h->b_block;
    ba.n_seg = h->b_nseg;
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
                         int32_t(h->max_norm_idx_evw < 0 ? 0 : h->min_norm_idx), h->d_evw, want_weight ? h->osc_f32() : nullptr, h->stream);
      h->evw_weight = want_weight;
    }
    h->evw_dirty = false;
  }
  const bool word_valid = !h->f64 && !h->h_bricks.empty() && h->evw_ok && packed_fit && h->d_evw;
  // Functional shifts: every event is re-binned from its shifted kinematics (ApplyShifts + FindBin).  Uniform tiles do it in
  // their kernel's epilogue.  Samples with bricks (and the float_t = double variant) by a pass of its own in front of the
  // reweight kernel: with the binning tables in LDS when they fit (uniform samples, <= 2048 edges) and, when the packed words
  // exist, straight into their bin field -- 8 B of kinematics + the word, read and written, per event; k_rebin_pack /
  // k_find_bins from global tables otherwise (M3HIP_REBIN=global forces those: tests).
  const bool dyn_any = !h->h_shifts.empty();
  bool nonuniform = false;
  for (const BinSample& bs : h->bb_samples) nonuniform = nonuniform || !bs.uniform;
  const bool lds_ok = rebin_tables_fit_lds(int(h->bb_samples.size()), int(h->bb_axes.size()), int(h->bb_edges.size()), int(h->h_shifts.size()),
                                           int(h->h_cuts.size()), nonuniform);
  const bool use_lds = lds_ok && h->sw_rebin != 'g';
  const bool dyn_pass = dyn_any && (h->f64 || !h->h_bricks.empty());
  const bool dyn_bins = dyn_any && !dyn_pass;  // fused into the uniform-tile kernel's epilogue
  if (dyn_pass) {
    BinArgs b = make_bin_args(h);
    b.compute_nominal = 0;
    b.word_bin_mask = h->evw_weight ? kEvwBinMaskWeight : kEvwBinMask;
    if (word_valid && h->h_tiles.empty()) {  // bricks only: nothing but the words needs the bins
      if (use_lds) {
        if (h->sw_rebin != 'l' && rebin_fits_1d(h->bb_samples.data(), int(h->bb_samples.size()), h->bb_axes.data(), h->h_shifts.data(), int(h->h_shifts.size()),
                                                h->h_cuts.data(), int(h->h_cuts.size()), h->kin_stride))  // (M3HIP_REBIN=lds: the general pass; tests)
          launch_rebin_1d(b, h->bb_samples[0], h->bb_axes[size_t(h->bb_samples[0].axis_off)], h->h_shifts.data(), h->h_cuts.data(), h->d_evw, h->stream);
        else
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
                           int32_t(h->max_norm_idx_evw < 0 ? 0 : h->min_norm_idx), h->d_evw, h->evw_weight ? h->osc_f32() : nullptr, h->stream);
    }
  }
  const double t_c = now_s();

  if (h->n_osc) {  // the oscillation weights in the precision this build of the sample side multiplies in
    // (the source may be the caller's device array, m3hip_set_osc_table_device; the converted copy is the engine's own)
    if (h->f64 && !h->osc_valid_f64) {
      launch_osc_convert(h->osc_f32(), nullptr, h->d_osc_table_d, h->n_osc, true, h->stream);
      h->osc_valid_f64 = true;
    } else if (!h->f64 && !h->osc_valid_f32) {
      launch_osc_convert(nullptr, h->osc_f64(), h->d_osc_table, h->n_osc, false, h->stream);
      h->osc_valid_f32 = true;
    }
  }
  // Engines whose brick kernel reads the UNPACKED sample side (a sample that does not fit the word's fields, or a bin space of
  // 2^14 and more): a large static table is gathered ONCE into a per-event column the kernel streams (4 B per event) -- it does not
  // stay in L2 beside the streams, and every event's gather from it then goes to memory (round 4, one GPU's share of config 5:
  // 5.5 % of the kernel).  With the packed word the weight travels in the word itself (above).
  if (osc_static_counts && !(word_valid && h->evw_weight)) {
    const bool large = sizeof(float) * h->n_osc > 64 * 1024;
    // (a table that MOVES is gathered from in the kernel's tail: expanding it into the column every step measured slower -- 60 us of
    // pass + 466 us of kernel against 486 us on one GPU's share of config 5, scripts/experiments/README.md.  What makes a moving
    // table free is the MC's order: SampleHandlerFD::EventOrderByOscillationBin)
    if (!h->oscw_valid && ((h->oscw_mode == 1) || (large && osc_is_static))) {
      if (!h->d_oscw && (rc = dev_alloc(h, &h->d_oscw, size_t(h->E_pad)))) return rc;
      launch_osc_expand(h->d_osc, h->osc_f32(), h->d_oscw, h->E_pad, h->stream);
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
      h->pos_dirty = true;
    }
  }
  // A binned-only sample whose events share their pointer lists (what MaCh3Tutorial's far-detector sample is) runs k_reweight_lists
  // on the event side in processing order; packed here whenever a per-event array, the lists or the order changed (set-up time).
  // Functional shifts re-bin inside the uniform-tile kernel's epilogue, the deterministic fill orders its waves there: both keep it.
  const bool use_lists = h->bl_mode == 1 && h->S + h->T == 0 && h->h_bricks.empty() && h->d_perm && h->d_bl_id && h->d_bl_off && h->h_shifts.empty() &&
                         !h->deterministic && h->sw_lists != 0;
  if (use_lists && h->pos_dirty) {
    if (!h->d_pos_bin && (rc = dev_alloc(h, &h->d_pos_bin, size_t(h->E_pad)))) return rc;
    if (!h->d_pos_osc && (rc = dev_alloc(h, &h->d_pos_osc, size_t(h->E_pad)))) return rc;
    if (!h->d_pos_loff && (rc = dev_alloc(h, &h->d_pos_loff, size_t(h->E_pad)))) return rc;
    if (!h->d_pos_llen && (rc = dev_alloc(h, &h->d_pos_llen, size_t(h->E_pad)))) return rc;
    launch_pack_positions(h->d_perm, h->d_bin, h->d_osc, h->d_bl_id, h->d_bl_off, h->E, h->d_pos_bin, h->d_pos_osc, h->d_pos_loff, h->d_pos_llen, h->stream);
    h->pos_dirty = false;
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
  if (use_lists) {
    PosArgs pa;
    pa.ev = h->d_perm;
    pa.bin = h->d_pos_bin;
    pa.osc = h->d_pos_osc;
    pa.loff = h->d_pos_loff;
    pa.llen = h->d_pos_llen;
    pa.n_pos = h->E;
    launch_reweight_lists(a, pa, mode, h->f64, ku, h->stream);
  } else if (h->f64) {
    launch_reweight_uniform_f64(a, n_tiles, mode, ku, h->stream);
  } else {
    launch_reweight_uniform(a, n_tiles, mode, ku, h->stream, dyn_bins);
    launch_reweight_brick(a, n_bricks, mode, kb, h->stream);
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
  if (int rc_g = get_events(h, out, h->d_spline_w, event_offset, n, sizeof(float))) return rc_g;
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
  if (int rc_g = get_events(h, out, h->d_event_w, event_offset, n, sizeof(float))) return rc_g;
  return M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_get_event_weights_f64(m3hip_handle h, double* out, uint64_t event_offset, uint64_t n) try {
  if (!h || !out) return M3HIP_EINVAL;
  if (!h->mono_done || event_offset + n > h->E) return fail(h, M3HIP_EINVAL, "get_event_weights_f64: bad range");
  if (!h->d_event_w_d) return fail(h, M3HIP_ESTATE, "get_event_weights_f64: needs m3hip_binned_upload_f64 and m3hip_record_event_weights(h, 1) before the reweight");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  collect_timing(h);
  if (int rc_g = get_events(h, out, h->d_event_w_d, event_offset, n, sizeof(double))) return rc_g;
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

}  // extern "C"
