// m3hip_upload.hip -- the one-time upload of include/m3hip.h: the flattened monolith in event chunks (re-layout through
// m3hip_layout.cpp, grouping by parameter list, short bricks), seam 2's evaluation, binned splines, the host-only probes.
#include "m3hip_ctx.h"

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
  dev_free(h->d_b_par); h->d_b_par = nullptr;
  dev_free(h->d_b_target); h->d_b_target = nullptr;
  dev_free(h->d_b_xk); h->d_b_xk = nullptr;
  dev_free(h->d_b_many); h->d_b_many = nullptr;
  dev_free(h->d_b_w); h->d_b_w = nullptr;
  dev_free(h->d_b_xk_d); h->d_b_xk_d = nullptr;
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
  dev_free(h->d_pos_bin); h->d_pos_bin = nullptr;
  dev_free(h->d_pos_osc); h->d_pos_osc = nullptr;
  dev_free(h->d_pos_loff); h->d_pos_loff = nullptr;
  dev_free(h->d_pos_llen); h->d_pos_llen = nullptr;
  h->pos_dirty = true;
  h->grouped.clear();
  h->keyed.clear();
  h->next_key.clear();
  h->keyed_acc = KeyedChunks();
  h->h_c2p.clear();
  dev_free(h->d_c2p); h->d_c2p = nullptr;
  h->n_keyed_chunks = h->n_key_refused = 0;
  h->osc_lines_stale = false;
  h->osc_line_sum = h->osc_line_blocks = 0;
  h->perm_monolith = false;
  h->n_grouped_chunks = 0;
  h->brick_groups = 0;
  h->h_bl_id.clear();
  h->perm_dirty = false;
  h->bl_mode = -1;
  h->bl_map.clear(); h->bl_off.clear(); h->bl_idx.clear();
  h->b_M = h->b_U = h->b_C = 0;
  h->b_block = h->b_nseg = 0;
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

// The handler's tables in the engine's layout (BinnedArgs, m3hip_kernels.h).  T = float / double.
template <class T>
static int binned_planes(m3hip_handle h, int n_params, int max_knots, const int16_t* npts, const int32_t* coeffindexvec, const int32_t* uniquesplinevec,
                         int n_unique, const int32_t* uniquecoeffindices, const T* xcoeff, const T* manycoeff, T** d_xk, void** d_many) {
  const int n_seg = std::max(1, max_knots - 1);
  // a block long enough for a parameter's run inside it to span whole waves (the quads of a run are one segment plane: contiguous),
  // short enough for the block's window of the weight vector to stay in L2 while its workgroups write into it
  int n_used = 0;
  {
    std::vector<char> used(size_t(n_params), 0);
    for (int u = 0; u < n_unique; ++u) used[size_t(uniquesplinevec[uniquecoeffindices[u]])] = 1;
    for (char c : used) n_used += c;
  }
  int block = std::min(65536, std::max(4096, 128 * std::max(1, n_used)));
  block = (block + 255) / 256 * 256;
  if (n_unique < block) block = std::max(256, (n_unique + 255) / 256 * 256);
  const size_t n_blocks = (size_t(n_unique) + size_t(block) - 1) / size_t(block);
  const size_t plane_len = n_blocks * size_t(n_seg) * size_t(block);
  // Every bin-spline gets max_knots - 1 segment planes, whatever its own knot count (a short one repeats its last segment): the
  // store is 5 * sizeof(T) * (max_knots - 1) bytes per non-flat bin-spline, on the device and once more in the staging vectors
  // below -- a handler that mixes a few many-knot parameters with many three-knot ones pays the many-knot price for all of them.
  // Said here rather than found out by a failed allocation half-way through.
  {
    size_t free_b = 0, total_b = 0;
    const size_t need = plane_len * 5 * sizeof(T);
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > free_b / 10 * 9)
      return fail(h, M3HIP_ENOMEM, "binned_upload: the segment-major tables need " + std::to_string(need >> 20) + " MiB (" + std::to_string(5 * sizeof(T) * size_t(n_seg)) +
                                       " B per non-flat bin-spline = 5 x sizeof(float_t) x (max_knots - 1), max_knots = " + std::to_string(max_knots) + ", " +
                                       std::to_string(n_unique) + " bin-splines), the device has " + std::to_string(free_b >> 20) + " MiB free");
  }
  std::vector<int16_t> par(static_cast<size_t>(n_unique));
  std::vector<int32_t> target(static_cast<size_t>(n_unique));
  std::vector<T> xk(plane_len, T(0)), many(plane_len * 4, T(0));
  std::vector<int32_t> ord;
  for (size_t b = 0; b < n_blocks; ++b) {
    const int j0 = int(b * size_t(block)), j1 = std::min(n_unique, j0 + block);
    ord.resize(size_t(j1 - j0));
    for (int j = j0; j < j1; ++j) ord[size_t(j - j0)] = j;
    std::stable_sort(ord.begin(), ord.end(), [&](int32_t x, int32_t y) { return uniquesplinevec[uniquecoeffindices[x]] < uniquesplinevec[uniquecoeffindices[y]]; });
    for (int jj = 0; jj < j1 - j0; ++jj) {
      const int i = uniquecoeffindices[ord[size_t(jj)]], p = uniquesplinevec[i];
      const size_t q = size_t(j0) + size_t(jj);
      par[q] = int16_t(p);
      target[q] = i;
      const int segs = npts[p] - 1;  // (checked by the caller: npts >= 2)
      for (int k = 0; k < n_seg; ++k) {
        const size_t src = size_t(coeffindexvec[i]) + size_t(std::min(k, segs - 1)), at = (b * size_t(n_seg) + size_t(k)) * size_t(block) + size_t(jj);
        xk[at] = xcoeff[src];
        for (int c = 0; c < 4; ++c) many[4 * at + size_t(c)] = manycoeff[4 * src + size_t(c)];
      }
    }
  }
  int rc;
  if ((rc = dev_alloc(h, &h->d_b_par, size_t(n_unique)))) return rc;
  if ((rc = dev_alloc(h, &h->d_b_target, size_t(n_unique)))) return rc;
  if ((rc = dev_alloc(h, d_xk, plane_len))) return rc;
  T* dm = nullptr;
  if ((rc = dev_alloc(h, &dm, plane_len * 4))) return rc;
  *d_many = dm;
  if (n_unique) {
    HIPCHK(h, hipMemcpy(h->d_b_par, par.data(), sizeof(int16_t) * par.size(), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_b_target, target.data(), sizeof(int32_t) * target.size(), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(*d_xk, xk.data(), sizeof(T) * xk.size(), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(dm, many.data(), sizeof(T) * many.size(), hipMemcpyHostToDevice));
  }
  h->b_block = block;
  h->b_nseg = n_seg;
  return M3HIP_OK;
}

extern "C" {

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
  { const char* f = getenv("M3HIP_LISTS_KERNEL"); h->sw_lists = f && f[0] ? atoi(f) : -1; }
  h->upload_expect = h->expect_events;  // the hint belongs to this upload, whether it completes or not
  h->short_bricks = ShortBrickPlan();
  { const char* f = getenv("M3HIP_NT_MASK"); h->nt_force = f && f[0] ? atoi(f) : -1; }
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

int m3hip_monolith_order_key(m3hip_handle h, const int32_t* key, uint64_t n_events) try {
  if (!h) return M3HIP_EINVAL;
  if (!h->mono_open) return fail(h, M3HIP_ESTATE, "monolith_order_key outside monolith_begin / monolith_end");
  if (!key && n_events) return fail(h, M3HIP_EINVAL, "monolith_order_key: NULL key");
  h->next_key.assign(key, key + n_events);
  return M3HIP_OK;
} M3_API_CATCH(h)

// One chunk through the re-layout and onto the device.  `key`: the order key of its events (m3hip_monolith_order_key) or null.
static int append_chunk(m3hip_handle h, ChunkInput in, const std::vector<int32_t>* key) {
  const uint64_t n_events = in.n_events;
  const int16_t *param_no = in.param_no, *param_no_tf1 = in.param_no_tf1;
  const unsigned int *n_param_per_event = in.npe, *n_param_per_event_tf1 = in.npe_tf1;  // (in the caller's event order, whatever `in` turns into below)
  HIPCHK(h, hipSetDevice(h->device));
  ChunkPlan plan;
  std::string err;
  { const char* f = getenv("M3HIP_PAD_MIN_FILL"); g_pad_min_fill = f ? atof(f) : kPadMinFillDefault; }  // experiments: > 1 disables the identity padding
  { const char* f = getenv("M3HIP_PAD_RUNS"); g_pad_runs = f ? atoi(f) : 2; }
  // G of the bricks, fixed by the first chunk (all bricks of an engine share one kernel instantiation)
  if (h->brick_groups == 0) h->brick_groups = choose_brick_groups(in, h->upload_expect);
  const int G = h->brick_groups;
  if (!h->short_bricks.decided) h->short_bricks.begin(h->upload_expect, n_events, G, h->n_cu);
  uint64_t short_from, short_to;  // chunk-local
  h->short_bricks.take(h->E, n_events, &short_from, &short_to);
  if (!plan_chunk(h->P, h->K, h->h_npts.data(), in, &plan, &err, G, short_from, short_to)) return fail(h, M3HIP_EINVAL, "monolith_append: " + err);

  // A candidate lay-out of the chunk: the {count,start} pairs in some order of the events (they point into the flat response
  // arrays, which stay as they are) and its plan.
  struct Candidate {
    ChunkInput in;
    ChunkPlan plan;
    std::vector<unsigned int> npe, npe_tf1;  // storage behind in.npe / in.npe_tf1 once the events were permuted
    std::vector<uint32_t> group_order;       // processing position -> event of `in`, when the grouping below was taken
    void permute(const std::vector<uint32_t>& ord) {  // events of `in` re-ordered: new event pos = old event ord[pos]
      std::vector<unsigned int> a(2 * ord.size()), b(2 * ord.size());
      for (size_t pos = 0; pos < ord.size(); ++pos) {
        const uint32_t e = ord[pos];
        a[2 * pos] = in.npe[2 * e];
        a[2 * pos + 1] = in.npe[2 * e + 1];
        b[2 * pos] = in.npe_tf1[2 * e];
        b[2 * pos + 1] = in.npe_tf1[2 * e + 1];
      }
      npe.swap(a);
      npe_tf1.swap(b);
      in.npe = npe.data();
      in.npe_tf1 = npe_tf1.data();
    }
  };
  auto bytes_of = [](const ChunkPlan& p) { return double(p.spl_float4) * 16 + double(p.tf1_float2) * 8 + double(p.meta_int32) * 4 + double(p.brick_words.size()) * 4; };
  // Grouping.  When a sizeable part of the chunk would end up in bricks, try the chunk with its events grouped by
  // parameter list (stable sort): events of one interaction mode then sit together and form uniform (or padded)
  // tiles.  The kernels map a lane position back to its event through the permutation (ReweightArgs::perm), so the
  // per-event arrays stay in event order.
  auto group = [&](Candidate& c) {
    uint64_t ragged = 0;
    for (const BrickHost& bh : c.plan.bricks) ragged += bh.n_events;
    const char* f = getenv("M3HIP_GROUP_EVENTS");
    const bool enabled = !(f && f[0] == '0');
    if (!enabled || ragged * 10 <= n_events) return;
    const unsigned int *npe_in = c.in.npe, *npe_tf1_in = c.in.npe_tf1;
    std::vector<uint32_t> ord(n_events);
    for (uint64_t e = 0; e < n_events; ++e) ord[e] = uint32_t(e);
    auto less = [&](uint32_t x, uint32_t y) {  // lexicographic on (spline parameters, TF1 parameters)
      const unsigned int cx = npe_in[2 * x], sx = npe_in[2 * x + 1];
      const unsigned int cy = npe_in[2 * y], sy = npe_in[2 * y + 1];
      if (std::lexicographical_compare(param_no + sx, param_no + sx + cx, param_no + sy, param_no + sy + cy)) return true;
      if (std::lexicographical_compare(param_no + sy, param_no + sy + cy, param_no + sx, param_no + sx + cx)) return false;
      const unsigned int tx = npe_tf1_in[2 * x], ux = npe_tf1_in[2 * x + 1];
      const unsigned int ty = npe_tf1_in[2 * y], uy = npe_tf1_in[2 * y + 1];
      return std::lexicographical_compare(param_no_tf1 + ux, param_no_tf1 + ux + tx, param_no_tf1 + uy, param_no_tf1 + uy + ty);
    };
    std::stable_sort(ord.begin(), ord.end(), less);
    Candidate g;
    g.in = c.in;
    g.permute(ord);
    std::string err_g;
    if (!plan_chunk(h->P, h->K, h->h_npts.data(), g.in, &g.plan, &err_g, G, short_from, short_to)) return;
    uint64_t ragged_g = 0;
    for (const BrickHost& bh : g.plan.bricks) ragged_g += bh.n_events;
    if (ragged_g * 2 > ragged) return;  // (taken when it at least halves what goes through bricks)
    c.npe.swap(g.npe);
    c.npe_tf1.swap(g.npe_tf1);
    c.in.npe = c.npe.data();
    c.in.npe_tf1 = c.npe_tf1.data();
    c.plan = std::move(g.plan);
    c.group_order = std::move(ord);
  };
  Candidate cand;
  cand.in = in;
  cand.plan = std::move(plan);
  group(cand);
  // Order key (m3hip_monolith_order_key).  The chunk's events are STORED in the order of the caller's key (a stable sort: a chunk
  // already in order stays as it is), then grouped as above (stable again: inside a group the key's order stays) -- unless that
  // lay-out is more than 2 % larger than the one of the caller's order (a key that tears runs of equal parameter lists apart in
  // a way the grouping does not mend; m3hip_info::n_key_refused counts those chunks).  From here on "event" means the physical
  // one: the device never sees the caller's numbering, the API's per-event transfers translate (put_events / get_events).
  std::vector<uint32_t> key_order;
  if (key) {
    std::vector<uint32_t> ord(n_events);
    for (uint64_t e = 0; e < n_events; ++e) ord[e] = uint32_t(e);
    std::stable_sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) { return uint32_t((*key)[x]) < uint32_t((*key)[y]); });  // (negative keys last)
    bool moved = false;
    for (uint64_t pos = 0; pos < n_events && !moved; ++pos) moved = ord[pos] != pos;
    if (moved) {
      Candidate k;
      k.in = in;
      k.permute(ord);
      std::string err_k;
      bool ok = plan_chunk(h->P, h->K, h->h_npts.data(), k.in, &k.plan, &err_k, G, short_from, short_to);
      if (ok) group(k);
      if (ok && bytes_of(k.plan) <= 1.02 * bytes_of(cand.plan)) {
        cand.npe.swap(k.npe);
        cand.npe_tf1.swap(k.npe_tf1);
        cand.in = k.in;
        cand.in.npe = cand.npe.data();
        cand.in.npe_tf1 = cand.npe_tf1.data();
        cand.plan = std::move(k.plan);
        cand.group_order = std::move(k.group_order);
        key_order = std::move(ord);
      } else {
        h->n_key_refused++;
      }
    }
  }
  in = cand.in;
  plan = std::move(cand.plan);
  std::vector<uint32_t> order = std::move(cand.group_order);
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
  if (!key_order.empty()) {
    h->n_keyed_chunks++;
    h->keyed.emplace_back(h->E, std::move(key_order));
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
    bd.pad_[1] = (bh.tf1_sorted ? 1u : 0u) | (bh.tf1_cnt_in_sigma ? 2u : 0u);
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
}

// Chunks that come with an order key are held back -- re-packed into the engine's own host arrays, because the caller's are only
// good for the duration of the call -- until 2^20 events are together, and go through append_chunk as ONE chunk: the key orders the
// events of a chunk, and how well the neighbours of an oscillation-table line end up together depends on how many events that is
// (one GPU's share of BASELINE configs[4]: chunks of 131 072 events leave 51 table lines per 1024 stored events, 2^20 leave 7),
// not on how the caller happened to slice its upload.  A chunk of that size by itself goes straight through.
constexpr uint64_t kKeyedChunkEvents = uint64_t(1) << 20;
static int flush_keyed(m3hip_handle h) {
  KeyedChunks& a = h->keyed_acc;
  if (a.n_events == 0) return M3HIP_OK;
  ChunkInput in;
  in.n_events = a.n_events;
  in.coeff_many = a.coeff_many.data();
  in.total_nknots = a.coeff_many.size() / 4;
  in.nknots_arr = a.nknots_arr.data();
  in.param_no = a.param_no.data();
  in.n_splines = a.param_no.size();
  in.coeff_tf1 = a.coeff_tf1.data();
  in.param_no_tf1 = a.param_no_tf1.data();
  in.n_tf1 = a.param_no_tf1.size();
  in.npe = a.npe.data();
  in.npe_tf1 = a.npe_tf1.data();
  const int rc = append_chunk(h, in, &a.key);
  a = KeyedChunks();  // (whatever happened: a chunk that failed leaves no trace)
  return rc;
}
// the responses of a chunk's events, copied out of the caller's arrays (which may be those of a larger sample: only what the
// {count,start} pairs of THESE events index is taken) behind what is already held back
static int hold_back_keyed(m3hip_handle h, const ChunkInput& in, const std::vector<int32_t>& key) {
  KeyedChunks& a = h->keyed_acc;
  const size_t s0 = a.param_no.size(), t0 = a.param_no_tf1.size(), k0 = a.coeff_many.size() / 4;
  size_t s_add = 0, t_add = 0, k_add = 0;
  for (uint64_t e = 0; e < in.n_events; ++e) {
    const uint64_t c = in.npe[2 * e], st = in.npe[2 * e + 1], ct = in.npe_tf1[2 * e], stt = in.npe_tf1[2 * e + 1];
    if (st + c > in.n_splines || stt + ct > in.n_tf1) return fail(h, M3HIP_EINVAL, "monolith_append: an event's {count,start} pair points past the response arrays");
    for (uint64_t i = 0; i < c; ++i) {
      const int p = in.param_no[st + i];
      if (p < 0 || p >= h->P || h->h_npts[size_t(p)] < 1) return fail(h, M3HIP_EINVAL, "monolith_append: a spline names a parameter without knots");
      if (uint64_t(in.nknots_arr[st + i]) + uint64_t(h->h_npts[size_t(p)]) > in.total_nknots) return fail(h, M3HIP_EINVAL, "monolith_append: a spline's knots lie past the coefficient array");
      k_add += size_t(h->h_npts[size_t(p)]);
    }
    s_add += c;
    t_add += ct;
  }
  if ((k0 + k_add) > 0xFFFFFFFFull || (s0 + s_add) > 0xFFFFFFFFull || (t0 + t_add) > 0xFFFFFFFFull)
    return fail(h, M3HIP_EINVAL, "monolith_append: more than 2^32 coefficients in one chunk of keyed events");
  a.param_no.reserve(s0 + s_add);
  a.nknots_arr.reserve(s0 + s_add);
  a.coeff_many.reserve(4 * (k0 + k_add));
  a.param_no_tf1.reserve(t0 + t_add);
  a.coeff_tf1.reserve(2 * (t0 + t_add));
  a.npe.reserve(a.npe.size() + 2 * in.n_events);
  a.npe_tf1.reserve(a.npe_tf1.size() + 2 * in.n_events);
  for (uint64_t e = 0; e < in.n_events; ++e) {
    const uint64_t c = in.npe[2 * e], st = in.npe[2 * e + 1], ct = in.npe_tf1[2 * e], stt = in.npe_tf1[2 * e + 1];
    a.npe.push_back(static_cast<unsigned int>(c));
    a.npe.push_back(static_cast<unsigned int>(a.param_no.size()));
    for (uint64_t i = 0; i < c; ++i) {
      const int p = in.param_no[st + i];
      const size_t nk = size_t(h->h_npts[size_t(p)]);
      a.param_no.push_back(int16_t(p));
      a.nknots_arr.push_back(static_cast<unsigned int>(a.coeff_many.size() / 4));
      const float* src = in.coeff_many + 4 * size_t(in.nknots_arr[st + i]);
      a.coeff_many.insert(a.coeff_many.end(), src, src + 4 * nk);
    }
    a.npe_tf1.push_back(static_cast<unsigned int>(ct));
    a.npe_tf1.push_back(static_cast<unsigned int>(a.param_no_tf1.size()));
    for (uint64_t i = 0; i < ct; ++i) {
      a.param_no_tf1.push_back(in.param_no_tf1[stt + i]);
      a.coeff_tf1.push_back(in.coeff_tf1[2 * (stt + i)]);
      a.coeff_tf1.push_back(in.coeff_tf1[2 * (stt + i) + 1]);
    }
  }
  a.key.insert(a.key.end(), key.begin(), key.end());
  a.n_events += in.n_events;
  return M3HIP_OK;
}

int m3hip_monolith_append(m3hip_handle h, uint64_t n_events, const float* coeff_many, uint64_t total_nknots,
                          const unsigned int* nknots_arr, const int16_t* param_no, uint64_t n_splines,
                          const float* coeff_tf1, const int16_t* param_no_tf1, uint64_t n_tf1,
                          const unsigned int* n_param_per_event, const unsigned int* n_param_per_event_tf1) try {
  if (!h) return M3HIP_EINVAL;
  if (!h->mono_open) return fail(h, M3HIP_ESTATE, "monolith_append before monolith_begin");
  const uint64_t held = h->keyed_acc.n_events;
  if ((h->E + held) % M3_GROUP != 0) return fail(h, M3HIP_EINVAL, "only the last chunk may hold a non-multiple of 1024 events");
  if (h->E + held + n_events > 0xFFFFF000ull) return fail(h, M3HIP_EINVAL, "more than 2^32 events on one device");
  if (n_events == 0) return M3HIP_OK;
  if ((n_splines && (!coeff_many || !nknots_arr || !param_no)) || (n_tf1 && (!coeff_tf1 || !param_no_tf1)) || !n_param_per_event || !n_param_per_event_tf1)
    return fail(h, M3HIP_EINVAL, "monolith_append: NULL coefficient arrays");
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
  if (h->next_key.empty()) {  // no key: whatever is held back goes first (the events keep their places)
    if (int rc = flush_keyed(h)) return rc;
    return append_chunk(h, in, nullptr);
  }
  const std::vector<int32_t> key = std::move(h->next_key);
  h->next_key.clear();
  if (key.size() != n_events) return fail(h, M3HIP_EINVAL, "monolith_append: the order key was given for another number of events");
  if (held == 0 && n_events >= kKeyedChunkEvents) return append_chunk(h, in, &key);
  if (int rc = hold_back_keyed(h, in, key)) return rc;
  return h->keyed_acc.n_events >= kKeyedChunkEvents ? flush_keyed(h) : M3HIP_OK;
} M3_API_CATCH(h)

int m3hip_monolith_end(m3hip_handle h) try {
  if (!h) return M3HIP_EINVAL;
  if (!h->mono_open) return fail(h, M3HIP_ESTATE, "monolith_end before monolith_begin");
  HIPCHK(h, hipSetDevice(h->device));
  if (int rc_k = flush_keyed(h)) return rc_k;
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
  if (!h->keyed.empty()) {  // caller's event -> physical event; identity over the chunks that kept their order
    h->h_c2p.resize(size_t(h->E));
    for (size_t i = 0; i < h->h_c2p.size(); ++i) h->h_c2p[i] = uint32_t(i);
    for (const auto& k : h->keyed)
      for (size_t pos = 0; pos < k.second.size(); ++pos) h->h_c2p[size_t(k.first) + k.second[pos]] = uint32_t(k.first + pos);
    if ((rc = dev_alloc(h, &h->d_c2p, size_t(h->E)))) return rc;
    HIPCHK(h, hipMemcpy(h->d_c2p, h->h_c2p.data(), sizeof(uint32_t) * h->E, hipMemcpyHostToDevice));
  }
  h->keyed.clear();
  h->keyed.shrink_to_fit();
  h->next_key.clear();
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
  if ((rc = binned_planes<float>(h, n_params, max_knots, npts, coeffindexvec, uniquesplinevec, n_unique, uniquecoeffindices, xcoeff, manycoeff, &h->d_b_xk,
                                 reinterpret_cast<void**>(&h->d_b_many))))
    return rc;
  if ((rc = dev_alloc(h, &h->d_b_w, size_t(monolith_size)))) return rc;
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
  if ((rc = binned_planes<double>(h, n_params, max_knots, npts, coeffindexvec, uniquesplinevec, n_unique, uniquecoeffindices, xcoeff, manycoeff, &h->d_b_xk_d,
                                  reinterpret_cast<void**>(&h->d_b_many_d))))
    return rc;
  if ((rc = dev_alloc(h, &h->d_b_w_d, size_t(monolith_size)))) return rc;
  HIPCHK(h, hipMemcpy(h->d_coeff_x_d, xpts, sizeof(double) * xf.size(), hipMemcpyHostToDevice));
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
    h->pos_dirty = true;
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

}  // extern "C"
