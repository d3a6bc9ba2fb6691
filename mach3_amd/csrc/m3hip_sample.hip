// m3hip_sample.hip -- set-up of the sample side (seam 3 of include/m3hip.h): parameter map, bin space, per-event arrays,
// device binning with shifts and cuts, oscillation tables, experiment weights, data, test statistic.
#include "m3hip_ctx.h"

// ---- per-event transfers of an engine whose upload re-ordered events (m3hip_monolith_order_key): set-up and diagnostics only
template <class T>
__global__ __launch_bounds__(256) void k_scatter_rows(T* __restrict__ dst, const T* __restrict__ src, const uint32_t* __restrict__ c2p, uint64_t n, uint32_t width) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * width) return;
  const uint64_t e = i / width, k = i - e * width;
  dst[(uint64_t)c2p[e] * width + k] = src[i];
}
template <class T>
__global__ __launch_bounds__(256) void k_gather_events(T* __restrict__ dst, const T* __restrict__ src, const uint32_t* __restrict__ c2p, uint64_t n) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[c2p[i]];
}
static int xfer_reserve(m3hip_handle h, size_t bytes) {
  if (h->xfer_cap >= bytes) return M3HIP_OK;
  dev_free(h->d_xfer);
  h->d_xfer = nullptr;
  h->xfer_cap = 0;
  HIPCHK(h, hipMalloc(&h->d_xfer, bytes));
  h->xfer_cap = bytes;
  return M3HIP_OK;
}
int put_events(m3hip_handle h, void* dst_base, uint64_t off, const void* src, uint64_t n, size_t elem, uint32_t width) {
  if (n == 0) return M3HIP_OK;
  const size_t bytes = size_t(n) * width * elem;
  if (h->h_c2p.empty()) {
    HIPCHK(h, hipMemcpy(static_cast<char*>(dst_base) + size_t(off) * width * elem, src, bytes, hipMemcpyHostToDevice));
    return M3HIP_OK;
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));  // the staging buffer may still feed an asynchronous copy (m3hip_eval_splines' weights)
  if (int rc = xfer_reserve(h, bytes)) return rc;
  HIPCHK(h, hipMemcpy(h->d_xfer, src, bytes, hipMemcpyHostToDevice));
  const dim3 g((unsigned)((n * width + 255) / 256)), b(256);
  const uint32_t* map = h->d_c2p + off;
  if (elem == 2) hipLaunchKernelGGL(k_scatter_rows<uint16_t>, g, b, 0, h->stream, static_cast<uint16_t*>(dst_base), static_cast<const uint16_t*>(h->d_xfer), map, n, width);
  else if (elem == 4) hipLaunchKernelGGL(k_scatter_rows<uint32_t>, g, b, 0, h->stream, static_cast<uint32_t*>(dst_base), static_cast<const uint32_t*>(h->d_xfer), map, n, width);
  else hipLaunchKernelGGL(k_scatter_rows<uint64_t>, g, b, 0, h->stream, static_cast<uint64_t*>(dst_base), static_cast<const uint64_t*>(h->d_xfer), map, n, width);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));  // the staging buffer is reused by the next transfer
  return M3HIP_OK;
}
// elements [off, off + n) in the caller's order -> d_xfer, enqueued on the handle's stream
static int gather_into_xfer(m3hip_handle h, const void* src_base, uint64_t off, uint64_t n, size_t elem) {
  if (int rc = xfer_reserve(h, size_t(n) * elem)) return rc;
  const dim3 g((unsigned)((n + 255) / 256)), b(256);
  const uint32_t* map = h->d_c2p + off;
  if (elem == 4) hipLaunchKernelGGL(k_gather_events<uint32_t>, g, b, 0, h->stream, static_cast<uint32_t*>(h->d_xfer), static_cast<const uint32_t*>(src_base), map, n);
  else hipLaunchKernelGGL(k_gather_events<uint64_t>, g, b, 0, h->stream, static_cast<uint64_t*>(h->d_xfer), static_cast<const uint64_t*>(src_base), map, n);
  HIPCHK(h, hipGetLastError());
  return M3HIP_OK;
}
int gather_events_async(m3hip_handle h, const void* src_base, uint64_t n, size_t elem) { return gather_into_xfer(h, src_base, 0, n, elem); }
int get_events(m3hip_handle h, void* out, const void* src_base, uint64_t off, uint64_t n, size_t elem) {
  if (n == 0) return M3HIP_OK;
  const size_t bytes = size_t(n) * elem;
  if (h->h_c2p.empty()) {
    HIPCHK(h, hipMemcpy(out, static_cast<const char*>(src_base) + size_t(off) * elem, bytes, hipMemcpyDeviceToHost));
    return M3HIP_OK;
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));  // (as in put_events: before the staging buffer is resized or rewritten)
  if (int rc = gather_into_xfer(h, src_base, off, n, elem)) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipMemcpy(out, h->d_xfer, bytes, hipMemcpyDeviceToHost));
  return M3HIP_OK;
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

extern "C" {

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
  // Round 5: a brick flushes EVERY bin of a several-sample space (its events belong to all samples), so with more bins the big
  // samples want more addresses, not the same number: config 4's share of one GPU (5 120 bricks, 800 bins), 8 / 16 / 32 / 64
  // replicas: kernel 358.5 / 355.3 / 348.6 / 348.8 us, step 389.7 / 390.1 / 383.9 / 392.1 -> 25 600 addresses for them (32 at 800 bins).
  if (!h->hist_rep_forced) {
    h->hist_rep = std::max(1, std::min(std::min(h->hist_rep_cfg, 32), 6400 / nb));
    h->hist_rep_big = std::max(h->hist_rep, std::min(h->hist_rep_cfg, 25600 / nb));
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
  h->pos_dirty = true;
  if (int rc_p = put_events(h, h->d_bin, event_offset, bin_idx, n, sizeof(int32_t))) return rc_p;
  if (osc_idx) {
    h->oscw_valid = false;
    h->osc_static_steps = 0;
    for (uint64_t i = 0; i < n; ++i) {
      // -1 = no oscillation weight; anything below would run into the packed sample word's neighbouring fields (k_pack_events)
      if (osc_idx[i] < -1) return fail(h, M3HIP_EINVAL, "upload_events: oscillation index below -1");
      h->max_osc_idx = std::max<int64_t>(h->max_osc_idx, osc_idx[i]);
    }
    if (int rc_p = put_events(h, h->d_osc, event_offset, osc_idx, n, sizeof(int32_t))) return rc_p;
    // How local are the look-ups?  Distinct 128-byte lines of a float table per 1024 consecutive events (the events a brick's
    // slab holds: its waves gather from these lines).  m3hip_info::osc_lines_per_kevent; SampleHandlerFD::EventOrderByOscillationBin.
    if (event_offset == 0) h->osc_line_sum = h->osc_line_blocks = 0;
    h->osc_lines_stale = !h->h_c2p.empty();  // (the engine's own event order: counted from d_osc when m3hip_get_info asks)
    std::vector<int32_t> ln(1024);
    for (uint64_t b0 = 0; !h->osc_lines_stale && b0 + 1024 <= n; b0 += 1024) {
      size_t m = 0;
      for (uint64_t i = b0; i < b0 + 1024; ++i)
        if (osc_idx[i] >= 0) ln[m++] = osc_idx[i] >> 5;
      std::sort(ln.begin(), ln.begin() + m);
      h->osc_line_sum += uint64_t(std::unique(ln.begin(), ln.begin() + m) - ln.begin());
      h->osc_line_blocks += 1;
    }
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
        if (int rc_p = put_events(h, h->d_nidx + size_t(m) * h->E_pad, event_offset, plane.data(), n, sizeof(int16_t))) return rc_p;
      }
    }
  }
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
    if (int rc_p = put_events(h, h->d_sample_id, event_offset, sample_id, n, sizeof(int32_t))) return rc_p;
  }
  if (int rc_p = put_events(h, h->d_kin, event_offset, kin, n, sizeof(double), uint32_t(kin_stride))) return rc_p;
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
  h->pos_dirty = true;
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
  if (int rc_g = get_events(h, out, h->d_bin, event_offset, n, sizeof(int32_t))) return rc_g;
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
  h->d_osc_ext = nullptr;
  h->d_osc_ext_d = nullptr;
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
  h->d_osc_ext = nullptr;
  h->d_osc_ext_d = nullptr;
  h->osc_valid_f64 = true;
  h->osc_valid_f32 = false;
  h->oscw_valid = false;
  h->osc_static_steps = 0;
  if (h->evw_weight) h->evw_dirty = true;
  return M3HIP_OK;
} M3_API_CATCH(h)

// The table where a GPU calculator left it (NuOscillator's CUDAProb3-style back ends keep their weight array on the device, and
// the reference's events alias that array: Samples/SampleHandlerFD.cpp:1108-1122): no copy, the kernels of the following reweights
// read the caller's memory.  The engine's own buffers are still sized to n -- the other precision of M3::float_t is derived into
// them when a reweight needs it.  Whether `d_table` really is device memory is checked once per call (a host pointer here would
// fault inside a kernel, for everyone on the node).
static int osc_bind_device(m3hip_handle h, const void* d_table, uint64_t n, bool is_f64, const char* who) {
  if (!d_table || n == 0) return fail(h, M3HIP_EINVAL, std::string(who) + ": empty table");
  HIPCHK(h, hipSetDevice(h->device));
  const bool seen = n == h->n_osc && d_table == (is_f64 ? (const void*)h->d_osc_ext_d : (const void*)h->d_osc_ext);  // the per-step call of a fit: checked before
  hipPointerAttribute_t at;
  memset(&at, 0, sizeof(at));
  const hipError_t e = seen ? hipSuccess : hipPointerGetAttributes(&at, d_table);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(h, M3HIP_EINVAL, std::string(who) + ": not a device pointer (" + hipGetErrorString(e) + "); host tables go through m3hip_set_osc_table");
  }
  if (!seen && at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeManaged)
    return fail(h, M3HIP_EINVAL, std::string(who) + ": the table is not in device memory; host tables go through m3hip_set_osc_table");
  if (!seen && at.type == hipMemoryTypeDevice && at.device != h->device)
    return fail(h, M3HIP_EINVAL, std::string(who) + ": the table lives on device " + std::to_string(at.device) + ", the engine on device " + std::to_string(h->device));
  int rc = osc_table_resize(h, n);
  if (rc) return rc;
  h->d_osc_ext = is_f64 ? nullptr : static_cast<const float*>(d_table);
  h->d_osc_ext_d = is_f64 ? static_cast<const double*>(d_table) : nullptr;
  h->osc_valid_f32 = !is_f64;
  h->osc_valid_f64 = is_f64;
  h->oscw_valid = false;
  h->osc_static_steps = 0;
  if (h->evw_weight) h->evw_dirty = true;  // the packed words carry weights of the table that was just replaced
  return M3HIP_OK;
}

int m3hip_set_osc_table_device(m3hip_handle h, const float* d_table, uint64_t n) try {
  if (!h) return M3HIP_EINVAL;
  return osc_bind_device(h, d_table, n, false, "set_osc_table_device");
} M3_API_CATCH(h)

int m3hip_set_osc_table_device_f64(m3hip_handle h, const double* d_table, uint64_t n) try {
  if (!h) return M3HIP_EINVAL;
  return osc_bind_device(h, d_table, n, true, "set_osc_table_device_f64");
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
  if (int rc_p = put_events(h, x.d_col, event_offset, w, n, sizeof(float))) return rc_p;
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
  if (int rc_p = put_events(h, x.d_idx, event_offset, idx, n, sizeof(int32_t))) return rc_p;
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

}  // extern "C"
