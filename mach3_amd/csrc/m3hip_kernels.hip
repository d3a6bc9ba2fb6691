// m3hip_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the per-step likelihood path.
//
//   k_prep              a1  SplineBase::FindSplineSegment (Splines/SplineBase.cpp:44-109)
//                       a6  SampleHandlerFD::ResetHistograms (Samples/SampleHandlerFD.cpp:454-463)
//   k_reweight_uniform  a2  SMonolith::CalcSplineWeights   (Splines/SplineMonolith.cpp:727-788)
//                       a3  SMonolith::CalcTotalEventWeight (:792-830)
//                       a7  SampleHandlerFD::CalcWeightTotal (Samples/SampleHandlerFD.cpp:568-594)
//                       a8  SampleHandlerFD::FillArray      (:352-383)
//   k_llh               a10-a12 GetLikelihood / GetTestStatLLH (Samples/SampleHandlerFD.cpp:1284-1300,
//                       Samples/SampleHandlerBase.cpp:17-192)
//
// Floating point: compiled with -ffp-contract=off; the only fused operations are the
// explicit fmaf() calls the reference itself makes, so per-event float weights are
// bit-identical to a serial CPU evaluation in the reference's operation order.
#include "m3hip_kernels.h"
#include <cstring>

#include <hip/hip_ext.h>

#include <type_traits>

namespace m3hip {

// ------------------------------------------------------------------------------------
// k_prep: one workgroup.  Parameter-only work of a step.
// ------------------------------------------------------------------------------------
// (Two variants with fewer dependent round trips on paper -- the proposal staged through LDS, and the knots of a
// parameter preloaded into registers before the PCIe wait -- were each measured 2.5-3.5 us SLOWER per step in
// same-box A/B runs (scripts/ab_step.py); the simple version stays.)
__global__ __launch_bounds__(1024) void k_prep(PrepArgs a) {
  const int tid = threadIdx.x;
  const int nt = blockDim.x;  // 256 up to P = 256, else enough lanes for one parameter each (<= 1024): see launch_prep
  if (a.dbg_clock && tid == 0) a.dbg_clock[0] = wall_clock64();
  // ResetHistograms: mc always, w2 only while FirstTimeW2 (SampleHandlerFD.cpp:454-463)
  if (a.reset_hist) {
    const int n = a.hist_rep * a.n_bins;
    for (int b = tid; b < n; b += nt) {
      a.hist[b] = 0.0;
      if (a.reset_w2) a.hist[n + b] = 0.0;
    }
  }
  if (a.params_src) {
    for (int i = tid; i < a.n_par; i += nt) {
      const double v = a.params_src[i];
      a.params_dev[i] = v;
      if (a.params_f32) a.params_f32[i] = static_cast<float>(v);
    }
    if (tid == 0) {
      a.params_dev[a.n_par] = 1.0;
      if (a.params_f32) a.params_f32[a.n_par] = 1.0f;
    }
    __syncthreads();  // from here on read the device copy, not host memory over PCIe
  }
  const int K = a.max_knots;
  for (int i = tid; i < a.n_params; i += nt) {
    float xvar;
    int segment = 0;
    const int nPoints = a.npts[i];
    const float* xArray = a.coeff_x + (size_t)i * K;
    if (a.vals_in) {
      // host already ran FindSplineSegment (RunGPU_SplineMonolith inputs, gpuSplineUtils.cu:444-465)
      xvar = a.vals_in[i];
      segment = a.seg_in[i];
      if (nPoints > 0) a.curr_seg[i] = (int16_t)segment;  // keep FastSplineInfo.CurrSegment in step with the host's
    } else {
      xvar = float(a.params_dev[a.par_map[i]]);  // SplineBase.cpp:54 (double -> float)
      if (nPoints > 0) {                         // :60 parameters without knots are skipped
        int kHigh = nPoints - 1;
        const int PreviousSegment = a.curr_seg[i];
        // comparisons run in double: exact for float knots (both sides promote exactly), and what the
        // float_t = double build does with its double knots
        const double* xd = a.coeff_x_d ? a.coeff_x_d + (size_t)i * K : nullptr;
        auto X = [&](int k) -> double { return xd ? xd[k] : (double)xArray[k]; };
        const double xv = (double)xvar;
        if (xv <= X(0)) {
          segment = 0;
        } else if (xv >= X(nPoints - 1)) {
          segment = kHigh;
        } else if (X(PreviousSegment + 1) > xv && xv >= X(PreviousSegment)) {  // :76
          segment = PreviousSegment;
        } else {
          while (kHigh - segment > 1) {  // :84-94
            const int kHalf = (segment + kHigh) / 2;
            if (xv > X(kHalf)) segment = kHalf;
            else kHigh = kHalf;
          }
        }
        if (segment >= nPoints - 1 && nPoints > 1) segment = nPoints - 2;  // :97
        a.curr_seg[i] = (int16_t)segment;
      } else {
        segment = a.curr_seg[i];  // untouched, as in the reference (SplineSegments keeps its value)
      }
    }
    a.pval[i] = xvar;
    a.seg_out[i] = (int16_t)segment;
    PTab t;
    t.seg = segment;
    // SplineMonolith.cpp:747,759: dx = ParamValues[Param] - coeff_x[Param*_max_knots + segment]
    t.dx = (nPoints > 0) ? (xvar - xArray[segment]) : 0.f;
    a.ptab[i] = t;
  }
  if (a.n_res > 0) {  // few unique slot lists: resolve them here and save a launch
    __syncthreads();
    for (int i = tid; i < a.n_res; i += nt) {
      const int p = a.u_param[i];
      const int pl0 = a.u_plane0[i];
      RSlot r;
      if (pl0 >= 0) {
        const PTab t = a.ptab[p];
        r.plane = pl0 + t.seg;
        r.f = t.dx;
      } else {
        r.plane = 0;
        r.f = a.pval[p];
      }
      a.resolved[i] = r;
    }
  }
}

// many unique slot lists (ragged samples): one thread per slot, after k_prep
__global__ __launch_bounds__(256) void k_resolve(PrepArgs a, int32_t n_res) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_res) return;
  const int p = a.u_param[i];
  const int pl0 = a.u_plane0[i];
  RSlot r;
  if (pl0 >= 0) {
    const PTab t = a.ptab[p];
    r.plane = pl0 + t.seg;
    r.f = t.dx;
  } else {
    r.plane = 0;
    r.f = a.pval[p];
  }
  a.resolved[i] = r;
}

void launch_prep(const PrepArgs& a, hipStream_t s) {
  // every pass over the parameters is a chain of dependent loads (map -> value -> knots): with P > 256 one
  // lane per parameter keeps it to a single pass (A/B on one box: -3 us per step at P = 512, -8 us at P = 1024)
  const int nt = a.n_params <= 256 ? 256 : (a.n_params <= 512 ? 512 : 1024);
  hipLaunchKernelGGL(k_prep, dim3(1), dim3(nt), 0, s, a);
}
void launch_resolve(const PrepArgs& a, int32_t n_res, hipStream_t s) {
  if (n_res > 0) hipLaunchKernelGGL(k_resolve, dim3((n_res + 255) / 256), dim3(256), 0, s, a, n_res);
}

// ---- per-event binning (a9), used by k_find_bins and by the fused epilogue when functional shifts are registered
__device__ __forceinline__ int d_upper_bound_bin(const double* e, int n_edges, double v) {
  // std::distance(begin, std::upper_bound(begin, end, v)) - 1
  int lo = 0, hi = n_edges;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (v < e[mid]) hi = mid;
    else lo = mid + 1;
  }
  return lo - 1;
}

__device__ int d_find_bin(const double* e, int N_Bins, double KinVar, int NomBin) {
  // Samples/SampleStructs.h:577-613.  The range test is written so that a NaN (a non-finite shift parameter, a bad
  // input) is out of range: in the reference's form `x < lo || x >= hi` a NaN passes, upper_bound then returns the
  // last edge and the event lands in bin N_Bins -- the next sample's first bin, or past the bin vector.
  // The nominal bin first, with two look-ups: a value inside [e[NomBin], e[NomBin + 1]) is inside the axis (the edges ascend), so
  // the reference's order -- range test, then nominal bin -- gives the same index; a NaN fails both comparisons and reaches the
  // range test below.  (The re-binning pass walks these tables in LDS, the bins of a wave's 64 events are all over them, and its
  // time was that of its LDS reads: six per event before, two for an event that stays where it was.)
  if (NomBin > -1) {
    const double lower = e[NomBin], upper = e[NomBin + 1];
    if (KinVar < upper && KinVar >= lower) return NomBin;
  }
  if (!(KinVar >= e[0] && KinVar < e[N_Bins])) return -1;
  if (NomBin > -1) {
    const double lower = e[NomBin], upper = e[NomBin + 1];
    const double lower_lower = NomBin == 0 ? e[0] : e[NomBin - 1];                       // :615-646
    const double upper_upper = (NomBin + 2 < N_Bins) ? e[NomBin + 2] : ((NomBin + 1 < N_Bins) ? e[NomBin + 1] : -999999.123456);
    if (KinVar < lower && KinVar >= lower_lower) return NomBin - 1;
    if (KinVar < upper_upper && KinVar >= upper) return NomBin + 1;
  }
  return d_upper_bound_bin(e, N_Bins + 1, KinVar);
}

// FindGlobalBin of one event in the per-step pass: registered shifts applied to the nominal kinematics
// (ApplyShifts, SampleHandlerFD.cpp:545-564), the selection cuts on the shifted values (IsEventSelected,
// :281-294), then nominal bin -> +-1 neighbour -> binary search per axis and, for non-uniform samples, the
// candidate scan of the mega-bin (BinningHandler.cpp:257-291).  Shared by k_find_bins and the fused epilogue.
__device__ int d_event_bin(const BinArgs& a, uint64_t ev) {
  const int sid = a.sample_id ? a.sample_id[ev] : 0;  // (null: one sample)
  const BinSample sm = a.samples[sid];
  double x[M3_MAX_DIM];
  for (int c = 0; c < a.kin_stride; ++c) x[c] = a.kin[ev * a.kin_stride + c];
  for (int s = 0; s < a.n_shifts; ++s) {
    const KinShift sh = a.shifts[s];
    if (sh.dim < a.kin_stride && (sh.sample < 0 || sh.sample == sid)) {
      const double th = a.params[sh.param];
      const int kind = sh.kind & 0xFF, aux = sh.kind >> 8;
      // 1: x *= 1 + theta; 2: x += theta; 3: x += theta * (nominal value of column aux), e.g. scaling the hadronic part of E_rec
      x[sh.dim] = kind == 1 ? x[sh.dim] * (1.0 + th) : (kind == 2 ? x[sh.dim] + th : x[sh.dim] + th * a.kin[ev * a.kin_stride + aux]);
    }
  }
  bool out = false;
  for (int c = 0; c < a.n_cuts; ++c) {
    const KinCut ct = a.cuts[c];
    if (ct.column < a.kin_stride && (ct.sample < 0 || ct.sample == sid)) {
      const double v = x[ct.column];
      if ((v < ct.lo) || (v >= ct.hi)) out = true;
    }
  }
  int global = 0;
  for (int d = 0; d < sm.n_dim; ++d) {
    const BinAxis ax = a.axes[sm.axis_off + d];
    // (no cached nominal bins: the cascade's last route, the binary search -- every route gives the same index)
    const int b = d_find_bin(a.edges + ax.edge_off, ax.n_bins, x[d], a.nom_bin ? a.nom_bin[ev * a.kin_stride + d] : -1);
    if (b < 0) out = true;  // BinningHandler.cpp:269: one axis out of range -> UnderOverFlowBin
    global += b * ax.stride;
  }
  if (!sm.uniform && !out) {
    // BinningHandler.cpp:277-290: `global` is the mega-bin; scan its candidate boxes in order, first
    // box with lo < x <= hi on every axis (BinInfo::IsEventInside, SampleStructs.h:186-220) wins
    const int32_t* go = a.grid_off + sm.map_off + global;
    const int c0 = go[0], c1 = go[1];
    int found = -1;
    for (int c = c0; c < c1 && found < 0; ++c) {
      const int box = a.grid_list[c];
      const double* ext = a.boxes + ((size_t)sm.box_off + box) * sm.n_dim * 2;
      bool inside = true;
      for (int d = 0; d < sm.n_dim; ++d) inside &= (x[d] > ext[2 * d]) & (x[d] <= ext[2 * d + 1]);
      if (inside) found = box;
    }
    global = found;
    if (found < 0) out = true;
  }
  return out ? -1 : global + sm.global_offset;
}

// ------------------------------------------------------------------------------------
// k_reweight_uniform: one workgroup (256 lanes = 256 events) per uniform tile.
//
// HBM traffic per event: one float4 per spline slot (the active segment's plane), one
// float2 per TF1 slot, + 8 B of bin/osc indices and 2 B per norm slot when fused.
// Every vector load is lane-contiguous (64 x 16 B = 1 KiB per wave instruction); all
// parameter-dependent state (segment, dx, value) is wave-uniform and travels in SGPRs.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float spline_poly(float dx, const float4 c) {
  // SplineMonolith.cpp:762  fmaf(dx, fmaf(dx, fmaf(dx, fD, fC), fB), fY); c = {y,b,c,d}
  return fmaf(dx, fmaf(dx, fmaf(dx, c.w, c.z), c.y), c.x);
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// coefficient planes are read exactly once per launch: NT marks them streaming so they do
// not displace the small, hot per-step tables in L2
// Pointers that reach a kernel inside a descriptor loaded from memory (TileDev / BrickDev) are generic ("flat")
// to the compiler: it then emits flat_load (which also ties up the LDS counter) and cannot use scalar loads for
// wave-uniform reads.  These stores all live in device global memory, so say so.
#define M3_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ const M3_GLOBAL T* as_global(const T* p) {
  return (const M3_GLOBAL T*)p;
}

template <bool NT>
__device__ __forceinline__ float4 stream_load(const float4* p) {
  const M3_GLOBAL f32x4_t* g = (const M3_GLOBAL f32x4_t*)p;
  const f32x4_t v = NT ? __builtin_nontemporal_load(g) : *g;
  return make_float4(v.x, v.y, v.z, v.w);
}
template <bool NT>
__device__ __forceinline__ float2 stream_load(const float2* p) {
  const M3_GLOBAL f32x2_t* g = (const M3_GLOBAL f32x2_t*)p;
  const f32x2_t v = NT ? __builtin_nontemporal_load(g) : *g;
  return make_float2(v.x, v.y);
}

// Per-event gathers of the fused epilogue, batched so that several independent loads are in
// flight per lane (a plain loop over a run-time slot count serialises index load -> gather).
// Multiplication order is the slot order, as CalcWeightTotal walks its pointer vectors.
// WT = M3::float_t of the sample side: float (low-memory build, the one SMonolith needs) or double
template <class WT>
__device__ __forceinline__ const WT* binned_table(const ReweightArgs& a);
template <>
__device__ __forceinline__ const float* binned_table<float>(const ReweightArgs& a) { return a.bw; }
template <>
__device__ __forceinline__ const double* binned_table<double>(const ReweightArgs& a) { return a.bw_d; }

template <class WT>
__device__ __forceinline__ WT mul_binned_weights(WT tw, const ReweightArgs& a, uint64_t ev) {
  const WT* __restrict__ bw = binned_table<WT>(a);
  if (a.bl_id) {  // shared lists: lanes of a wave mostly walk the same list, so both look-ups are near-uniform
    const int id = a.bl_id[ev];
    if (id < 0) return tw;
    int off = a.bl_off[id];
    const int end = a.bl_off[id + 1];
    for (; off + 4 <= end; off += 4) {
      int bi[4];
      WT wv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) bi[u] = a.bl_idx[off + u];
#pragma unroll
      for (int u = 0; u < 4; ++u) wv[u] = bw[bi[u]];
#pragma unroll
      for (int u = 0; u < 4; ++u) tw *= wv[u];
    }
    for (; off < end; ++off) tw *= bw[a.bl_idx[off]];
    return tw;
  }
  uint32_t k = 0;
  for (; k + 4 <= a.n_bw_slots; k += 4) {
    int bi[4];
    WT wv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) bi[u] = a.bwidx[(uint64_t)(k + u) * a.e_pad + ev];
#pragma unroll
    for (int u = 0; u < 4; ++u) wv[u] = bi[u] >= 0 ? bw[bi[u]] : WT(1);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (bi[u] >= 0) tw *= wv[u];
  }
  for (; k < a.n_bw_slots; ++k) {
    const int bi = a.bwidx[(uint64_t)k * a.e_pad + ev];
    if (bi >= 0) tw *= bw[bi];
  }
  return tw;
}

// The same product where the lanes of a wave mostly SHARE their pointer list (shared-list form; the engine processes events with
// equal lists next to each other): the wave walks each distinct list ONCE -- lane j loads entry j's index and gathers its weight,
// two vector loads per list instead of two per entry and lane -- and every lane that carries the list multiplies the entries in,
// in list order, out of the loading lanes' registers (v_readlane).  Same factors in the same order as the per-lane loop: the
// product is bit-identical.  Must be called by all 64 lanes of a wave together.  Lanes whose list has not come up after
// kWaveLists rounds (a wave of unrelated events) walk theirs alone.
constexpr int kWaveLists = 6;
__device__ __forceinline__ float lane_value(float v, int j) { return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), j)); }
__device__ __forceinline__ double lane_value(double v, int j) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), j), hi = __builtin_amdgcn_readlane(__double2hiint(v), j);
  return __hiloint2double(hi, lo);
}
template <class WT>
__device__ __forceinline__ WT mul_binned_lists_wave(WT tw, const ReweightArgs& a, uint64_t ev) {
  const WT* __restrict__ bw = binned_table<WT>(a);
  const int id = a.bl_id[ev];
  const int lane64 = (int)(threadIdx.x & 63u);
  uint64_t todo = __ballot(id >= 0);
  for (int round = 0; todo != 0ull && round < kWaveLists; ++round) {
    const int lid = __builtin_amdgcn_readlane(id, (int)__ffsll((unsigned long long)todo) - 1);  // wave-uniform: the first waiting lane's list
    const bool mine = id == lid;
    const int off = a.bl_off[lid], end = a.bl_off[lid + 1];
    for (int o = off; o < end; o += 64) {
      const int n = end - o < 64 ? end - o : 64;
      const int bi = a.bl_idx[o + (lane64 < n ? lane64 : 0)];
      const WT wv = bw[bi];
      for (int j = 0; j < n; ++j) {
        const WT f = lane_value(wv, j);
        tw = mine ? tw * f : tw;
      }
    }
    todo &= ~__ballot(mine);
  }
  if ((todo >> lane64) & 1ull) {  // (rare) the lane's own walk
    int off = a.bl_off[id];
    const int end = a.bl_off[id + 1];
    for (; off < end; ++off) tw *= bw[a.bl_idx[off]];
  }
  return tw;
}

// experiment weights behind the spline weight in total_weight_pointers, in the order they were appended (slot order)
__device__ __forceinline__ float mul_extra_weights(float tw, const ReweightArgs& a, uint64_t ev) {
  for (uint32_t s = 0; s < a.n_extra; ++s) {
    const ExtraSlot x = a.extras[s];
    if (x.col) {
      tw *= x.col[ev];
    } else if (x.idx) {
      const int i = x.idx[ev];
      if (i >= 0) tw *= x.tab[i];
    }
  }
  return tw;
}

template <class WT>
__device__ __forceinline__ WT mul_norm_weights(WT tw, const ReweightArgs& a, uint64_t ev, uint32_t m = 0) {
  for (; m + 4 <= a.n_norm_slots; m += 4) {
    int idx[4];
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) idx[u] = a.nidx[(uint64_t)(m + u) * a.e_pad + ev];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = idx[u] >= 0 ? a.params[idx[u]] : 1.0;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (idx[u] >= 0) tw *= static_cast<WT>(v[u]);  // static_cast<M3::float_t> per pointer (SampleHandlerFD.cpp:580)
  }
  for (; m < a.n_norm_slots; ++m) {
    const int idx = a.nidx[(uint64_t)m * a.e_pad + ev];
    if (idx >= 0) tw *= static_cast<WT>(a.params[idx]);
  }
  return tw;
}

template <int MODE, int UNROLL, bool NT, bool F64 = false, bool DYN = false>
__global__ __launch_bounds__(256) void k_reweight_uniform(ReweightArgs a) {
  extern __shared__ double lds_hist[];  // [n_bins] mc, [n_bins] w2 (kModeFill only)
  const uint32_t tile = a.tile_base + blockIdx.x;
  const uint32_t lane = threadIdx.x;
  const TileDev td = a.tiles[tile];

  if ((MODE & kModeFill) && a.hist_lds) {
    const int nb = a.fill_w2 ? 2 * a.n_bins : a.n_bins;
    for (int b = lane; b < nb; b += 256) lds_hist[b] = 0.0;
  }

  // ---- kernel (i): TSpline3 responses, then TF1 responses, product in the reference's order.
  // The slot list is wave-uniform: it is fetched with scalar loads (16 dwords = 8 slots at a
  // time) and carries the already-resolved plane index and dx, so there is no dependent
  // look-up between a slot and its vector load.
  float w = 1.0f;  // SplineMonolith.cpp:798
  {
    const float4* __restrict__ sp = td.spl + lane;
    const M3_GLOBAL RSlot* __restrict__ rs = as_global(td.res);
    const uint32_t J = td.n_spl_slots;
    uint32_t j = 0;
    for (; j + UNROLL <= J; j += UNROLL) {
      float4 c[UNROLL];
      float dx[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        RSlot r;
        r.plane = rs[j + u].plane;
        r.f = rs[j + u].f;
        c[u] = stream_load<NT>(&sp[(size_t)(uint32_t)r.plane * M3_TE]);
        dx[u] = r.f;
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) w *= spline_poly(dx[u], c[u]);
    }
    for (; j < J; ++j) {
      RSlot r;
      r.plane = rs[j].plane;
      r.f = rs[j].f;
      w *= spline_poly(r.f, stream_load<NT>(&sp[(size_t)(uint32_t)r.plane * M3_TE]));
    }
  }
  {
    const float2* __restrict__ tp = td.tf1 + lane;
    const M3_GLOBAL RSlot* __restrict__ rt = as_global(td.res) + td.n_spl_slots;
    const uint32_t J = td.n_tf1_slots;
    uint32_t j = 0;
    for (; j + UNROLL <= J; j += UNROLL) {
      float2 c[UNROLL];
      float x[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        x[u] = rt[j + u].f;
        c[u] = stream_load<NT>(&tp[(size_t)(j + u) * M3_TE]);
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) w *= fmaf(c[u].x, x[u], c[u].y);  // SplineMonolith.cpp:779
    }
    for (; j < J; ++j) {
      const float x = rt[j].f;
      const float2 c = stream_load<NT>(&tp[(size_t)j * M3_TE]);
      w *= fmaf(c.x, x, c.y);
    }
  }

  const uint64_t pos = (uint64_t)td.first_event + lane;
  const uint64_t ev = a.perm ? (uint64_t)a.perm[pos] : pos;  // processing order -> event (identity unless a binned-only space was grouped)
  if (MODE & kModeEval) a.spline_w[ev] = w;  // cpu_total_weights[EventNum] (SplineMonolith.cpp:828)

  if (MODE & (kModeFill | kModeEventW)) {
    // ---- kernel (ii): CalcWeightTotal (SampleHandlerFD.cpp:568-594): norm pointers first
    // (double -> float cast each), then total_weight_pointers = {osc, spline total weight}.
    using WT = typename std::conditional<F64, double, float>::type;  // M3::float_t
    WT tw = mul_norm_weights<WT>(WT(1), a, ev);  // -1: empty ELL slot
    const int oi = a.osc[ev];
    if (oi >= 0) tw *= F64 ? (WT)a.osc_table_d[oi] : (WT)a.osc_table[oi];
    // bin-spline weights, in SetSplinePointers order (all 256 lanes are here: the shared-list form is walked wave by wave)
    tw = a.bl_id ? mul_binned_lists_wave<WT>(tw, a, ev) : mul_binned_weights<WT>(tw, a, ev);
    tw *= (WT)w;                              // F64: no event-by-event splines in that build, w == 1
    if (!F64 && a.n_extra) tw = (WT)mul_extra_weights((float)tw, a, ev);  // AddAdditionalWeightPointers: behind the splines
    if (MODE & kModeEventW) {
      if (F64) a.event_w_d[ev] = (double)tw;
      else a.event_w[ev] = (float)tw;
    }
    if (MODE & kModeFill) {
      __syncthreads();
      int b;
      if (DYN) {  // functional parameters: FindGlobalBin on the shifted kinematics, here instead of in a kernel of its own
        b = lane < td.n_events ? d_event_bin(a.dyn, ev) : -1;
        a.dyn.bin[ev] = b;
      } else {
        b = a.bin[ev];
      }
      // SampleHandlerFD.cpp:432 `if (totalweight <= 0.) continue;`  :443 `if (GlobalBin > -1)`
      const bool fills = lane < td.n_events && !(tw <= WT(0)) && b >= 0;  // written as the reference's test: a NaN weight fills
      const int rep = (int)(blockIdx.x % (uint32_t)a.hist_rep);
      if (!a.hist_lds) {  // many bins: straight into this workgroup's replica
        if (fills) {
          atomicAdd(&a.hist[(size_t)rep * a.n_bins + b], (double)tw);
          if (a.fill_w2) atomicAdd(&a.hist[(size_t)(a.hist_rep + rep) * a.n_bins + b], (double)(tw * tw));
        }
        return;
      }
      const int nb = a.fill_w2 ? 2 * a.n_bins : a.n_bins;
      if (a.partial) {
        // deterministic mode (m3hip_set_deterministic): the waves add in turn, so the order of the fp64 adds inside the
        // workgroup is fixed (within one wave the LDS serves the lanes of an instruction in lane order); the workgroup's
        // bin vector then goes to its own row of `partial` with plain stores, and k_fold_partials sums the rows in order
        for (uint32_t wv = 0; wv < 256 / 64; ++wv) {
          if ((lane >> 6) == wv && fills) {
            atomicAdd(&lds_hist[b], (double)tw);
            if (a.fill_w2) atomicAdd(&lds_hist[a.n_bins + b], (double)(tw * tw));
          }
          __syncthreads();
        }
        double* row = a.partial + (size_t)tile * (size_t)(2 * a.n_bins);
        for (int i = lane; i < nb; i += 256) row[i] = lds_hist[i];
        return;
      }
      if (fills) {
        atomicAdd(&lds_hist[b], (double)tw);
        if (a.fill_w2) atomicAdd(&lds_hist[a.n_bins + b], (double)(tw * tw));  // square in float_t, then widened (:445)
      }
      __syncthreads();
      // one flush per workgroup, spread over hist_rep replicas of the bin vector: thousands of
      // workgroups adding into the same few hundred addresses serialise at the memory side
      for (int i = lane; i < nb; i += 256) {
        const double v = lds_hist[i];
        const int slot = i < a.n_bins ? rep * a.n_bins + i : (a.hist_rep + rep) * a.n_bins + (i - a.n_bins);
        if (v != 0.0) atomicAdd(&a.hist[slot], v);
      }
    }
  }
}

// plain launch, or -- when the caller wants the dispatch timed -- the extended launch that stamps two events
#define M3_LAUNCH(kernel, g, b, lds, s, k, a)                                                                   \
  do {                                                                                                          \
    if ((k).ev_start || (k).ev_stop) hipExtLaunchKernelGGL((kernel), g, b, (uint32_t)(lds), s, (k).ev_start, (k).ev_stop, 0, a); \
    else hipLaunchKernelGGL((kernel), g, b, lds, s, a);                                                          \
  } while (0)

template <int MODE>
static void launch_mode(const ReweightArgs& a, dim3 g, size_t lds, const ReweightKnobs& k, hipStream_t s) {
  const dim3 b(256);
  if (k.unroll >= 16) {
    if (k.nt) M3_LAUNCH((k_reweight_uniform<MODE, 16, true>), g, b, lds, s, k, a);
    else M3_LAUNCH((k_reweight_uniform<MODE, 16, false>), g, b, lds, s, k, a);
  } else {
    if (k.nt) M3_LAUNCH((k_reweight_uniform<MODE, 8, true>), g, b, lds, s, k, a);
    else M3_LAUNCH((k_reweight_uniform<MODE, 8, false>), g, b, lds, s, k, a);
  }
}

void launch_reweight_uniform(const ReweightArgs& a, uint32_t n_tiles, int mode, const ReweightKnobs& k, hipStream_t s, bool dyn_bins) {
  if (n_tiles == 0) return;
  size_t lds = ((mode & kModeFill) && a.hist_lds) ? sizeof(double) * (a.fill_w2 ? 2 : 1) * (size_t)a.n_bins : 0;
  if (k.lds_pad > 0 && lds < (size_t)k.lds_pad) lds = (size_t)k.lds_pad;  // occupancy throttle (experiments)
  const dim3 g(n_tiles);
  if (dyn_bins && (mode & kModeFill)) {
    if (mode == kModeFill) M3_LAUNCH((k_reweight_uniform<kModeFill, 8, true, false, true>), g, dim3(256), lds, s, k, a);
    else M3_LAUNCH((k_reweight_uniform<kModeFill | kModeEval | kModeEventW, 8, true, false, true>), g, dim3(256), lds, s, k, a);
    return;
  }
  if (mode == kModeEval) launch_mode<kModeEval>(a, g, lds, k, s);
  else if (mode == kModeFill) launch_mode<kModeFill>(a, g, lds, k, s);
  else launch_mode<kModeFill | kModeEval | kModeEventW>(a, g, lds, k, s);
}


void launch_reweight_uniform_f64(const ReweightArgs& a, uint32_t n_tiles, int mode, const ReweightKnobs& k, hipStream_t s) {
  if (n_tiles == 0) return;
  const size_t lds = a.hist_lds ? sizeof(double) * (a.fill_w2 ? 2 : 1) * (size_t)a.n_bins : 0;
  const dim3 g(n_tiles), b(256);
  if (mode == kModeFill) M3_LAUNCH((k_reweight_uniform<kModeFill, 8, true, true>), g, b, lds, s, k, a);
  else M3_LAUNCH((k_reweight_uniform<kModeFill | kModeEval | kModeEventW, 8, true, true>), g, b, lds, s, k, a);
}

// ------------------------------------------------------------------------------------
// k_reweight_lists: the event side of a BINNED-ONLY sample whose events share their pointer lists -- what MaCh3Tutorial's far-detector
// sample runs: SampleHandlerFD::FillArray over events whose total_weight_pointers are {osc, bin-spline weights...}
// (Samples/SampleHandlerFD.cpp:390-448, 568-594, 1196-1243).  No event-by-event splines, static bins.
//
// The uniform-tile kernel did this with zero spline slots: a 256-event workgroup per tile, whose every lane walked
// position -> event -> {bin, osc, list id} -> list bounds -> entry -> weight, six dependent loads deep, for 16 bytes per event -- bound by
// that chain's latency (22 us per 1e6 events with the MC in list order, 43 us in any other) and by 3 900 flushes of the bin vector.
// Here the event side is read in PROCESSING order (PosArgs, packed by k_pack_positions when an input changes: whatever order the MC
// was registered in, every load below is coalesced), a workgroup takes 1 024 consecutive positions (a quarter of the flushes), every
// lane has its four events' loads in flight together, and the lists are walked by the WAVE: the distinct lists among 64 neighbouring
// positions (one to three when the order is by list) are found from the registers, lane j fetches entry j of each, and every lane
// that carries a list multiplies its entries in, in list order, out of the fetching lanes' registers.  Chain: position record ->
// list entry -> weight.  Same factors in the same order as CalcWeightTotal: weights bit-identical to the oracle's.
// ------------------------------------------------------------------------------------
constexpr int kListsEPL = 4;     // events per lane
constexpr int kListsRounds = 3;  // distinct lists of a wave's 64 positions fetched ahead; lanes whose list is a fourth (or later) walk theirs alone

template <bool F64>
__global__ __launch_bounds__(256) void k_reweight_lists(ReweightArgs a, PosArgs p, int mode) {
  using WT = typename std::conditional<F64, double, float>::type;  // M3::float_t
  extern __shared__ double lds_hist[];  // [n_bins] mc, [n_bins] w2 when the bin vector is workgroup-private; then the list staging
  const uint32_t lane = threadIdx.x;
  const int l64 = (int)(lane & 63u);
  const int nb = a.fill_w2 ? 2 * a.n_bins : a.n_bins;
  if (a.hist_lds)
    for (int b = lane; b < nb; b += 256) lds_hist[b] = 0.0;
  // staging of the lists' weights: one region of 64 per (wave, row, list fetched ahead) + one region of ones per wave
  WT* const stage = reinterpret_cast<WT*>(lds_hist + (a.hist_lds ? ((nb + 1) & ~1) : 0)) + (size_t)(lane >> 6) * ((kListsEPL * kListsRounds + 1) * 64);
  WT* const ones = stage + kListsEPL * kListsRounds * 64;
  ones[l64] = WT(1);
  const WT* __restrict__ bw = binned_table<WT>(a);
  const uint64_t base = (uint64_t)blockIdx.x * (256u * kListsEPL);
  uint32_t ev[kListsEPL], loff[kListsEPL], llen[kListsEPL];
  int32_t bin[kListsEPL], osc[kListsEPL];
  bool valid[kListsEPL];
#pragma unroll
  for (int u = 0; u < kListsEPL; ++u) {  // the position records: five coalesced loads per event, all in flight together
    const uint64_t pos = base + (uint64_t)u * 256u + lane;
    valid[u] = pos < p.n_pos;
    const uint64_t pc = valid[u] ? pos : 0;
    ev[u] = p.ev[pc];
    bin[u] = p.bin[pc];
    osc[u] = p.osc[pc];
    loff[u] = p.loff[pc];
    llen[u] = valid[u] ? (uint32_t)p.llen[pc] : 0u;
  }
  // the distinct lists of each wave-row, from the registers: no memory involved
  uint32_t s_off[kListsEPL][kListsRounds], s_len[kListsEPL][kListsRounds];
  int my_round[kListsEPL];  // which of the row's lists is this lane's (-1: none, or one beyond those fetched ahead)
  bool alone[kListsEPL];
#pragma unroll
  for (int u = 0; u < kListsEPL; ++u) {
    uint64_t todo = __ballot(llen[u] > 0u);
    my_round[u] = -1;
#pragma unroll
    for (int r = 0; r < kListsRounds; ++r) {
      s_off[u][r] = s_len[u][r] = 0u;
      if (todo != 0ull) {  // (wave-uniform)
        const int leader = (int)__ffsll((unsigned long long)todo) - 1;
        s_off[u][r] = (uint32_t)__builtin_amdgcn_readlane((int)loff[u], leader);
        s_len[u][r] = (uint32_t)__builtin_amdgcn_readlane((int)llen[u], leader);
        const bool m = llen[u] > 0u && loff[u] == s_off[u][r];  // lists are stored once: equal first entry = the same list
        my_round[u] = m ? r : my_round[u];
        todo &= ~__ballot(m);
      }
    }
    alone[u] = ((todo >> l64) & 1ull) != 0ull;
  }
  // lane j fetches entry j of every list found (the first 64 entries; longer lists continue below), then its weight
  WT wv[kListsEPL][kListsRounds];
  {
    int32_t bi[kListsEPL][kListsRounds];
#pragma unroll
    for (int u = 0; u < kListsEPL; ++u)
#pragma unroll
      for (int r = 0; r < kListsRounds; ++r) {
        const uint32_t n = s_len[u][r] < 64u ? s_len[u][r] : 64u;
        bi[u][r] = n ? a.bl_idx[s_off[u][r] + ((uint32_t)l64 < n ? (uint32_t)l64 : 0u)] : 0;
      }
#pragma unroll
    for (int u = 0; u < kListsEPL; ++u)
#pragma unroll
      for (int r = 0; r < kListsRounds; ++r) wv[u][r] = s_len[u][r] ? bw[bi[u][r]] : WT(1);
  }
  // oscillation weights (a small table in L2 / L1) while the lists travel
  WT ow[kListsEPL];
#pragma unroll
  for (int u = 0; u < kListsEPL; ++u) {
    const int oi = osc[u] >= 0 ? osc[u] : 0;
    ow[u] = F64 ? (WT)(a.osc_table_d ? a.osc_table_d[oi] : 1.0) : (WT)(a.osc_table ? a.osc_table[oi] : 1.0f);
  }
  // the weights change lanes through the wave's staging regions, padded with ones to 64: a lane then reads ITS list's entries
  // front to back whatever its neighbours carry (a factor of exactly 1 leaves a product's bits alone)
#pragma unroll
  for (int u = 0; u < kListsEPL; ++u)
#pragma unroll
    for (int r = 0; r < kListsRounds; ++r) {
      const uint32_t n = s_len[u][r] < 64u ? s_len[u][r] : 64u;
      if (n) stage[(u * kListsRounds + r) * 64 + l64] = (uint32_t)l64 < n ? wv[u][r] : WT(1);  // (wave-uniform branch)
    }
  const int rep = (int)(blockIdx.x % (uint32_t)a.hist_rep);
  __syncthreads();  // the workgroup's bin vector is zeroed, the staging regions are written
#pragma unroll
  for (int u = 0; u < kListsEPL; ++u) {
    // CalcWeightTotal (SampleHandlerFD.cpp:568-594): norm pointers first (double -> float_t cast each), then
    // total_weight_pointers = {osc, bin-spline weights in SetSplinePointers order, [spline total weight: none here], extras}
    WT tw = mul_norm_weights<WT>(WT(1), a, ev[u]);
    if (osc[u] >= 0) tw *= ow[u];
    uint32_t longest = 0;  // (wave-uniform)
#pragma unroll
    for (int r = 0; r < kListsRounds; ++r) longest = s_len[u][r] > longest ? s_len[u][r] : longest;
    const int n64 = (int)(longest < 64u ? longest : 64u);
    const WT* mine = my_round[u] >= 0 ? stage + (u * kListsRounds + my_round[u]) * 64 : ones;
    constexpr int kVec = 16 / (int)sizeof(WT);  // entries per 16-byte LDS read
    for (int j = 0; j < n64; j += kVec) {
      WT f[kVec];
#pragma unroll
      for (int i = 0; i < kVec; ++i) f[i] = mine[j + i];
#pragma unroll
      for (int i = 0; i < kVec; ++i) tw *= f[i];
    }
    if (longest > 64u) {  // (rare) lists of more than 64 entries: the rest 64 at a time, list after list
#pragma unroll
      for (int r = 0; r < kListsRounds; ++r) {
        const bool m = my_round[u] == r;
        for (uint32_t o = 64u; o < s_len[u][r]; o += 64u) {
          const uint32_t nn = s_len[u][r] - o < 64u ? s_len[u][r] - o : 64u;
          const WT more = bw[a.bl_idx[s_off[u][r] + o + ((uint32_t)l64 < nn ? (uint32_t)l64 : 0u)]];
          for (int j = 0; j < (int)nn; ++j) {
            const WT f = lane_value(more, j);
            tw = m ? tw * f : tw;
          }
        }
      }
    }
    if (alone[u]) {  // (rare) a wave of unrelated events: the lane's own walk
      for (uint32_t o = loff[u]; o < loff[u] + llen[u]; ++o) tw *= bw[a.bl_idx[o]];
    }
    if (!F64 && a.n_extra) tw = (WT)mul_extra_weights((float)tw, a, ev[u]);  // AddAdditionalWeightPointers: behind the splines
    if (valid[u]) {
      if (mode & kModeEval) a.spline_w[ev[u]] = 1.0f;  // no event-by-event splines: SMonolith's total weight of every event is 1
      if (mode & kModeEventW) {
        if (F64) a.event_w_d[ev[u]] = (double)tw;
        else a.event_w[ev[u]] = (float)tw;
      }
      // SampleHandlerFD.cpp:432 `if (totalweight <= 0.) continue;`  :443 `if (GlobalBin > -1)`; a NaN weight fills, as there
      const int b = bin[u];
      if (!(tw <= WT(0)) && b >= 0) {
        if (a.hist_lds) {
          atomicAdd(&lds_hist[b], (double)tw);
          if (a.fill_w2) atomicAdd(&lds_hist[a.n_bins + b], (double)(tw * tw));  // square in float_t, then widened (:445)
        } else {  // many bins: straight into this workgroup's replica
          atomicAdd(&a.hist[(size_t)rep * a.n_bins + b], (double)tw);
          if (a.fill_w2) atomicAdd(&a.hist[(size_t)(a.hist_rep + rep) * a.n_bins + b], (double)(tw * tw));
        }
      }
    }
  }
  if (a.hist_lds) {
    __syncthreads();
    for (int i = lane; i < nb; i += 256) {
      const double v = lds_hist[i];
      const int slot = i < a.n_bins ? rep * a.n_bins + i : (a.hist_rep + rep) * a.n_bins + (i - a.n_bins);
      if (v != 0.0) atomicAdd(&a.hist[slot], v);
    }
  }
}

void launch_reweight_lists(const ReweightArgs& a, const PosArgs& p, int mode, bool f64, const ReweightKnobs& k, hipStream_t s) {
  if (p.n_pos == 0) return;
  const size_t nb = a.hist_lds ? (size_t)(a.fill_w2 ? 2 : 1) * (size_t)a.n_bins : 0;
  const size_t lds = sizeof(double) * ((nb + 1) & ~size_t(1)) + (f64 ? sizeof(double) : sizeof(float)) * 4 * (kListsEPL * kListsRounds + 1) * 64;
  const dim3 g((unsigned)((p.n_pos + 256u * kListsEPL - 1) / (256u * kListsEPL))), b(256);
  if (k.ev_start || k.ev_stop) {
    if (f64) hipExtLaunchKernelGGL((k_reweight_lists<true>), g, b, (uint32_t)lds, s, k.ev_start, k.ev_stop, 0, a, p, mode);
    else hipExtLaunchKernelGGL((k_reweight_lists<false>), g, b, (uint32_t)lds, s, k.ev_start, k.ev_stop, 0, a, p, mode);
  } else {
    if (f64) hipLaunchKernelGGL((k_reweight_lists<true>), g, b, lds, s, a, p, mode);
    else hipLaunchKernelGGL((k_reweight_lists<false>), g, b, lds, s, a, p, mode);
  }
}

// k_pack_positions: the event side in processing order (PosArgs) from the per-event arrays, which stay in the caller's order
__global__ __launch_bounds__(256) void k_pack_positions(const uint32_t* __restrict__ perm, const int32_t* __restrict__ bin, const int32_t* __restrict__ osc,
                                                        const int32_t* __restrict__ bl_id, const int32_t* __restrict__ bl_off, uint64_t n, int32_t* __restrict__ p_bin,
                                                        int32_t* __restrict__ p_osc, uint32_t* __restrict__ p_loff, uint16_t* __restrict__ p_llen) {
  const uint64_t pos = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (pos >= n) return;
  const uint32_t e = perm[pos];
  p_bin[pos] = bin[e];
  p_osc[pos] = osc[e];
  const int id = bl_id[e];
  const int o0 = id >= 0 ? bl_off[id] : 0, o1 = id >= 0 ? bl_off[id + 1] : 0;
  p_loff[pos] = (uint32_t)o0;
  p_llen[pos] = (uint16_t)(o1 - o0);
}
void launch_pack_positions(const uint32_t* perm, const int32_t* bin, const int32_t* osc, const int32_t* bl_id, const int32_t* bl_off, uint64_t n, int32_t* p_bin,
                           int32_t* p_osc, uint32_t* p_loff, uint16_t* p_llen, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_pack_positions, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, perm, bin, osc, bl_id, bl_off, n, p_bin, p_osc, p_loff, p_llen);
}

// ------------------------------------------------------------------------------------
// k_reweight_brick (description at the kernel)
// ------------------------------------------------------------------------------------
constexpr int kBrickThreads = 1024;                                // lane l owns events l, l + 1024, ...; 2 workgroups per CU = 8 waves per SIMD
constexpr int kBrickRPL = M3_STAGE_CAP / kBrickThreads;            // responses per lane and stage = coefficient loads in flight per lane
constexpr int kBrickHdrWords = M3_BRICK_HDR0 + M3_STAGE_HDR * (M3_BRICK_MAX_STAGES + 2);  // + 2: look-ahead reads past the last stage see zeros
constexpr int kBrickLdsParams = 1024;                              // per-step parameter table {segment, dx} kept in LDS up to this P
// a lane past a stage's end works on whatever word follows the stage's list; its run index can be off by up to 64 runs
constexpr int kBrickRunSlots = M3_STAGE_MAX_RUNS + 64 + 1;
constexpr int kBrickGrpPerWave = 8;                                // uint16 per wave and stage: one 16-byte LDS read
constexpr int kBrickTf1BatchMax = 16;                              // TF1 rows of the first batch (the registers the quads held); 12 at G = 4
constexpr int kBrickEllSlack = 16;                                 // an event's lane reads up to 16 slots from its first one
static_assert(M3_STAGE_CAP % kBrickThreads == 0, "a stage is a whole number of responses per lane");
static_assert(kBrickHdrWords <= kBrickThreads, "the stage headers are fetched by one load per lane");
static_assert(M3_STAGE_MAX_RUNS <= kBrickThreads && M3_STAGE_PAIR_WORDS <= kBrickThreads, "run entries and pair words: one per lane");
static_assert(M3_GROUP == kBrickThreads, "lane = event of a slab");
static_assert(kBrickRPL <= kBrickGrpPerWave, "group entries of a wave");
static_assert(16 * M3_BRICK_MAX_GROUPS <= M3_STAGE_PAIR_WORDS, "a wave-block's first slot travels in the high half of a pair word");
static_assert(4 * (M3_STAGE_CAP + kBrickEllSlack) <= 0x8000, "bits 2..14 of a response word are a byte offset into the scatter array");
// LDS budget: two workgroups per CU (what __launch_bounds__(1024, 8) and the 64-VGPR kernel are tuned for) leave each 80 KB of
// gfx950's 160 KB.  The static arrays below take 55 KB at any G (round 4: + the TF1 bookkeeping of slab 0 and the stages' count words); the dynamic bin vector of a fill launch is capped at
// M3_HIST_LDS_MAX_BYTES, so the sum fits whatever the bin space (round 2's [rank][event] scatter array did not leave that room).
constexpr size_t kBrickStaticLds = sizeof(float) * (M3_STAGE_CAP + kBrickEllSlack) + 8 * 2 * (size_t)kBrickRunSlots + 2 * 2 * 16 * kBrickGrpPerWave +
                                   2 * 2 * 16 * M3_BRICK_MAX_GROUPS + 4 * (size_t)kBrickHdrWords + 8 * (size_t)kBrickLdsParams + 4 * ((size_t)kBrickLdsParams + 1) +
                                   4 * 64 + 4 * ((size_t)kBrickThreads + 32) + 2 * 512 * (size_t)M3_BRICK_MAX_GROUPS;
static_assert(kBrickStaticLds + M3_HIST_LDS_MAX_BYTES <= 80 * 1024, "two brick workgroups per CU: static LDS + the largest LDS bin vector must fit 80 KB");
// Idle lanes do not branch around their loads (see the note in the kernel): they read a few KB past the list they index.
// m3hip_upload.hip allocates every chunk's stores with this much room behind them.
static_assert(M3_STORE_SLACK_BYTES >= 2 * M3_STAGE_CAP + 8 * kBrickThreads, "room behind the stores for idle lanes' reads");

// Diagnostic build only (-DM3_BRICK_STAMPS, `make stamps`): lane 0 of every brick records the 100 MHz wall clock at the
// kernel's phase boundaries into a device array read back with m3hip_debug_read_stamps.  The production build carries
// none of this.
#ifdef M3_BRICK_STAMPS
constexpr int kStampSlots = 16, kStampBricks = 16384;
__device__ unsigned long long g_brick_stamps[kStampBricks * kStampSlots];
#define M3_STAMP(k)                                                                                       \
  do {                                                                                                    \
    if (threadIdx.x == 0 && blockIdx.x < kStampBricks) g_brick_stamps[blockIdx.x * kStampSlots + (k)] = wall_clock64(); \
  } while (0)
extern "C" int m3hip_debug_read_stamps(unsigned long long* out, unsigned long long n_words) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_brick_stamps), sizeof(unsigned long long) * n_words, 0, hipMemcpyDeviceToHost);
}
// slot 12: where the workgroup ran -- HW_ID (hwreg 4: cu_id bits 8..11, sh_id 12, se_id 13..15) | XCC_ID (hwreg 20) << 32
#define M3_STAMP_HWID()                                                                                   \
  do {                                                                                                    \
    if (threadIdx.x == 0 && blockIdx.x < kStampBricks)                                                    \
      g_brick_stamps[blockIdx.x * kStampSlots + 12] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | \
                                                      ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); \
  } while (0)
#else
#define M3_STAMP(k) do {} while (0)
#define M3_STAMP_HWID() do {} while (0)
#endif
// Diagnostic build only (-DM3_BRICK_HOLLOW, `make hollow`): the brick kernel with its arithmetic and its LDS traffic taken
// out -- every load, wait and barrier stays, a loaded register is "used" by an empty asm instead of being evaluated, scattered,
// multiplied.  What such a launch takes is the memory side's share of the kernel's time (results are garbage).
#ifdef M3_BRICK_HOLLOW  // 1: as described; 2: also without the stage's second barrier; 3: also without the TF1 rows; 4: level 1 without the flush of the bin vector; 5: level 1 with the flush's barrier but without its atomics
#define M3_HOLLOW_USE4(v) asm volatile("" ::"v"((v).x), "v"((v).y), "v"((v).z), "v"((v).w))
#define M3_HOLLOW_USE1(v) asm volatile("" ::"v"(v))
#endif

// A wave-uniform address, pinned into an SGPR pair: the lane-dependent part of a load's address then stays a 32-bit
// offset register (`global_load v, v_off, s[base]`).  Without the pin hipcc hoists base + lane offset out of loops
// as a 64-bit register pair per stream.
template <class T>
__device__ __forceinline__ const M3_GLOBAL T* uniform_base(const M3_GLOBAL T* p) {
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (const M3_GLOBAL T*)(((uint64_t)hi << 32) | lo);
}

// Which of a brick's streams go around the Infinity Cache.  Non-temporal loads do not allocate in its 256 MB and a plain stream
// evicts it (tools/mall_probe.hip), so the two large streams a step reads once besides the coefficients -- the TF1 rows (1.28 GB
// per step of config 3) and the response words (350 MB) -- are loaded non-temporally like the coefficients, and the small ones a
// step re-reads -- the stages' tables, the sample words, the TF1 bookkeeping, and the TF1 codes when they still fit the engine's
// budget (template parameter NTC = they do not: m3hip_step.hip choose_nt_codes) -- then stay there from step to step and come back
// in 545 instead of 900+ cycles, which is what a compute unit's outstanding misses turn into bandwidth (round 4, config 3:
// 712 -> 682-690 us).  NTC is a template parameter: as a kernel argument with a branch per load it cost 1 %.
struct RRun {      // per-step resolved run: quad address = base + r, dx of the run's parameter
  int32_t base;    // coef_off - start_r + seg * stride   (float4 units from the brick's store)
  float dx;
};

typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));
typedef u32x3_t u32x3_u __attribute__((aligned(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// k_reweight_brick: one 1024-lane workgroup per brick (<= G x 1024 events with ragged parameter lists); lane l owns the
// events l + 1024 h of the brick's slabs h = 0 .. G-1.
//
// TSpline3 responses, stage by stage (<= M3_STAGE_CAP responses): phase A evaluates the stage's responses
// lane-per-response in PARAMETER-major order -- a run's active segment plane is contiguous, so the float4 loads
// coalesce -- and scatters each weight to its slot of a compact LDS array (one float per response; the slot is the
// response's metadata word); phase B lets every event multiply its own responses, ascending parameter = the
// reference's order: an event's slots lie next to each other, its first one at the wave-block's first slot + the
// exclusive prefix sum of the events' 4-bit counts over the wave's lanes (a DPP scan), so a row costs one LDS read with
// an immediate offset, one select and one multiply.
// TF1 responses are read lane-per-event from the wave-block's jagged rows straight into the register product (no
// segment to select, so no regrouping and no LDS), slab after slab, each followed by its events' CalcWeightTotal + fill.
// HBM bytes per TSpline3 response: 16 (quad) + 2 (word) (+ the padding of the run's plane to a 128-B line);
// per TF1 response: 8 + 1; per run: 8; per event and stage: 1/2; per event: 1 + the sample side.
//
// Latency structure.  The coefficient loads of stage s+1 are issued BEFORE phase B of stage s (right after the barrier
// that publishes the stage's resolved run table), so a workgroup has quads in flight through its barriers and LDS
// phases; everything those loads need -- response words, run entries -- is requested one stage earlier still, and the
// parameters' {segment, dx} sit in LDS, so no dependent round trip lies between two stages' coefficient loads.
// What a workgroup pays once -- the first round trips, filling and draining that pipeline, the parameter table, the
// bin vector's flush, its own dispatch -- is spread over G x 1024 events; the LDS it needs does not grow with G.
//
// Written for instruction count: a response word is already the byte offset of its slot; idle lanes load from a clamped
// index instead of branching around the load.
//
// BIGP: more spline parameters than the LDS tables hold (kBrickLdsParams): {segment, dx} and the TF1 parameter values are
// then read from global memory where they are needed -- a dependent round trip per stage and per TF1 response; a
// separate instantiation, because a run-time branch around those loads would cost the common case its counted waits
// PACK: the event's bin, oscillation index and norm indices come as one 8-byte word (ReweightArgs::evw) instead of from three
// arrays; a template parameter, because both forms in one kernel cost the slabs' tails more registers than there are
template <int MODE, bool NT, bool BIGP = false, int G = 1, bool PACK = false, bool NTC = false>
__global__ __launch_bounds__(kBrickThreads, 8) void k_reweight_brick(ReweightArgs a) {
  __shared__ __attribute__((aligned(16))) float ell[M3_STAGE_CAP + kBrickEllSlack];  // compact scatter array: one slot per response of the stage
  __shared__ RRun lds_runs[2 * kBrickRunSlots];   // resolved run tables of two stages
  __shared__ __attribute__((aligned(16))) uint16_t lds_grp[2 * 16 * kBrickGrpPerWave];  // [stage & 1][wave][u]: run of the group's first response
  __shared__ __attribute__((aligned(4))) uint16_t lds_woff[2 * 16 * G];       // [stage & 1][slab][wave]: first slot of the wave-block's rows
  __shared__ __attribute__((aligned(16))) uint32_t lds_hdr[kBrickHdrWords];
  __shared__ PTab lds_ptab[kBrickLdsParams];
  __shared__ float lds_tf1x[kBrickLdsParams + 1];  // [1 + k]: parameter value of entry k of the TF1 dictionary
  __shared__ u32x2_t lds_cnt[2 * 64 * G];  // [stage & 1]: the events' response counts of the stage, as the planner stores them
  __shared__ uint32_t lds_book[kBrickThreads];  // slab 0's TF1 bookkeeping per lane-slot: code count | the slot's event << 16
  __shared__ uint32_t lds_bookw[32];            // [wave]: first float2 of the wave-block's rows, [16 + wave]: first code dword
  __shared__ float lds_nv[64];  // PACK: the norm parameters the packed words' 6-bit codes can name, float(params[evw_norm_base + k]) (SampleHandlerFD.cpp:575: the cast the reference makes per event)
  extern __shared__ double lds_hist[];
  const uint32_t lane = threadIdx.x;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(lane >> 6);
  const BrickDev bd = a.bricks[a.tile_base + blockIdx.x];
  const M3_GLOBAL uint32_t* __restrict__ words = as_global(bd.words);
  const M3_GLOBAL char* __restrict__ wbytes = (const M3_GLOBAL char*)words;
  const M3_GLOBAL PTab* __restrict__ ptab = as_global(a.ptab);
  const uint32_t n_stages = bd.n_stages;
  // The kernel's arguments, re-read from the kernarg segment where they are used.  hipcc loads every argument at the kernel's
  // top and keeps it in SGPRs to its last use -- for the sample side's two dozen pointers that is the end of the last slab,
  // and the scalar registers they would occupy through the stage loop spill into vector registers the loop does not have.
  // Behind the empty asm the loads (scalar, served by the constant cache) cannot be hoisted.
  auto late_args = [&]() -> const ReweightArgs& {
    const __attribute__((address_space(4))) char* p = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const ReweightArgs*)p;
  };
  constexpr bool tab_in_lds = !BIGP;
  constexpr int kWordRegs = (kBrickRPL + 1) / 2;

  // What one stage's coefficient loads need, requested ahead of time (global loads only, nothing here waits).
  // NOTE: no branch may enclose a vector-memory instruction between the prologue and the last stage: hipcc's wait-count
  // insertion falls back to vmcnt(0) at control-flow joins with loads pending, which would drain the quads in flight at
  // every table build.  Hence no predicated loads: a lane past the end of a list reads what lies behind it (stages that
  // do not exist have an all-zero header and read the brick's first words), and nothing is ever done with the value.
  struct Meta {
    uint32_t w[kWordRegs];  // response words, two per register
    u32x2_t run;            // the stage's tables, 8 bytes per lane: lanes < n_runs the raw entry of run `lane`; lanes 512 .. 551 two pair
                            // words each (run of a group's first response | first slot of a wave-block << 16); lanes 576 .. the events' counts
  };
  struct StageHdr { uint32_t h0, off_runs, off_pairs, off_resp; };  // wave-uniform (SGPRs): bases of SGPR + lane-offset loads
  auto fetch_meta = [&](const StageHdr& hd, Meta& m) {
    // lanes past the end of a list re-read its last element (one line): an unclamped read would drag 10+ KB of foreign
    // lines per stage through the memory system
    const uint32_t n_resp = hd.h0 & 0xFFFFu, n_runs = (hd.h0 >> 16) & 0x3FFu;
    {
      // the lane's response words: ONE load of three dwords from its record of 1 .. 5 uint16 (m3hip_layout.cpp) -- every other
      // record of an odd length starts on a 2-byte boundary: the load is declared so and stays one global_load_dwordx3.  What lies
      // behind a shorter record belongs to the next lane and is never looked at as a response (scatter and ballots check the count).
      // Lanes without a response re-read the last record.
      const uint32_t rec = 2u * ((n_resp + 1023u) / 1024u);  // bytes, scalar
      const uint32_t lanes = n_resp < (uint32_t)kBrickThreads ? (n_resp ? n_resp : 1u) : (uint32_t)kBrickThreads;
      const uint32_t l = lane < lanes - 1u ? lane : lanes - 1u;
      const u32x3_t v = __builtin_nontemporal_load((const M3_GLOBAL u32x3_u*)(uniform_base(wbytes + 4u * hd.off_resp) + rec * l));
      static_assert(kWordRegs == 3, "a lane's response words of a stage travel as three dwords");
      m.w[0] = v.x;
      m.w[1] = v.y;
      m.w[2] = v.z;
    }
    {
      // ONE 8-byte load per lane for the stage's tables: lanes 0 .. 511 the run entries, 512 .. 551 the pair words (two each),
      // 576 .. 576 + 64 G - 1 the events' counts (they follow the pair words); every other lane re-reads the last run entry.
      // (A load per table cost every wave three vector-memory instructions a stage, two of them idle in 14 waves of 16.)
      const uint32_t last = n_runs ? n_runs - 1u : 0u, l = lane < last ? lane : last;
      uint32_t off = 4u * hd.off_runs + 8u * l;
      off = lane - 512u < (uint32_t)(M3_STAGE_PAIR_WORDS / 2) ? 4u * hd.off_pairs + 8u * (lane - 512u) : off;
      off = lane - 576u < 64u * G ? 4u * (hd.off_pairs + (uint32_t)M3_STAGE_PAIR_WORDS) + 8u * (lane - 576u) : off;
      m.run = *(const M3_GLOBAL u32x2_t*)(uniform_base(wbytes) + off);
    }
  };
  auto word = [&](const Meta& m, int u) -> uint32_t { return (u & 1) ? (m.w[u / 2] >> 16) : (m.w[u / 2] & 0xFFFFu); };
  // the responses this lane's G events hold in a stage: a nibble each of the stage's count word (lds_cnt)
  auto count_of = [&](uint32_t packed, int h) -> uint32_t {
    if (G != 1) return (packed >> (4 * h)) & 15u;
    // G = 1: two events share a byte.  The lane's shift is recomputed where it is used: hoisted out of the stage loop it would
    // cost a register the loop does not have (a spill there means a scratch reload = a wait for every quad in flight)
    uint32_t l = lane;
    asm volatile("" : "+v"(l));
    return (packed >> ((l >> 7) & 4u)) & 15u;
  };

  M3_STAMP(0);
  M3_STAMP_HWID();
  // ---- prologue: ONE round trip and ONE barrier before the first coefficient loads.  The descriptor carries the headers of
  // stages 0 and 1, so their metadata is requested together with the header block, the parameter table and the TF1 side's
  // bookkeeping; stage 0's runs take their {segment, dx} straight from global memory (a second, L2-served round trip for
  // the few lanes that hold a run) instead of waiting for the table to be visible in LDS behind a barrier of its own.
  uint32_t hdr_w = 0;
  if (lane < kBrickHdrWords) hdr_w = lane < M3_BRICK_HDR0 + M3_STAGE_HDR * n_stages ? words[lane] : 0u;
  Meta m_cur, m_nxt;
  fetch_meta(StageHdr{bd.st0[0], bd.st0[1], bd.st0[2], bd.st0[3]}, m_cur);
  // stage 1's metadata is requested BEFORE stage 0's coefficient loads: vector memory returns in order, so whatever a
  // stage's tables need has to be older than the quads that are in flight while the tables are built
  fetch_meta(StageHdr{bd.st1[0], bd.st1[1], bd.st1[2], bd.st1[3]}, m_nxt);
  const M3_GLOBAL uint32_t* __restrict__ tf1h = words + bd.off_tf1;
  // TF1 bookkeeping of a slab: the code count of the lane's event (a byte) and the wave-block's first slot / first code dword.
  // Slab 0's is requested here, slab h + 1's when slab h begins.
  // cnt: code count of the lane-slot | the slot's event (relative to the slab) << 16.  The slots of a slab take its events in
  // order of descending count (m3hip_layout.cpp), so that the events of a wave have about equal counts
  struct Tf1Book { uint32_t cnt, cbase, dbase; };
  // Where no event of the brick has more than 63 codes (bd.pad_[1] bit 1) the count sits in the sigma word's bits 10..15 and the
  // byte array is not in the stream: a byte per event less to read.  Brick-uniform, but neither a branch (see the note at
  // fetch_meta) nor a select (the instantiations without the packed sample word have no register for its mask): the byte load of
  // such a brick reads, in every lane, a byte of the stream's header that is zero, and the count is the sum of the two fields.
  // (Not for bricks of 2048 events, m3hip_layout.h kTf1CntInSigmaG2: those instantiations without the packed sample word are the
  // ones the two scalars below push into scratch.)
  auto book_word = [&](const M3_GLOBAL char* cb, uint32_t k) -> uint32_t {
    if (G == 2 && !kTf1CntInSigmaG2) {
      const uint32_t c8 = *(const M3_GLOBAL uint8_t*)(cb + k);
      const uint32_t s16 = *(const M3_GLOBAL uint16_t*)(cb + 1024u * G + 2u * k);
      return c8 | s16 << 16;
    }
    const uint32_t in_sigma = (bd.pad_[1] >> 1) & 1u;  // scalar
    const uint32_t c8 = *(const M3_GLOBAL uint8_t*)(uniform_base(in_sigma ? wbytes + 4u * 6u : cb) + k * (1u - in_sigma));
    const uint32_t s16 = *(const M3_GLOBAL uint16_t*)(uniform_base(cb + (in_sigma ? 0u : 1024u * G)) + 2u * k);
    return c8 | s16 << 16;  // count in bits 0..7 or in bits 26..31 (the other field 0) | the slot's event in bits 16..25
  };
  auto book_count = [](uint32_t bw) -> uint32_t { return (bw & 0xFFu) + (bw >> 26); };
  auto book_event = [](uint32_t bw) -> uint32_t { return (bw >> 16) & 1023u; };
  auto tf1_book = [&](uint32_t h) -> Tf1Book {
    Tf1Book b;
    uint32_t le = lane;
    asm volatile("" : "+v"(le));  // as in tail_issue_idx: no hoisting of G slabs' addresses
    const M3_GLOBAL char* cb = uniform_base(wbytes + 4u * (bd.off_tf1 + 32u * G));  // SGPR base + 32-bit offsets
    b.cnt = book_word(cb, 1024u * h + le);
    b.cbase = __builtin_amdgcn_readfirstlane(tf1h[16u * h + wave]);
    b.dbase = __builtin_amdgcn_readfirstlane(tf1h[16u * G + 16u * h + wave]);
    return b;
  };
  Tf1Book book_nxt = {0u, 0u, 0u};
  // (a constant since round 4's last kernel step -- every G has the register now; the other path is kept as written because
  // removing it perturbs the G = 4 instantiations' register allocation, which could not be re-measured)
  constexpr bool kBookLds = true;
  // slab 0's bookkeeping comes in with the prologue's round trip and waits in LDS for the last stage: requested there (as
  // slabs h > 0 still are) it put a round trip of its own in front of the first TF1 rows
  uint32_t book0_cnt = 0, book0_w = 0;
  if (kBookLds) {
    const M3_GLOBAL char* cb = uniform_base(wbytes + 4u * (bd.off_tf1 + 32u * G));
    book0_cnt = book_word(cb, lane);
    if (lane < 32) book0_w = tf1h[(lane < 16 ? 0u : 16u * G - 16u) + lane];
  } else if (n_stages == 0) {
    book_nxt = tf1_book(0);
  }
  const M3_GLOBAL RSlot* __restrict__ tf1x = as_global(bd.tf1_x);  // parameter value per dictionary entry (k_prep / k_resolve)
  constexpr bool x_in_lds = !BIGP;  // the dictionary holds spline parameters, so it fits whenever they do
  float x_own = 0.f;
  if (x_in_lds && lane < bd.n_dict) x_own = tf1x[lane].f;
  // The packed word's norm codes index a window of 63 parameters.  Their values (k_prep wrote float(value) beside the doubles:
  // the cast of SampleHandlerFD.cpp:575) come in with the prologue's round trip and sit in LDS: the slabs' tails take them from
  // there -- no global look-up behind the word (the tail's dependent chain is word -> weight, not word -> index -> value).
  float nv_own = 1.0f;
  if (PACK && (MODE & (kModeFill | kModeEventW)) && lane < 64 && a.params_f32) {
    const int32_t k = a.evw_norm_base + (int32_t)lane;
    nv_own = a.params_f32[k < a.n_par_total ? k : a.n_par_total];  // ([n_par_total] = 1.0)
  }
  PTab t_own;
  t_own.dx = 0.f;
  t_own.seg = 0;
  if (tab_in_lds && lane < (uint32_t)a.n_params) {
    t_own.dx = ptab[lane].dx;
    t_own.seg = ptab[lane].seg;
  }
  if ((MODE & kModeFill) && a.hist_lds) {
    const int nb = a.fill_w2 ? 2 * a.n_bins : a.n_bins;
    for (int b = lane; b < nb; b += kBrickThreads) lds_hist[b] = 0.0;
  }
  // stage s's run table, group table and wave-block offsets -> LDS; the parameters' {segment, dx} from LDS (FROM_LDS) or
  // from global memory
  auto publish = [&](uint32_t s, uint32_t h0, const Meta& m, auto from_lds_tag) {
    constexpr bool FROM_LDS = decltype(from_lds_tag)::value;
    const uint32_t n_runs = (h0 >> 16) & 0x3FFu;
    uint32_t lane = threadIdx.x;
    if (G != 1) asm volatile("" : "+v"(lane));  // G = 2 / 4 have no register to spare for LDS addresses hoisted out of the stage loop
    if (lane < n_runs + 65u) {
      const uint32_t p = lane < n_runs ? m.run.y >> 16 : 0u;  // (lanes past the runs may hold pair words or counts)
      PTab t;
      if (FROM_LDS) {
        t = lds_ptab[p];
      } else {  // stage 0, or more parameters than the LDS table holds: a dependent (L2) round trip
        t.dx = ptab[p].dx;
        t.seg = ptab[p].seg;
      }
      RRun rr;
      // entries behind the last run: what lanes past the stage's end index (their words are not theirs).  The base
      // sends their loads far past the store, where the offset clamp turns all of them into one and the same line.
      rr.base = lane < n_runs ? (int32_t)m.run.x + t.seg * (int32_t)(m.run.y & 0xFFFFu) : 0x07000000;
      rr.dx = t.dx;
      lds_runs[(s & 1u) * kBrickRunSlots + lane] = rr;
    }
    // group g = u * 16 + wave -> [wave][u]: a wave fetches its entries of a stage with one 16-byte read
    if (lane - 512u < (uint32_t)(M3_STAGE_PAIR_WORDS / 2)) {
      const uint32_t g = 2u * (lane - 512u);  // this lane holds pair words g and g + 1: (g & 15) + 1 <= 15, same g >> 4
      uint16_t* grp = lds_grp + (s & 1u) * 16 * kBrickGrpPerWave + (g >> 4);
      grp[(g & 15u) * kBrickGrpPerWave] = (uint16_t)m.run.x;
      grp[((g & 15u) + 1u) * kBrickGrpPerWave] = (uint16_t)m.run.y;
      if (g < (uint32_t)(16 * G)) *(uint32_t*)(lds_woff + (s & 1u) * 16 * G + g) = (m.run.x >> 16) | (m.run.y & 0xFFFF0000u);
    }
    if (lane - 576u < 64u * G) lds_cnt[(s & 1u) * 64 * G + (lane - 576u)] = m.run;
  };
  publish(0, bd.st0[0], m_cur, std::false_type{});
  if (lane < kBrickHdrWords) lds_hdr[lane] = hdr_w;
  if (x_in_lds && lane < bd.n_dict) lds_tf1x[1 + lane] = x_own;
  if (lane == 0) lds_tf1x[0] = 0.f;
  if (PACK && lane < 64) lds_nv[lane] = nv_own;
  if (tab_in_lds && lane < (uint32_t)a.n_params) lds_ptab[lane] = t_own;
  if (kBookLds) {
    lds_book[lane] = book0_cnt;
    if (lane < 32) lds_bookw[lane] = book0_w;
  }
  __syncthreads();  // stage 0's tables, the headers and the parameter table visible
  M3_STAMP(1);
  M3_STAMP(2);
  auto hdr = [&](uint32_t s) -> StageHdr {
    const u32x4_t v = *(const u32x4_t*)(lds_hdr + M3_BRICK_HDR0 + M3_STAGE_HDR * s);
    return StageHdr{(uint32_t)__builtin_amdgcn_readfirstlane(v.x), (uint32_t)__builtin_amdgcn_readfirstlane(v.y),
                    (uint32_t)__builtin_amdgcn_readfirstlane(v.z), (uint32_t)__builtin_amdgcn_readfirstlane(v.w)};
  };

  float4 c[kBrickRPL];
  float dx[kBrickRPL];
  const M3_GLOBAL char* __restrict__ spl_bytes = (const M3_GLOBAL char*)as_global(bd.spl);
  const uint32_t spl_last = bd.spl_last_off;  // byte offset of the store's last quad
  // coefficient loads of stage s (its tables visible): response lane + u * 1024
  auto issue_quads = [&](uint32_t s, const Meta& m, auto&& between) {
    const u32x4_t g4 = *(const u32x4_t*)(lds_grp + (s & 1u) * 16 * kBrickGrpPerWave + wave * kBrickGrpPerWave);
    const uint32_t g[4] = {g4.x, g4.y, g4.z, g4.w};
    const RRun* runs = lds_runs + (s & 1u) * kBrickRunSlots;
#pragma unroll
    for (int u = 0; u < kBrickRPL; ++u) {
      between(u);
      // the run of a response: the run of its group's first response + the runs that END before it inside the group
      const unsigned long long ends = __ballot((m.w[u / 2] & (1u << (16 * (u & 1)))) != 0);
      const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(ends >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ends, 0u));
      const uint32_t first = __builtin_amdgcn_readfirstlane((g[u / 2] >> (16 * (u & 1))) & 0xFFFFu);
      const RRun rr = runs[first + before];
      dx[u] = rr.dx;
      // SGPR base + 32-bit byte offset (a brick's store is far below 4 GB).  A lane past the stage's end indexes one
      // of the entries behind the last run: its offset is clamped to the store's last quad, loaded, never used.
      uint32_t off = ((uint32_t)rr.base + (lane + u * kBrickThreads)) << 4;
      off = off < spl_last ? off : spl_last;
      const M3_GLOBAL f32x4_t* gp = (const M3_GLOBAL f32x4_t*)(spl_bytes + off);
      const f32x4_t v = NT ? __builtin_nontemporal_load(gp) : *gp;
      c[u] = make_float4(v.x, v.y, v.z, v.w);
    }
  };
  issue_quads(0, m_cur, [](int) {});

  // ---- TF1 side (SplineMonolith.cpp:818-825: after the TSpline3s, in the event's parameter order): one batch of rows of
  // the CURRENT slab's wave-block
  constexpr int kBrickTf1Batch = G == 4 ? 12 : kBrickTf1BatchMax;  // four products live through the slabs leave G = 4 a register short of 16 rows
  float2 ab[kBrickTf1Batch];
  uint32_t code[kBrickTf1Batch / 4];
  bool tf1_on[kBrickTf1Batch / 4];
  uint32_t tf1_cnt = 0, tf1_cbase = 0, tf1_dbase = 0;
  uint32_t ev_slot = 0;  // the event (relative to its slab) this lane works on from here on: sigma[slot]
  const M3_GLOBAL char* __restrict__ tf1_bytes = (const M3_GLOBAL char*)as_global(bd.tf1);
  // loads of rows j0 .. j0 + N - 1 into registers R0 .. R0 + N - 1 of the batch (nothing here waits)
  // kTrim (bricks of several slabs): four rows none of the wave's events has are not requested at all -- the slots are sorted by
  // code count, so the waves at the end of a slab stop early, and with 8 codes per event (config 5) most waves of a slab do: one
  // GPU's share of config 5 480 -> 475 us.  The branch costs the waits behind it nothing (they are for everything anyway).  At
  // G = 1 (config 3: 16 codes per event, the first batch is full in nearly every wave) it measured level to 0.3 % worse: off.
  constexpr bool kTrim = G != 1;
  auto tf1_issue = [&](uint32_t j0, auto nrows_tag, auto r0_tag) {
    constexpr int N = decltype(nrows_tag)::value, R0 = decltype(r0_tag)::value;
    static_assert(N % 4 == 0 && R0 % 4 == 0, "a code dword covers four rows");
    auto row = [&](int i) {
      const unsigned long long act = __ballot(tf1_cnt > j0 + i);
      const uint32_t pos = tf1_cbase + __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
      // non-temporal: a seam line wanted by two rows is asked for twice (cached loads were 0.7 % ahead on that in round 3), but the
      // rows are 1.3 GB a step that would sweep the Infinity Cache clean of what the kernel re-reads (see the note above RRun)
      const f32x2_t v = __builtin_nontemporal_load((const M3_GLOBAL f32x2_t*)(tf1_bytes + 8u * pos));
      ab[R0 + i] = make_float2(v.x, v.y);
      tf1_cbase += (uint32_t)__popcll(act);
    };
#pragma unroll
    for (int q = 0; q < N / 4; ++q) {
      const unsigned long long act = __ballot(tf1_cnt > j0 + 4 * q);
      tf1_on[R0 / 4 + q] = tf1_cnt > j0 + 4 * q;
      if (kTrim && q > 0 && act == 0ull) {
        code[R0 / 4 + q] = 0u;
        continue;
      }
      const uint32_t pos = tf1_dbase + __builtin_amdgcn_mbcnt_hi((uint32_t)(act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)act, 0u));
      const M3_GLOBAL uint32_t* cp = (const M3_GLOBAL uint32_t*)(wbytes + 4u * pos);
      code[R0 / 4 + q] = NTC ? __builtin_nontemporal_load(cp) : *cp;  // idle lanes: some word behind the row ...
      tf1_dbase += (uint32_t)__popcll(act);
      if (kTrim) {
#pragma unroll
        for (int i = 4 * q; i < 4 * q + 4; ++i) row(i);
      }
    }
    if (!kTrim) {
#pragma unroll
      for (int i = 0; i < N; ++i) row(i);
    }
  };
  constexpr std::integral_constant<int, 0> kR0{};
  constexpr std::integral_constant<int, kBrickTf1Batch> kFirstBatch{};
  // ---- sample side of the slab's event (CalcWeightTotal's inputs), requested while the TF1 rows stream: first the event's
  // indices (with the first TF1 rows), then what they point at (small tables that live in L2); the tail finds both there
  constexpr int kTailNorm = 4;  // norm slots fetched ahead; further ones are read in the tail itself
  uint32_t ev_own = 0;  // this lane's event of the slab (an engine holds fewer than 2^32)
  bool ev_valid = false;
  int32_t t_bin = -1, t_osc = -1, t_ni[kTailNorm];
  uint32_t t_codes = 0;
  uint64_t t_pack = 0;
  float t_oscw = 1.f;
  double t_nv[kTailNorm];
  constexpr bool kEvwLin = G == 1 && PACK && (MODE & (kModeFill | kModeEventW)) != 0;
  auto tail_issue_idx = [&](uint32_t h) {
    const ReweightArgs& a = late_args();
    // (the lane number passes through an empty asm: the addresses below must be formed HERE, slab by slab; hoisted to the
    // kernel's top for all G slabs at once they cost more registers than the stage loop leaves, i.e. spills)
    uint32_t le = ev_slot + 1024u * h;
    ev_valid = le < bd.n_events;
    ev_own = bd.first_event + (ev_valid ? le : 0u);  // idle lanes shadow the brick's first event
    if (a.perm) ev_own = a.perm[ev_own];  // processing order -> event; kernel-uniform branch: only grouped uploads pay this dependent load
    if (MODE & (kModeFill | kModeEventW)) {
      // the event's oscillation weight itself when the engine has expanded a large, static table into a per-event column
      // (ReweightArgs::oscw): requested with the indices, not behind them.  No branch around the load: without the column
      // every lane reads the same word of the brick descriptors and drops it.
      if (a.oscw) t_oscw = a.oscw[ev_own];  // (kernel-uniform branch; the waits behind it are for everything anyway)
    }
    if ((MODE & (kModeFill | kModeEventW)) && PACK) {  // the event's indices in one 8-byte word
      if (kEvwLin) {
        // ... of the event the lane held through the stages (lane = event: one coalesced load); the words change lanes with
        // the TSpline3 products.  Gathered by lane-slot, every wave asked for most of the brick's 64 lines of them.
        uint32_t ll = lane;
        asm volatile("" : "+v"(ll));
        uint32_t ev_lin = bd.first_event + (ll < bd.n_events ? ll : 0u);
        if (a.perm) ev_lin = a.perm[ev_lin];
        t_pack = a.evw[ev_lin];
      } else {
        t_pack = a.evw[ev_own];
      }
      return;
    }
    if (MODE & (kModeFill | kModeEventW)) {
      t_osc = a.osc[ev_own];
#pragma unroll
      for (int m = 0; m < kTailNorm; ++m) {  // no branch around the load (wait counts, see above): absent slots re-read a valid word
        const bool has = (uint32_t)m < a.n_norm_slots;
        const int16_t* src = has ? a.nidx + (uint64_t)m * a.e_pad + ev_own : (const int16_t*)a.bricks;
        const int16_t v = *src;
        t_ni[m] = has ? (int32_t)v : -1;
      }
    }
    if (MODE & kModeFill) t_bin = a.bin[ev_own];
  };
  auto tail_issue_val = [&]() {
    const ReweightArgs& a = late_args();
    bool have_w = false;  // (kernel-uniform)
    if ((MODE & (kModeFill | kModeEventW)) && PACK) {
      const uint32_t lo = (uint32_t)t_pack, hi = (uint32_t)(t_pack >> 32);
      if (a.evw_weight) {  // the word carries the oscillation weight itself: nothing to look up
        t_bin = (int32_t)(lo & 0x3FFFu) - 1;
        t_codes = lo >> 14;
        t_oscw = __uint_as_float(hi);
        t_osc = 0;  // (1.0f stands for "no oscillation pointer")
        have_w = true;
      } else {
        t_bin = (int32_t)(lo & 0xFFFFFu) - 1;
        t_osc = (int32_t)(((lo >> 20) | (hi << 12)) & 0xFFFFFFu) - 1;
        t_codes = hi >> 12;  // three 6-bit norm codes: resolved against lds_nv in the tail
      }
    }
    if (MODE & (kModeFill | kModeEventW)) {
      // a sample without oscillated events has no table, one without norm parameters no index array: absent entries
      // read (and drop) the first bytes of the brick descriptors instead -- again no branch around the loads
      const float* ot = a.osc_table ? a.osc_table : (const float*)a.bricks;
      const double* pp = a.params ? a.params : (const double*)a.bricks;
      if (!a.oscw && !have_w) t_oscw = ot[t_osc >= 0 ? t_osc : 0];
      if (!PACK) {
#pragma unroll
        for (int m = 0; m < kTailNorm; ++m) t_nv[m] = pp[t_ni[m] >= 0 ? t_ni[m] : 0];
      }
    }
  };
  auto tf1_book0 = [&]() -> Tf1Book {  // slab 0's, from LDS
    Tf1Book b;
    uint32_t le = lane;
    asm volatile("" : "+v"(le));
    b.cnt = lds_book[le];
    b.cbase = __builtin_amdgcn_readfirstlane(lds_bookw[wave]);
    b.dbase = __builtin_amdgcn_readfirstlane(lds_bookw[16u + wave]);
    return b;
  };
  auto slab_begin = [&]() {  // the slab whose bookkeeping was requested last becomes the current one
    tf1_cnt = book_count(book_nxt.cnt);
    ev_slot = book_event(book_nxt.cnt);
    tf1_cbase = book_nxt.cbase;
    tf1_dbase = book_nxt.dbase;
  };
#if defined(M3_BRICK_HOLLOW) && M3_BRICK_HOLLOW == 3  // (hollow level 3: without the TF1 rows)
  const bool any_tf1 = false;
#else
  const bool any_tf1 = bd.n_dict > 0;  // kernel-uniform
#endif

  float w[G];  // SplineMonolith.cpp:798, one product per slab
#pragma unroll
  for (int h = 0; h < G; ++h) w[h] = 1.0f;
  // inclusive prefix sum over the wave's 64 lanes (row_shr 1, 2, 4, 8 inside the rows of 16, then the rows' totals)
  auto wave_scan = [](uint32_t x) -> uint32_t {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1 and 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2 and 3
    return x;
  };
  auto column = [&](const uint32_t (&block0)[G], uint32_t counts) {
#ifdef M3_BRICK_HOLLOW
    M3_HOLLOW_USE1(counts);
    return;
#endif
    uint32_t cnt[G], excl[G];
#pragma unroll
    for (int h = 0; h < G; ++h) cnt[h] = count_of(counts, h);
#pragma unroll
    for (int h = 0; h < G; h += 2) {  // sums stay below 64 * 15 < 2^16
      const uint32_t packed = cnt[h] | (h + 1 < G ? cnt[(h + 1) % G] << 16 : 0u);
      const uint32_t incl = wave_scan(packed);
      excl[h] = (incl & 0xFFFFu) - cnt[h];
      if (h + 1 < G) excl[(h + 1) % G] = (incl >> 16) - cnt[(h + 1) % G];
    }
#pragma unroll
    for (int h = 0; h < G; ++h) {
      const float* mine = ell + (block0[h] + excl[h]);
      float acc = w[h];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (__ballot(cnt[h] > (uint32_t)(4 * q)) == 0ull) break;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = mine[4 * q + i];  // lanes with fewer responses read a neighbour's slot (or the slack) and drop it
#pragma unroll
        for (int i = 0; i < 4; ++i) acc *= cnt[h] > (uint32_t)(4 * q + i) ? v[i] : 1.0f;
      }
      w[h] = acc;
    }
  };
  // One stage: tables of the next stage -> LDS while this stage's quads are in flight; then every quad is evaluated,
  // scattered to its slot of the scatter array, and its register reused at once for the next stage's load (LAST: for the
  // first TF1 rows instead); barrier; every event's responses of the stage, ascending parameter = reference order.
  // kind 0: a stage with at least two behind it; 1: the last but one (nothing left to request for a stage s + 2: without the
  // peel all 16 waves load five words of a header of zeros); 2: the last
  auto stage = [&](uint32_t s, auto kind_tag) {
    constexpr int KIND = decltype(kind_tag)::value;
    constexpr bool LAST = KIND == 2;
    constexpr bool AHEAD2 = KIND == 0;
    const uint32_t h0 = hdr(s).h0;
    const uint32_t n_resp = h0 & 0xFFFFu;
    Meta m_nn;
    if (LAST && !kBookLds) book_nxt = tf1_book(0);
    if (!LAST) {
      publish(s + 1, hdr(s + 1).h0, m_nxt, std::integral_constant<bool, tab_in_lds>{});  // waits for stage s+1's metadata only: it is older than the quads in flight
      if (AHEAD2) fetch_meta(hdr(s + 2), m_nn);
    }
    __syncthreads();  // stage s+1's tables visible; every wave is through its events' products of stage s-1: the scatter array is free
    // first slots of this wave's blocks in stage s.  Read HERE: the buffer they sit in is overwritten by the publish of stage
    // s+2, which a fast wave reaches right after the next barrier -- possibly while a slow one is still multiplying
    uint32_t block0[G];
#pragma unroll
    for (int h = 0; h < G; ++h) block0[h] = __builtin_amdgcn_readfirstlane((uint32_t)lds_woff[(s & 1u) * 16 * G + 16 * h + wave]);
    // the lane's events' counts of this stage (published with the stage's tables); read here for the same reason
    uint32_t counts;
    {
      uint32_t lc = threadIdx.x;
      asm volatile("" : "+v"(lc));
      const char* cb = (const char*)(lds_cnt + (s & 1u) * 64 * G);
      if (G == 1) counts = *(const uint8_t*)(cb + (lc & 511u));
      else if (G == 2) counts = *(const uint8_t*)(cb + lc);
      else counts = *(const uint16_t*)(cb + 2u * lc);
    }
    auto scatter = [&](int u) {
      // bits 2..14 of a response word: byte offset of its slot.  Lanes past the stage's end hold no response.
#ifdef M3_BRICK_HOLLOW
      const float4 cu = c[u];
      const float du = dx[u];
      M3_HOLLOW_USE4(cu);
      M3_HOLLOW_USE1(du);
      (void)n_resp;
      (void)word;
#else
      if (lane + u * kBrickThreads < n_resp)
        *(float*)((char*)ell + (word(m_cur, u) & 0x7FFCu)) = spline_poly(dx[u], c[u]);
#endif
    };
    if (!LAST) {
      issue_quads(s + 1, m_nxt, scatter);
    } else {
#pragma unroll
      for (int u = 0; u < kBrickRPL; ++u) scatter(u);
      if (kBookLds) book_nxt = tf1_book0();
      slab_begin();
      // the event's sample word BEFORE the rows: behind them (and behind the join of the branch around them) its request waited
      // for every row to arrive -- one more round trip in front of the oscillation look-up
      tail_issue_idx(0);
      if (any_tf1) tf1_issue(0, kFirstBatch, kR0);
    }
#if !defined(M3_BRICK_HOLLOW) || M3_BRICK_HOLLOW != 2  // (hollow level 2: without this barrier -- nothing is scattered there)
    __syncthreads();  // the stage's scatter is complete
#endif
    // every event's responses: they lie next to each other in the scatter array, from the wave-block's first slot + the
    // exclusive prefix sum of the counts over the wave's lanes on (a DPP scan; two slabs' counts share a register)
    column(block0, counts);
    if (!LAST) {
      m_cur = m_nxt;
      if (AHEAD2) m_nxt = m_nn;
    }
  };
  if (n_stages > 0) {
    for (uint32_t s = 0; s + 2 < n_stages; ++s) {
      stage(s, std::integral_constant<int, 0>{});
      if (s < 4) M3_STAMP(3 + s);
    }
    if (n_stages > 1) stage(n_stages - 2, std::integral_constant<int, 1>{});
    stage(n_stages - 1, std::integral_constant<int, 2>{});
    M3_STAMP(8);
  } else {  // (a brick without TSpline3 stages)
    if (kBookLds) book_nxt = tf1_book0();
    slab_begin();
    if (any_tf1) tf1_issue(0, kFirstBatch, kR0);
    tail_issue_idx(0);
  }

  // ---- the TSpline3 products change lanes: up to here lane l held event l of every slab, from here on it holds the event of
  // its lane-slot (sigma).  Through the scatter array, which the stages are done with.
  const bool resorted = (bd.pad_[1] & 1u) != 0;  // brick-uniform; otherwise sigma is the identity and the products stay where they are
  if ((n_stages > 0 || kEvwLin) && resorted) {
    __syncthreads();  // every wave is through its events' products of the last stage
#pragma unroll
    for (int h = 0; h < G; ++h) ell[1024 * h + lane] = w[h];
    if (kEvwLin) ((uint64_t*)(ell + 1024))[lane] = t_pack;  // (waits for the word; the first TF1 rows, younger, stay in flight)
    __syncthreads();
    if (kEvwLin) t_pack = ((const uint64_t*)(ell + 1024))[ev_slot];
  }
  // ---- slab after slab: the events' TF1 rows, then CalcWeightTotal and the fill.  The next slab's first rows and indices
  // are requested once this slab's tail is through (holding 16 rows in flight across the tail costs it its registers).
#pragma unroll
  for (uint32_t h = 0; h < (uint32_t)G; ++h) {
    if (h + 1 < (uint32_t)G) book_nxt = tf1_book(h + 1);
    tail_issue_val();  // waits for the indices (and with them the first TF1 rows, consumed next anyway)
    float wh = w[0];
    if (G > 1 && h == 1) wh = w[1 % G];
    if (G > 2 && h == 2) wh = w[2 % G];
    if (G > 2 && h == 3) wh = w[3 % G];
    if ((n_stages > 0 || kEvwLin) && resorted) wh = ell[1024u * h + ev_slot];
    if (any_tf1) {
      uint32_t kd4 = 0;  // 4 * (1 + position in the dictionary)
      auto tf1_rows = [&](auto nrows_tag, auto r0_tag) {  // registers R0 .. R0 + N - 1 of the batch
        constexpr int N = decltype(nrows_tag)::value, R0 = decltype(r0_tag)::value;
#pragma unroll
        for (int q = 0; q < N / 4; ++q) code[R0 / 4 + q] = tf1_on[R0 / 4 + q] ? code[R0 / 4 + q] : 0u;  // ... dropped here
#pragma unroll
        for (int i = R0; i < R0 + N; ++i) {
#ifdef M3_BRICK_HOLLOW
          const float2 abi = ab[i];
          const uint32_t cdw = code[i / 4];
          M3_HOLLOW_USE1(abi.x);
          M3_HOLLOW_USE1(abi.y);
          M3_HOLLOW_USE1(cdw);
          continue;
#endif
          // code 0: no response in this row (the event's list has ended: the planner zero-fills); M3_TF1_SKIP: a step of 254
          // without a response
          const uint32_t cd = (code[i / 4] >> (8 * (i % 4))) & 0xFFu;
          kd4 += (cd < 254u ? cd : 254u) << 2;
          const float x = x_in_lds ? *(const float*)((const char*)lds_tf1x + kd4) : tf1x[kd4 ? (kd4 >> 2) - 1u : 0u].f;
          const float t = fmaf(ab[i].x, x, ab[i].y);  // SplineMonolith.cpp:779
          wh *= (cd - 1u < (uint32_t)(M3_TF1_SKIP - 1)) ? t : 1.0f;
        }
      };
      tf1_rows(kFirstBatch, kR0);
      // further rows in batches of 8 (16 at a time measured 1.6 % slower on config 3: the extra loads of rows nobody has; requested
      // ahead by every wave as the first batch's registers come free, without a branch: +8 % -- round 4, 380 more loads per brick)
      constexpr std::integral_constant<int, 8> kNext{};
      for (uint32_t j0 = kBrickTf1Batch; __any(tf1_cnt > j0); j0 += 8) {
        tf1_issue(j0, kNext, kR0);
        tf1_rows(kNext, kR0);
      }
    }
    if (h == (uint32_t)(G - 1)) M3_STAMP(9);
    // what the requested values give: the norm and oscillation part of CalcWeightTotal (SampleHandlerFD.cpp:568-594: norm
    // pointers first, double -> float cast each, then total_weight_pointers = {osc, bin-spline weights, spline total weight})
    const uint64_t ev = ev_own;
    const bool valid = ev_valid;
    const int32_t bin_h = t_bin;
    float tw = 1.0f;
    if ((MODE & (kModeFill | kModeEventW)) && PACK) {
#pragma unroll
      for (int m = 0; m < M3_EVW_NORM_SLOTS; ++m) {
        const uint32_t cd = (t_codes >> (6 * m)) & 63u;
        const float nv = lds_nv[cd ? cd - 1u : 0u];
        if (cd) tw *= nv;
      }
    } else if (MODE & (kModeFill | kModeEventW)) {
#pragma unroll
      for (int m = 0; m < kTailNorm; ++m)
        if (t_ni[m] >= 0) tw *= static_cast<float>(t_nv[m]);
    }
    const float oscw = t_oscw;
    const bool has_osc = t_osc >= 0;
    const ReweightArgs& a = late_args();
    // a brick may hold fewer than G x 1024 events and start on any tile boundary: lanes past its last event
    // belong to somebody else's events and must touch nothing
    bool det_fill = false;
    int det_b = -1;
    float det_w = 0.f;
    if (valid) {
      if (MODE & kModeEval) a.spline_w[ev] = wh;
      if (MODE & (kModeFill | kModeEventW)) {
        // slots the look-ahead did not cover (the packed word holds an event's first M3_EVW_NORM_SLOTS, or none): read here, in slot order
        const uint32_t ahead = PACK ? a.evw_norm_slots : (uint32_t)kTailNorm;
        if (a.n_norm_slots > ahead) tw = mul_norm_weights(tw, a, ev, ahead);
        if (has_osc) tw *= oscw;
        tw = mul_binned_weights(tw, a, ev);
        tw *= wh;
        if (a.n_extra) tw = mul_extra_weights(tw, a, ev);  // AddAdditionalWeightPointers: behind the splines
        if (MODE & kModeEventW) a.event_w[ev] = tw;
        if (MODE & kModeFill) {
          const int b = bin_h;
          if (a.partial) {  // deterministic mode: filled below, wave after wave
            det_fill = !(tw <= 0.f) && b >= 0;
            det_b = b;
            det_w = tw;
          } else if (!(tw <= 0.f) && b >= 0) {  // a NaN weight fills, as in the reference
            if (a.hist_lds) {
              atomicAdd(&lds_hist[b], (double)tw);
              if (a.fill_w2) atomicAdd(&lds_hist[a.n_bins + b], (double)(tw * tw));
            } else {  // many bins: straight into this workgroup's replica
              const int rep = (int)(blockIdx.x % (uint32_t)a.hist_rep);
              atomicAdd(&a.hist[(size_t)rep * a.n_bins + b], (double)tw);
              if (a.fill_w2) atomicAdd(&a.hist[(size_t)(a.hist_rep + rep) * a.n_bins + b], (double)(tw * tw));
            }
          }
        }
      }
    }
    if ((MODE & kModeFill) && a.hist_lds && a.partial) {
      // deterministic mode, as in k_reweight_uniform: the waves add in turn, slab after slab (kernel-uniform branch)
      for (uint32_t wv = 0; wv < kBrickThreads / 64; ++wv) {
        if (wave == wv && det_fill) {
          atomicAdd(&lds_hist[det_b], (double)det_w);
          if (a.fill_w2) atomicAdd(&lds_hist[a.n_bins + det_b], (double)(det_w * det_w));
        }
        __syncthreads();
      }
    }
    if (h + 1 < (uint32_t)G) {
      if (h == 0) M3_STAMP(13);  // diagnostics: slab 0 is through
      slab_begin();
      if (any_tf1) tf1_issue(0, kFirstBatch, kR0);
      tail_issue_idx(h + 1);
    }
  }
  const ReweightArgs& af = late_args();
  if ((MODE & kModeFill) && af.hist_lds && af.partial) {  // deterministic mode: the bin vector goes to the brick's row of `partial`
    const int nb = af.fill_w2 ? 2 * af.n_bins : af.n_bins;
    double* row = af.partial + ((size_t)af.partial_row0 + blockIdx.x) * (size_t)(2 * af.n_bins);
    for (int i = lane; i < nb; i += kBrickThreads) row[i] = lds_hist[i];
    return;
  }
#if defined(M3_BRICK_HOLLOW) && M3_BRICK_HOLLOW == 4  // (hollow level 4: without the flush of the bin vector)
  if (false) {
#else
  if ((MODE & kModeFill) && af.hist_lds) {
#endif
    __syncthreads();
    M3_STAMP(10);
    const int nb = af.fill_w2 ? 2 * af.n_bins : af.n_bins;
    const int rep = (int)(blockIdx.x % (uint32_t)af.hist_rep);
    for (int i = lane; i < nb; i += kBrickThreads) {
      const double v = lds_hist[i];
      const int slot = i < af.n_bins ? rep * af.n_bins + i : (af.hist_rep + rep) * af.n_bins + (i - af.n_bins);
#if defined(M3_BRICK_HOLLOW) && M3_BRICK_HOLLOW == 5  // (hollow level 5: the flush's barrier and LDS reads, not its atomics)
      if (v == 123.456) atomicAdd(&af.hist[slot], v);
#else
      if (v != 0.0) atomicAdd(&af.hist[slot], v);
#endif
    }
  }
  M3_STAMP(11);
}

template <int MODE, int G>
static void launch_brick_g(const ReweightArgs& a, dim3 g, size_t lds, const ReweightKnobs& k, hipStream_t s) {
  const dim3 b(kBrickThreads);
  constexpr bool kSampleSide = (MODE & (kModeFill | kModeEventW)) != 0;  // only these read the packed word
  if (kSampleSide && a.evw) {
    if (a.n_params > kBrickLdsParams) M3_LAUNCH((k_reweight_brick<MODE, true, true, G, kSampleSide>), g, b, lds, s, k, a);
    else if (a.nt_codes) M3_LAUNCH((k_reweight_brick<MODE, true, false, G, kSampleSide, true>), g, b, lds, s, k, a);
    else M3_LAUNCH((k_reweight_brick<MODE, true, false, G, kSampleSide>), g, b, lds, s, k, a);
    return;
  }
  if (a.n_params > kBrickLdsParams) M3_LAUNCH((k_reweight_brick<MODE, true, true, G>), g, b, lds, s, k, a);
  else if (k.nt || G != 1) M3_LAUNCH((k_reweight_brick<MODE, true, false, G>), g, b, lds, s, k, a);
  else M3_LAUNCH((k_reweight_brick<MODE, false, false, 1>), g, b, lds, s, k, a);
}

template <int MODE>
static void launch_brick_mode(const ReweightArgs& a, dim3 g, size_t lds, const ReweightKnobs& k, hipStream_t s) {
  if (a.brick_groups == 4) launch_brick_g<MODE, 4>(a, g, lds, k, s);
  else if (a.brick_groups == 2) launch_brick_g<MODE, 2>(a, g, lds, k, s);
  else launch_brick_g<MODE, 1>(a, g, lds, k, s);
}

// (Samples with bricks and functional shifts are re-binned by a pass of its own in front of this kernel -- k_rebin_lds, k_rebin_pack,
// k_find_bins below.  Rounds 3-4 could also re-bin in this kernel's tail -- G = 1 with the packed word, no scratch -- and measured it
// at +159 us per step on config 3 against +60 for the pass: the cascade's look-ups come from L2 there.  Removed in round 5.)
void launch_reweight_brick(const ReweightArgs& a, uint32_t n_bricks, int mode, const ReweightKnobs& k, hipStream_t s) {
  if (n_bricks == 0) return;
  size_t lds = ((mode & kModeFill) && a.hist_lds) ? sizeof(double) * (a.fill_w2 ? 2 : 1) * (size_t)a.n_bins : 0;
  if (k.lds_pad > 0 && lds < (size_t)k.lds_pad) lds = (size_t)k.lds_pad;  // occupancy throttle (experiments)
  const dim3 g(n_bricks);
  if (mode == kModeEval) launch_brick_mode<kModeEval>(a, g, lds, k, s);
  else if (mode == kModeFill) launch_brick_mode<kModeFill>(a, g, lds, k, s);
  else launch_brick_mode<kModeFill | kModeEval | kModeEventW>(a, g, lds, k, s);
}

// ------------------------------------------------------------------------------------
// k_rebin_pack: the per-step re-binning of samples whose brick kernel cannot do it in its tail (G > 1): the event's bin from
// its shifted kinematics (d_event_bin) straight into the bin field of its packed sample word -- one pass, 8 B read + 8 B
// written per event beside the kinematics, instead of k_find_bins followed by a full k_pack_events (which re-read the
// oscillation and norm indices of every event every step: +27 % on a config-3 step)
__global__ __launch_bounds__(256) void k_rebin_pack(BinArgs a, uint64_t* evw) {
  const uint64_t ev = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (ev >= a.n_events) return;
  const uint64_t w = evw[ev];  // requested before the binning cascade's dependent look-ups
  const int b = d_event_bin(a, ev);
  a.bin[ev] = b;
  evw[ev] = (w & ~a.word_bin_mask) | (uint64_t)(uint32_t)(b + 1);
}
// The same with the binning tables in LDS (samples, axes, edges, shifts with their parameter values, cuts): the cascade of an
// event is a chain of dependent look-ups into tables of a few KB, and from L2 that chain -- not the 24-36 B of HBM traffic per
// event -- set the pass's time (160 us per 1e7 events, as a pass or in the brick kernel's tail).  A block brings the tables in
// once for 4096 events; the cached nominal bins are not read at all (the binary search over LDS edges gives the same index, and
// 4 B per event and axis stay in HBM).  WORD: the bin goes into the packed sample word's bin field only (m3hip_get_event_bins
// reads it from there); else into `bin`.
struct RebinLimits { static constexpr int kSamples = 32, kAxes = 64, kEdges = 2048, kShifts = 16, kCuts = 16; };
// d_event_bin on kinematics that are already in registers (xn: the event's nominal columns)
// `last_global`: the event's global bin of the previous step (-1: none) -- for 1-D samples the cascade's "nominal" bin: parameters
// move little from step to step, so the +-1 neighbour test (five independent LDS reads) nearly always replaces the binary search
// (eight dependent ones); every route gives the same index
template <int STRIDE>
__device__ __forceinline__ int d_event_bin_x(const BinArgs& a, int sid, const double (&xn)[STRIDE], int last_global) {
  const BinSample sm = a.samples[sid];
  double x[STRIDE];
#pragma unroll
  for (int c = 0; c < STRIDE; ++c) x[c] = xn[c];
  for (int s = 0; s < a.n_shifts; ++s) {
    const KinShift sh = a.shifts[s];
    if (sh.dim < STRIDE && (sh.sample < 0 || sh.sample == sid)) {
      const double th = a.params[sh.param];
      const int kind = sh.kind & 0xFF, aux = sh.kind >> 8;
#pragma unroll
      for (int c = 0; c < STRIDE; ++c) {  // (no dynamic index into the register arrays)
        if (c != sh.dim) continue;
        double xa = xn[0];
#pragma unroll
        for (int q = 1; q < STRIDE; ++q) xa = aux == q ? xn[q] : xa;
        x[c] = kind == 1 ? x[c] * (1.0 + th) : (kind == 2 ? x[c] + th : x[c] + th * xa);
      }
    }
  }
  bool out = false;
  for (int ci = 0; ci < a.n_cuts; ++ci) {
    const KinCut ct = a.cuts[ci];
    if (ct.column < STRIDE && (ct.sample < 0 || ct.sample == sid)) {
      double v = x[0];
#pragma unroll
      for (int q = 1; q < STRIDE; ++q) v = ct.column == q ? x[q] : v;
      if ((v < ct.lo) || (v >= ct.hi)) out = true;
    }
  }
  int global = 0;
#pragma unroll
  for (int d = 0; d < STRIDE; ++d) {
    if (d < sm.n_dim) {
      const BinAxis ax = a.axes[sm.axis_off + d];
      int hint = -1;
      if (d == 0 && sm.n_dim == 1 && last_global >= sm.global_offset && last_global - sm.global_offset < ax.n_bins) hint = last_global - sm.global_offset;
      const int b = d_find_bin(a.edges + ax.edge_off, ax.n_bins, x[d], hint);
      if (b < 0) out = true;
      global += b * ax.stride;
    }
  }
  return out ? -1 : global + sm.global_offset;  // (uniform samples only: rebin_tables_fit_lds)
}
// STRIDE = kin_stride when it is 1 or 2: the workgroups are persistent (the tables come in once), a lane works on U events per
// round whose loads all go out before the first is looked at; 0: any stride, one event after the other
template <bool WORD, int STRIDE>
__global__ __launch_bounds__(1024, 8) void k_rebin_lds(BinArgs a, uint64_t* __restrict__ evw, int n_samples, int n_axes, int n_edges) {
  __shared__ BinSample s_samples[RebinLimits::kSamples];
  __shared__ BinAxis s_axes[RebinLimits::kAxes];
  __shared__ double s_edges[RebinLimits::kEdges];
  __shared__ KinShift s_shifts[RebinLimits::kShifts];
  __shared__ double s_theta[RebinLimits::kShifts];
  __shared__ KinCut s_cuts[RebinLimits::kCuts];
  __shared__ int s_moves[RebinLimits::kSamples];  // whether a registered shift touches the sample: the others keep their nominal bins
  const int t = threadIdx.x;
  if (t < n_samples) {
    s_samples[t] = a.samples[t];
    int moves = 0;
    for (int q = 0; q < a.n_shifts; ++q) moves |= (a.shifts[q].sample < 0 || a.shifts[q].sample == t) ? 1 : 0;
    s_moves[t] = moves;
  }
  if (t < n_axes) s_axes[t] = a.axes[t];
  for (int i = t; i < n_edges; i += 1024) s_edges[i] = a.edges[i];
  if (t < a.n_shifts) {
    KinShift sh = a.shifts[t];
    s_theta[t] = a.params[sh.param];
    sh.param = t;  // index into s_theta
    s_shifts[t] = sh;
  }
  if (t < a.n_cuts) s_cuts[t] = a.cuts[t];
  __syncthreads();
  BinArgs al = a;
  al.samples = s_samples;
  al.axes = s_axes;
  al.edges = s_edges;
  al.shifts = s_shifts;
  al.params = s_theta;
  al.cuts = s_cuts;
  al.nom_bin = nullptr;
  if (n_samples == 1) al.sample_id = nullptr;
  const uint64_t step = (uint64_t)gridDim.x * 1024u;
  if constexpr (STRIDE == 0) {
    for (uint64_t ev = (uint64_t)blockIdx.x * 1024u + t; ev < a.n_events; ev += step) {
      if (!s_moves[al.sample_id ? al.sample_id[ev] : 0]) continue;
      uint64_t w = 0;
      if (WORD) w = evw[ev];
      const int b = d_event_bin(al, ev);
      if (WORD) evw[ev] = (w & ~a.word_bin_mask) | (uint64_t)(uint32_t)(b + 1);
      else a.bin[ev] = b;
    }
  } else {
    // U events per lane and round, and two rounds in flight: round r + 1's loads go out before round r is looked at.  The pass
    // is a stream of 8 B of kinematics per axis + the 8-byte word per event, and what bounds a stream is the bytes a compute
    // unit keeps in flight.  Round 4's form -- ONE event per lane, the next one requested ahead -- kept 16 B per lane in flight
    // (56 us per 1e7 events = 2.8 TB/s of its own bytes); a round of U without look-ahead leaves the memory idle while the
    // workgroup's waves, all in the same phase, walk the tables.  The wait in front of the conditional stores (a join with a
    // store on one side: the wait-count pass gives up its count) now comes AFTER the round's table walks, when the look-ahead
    // has had their time to arrive.
    constexpr int U = 2;
    const double* __restrict__ kin = a.kin;
    const int32_t* __restrict__ sidp = al.sample_id;
    // every workgroup owns ONE contiguous share of the events, in slabs of 1024: the shares differ by a slab at most (dealt round
    // by round, 1e7 events on 512 workgroups are 3.2 rounds of 6 slabs -- four rounds for some, three for others)
    const uint64_t n = a.n_events, slabs = (n + 1023u) / 1024u, per = (slabs + gridDim.x - 1) / gridDim.x;
    const uint64_t s0 = (uint64_t)blockIdx.x * per, s1 = s0 + per < slabs ? s0 + per : slabs;
    if (s0 >= s1) return;
    const uint64_t lim = s1 * 1024u < n ? s1 * 1024u : n;  // the share's end
    const uint64_t bin_mask = a.word_bin_mask;
    struct Round { double x[U][STRIDE]; uint64_t w[U]; int sid[U]; };
    auto issue = [&](uint64_t sl, Round& r) {  // loads only (past the share's end: a valid event, dropped)
      const uint64_t base = sl * 1024u;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t ev = base + (uint64_t)u * 1024u + t;
        const uint64_t e = ev < lim ? ev : lim - 1;
        r.sid[u] = 0;
        bool moves = true;
        if (sidp) {  // (kernel-uniform) several samples: only what a registered shift touches is read and written
          r.sid[u] = sidp[e];
          moves = s_moves[r.sid[u]] != 0;
          if (!moves) r.sid[u] = -1;
        }
#pragma unroll
        for (int c = 0; c < STRIDE; ++c) r.x[u][c] = moves ? kin[e * STRIDE + c] : 0.0;
        r.w[u] = WORD && moves ? evw[e] : 0;
      }
    };
    auto finish = [&](uint64_t sl, const Round& r) {
      const uint64_t base = sl * 1024u;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint64_t ev = base + (uint64_t)u * 1024u + t;
        if (ev < lim && r.sid[u] >= 0) {
          const int b = d_event_bin_x<STRIDE>(al, r.sid[u], r.x[u], WORD ? (int)(uint32_t)(r.w[u] & bin_mask) - 1 : -1);
          if (WORD) {  // only events that changed bin are written back
            const uint64_t wn = (r.w[u] & ~bin_mask) | (uint64_t)(uint32_t)(b + 1);
            if (wn != r.w[u]) evw[ev] = wn;
          } else {
            a.bin[ev] = b;
          }
        }
      }
    };
    Round ra, rb;
    issue(s0, ra);
    for (uint64_t sl = s0; sl < s1; sl += 2 * U) {
      issue(sl + U, rb);
      finish(sl, ra);
      issue(sl + 2 * U, ra);
      finish(sl + U, rb);
    }
  }
}
// k_rebin_1d: the same pass for the shape real far-detector samples mostly have -- ONE sample binned in ONE variable (kin_stride 1), a
// handful of functional parameters on it.  k_rebin_lds is written for any number of samples, axes, shifts and cuts: per event it reads
// the sample's and the axis' records, every shift record and its parameter from LDS and walks run-time loops over them -- some 200
// instructions per event, and at 1e7 events the pass took what those instructions take (40-52 us), not what its 160 MB do.  Here
// everything that is the same for every event -- the axis, the shifts with 1 + theta formed once, the cuts -- sits in registers for the
// whole launch, and only the edges are in LDS.  FindBin (Samples/SampleStructs.h:577-613) as before -- the event's last bin first (two
// edges), the range test, the +-1 neighbours -- then, for an event that moved further, ONE guess from the axis' mean bin width,
// verified against its two edges, before the binary search: every route returns the bin with e[b] <= x < e[b + 1], so the index is
// the reference's whichever route finds it.  The kinematics are read around the Infinity Cache (once per step, nothing re-reads
// them); the words stay in it for the brick kernel that follows.
struct Rebin1D {
  const double* kin;     // [E] the binning variable
  const double* edges;   // [n_bins + 1]
  const double* params;  // device parameter vector
  uint64_t n_events, bin_mask;
  int32_t n_bins, global_offset, axis_stride, n_shifts, n_cuts;
  int32_t kind[4], param[4];  // shifts in registration order: 1 x *= 1 + theta, 2 x += theta, 3 x += theta * (nominal x)
  double cut_lo[2], cut_hi[2];
};
constexpr int kRebin1DEdges = 2049;
template <int NS>  // the number of shifts: the loop over them is unrolled, nothing of an absent one is computed
__global__ __launch_bounds__(1024, 8) void k_rebin_1d(Rebin1D a, uint64_t* __restrict__ evw) {
  __shared__ double s_e[kRebin1DEdges];
  const int t = threadIdx.x;
  for (int i = t; i <= a.n_bins; i += 1024) s_e[i] = a.edges[i];
  double th[NS], opt[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    th[q] = a.params[a.param[q]];
    opt[q] = 1.0 + th[q];
  }
  __syncthreads();
  const int nb = a.n_bins;
  const double e_first = s_e[0], e_last = s_e[nb];
  const double inv_w = (double)nb / (e_last - e_first);
  constexpr int U = 2;
  const uint64_t n = a.n_events, slabs = (n + 1023u) / 1024u, per = (slabs + gridDim.x - 1) / gridDim.x;
  const uint64_t s0 = (uint64_t)blockIdx.x * per, s1 = s0 + per < slabs ? s0 + per : slabs;
  if (s0 >= s1) return;
  const uint64_t lim = s1 * 1024u < n ? s1 * 1024u : n;
  const uint64_t bin_mask = a.bin_mask;
  struct Round { double x[U]; uint64_t w[U]; };
  auto issue = [&](uint64_t sl, Round& r) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t ev = (sl + (uint64_t)u) * 1024u + t;
      const uint64_t e = ev < lim ? ev : lim - 1;
      r.x[u] = __builtin_nontemporal_load(a.kin + e);
      r.w[u] = evw[e];
    }
  };
  auto finish = [&](uint64_t sl, const Round& r) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t ev = (sl + (uint64_t)u) * 1024u + t;
      const double x0 = r.x[u];
      double x = x0;
#pragma unroll
      for (int q = 0; q < NS; ++q)  // ApplyShifts, in registration order (the kinds are launch-uniform)
        x = a.kind[q] == 1 ? x * opt[q] : (a.kind[q] == 2 ? x + th[q] : x + th[q] * x0);
      bool out = false;
#pragma unroll
      for (int c = 0; c < 2; ++c)
        if (c < a.n_cuts) out = out || (x < a.cut_lo[c]) || (x >= a.cut_hi[c]);
      const int last = (int)(uint32_t)(r.w[u] & bin_mask) - 1 - a.global_offset;  // the event's bin of the step before
      const int hint = last >= 0 && last < nb ? last : -1;
      int b;
      {
        const int h0 = hint >= 0 ? hint : 0;
        const double lower = s_e[h0], upper = s_e[h0 + 1];
        if (hint >= 0 && x < upper && x >= lower) {
          b = hint;
        } else if (!(x >= e_first && x < e_last)) {
          b = -1;
        } else {
          b = -2;
          if (hint >= 0) {  // SampleStructs.h:615-646
            const double lower_lower = hint == 0 ? e_first : s_e[hint - 1];
            const double upper_upper = (hint + 2 < nb) ? s_e[hint + 2] : ((hint + 1 < nb) ? s_e[hint + 1] : -999999.123456);
            if (x < lower && x >= lower_lower) b = hint - 1;
            else if (x < upper_upper && x >= upper) b = hint + 1;
          }
          if (b == -2) {
            int g = (int)((x - e_first) * inv_w);
            g = g < 0 ? 0 : (g >= nb ? nb - 1 : g);
            if (x >= s_e[g] && x < s_e[g + 1]) b = g;
            else b = d_upper_bound_bin(s_e, nb + 1, x);
          }
        }
      }
      const int global = (out || b < 0) ? -1 : b * a.axis_stride + a.global_offset;
      const uint64_t wn = (r.w[u] & ~bin_mask) | (uint64_t)(uint32_t)(global + 1);
      // only events that changed bin are written back -- and, where at least half of a 128-byte line's sixteen words changed, the
      // whole line (the others with the value they had): a line written whole needs no merge with what memory holds
      const bool changed = ev < lim && wn != r.w[u];
      const uint32_t line = (uint32_t)(__ballot(changed) >> (t & 48)) & 0xFFFFu;
      if (changed || (ev < lim && __popc(line) >= 8)) evw[ev] = wn;
    }
  };
  Round ra, rb;
  issue(s0, ra);
  for (uint64_t sl = s0; sl < s1; sl += 2 * U) {
    issue(sl + U, rb);
    finish(sl, ra);
    issue(sl + 2 * U, ra);
    finish(sl + U, rb);
  }
}
void launch_rebin_1d(const BinArgs& b, const BinSample& sm, const BinAxis& ax, const KinShift* shifts, const KinCut* cuts, uint64_t* evw, hipStream_t s) {
  if (b.n_events == 0) return;
  Rebin1D a;
  memset(&a, 0, sizeof(a));
  a.kin = b.kin;
  a.edges = b.edges + ax.edge_off;
  a.params = b.params;
  a.n_events = b.n_events;
  a.bin_mask = b.word_bin_mask;
  a.n_bins = ax.n_bins;
  a.global_offset = sm.global_offset;
  a.axis_stride = ax.stride;
  for (int q = 0; q < b.n_shifts; ++q) {
    if (shifts[q].dim != 0 || shifts[q].sample > 0) continue;  // (a shift on a column or a sample that is not there is skipped, as d_event_bin_x does)
    a.kind[a.n_shifts] = shifts[q].kind & 0xFF;
    a.param[a.n_shifts] = shifts[q].param;
    ++a.n_shifts;
  }
  for (int c = 0; c < b.n_cuts; ++c) {
    if (cuts[c].column != 0 || cuts[c].sample > 0) continue;
    a.cut_lo[a.n_cuts] = cuts[c].lo;
    a.cut_hi[a.n_cuts] = cuts[c].hi;
    ++a.n_cuts;
  }
  const uint64_t chunks = (b.n_events + 1023) / 1024;
  const dim3 g((unsigned)(chunks < 512 ? chunks : 512)), blk(1024);
  if (a.n_shifts <= 1) hipLaunchKernelGGL(k_rebin_1d<1>, g, blk, 0, s, a, evw);  // (no shift on the axis: kind 0 = x + 0 * x)
  else if (a.n_shifts == 2) hipLaunchKernelGGL(k_rebin_1d<2>, g, blk, 0, s, a, evw);
  else if (a.n_shifts == 3) hipLaunchKernelGGL(k_rebin_1d<3>, g, blk, 0, s, a, evw);
  else hipLaunchKernelGGL(k_rebin_1d<4>, g, blk, 0, s, a, evw);
}
bool rebin_fits_1d(const BinSample* samples, int n_samples, const BinAxis* axes, const KinShift* shifts, int n_shifts, const KinCut* cuts, int n_cuts, int kin_stride) {
  if (n_samples != 1 || kin_stride != 1 || !samples[0].uniform || samples[0].n_dim != 1) return false;
  if (axes[samples[0].axis_off].n_bins + 1 > kRebin1DEdges || axes[samples[0].axis_off].n_bins < 1) return false;
  int ns = 0, nc = 0;
  for (int q = 0; q < n_shifts; ++q) ns += (shifts[q].dim == 0 && shifts[q].sample <= 0) ? 1 : 0;
  for (int c = 0; c < n_cuts; ++c) nc += (cuts[c].column == 0 && cuts[c].sample <= 0) ? 1 : 0;
  return ns <= 4 && nc <= 2;
}
bool rebin_tables_fit_lds(int n_samples, int n_axes, int n_edges, int n_shifts, int n_cuts, bool nonuniform) {
  return !nonuniform && n_samples <= RebinLimits::kSamples && n_axes <= RebinLimits::kAxes && n_edges <= RebinLimits::kEdges && n_shifts <= RebinLimits::kShifts &&
         n_cuts <= RebinLimits::kCuts;
}
void launch_rebin_lds(const BinArgs& a, uint64_t* evw, int n_samples, int n_axes, int n_edges, hipStream_t s) {
  if (a.n_events == 0) return;
  const uint64_t chunks = (a.n_events + 1023) / 1024;
  const dim3 g((unsigned)(chunks < 512 ? chunks : 512));  // two workgroups of 1024 lanes per CU; each strides over the events
  const dim3 b(1024);
#define M3_REBIN(WORD, STRIDE) hipLaunchKernelGGL((k_rebin_lds<WORD, STRIDE>), g, b, 0, s, a, evw, n_samples, n_axes, n_edges)
  if (evw) {
    if (a.kin_stride == 1) M3_REBIN(true, 1);
    else if (a.kin_stride == 2) M3_REBIN(true, 2);
    else M3_REBIN(true, 0);
  } else {
    if (a.kin_stride == 1) M3_REBIN(false, 1);
    else if (a.kin_stride == 2) M3_REBIN(false, 2);
    else M3_REBIN(false, 0);
  }
#undef M3_REBIN
}
void launch_rebin_pack(const BinArgs& a, uint64_t* evw, hipStream_t s) {
  if (a.n_events == 0) return;
  hipLaunchKernelGGL(k_rebin_pack, dim3((unsigned)((a.n_events + 255) / 256)), dim3(256), 0, s, a, evw);
}
// k_osc_expand: the events' oscillation weights as a column (1.0 for events without an oscillation pointer).  A table of
// hundreds of KB does not stay in L2 beside the streams, and the per-event gather from it cost 5.5 % of the brick kernel on one
// GPU's share of config 5 (204 000 entries): when the table has not changed for a few steps the engine gathers ONCE.
__global__ __launch_bounds__(256) void k_osc_expand(const int32_t* __restrict__ osc, const float* __restrict__ table, float* __restrict__ out, uint64_t n) {
  const uint64_t ev = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (ev >= n) return;
  const int32_t i = osc[ev];
  out[ev] = i >= 0 ? table[i] : 1.0f;
}
void launch_osc_expand(const int32_t* osc, const float* table, float* out, uint64_t n, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_osc_expand, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, osc, table, out, n);
}
// the bin field of the packed words back into `bin` (m3hip_get_event_bins after steps that re-binned into the words only)
__global__ __launch_bounds__(256) void k_unpack_bins(const uint64_t* evw, int32_t* bin, uint64_t n, uint64_t bin_mask) {
  const uint64_t ev = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (ev < n) bin[ev] = (int32_t)(uint32_t)(evw[ev] & bin_mask) - 1;
}
void launch_unpack_bins(const uint64_t* evw, int32_t* bin, uint64_t n, uint64_t bin_mask, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_unpack_bins, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, evw, bin, n, bin_mask);
}

// ------------------------------------------------------------------------------------
// k_pack_events: an event's bin, oscillation index and its first three norm-parameter indices as one 8-byte word
// (ReweightArgs::evw).  Set-up time: runs when one of the three arrays has changed.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_events(const int32_t* bin, const int32_t* osc, const int16_t* nidx, uint32_t n_norm_slots,
                                                     uint64_t e_pad, int32_t norm_base, uint64_t* out, const float* osc_table) {
  const uint64_t ev = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (ev >= e_pad) return;
  uint32_t codes = 0;
  for (uint32_t m = 0; m < n_norm_slots && m < (uint32_t)M3_EVW_NORM_SLOTS; ++m) {
    const int32_t i = nidx[(uint64_t)m * e_pad + ev];
    codes |= (i >= 0 ? (uint32_t)(i - norm_base + 1) : 0u) << (6 * m);
  }
  if (osc_table) {  // the word carries the oscillation weight itself (ReweightArgs::evw_weight); the bin space is below 2^14
    const int32_t o = osc[ev];
    const float ow = o >= 0 ? osc_table[o] : 1.0f;
    out[ev] = (uint64_t)(uint32_t)(bin[ev] + 1) | ((uint64_t)codes << 14) | ((uint64_t)__float_as_uint(ow) << 32);
    return;
  }
  out[ev] = (uint64_t)(uint32_t)(bin[ev] + 1) | ((uint64_t)(uint32_t)(osc[ev] + 1) << 20) | ((uint64_t)codes << 44);
}
void launch_pack_events(const int32_t* bin, const int32_t* osc, const int16_t* nidx, uint32_t n_norm_slots, uint64_t e_pad, int32_t norm_base,
                        uint64_t* out, const float* osc_table, hipStream_t s) {
  if (e_pad == 0) return;
  hipLaunchKernelGGL(k_pack_events, dim3((unsigned)((e_pad + 255) / 256)), dim3(256), 0, s, bin, osc, nidx, n_norm_slots, e_pad, norm_base, out, osc_table);
}

// ------------------------------------------------------------------------------------
// k_find_bins: a9, BinningHandler::FindGlobalBin / SampleBinningInfo::FindBin per event, on the
// device.  Set-up pass (compute_nominal): FindNominalBin = upper_bound - 1.  Per-step pass: apply
// the registered shifts to the nominal kinematics, then nominal bin -> +-1 neighbour -> binary
// search, exactly the reference's cascade (all routes give the same index; the integer result
// is what the parity tests pin).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_find_bins(BinArgs a) {
  const uint64_t ev = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (ev >= a.n_events) return;
  if (!a.compute_nominal) {  // per-step pass (kept for the paths that do not fuse it into the reweight epilogue)
    a.bin[ev] = d_event_bin(a, ev);
    return;
  }
  // set-up pass: BinningHandler::FindNominalBin per axis (BinningHandler.cpp:294-307) on the nominal kinematics
  const BinSample sm = a.samples[a.sample_id[ev]];
  for (int d = 0; d < sm.n_dim; ++d) {
    const BinAxis ax = a.axes[sm.axis_off + d];
    const double* e = a.edges + ax.edge_off;
    const double x = a.kin[ev * a.kin_stride + d];
    a.nom_bin[ev * a.kin_stride + d] = !(x >= e[0] && x < e[ax.n_bins]) ? -1 : d_upper_bound_bin(e, ax.n_bins + 1, x);  // NaN: out of range
  }
  // the global bin of the nominal event: the per-step cascade with the nominal bins just stored, no shifts
  BinArgs n = a;
  n.n_shifts = 0;
  a.bin[ev] = d_event_bin(n, ev);
}

void launch_find_bins(const BinArgs& a, hipStream_t s) {
  if (a.n_events == 0) return;
  hipLaunchKernelGGL(k_find_bins, dim3((unsigned)((a.n_events + 255) / 256)), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------
// k_binned_weights: BinnedSplineHandler::CalcSplineWeights (Splines/BinnedSplineHandler.cpp:305-341): one lane per non-flat
// bin-spline, weight = max(0, cubic), on the block-sorted segment-major tables described at BinnedArgs.  The workgroups of a
// block run on ONE XCD (chunk = the XCD's share, not the workgroup number): their scattered 4-byte writes into the block's window
// of the weight vector then merge in that XCD's L2 instead of leaving eight L2s with partial lines.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ int binned_chunk(int n_chunks) {
  // workgroup w lands on XCD w % 8: give XCD x the contiguous chunks [x * per, (x + 1) * per)
  const int per = (n_chunks + 7) / 8;
  return (int)(blockIdx.x % 8u) * per + (int)(blockIdx.x / 8u);
}
__global__ __launch_bounds__(256) void k_binned_weights(BinnedArgs a) {
  const int n_chunks = (a.n_unique + 255) / 256;
  const int q = binned_chunk(n_chunks) * 256 + (int)threadIdx.x;
  if (q >= a.n_unique) return;
  const int p = a.par[q];
  const int tgt = a.target[q];
  const PTab pt = a.ptab[p];
  const size_t at = ((size_t)(q / a.block) * a.n_seg + pt.seg) * a.block + (q % a.block);
  const float4 c = a.many[at];
  const float dx = a.pval[p] - a.xk[at];  // :327-329 (float_t = float)
  float weight = fmaf(dx, fmaf(dx, fmaf(dx, c.w, c.z), c.y), c.x);
  if (weight < 0) weight = 0.f;  // :337
  a.weightvec[tgt] = weight;
}

// float_t = double build: double tables, std::fma, and xvar is the parameter's own double value
// (BinnedSplineHandler.cpp:320-337 with M3::float_t = double, M3::fmaf_t = std::fma, Core.h:44-51)
__global__ __launch_bounds__(256) void k_binned_weights_f64(BinnedArgs a) {
  const int n_chunks = (a.n_unique + 255) / 256;
  const int q = binned_chunk(n_chunks) * 256 + (int)threadIdx.x;
  if (q >= a.n_unique) return;
  const int p = a.par[q];
  const int tgt = a.target[q];
  const size_t at = ((size_t)(q / a.block) * a.n_seg + a.ptab[p].seg) * a.block + (q % a.block);
  const double* c = a.many_d + at * 4;
  const double y = c[0], b = c[1], cc = c[2], d = c[3];
  const double xvar = a.params[a.par_map[p]];
  const double dx = xvar - a.xk_d[at];
  double weight = fma(dx, fma(dx, fma(dx, d, cc), b), y);
  if (weight < 0) weight = 0.;
  a.weightvec_d[tgt] = weight;
}

void launch_binned_weights(const BinnedArgs& a, hipStream_t s) {
  if (a.n_unique <= 0) return;
  const unsigned n_wg = (unsigned)(((a.n_unique + 255) / 256 + 7) / 8 * 8);  // a whole number of chunks per XCD (binned_chunk)
  if (a.weightvec_d) hipLaunchKernelGGL(k_binned_weights_f64, dim3(n_wg), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(k_binned_weights, dim3(n_wg), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------
// One-shot exchange ("avoid ring, use ... P2P all-gather + local sum"; the reference has one device only,
// Manager/gpuUtils.cu:71).  The message is 1.6 - 6.4 KB: a ring all-reduce is 2 (N-1) latency-bound hops, this is one.
// k_xch, blocks 0 .. N-1 (push): block t stores this rank's folded vector [mc | w2] into slot [step & 1][rank] of rank t's buffer
// (fine-grained memory mapped through IPC: the stores of N-1 blocks cross N-1 different xGMI links at once), fences at
// system scope and release-stores the step number into the slot's ticket word.
// k_xch, block N (pull): acquire-loads the N tickets of the parity in this rank's OWN buffer until all carry the step number, then
// adds the N vectors in rank order -- bit-identical bins on every rank by construction.
// Two parities suffice: a rank pushes step t+2 only after its pull of t+1, which needed every peer's push of t+1, which
// each peer issued behind its own pull of t on its stream.  Every spin has a time-out (`timeout_ticks` of the 100 MHz clock;
// 2 s unless m3hip_xch_set_timeout says otherwise).  A ticket must EQUAL the step: a slot that already carries a later step
// means this rank fell a whole step behind (host skew beyond the time-out on the peers' side) and the data is gone -- an error,
// not a match.
// ------------------------------------------------------------------------------------
// A slot is [2 * n_bins data | ticket | pad]: the ticket has a fixed place whatever the length of the step's message.
// push and pull as ONE launch: blocks 0 .. N-1 push, block N pulls (its own rank's push is one of the tickets it waits for;
// the grid is at most 65 blocks, all resident at once)
// `pushed`: counts this rank's finished push blocks (monotonic: n_ranks per step); the pull block overwrites the vector the
// push blocks read from, so it also waits for all of them.
__global__ __launch_bounds__(256) void k_xch(const double* src, double* dst, int n_msg, double* const* targets, int n_ranks,
                                             const double* slots, int slot_doubles, uint64_t step, uint64_t* status, unsigned long long* pushed,
                                             uint64_t timeout_ticks) {
  __shared__ int late;  // 1: a ticket did not arrive in time; 2: a slot holds a later step
  if ((int)blockIdx.x < n_ranks) {
    double* t = targets[blockIdx.x];
    for (int i = threadIdx.x; i < n_msg; i += 256) __hip_atomic_store(&t[i], src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_store((uint64_t*)(t + (slot_doubles - 2)), step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_fetch_add(pushed, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);  // this block is through with src
    }
    return;
  }
  if (threadIdx.x == 0) late = 0;
  __syncthreads();
  for (int r = threadIdx.x; r <= n_ranks; r += 256) {  // lanes 0 .. N-1: the peers' tickets; lane N: this rank's own push blocks
    const uint64_t t0 = wall_clock64();
    for (;;) {
      bool there;
      if (r < n_ranks) {
        const uint64_t tk = __hip_atomic_load((const uint64_t*)(slots + (size_t)r * slot_doubles + (slot_doubles - 2)), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (tk > step) {
          late = 2;
          break;
        }
        there = tk == step;
      } else {
        there = __hip_atomic_load(pushed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned long long)n_ranks * step;
      }
      if (there) break;
      if (wall_clock64() - t0 > timeout_ticks) {  // a peer that never pushed must not hang the device
        atomicMax(&late, 1);
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
  if (late) {
    if (threadIdx.x == 0) __hip_atomic_store(status, (step << 1) | (late == 2 ? 1u : 0u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  // src == dst: every push block of this rank has finished reading src (lane N above), so it may be overwritten
  for (int i = threadIdx.x; i < n_msg; i += 256) {
    double acc = 0.0;
    for (int r = 0; r < n_ranks; ++r) acc += __hip_atomic_load(&slots[(size_t)r * slot_doubles + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    dst[i] = acc;
  }
}
void launch_xch(const double* src, double* dst, int n_msg, double* const* targets, int n_ranks, const double* slots, int slot_doubles, uint64_t step,
                uint64_t* status, unsigned long long* pushed, uint64_t timeout_ticks, hipStream_t s) {
  hipLaunchKernelGGL(k_xch, dim3(n_ranks + 1), dim3(256), 0, s, src, dst, n_msg, targets, n_ranks, slots, slot_doubles, step, status, pushed, timeout_ticks);
}

// ------------------------------------------------------------------------------------
// k_llh: kernel (iii).  One workgroup; per-bin test statistic in fp64, wavefront shuffle
// reduction followed by an LDS tree across the four waves.  Fixed summation tree, so the
// result is run-to-run reproducible for a given bin vector.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ double d_poisson_llh(double data, double mc) {
  // Samples/SampleHandlerBase.cpp:17-31
  if (data == 0) return mc;
  if (mc < M3HIP_LOW_MC_BOUND_D) {
    if (data > M3HIP_LOW_MC_BOUND_D) return (M3HIP_LOW_MC_BOUND_D - data + data * log(data / M3HIP_LOW_MC_BOUND_D));
    else if (data >= mc) return 0.;
  }
  return (mc - data + data * log(data / mc));
}

__device__ double d_test_stat_llh(int kind, double data, double mc, double w2) {
  const double low = M3HIP_LOW_MC_BOUND_D;
  switch (kind) {
    case 1: {  // kBarlowBeeston, SampleHandlerBase.cpp:46-87
      double newmc = mc;
      if (mc < low) {
        if (data > low) newmc = low;
        else if (data >= mc) return 0.;
      }
      const double fractional = sqrt(w2) / newmc;
      const double fractional2 = fractional * fractional;
      const double temp = newmc * fractional2 - 1;
      const double temp2 = temp * temp + 4 * data * fractional2;
      if (temp2 < 0) return __builtin_nan("");  // the reference throws here
      const double beta = (-1 * temp + sqrt(temp2)) / 2.;
      double stat = mc * beta;
      if (data > 0) {
        newmc *= beta;
        stat = newmc - data + data * log(data / newmc);
      }
      double penalty = 0;
      if (fractional > 0) penalty = (beta - 1) * (beta - 1) / (2 * fractional2);
      return stat + penalty;
    }
    case 4: {  // kDembinskiAbdelmotteleb, :90-125
      if (w2 == 0) return d_poisson_llh(data, mc);
      double newmc = mc;
      if (mc < low) {
        if (data > low) newmc = low;
        else if (data >= mc) return 0.;
      }
      const double k = newmc * newmc / w2;
      const double beta = (data + k) / (newmc + k);
      newmc *= beta;
      const double penalty = k * beta - k + k * log(k / (k * beta));
      double stat = newmc;
      if (data > 0) stat = newmc - data + data * log(data / newmc);
      return stat + penalty;
    }
    case 2: {  // kIceCube, :133-159.  The reference evaluates this in x87 long double with lgammal; the device has no 80-bit
               // type, so the expression is rearranged to keep fp64 well-conditioned (round 3; agreement with the long double
               // evaluation 2e-11 relative over mc = 1e-2 .. 3e4, w2 = 1e-4 mc .. 3 mc, against 4e-8 for the literal form):
               //   -(a ln b + lnG(d+a) - lnG(d+1) - (d+a) ln(1+b) - lnG(a))
               //     = a log1p(1/b) + d log1p(b) - [lnG(a+d) - lnG(a)] + lnG(d+1),
               // and the bracket -- two huge, nearly equal numbers when a = mc^2/w2 + 1 is large -- from Stirling's series
               // with the leading terms combined through log1p: (a - 1/2) log1p(d/a) + d ln(a+d) - d + [S(a+d) - S(a)],
               // S(z) = 1/(12 z) - 1/(360 z^3) + 1/(1260 z^5) - 1/(1680 z^7)   (a >= 16; below that lgamma itself is harmless)
      // Single-exit form on purpose: with early returns hipcc (ROCm 7.2, gfx950, -O1..-O3)
      // returned the Poisson cap even when stat < cap; tests/test_gpu_parity.py pins this.
      double r;
      if (w2 == 0) {
        r = d_poisson_llh(data, mc);
      } else {
        const double b = mc / w2;
        const double a = mc * b + 1;
        double diff;
        if (a >= 16.0) {
          const double x = a + data;
          auto tail = [](double z) {
            const double z2 = 1.0 / (z * z);
            return (1.0 / z) * (1.0 / 12.0 - z2 * (1.0 / 360.0 - z2 * (1.0 / 1260.0 - z2 * (1.0 / 1680.0))));
          };
          diff = (a - 0.5) * log1p(data / a) + data * log(x) - data + (tail(x) - tail(a));
        } else {
          diff = lgamma(data + a) - lgamma(a);
        }
        const double stat = a * log1p(1.0 / b) + data * log1p(b) - diff + lgamma(data + 1);
        r = stat;
        if (mc <= data) {
          if (data <= low) {
            r = 0.;
          } else {
            const double poisson = d_poisson_llh(data, low);
            if (stat > poisson) r = poisson;
          }
        }
      }
      return r;
    }
    case 3: {  // kPearson, :162-176
      if (data == 0) return mc / 2.;
      if (mc < low) {
        if (data > low) return (data - low) * (data - low) / (2. * low);
        else if (data >= mc) return 0.;
      }
      return (data - mc) * (data - mc) / (2 * mc);
    }
    case 0:  // kPoisson, :178-183
      return d_poisson_llh(data, mc);
    default:
      return __builtin_nan("");
  }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// fold the flush replicas into replica 0 (same fixed order as k_llh), so that the multi-GPU
// all-reduce moves n_bins doubles instead of hist_rep * n_bins.  park_w2 (hist_rep > 1): w2's sum goes right behind
// mc's -- into mc's second replica, which this fold has just emptied -- so that ONE all-reduce moves [mc | w2].
__global__ __launch_bounds__(256) void k_fold_hist(double* hist, int hist_rep, int n_bins, int with_w2, int park_w2) {
  double sums[8];  // a slice of 8 x 256 values is folded into registers before any of its replicas is rewritten
  const int n = n_bins * (with_w2 ? 2 : 1);
  for (int i0 = 0; i0 < n; i0 += 256 * 8) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = i0 + q * 256 + (int)threadIdx.x;
      double m = 0.0;
      if (i < n) {
        const double* base = hist + (i < n_bins ? i : (size_t)hist_rep * n_bins + (i - n_bins));
        for (int r0 = 0; r0 < hist_rep; r0 += 8) {
          double am[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) am[u] = (r0 + u < hist_rep) ? base[(size_t)(r0 + u) * n_bins] : 0.0;
#pragma unroll
          for (int u = 0; u < 8; ++u) m += am[u];
        }
      }
      sums[q] = m;
    }
    __syncthreads();  // every sum of this slice is taken before any replica is rewritten (w2's parking place is an mc replica)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = i0 + q * 256 + (int)threadIdx.x;
      if (i < n) {
        double* base = hist + (i < n_bins ? i : (size_t)hist_rep * n_bins + (i - n_bins));
        for (int r = 0; r < hist_rep; ++r) base[(size_t)r * n_bins] = 0.0;
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = i0 + q * 256 + (int)threadIdx.x;
      if (i < n) {
        if (i < n_bins) hist[i] = sums[q];
        else if (park_w2) hist[(size_t)n_bins + (i - n_bins)] = sums[q];
        else hist[(size_t)hist_rep * n_bins + (i - n_bins)] = sums[q];
      }
    }
    __syncthreads();
  }
}

void launch_fold_hist(double* hist, int hist_rep, int n_bins, int with_w2, int park_w2, hipStream_t s) {
  hipLaunchKernelGGL(k_fold_hist, dim3(1), dim3(256), 0, s, hist, hist_rep, n_bins, with_w2, park_w2);
}

// after the all-reduce of [mc | w2]: w2 back from its parking place (mc's second replica) to its own first replica
__global__ __launch_bounds__(256) void k_unpark_w2(double* hist, int hist_rep, int n_bins) {
  for (int i = threadIdx.x; i < n_bins; i += 256) {
    hist[(size_t)hist_rep * n_bins + i] = hist[(size_t)n_bins + i];
    hist[(size_t)n_bins + i] = 0.0;
  }
}

void launch_unpark_w2(double* hist, int hist_rep, int n_bins, hipStream_t s) {
  hipLaunchKernelGGL(k_unpark_w2, dim3(1), dim3(256), 0, s, hist, hist_rep, n_bins);
}

// Deterministic mode: the bin vector as an ordered sum of the workgroups' rows.  Level 1: every 32 (kFoldGroup) consecutive rows into one
// (thread = bin, rows added in index order); level 2: one workgroup adds the level-1 rows in order into replica 0.  No
// atomics anywhere: the result depends on nothing but the rows.
constexpr int kFoldGroup = 32;
// m += rows[r][i] for r in [r0, r1), in index order; the loads of a batch are in flight together, the adds stay sequential
__device__ __forceinline__ double fold_rows(const double* rows, int r0, int r1, int row_len, int i) {
  double m = 0.0;
  for (int r = r0; r < r1; r += kFoldGroup) {
    double v[kFoldGroup];
#pragma unroll
    for (int u = 0; u < kFoldGroup; ++u) v[u] = r + u < r1 ? rows[(size_t)(r + u) * row_len + i] : 0.0;
#pragma unroll
    for (int u = 0; u < kFoldGroup; ++u) m += v[u];  // + 0.0 past the end changes nothing (m is never -0.0 here: it starts at +0.0)
  }
  return m;
}
__global__ __launch_bounds__(256) void k_fold_partials(const double* rows, int n_rows, int row_len, int n_used, double* out, int out_len) {
  const int g = blockIdx.y, r0 = g * kFoldGroup, r1 = min(n_rows, r0 + kFoldGroup);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n_used) out[(size_t)g * out_len + i] = fold_rows(rows, r0, r1, row_len, i);
}
__global__ __launch_bounds__(256) void k_fold_final(const double* rows, int n_rows, int row_len, double* hist, int hist_rep, int n_bins, int with_w2) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n_bins * (with_w2 ? 2 : 1))
    hist[i < n_bins ? i : (size_t)hist_rep * n_bins + (i - n_bins)] = fold_rows(rows, 0, n_rows, row_len, i);  // replica 0 (k_prep has emptied the others)
}
void launch_fold_partials(const double* partial, int n_rows, int n_bins, int with_w2, double* scratch, double* hist, int hist_rep, hipStream_t s) {
  const int row_len = 2 * n_bins, n_used = n_bins * (with_w2 ? 2 : 1), n_groups = (n_rows + kFoldGroup - 1) / kFoldGroup;
  const int bx = (n_used + 255) / 256;
  hipLaunchKernelGGL(k_fold_partials, dim3(bx, n_groups), dim3(256), 0, s, partial, n_rows, row_len, n_used, scratch, row_len);
  hipLaunchKernelGGL(k_fold_final, dim3(bx), dim3(256), 0, s, scratch, n_groups, row_len, hist, hist_rep, n_bins, with_w2);
}

// sum of one bin over the flush replicas, in replica order (fixed => reproducible), 32 loads in flight
__device__ __forceinline__ double fold_replicas(const double* base, int hist_rep, int n_bins) {
  double m = 0.0;
  for (int r0 = 0; r0 < hist_rep; r0 += 32) {
    double am[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) am[u] = (r0 + u < hist_rep) ? base[(size_t)(r0 + u) * n_bins] : 0.0;
#pragma unroll
    for (int u = 0; u < 32; ++u) m += am[u];
  }
  return m;
}

// GetTestStatLLH of bin i on the bin vectors as the reweight kernels left them (read from the flush replicas;
// W2 folded in place once, at its freeze)
__device__ __forceinline__ double llh_term(const LlhArgs& a, int i) {
  const double* mc = a.hist;
  double* w2 = const_cast<double*>(a.hist) + (size_t)a.hist_rep * a.n_bins;
  const double m = fold_replicas(mc + i, a.hist_rep, a.n_bins);
  double v;
  if (a.w2_mode == kW2Replica0) {
    v = w2[i];
  } else {
    v = fold_replicas(w2 + i, a.hist_rep, a.n_bins);
    if (a.w2_mode == kW2FoldInPlace) {  // W2 was frozen by this reweight: keep it as one plain vector
      w2[i] = v;
      for (int r = 1; r < a.hist_rep; ++r) w2[(size_t)r * a.n_bins + i] = 0.0;
    }
  }
  return d_test_stat_llh(a.kind, a.data[i], m, v);
}

// GetLikelihood() = sum over samples of GetSampleLikelihood(s) = sum over the sample's bins of GetTestStatLLH
// (SampleHandlerFD.cpp:1262-1300), from the per-bin terms: the waves take the samples round-robin, the lanes of a
// wave stride over the sample's bins (8 loads in flight), wave tree, then the samples in order -- a fixed order
// throughout, so the result is reproducible; its cost does not grow with the number of samples beyond their bins
// (a workgroup-wide reduction per sample cost 2.6 us each: 78 us at 30 samples).  Returns the total in thread 0.
template <int NT>
__device__ double sum_by_sample(const LlhArgs& a, const double* terms, double* sums /*[NT], LDS*/) {
  constexpr int NW = NT / 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double total = 0.0;
  for (int s0 = 0; s0 < a.n_samples; s0 += NT) {  // rounds of NT samples: in practice one
    const int s1 = s0 + NT < a.n_samples ? s0 + NT : a.n_samples;
    for (int s = s0 + wave; s < s1; s += NW) {
      const int b1 = a.sample_off[s + 1];
      int i = a.sample_off[s] + lane;
      double acc = 0.0;
      for (; i + 7 * 64 < b1; i += 8 * 64) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = terms[i + u * 64];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += t[u];
      }
      for (; i < b1; i += 64) acc += terms[i];
      acc = wave_sum(acc);
      if (lane == 0) {
        sums[s - s0] = acc;
        a.out[1 + s] = acc;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0)
      for (int k = 0; k < s1 - s0; ++k) total += sums[k];
    __syncthreads();
  }
  return total;
}

// results first, then the ticket the host is spinning on (system scope: host memory)
__device__ __forceinline__ void publish_llh(const LlhArgs& a, double total) {
  a.out[0] = total;
  if (a.dbg_clock) a.dbg_clock[1] = wall_clock64();
  if (a.done_flag) {
    __threadfence_system();
    *a.done_flag = a.ticket;
  }
}

// Up to M3_LLH_SPLIT_BINS bins: one workgroup, the terms pass through LDS.  NT lanes: 256 for the usual few hundred
// bins, 1024 beyond 1 024 (every pass over the bins is a chain of dependent loads: fewer passes is what counts).
// Nothing is written back to the bin vectors, so the call can be repeated.
template <int NT>
__global__ __launch_bounds__(NT) void k_llh(LlhArgs a) {
  __shared__ double terms[M3_LLH_SPLIT_BINS];
  __shared__ double sums[NT];
  for (int i = threadIdx.x; i < a.n_bins; i += NT) terms[i] = llh_term(a, i);
  __syncthreads();
  const double total = sum_by_sample<NT>(a, terms, sums);
  if (threadIdx.x == 0) publish_llh(a, total);
}

// Beyond that (2-D samples with thousands of bins): one lane per bin across as many workgroups as it takes, the
// terms pass through global memory, and a second small kernel sums them.
__global__ __launch_bounds__(M3_LLH_SPLIT_LANES) void k_llh_terms(LlhArgs a) {
  const int i = (int)blockIdx.x * M3_LLH_SPLIT_LANES + (int)threadIdx.x;
  if (i < a.n_bins) a.terms[i] = llh_term(a, i);
}
__global__ __launch_bounds__(M3_LLH_SPLIT_LANES) void k_llh_sum(LlhArgs a) {
  __shared__ double sums[M3_LLH_SPLIT_LANES];
  const double total = sum_by_sample<M3_LLH_SPLIT_LANES>(a, a.terms, sums);
  if (threadIdx.x == 0) publish_llh(a, total);
}

// oscillation-weight table in the other precision of M3::float_t (widening is exact; narrowing is what a
// float build would have stored)
__global__ __launch_bounds__(256) void k_osc_widen(const float* src, double* dst, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) dst[i] = (double)src[i];
}
__global__ __launch_bounds__(256) void k_osc_narrow(const double* src, float* dst, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) dst[i] = (float)src[i];
}
void launch_osc_convert(const float* f32, const double* f64, void* dst, uint64_t n, bool widen, hipStream_t s) {
  const unsigned g = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  if (widen) hipLaunchKernelGGL(k_osc_widen, dim3(g), dim3(256), 0, s, f32, reinterpret_cast<double*>(dst), n);
  else hipLaunchKernelGGL(k_osc_narrow, dim3(g), dim3(256), 0, s, f64, reinterpret_cast<float*>(dst), n);
}

// ------------------------------------------------------------------------------------
// The parameter handler's two O(nPar^2) sweeps of a Metropolis step, on the device (BASELINE configs[4] has
// 1 040 parameters: 2 x 8.65 MB of fp64 matrix per step, 0.6 ms on one host core -- longer than the GPU's step):
//   rows 0 .. n-1   the prior, Parameters/ParameterHandlerBase.cpp:816-841: sum over i, j <= i of (i == j ? 0.5 : 1) d_i d_j V^-1_ij
//                   (d = proposed - prior value, 0 where the prior is flat); row i contributes d_i (sum_{j<i} V^-1_ij d_j + 0.5 V^-1_ii d_i)
//   rows n .. 2n-1  the correlated throw, :726-745 with Parameters/ParameterHandlerUtils.h:118-135: corr_throw = L z, L the lower
//                   Cholesky factor of the throw matrix
// One wave per row, lanes stride over the row's lower-triangle part (512-byte coalesced reads), a fixed shuffle tree; the
// finishing kernel adds the prior's rows in index order groups and hands prior + throw to mapped host memory, then the ticket.
// Every sum has a fixed order: results are reproducible run to run (they differ from the host loops' by fp64 rounding only).
// ------------------------------------------------------------------------------------
// (the step's [d | z] arrives in mapped host memory; one block brings it on board first -- 2 080 waves reading it across PCIe
// would each pay that round trip)
__global__ __launch_bounds__(1024) void k_par_stage(const double* __restrict__ in_host, int n2, double* __restrict__ in_dev) {
  for (int i = threadIdx.x; i < n2; i += 1024) in_dev[i] = in_host[i];
}
__global__ __launch_bounds__(256) void k_par_rows(const double* __restrict__ inv, const double* __restrict__ chol, const double* __restrict__ in, int n,
                                                  double* __restrict__ rows) {
  const int r = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
  if (r >= 2 * n) return;
  const bool prior = r < n;
  const int i = prior ? r : r - n;
  const double* __restrict__ row = (prior ? inv : chol) + (size_t)i * n;
  const double* __restrict__ v = in + (prior ? 0 : n);  // d, then z
  const int last = prior ? i - 1 : i;                    // the prior's diagonal term carries its own factor
  // eight loads of the row in flight per lane (a row of 1 040 is 17 per lane: two round trips, not seventeen), summed in a
  // fixed order
  double s = 0.0;
  for (int j0 = lane; j0 <= last; j0 += 8 * 64) {
    double a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + 64 * u;
      const int jc = j <= last ? j : last;
      a[u] = row[jc];
      b[u] = j <= last ? v[jc] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += a[u] * b[u];
  }
  s = wave_sum(s);
  if (lane == 0) rows[r] = prior ? v[i] * (s + 0.5 * v[i] * row[i]) : s;
}
__global__ __launch_bounds__(1024) void k_par_finish(const double* __restrict__ rows, int n, double* __restrict__ out, volatile uint64_t* done_flag, uint64_t ticket) {
  __shared__ double part[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) s += rows[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  for (int i = threadIdx.x; i < n; i += 1024) out[1 + i] = rows[n + i];
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += part[w];
    out[0] = t;
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) *done_flag = ticket;
}
void launch_par_step(const double* inv, const double* chol, const double* in_host, double* in_dev, int n, double* rows, double* out,
                     volatile uint64_t* done_flag, uint64_t ticket, hipStream_t s) {
  hipLaunchKernelGGL(k_par_stage, dim3(1), dim3(1024), 0, s, in_host, 2 * n, in_dev);
  hipLaunchKernelGGL(k_par_rows, dim3((unsigned)((2 * n + 3) / 4)), dim3(256), 0, s, inv, chol, in_dev, n, rows);
  hipLaunchKernelGGL(k_par_finish, dim3(1), dim3(1024), 0, s, rows, n, out, done_flag, ticket);
}

void launch_llh(const LlhArgs& a, hipStream_t s) {
  if (a.n_bins > M3_LLH_SPLIT_BINS) {
    const int n_groups = (a.n_bins + M3_LLH_SPLIT_LANES - 1) / M3_LLH_SPLIT_LANES;
    hipLaunchKernelGGL(k_llh_terms, dim3(n_groups), dim3(M3_LLH_SPLIT_LANES), 0, s, a);
    hipLaunchKernelGGL(k_llh_sum, dim3(1), dim3(M3_LLH_SPLIT_LANES), 0, s, a);
  } else if (a.n_bins > 1024) {
    hipLaunchKernelGGL(k_llh<1024>, dim3(1), dim3(1024), 0, s, a);
  } else {
    hipLaunchKernelGGL(k_llh<256>, dim3(1), dim3(256), 0, s, a);
  }
}

}  // namespace m3hip
