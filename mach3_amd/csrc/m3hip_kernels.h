// m3hip_kernels.h -- device-side data contracts and kernel launchers (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "m3hip_layout.h"

#define M3HIP_LOW_MC_BOUND_D 0.00001 /* Manager/Core.h:83 */

namespace m3hip {

// Per-step, per-parameter table: everything kernel (i) needs that depends on the
// parameter only.  dx = ParamValues[p] - coeff_x[p*K + seg] is the same float subtraction
// the reference performs per spline (Splines/SplineMonolith.cpp:759), hoisted.
struct PTab {
  float dx;
  int32_t seg;
};

// Device-side descriptor of one uniform tile: absolute pointers into the chunk stores.
// Per-step resolved slot: what kernel (i) needs for one slot of a slot list, with the
// parameter indirection already applied (k_prep / k_resolve write these every step).
//   TSpline3 slot: plane = plane0[j] + seg[param], f = dx[param]
//   TF1 slot:      plane unused,                  f = ParamValues[param]
struct RSlot {
  int32_t plane;
  float f;
};

struct TileDev {
  const float4* spl;   // [(plane0[j] + k) * M3_TE + lane]
  const float2* tf1;   // [j * M3_TE + lane]
  const RSlot* res;    // resolved slot list of this tile: [J_s] spline slots, then [J_t] TF1 slots.
                       // Tiles with identical ordered parameter lists share one list, so in
                       // structured samples it stays resident in the scalar cache / L2.
  uint32_t n_spl_slots;
  uint32_t n_tf1_slots;
  uint32_t n_events;
  uint32_t first_event;  // padded global index of lane 0
};

// Device-side descriptor of one brick (ragged group of <= G x 1024 events), see m3hip_layout.h.  Read with scalar loads.
struct BrickDev {
  const float4* spl;
  const float2* tf1;
  const uint32_t* words;      // metadata stream
  const RSlot* tf1_x;         // the chunk's TF1 parameter dictionary, resolved per step: tf1_x[k].f = value of entry k
  uint32_t n_events;
  uint32_t first_event;
  uint32_t n_stages;          // TSpline3 stages
  uint32_t off_tf1;           // word offset of the TF1 section
  uint32_t st0[4];            // copies of the headers of stages 0 and 1: their metadata is requested without waiting for
  uint32_t st1[4];            // the header block
  uint32_t n_dict;
  uint32_t spl_last_off;      // byte offset of the spline store's last quad (idle lanes' offsets are clamped to it)
  uint32_t pad_[2];           // [0] G of the brick's format, [1] bit 0: the TF1 lane-slots are sorted by code count (sigma in use), bit 1: the slots' code counts are bits 10..15 of their sigma words
};

// ---- device-side binning (uniform N-D), ref: Samples/SampleStructs.h:577-613, BinningHandler.cpp:257-307
constexpr int M3_MAX_DIM = 4;
struct BinAxis {       // one axis of one sample
  int32_t edge_off;    // first edge in the concatenated edge array
  int32_t n_bins;
  int32_t stride;      // SampleBinningInfo::Strides
  int32_t pad_;
};
struct BinSample {
  int32_t n_dim;
  int32_t axis_off;       // first BinAxis of the sample (non-uniform: the axes of the mega-bin grid)
  int32_t global_offset;  // SampleBinningInfo::GlobalOffset
  int32_t uniform;        // SampleBinningInfo::Uniform
  int32_t map_off;        // non-uniform: first entry of this sample's BinGridMapping offsets (CSR, n_cells + 1)
  int32_t box_off;        // non-uniform: first box of this sample in `boxes` ([box][dim][2] doubles)
};
struct KinShift {      // closed set of functional-parameter shifts (stand-ins for the reference's
  int32_t sample;      // std::function callbacks, SampleHandlerFD.cpp:545-564); -1 = every sample
  int32_t dim;
  int32_t kind;        // low byte 1: x *= (1 + theta)   2: x += theta   3: x += theta * kin[aux], aux = kind >> 8
  int32_t param;       // index into the global parameter vector
};
struct KinCut {        // SampleHandlerFD::IsEventSelected (SampleHandlerFD.cpp:281-294): keep lo <= x < hi
  int32_t sample;      // -1 = every sample
  int32_t column;      // column of the kinematics array (binning axes first, extra cut variables after)
  double lo, hi;
};
struct BinArgs {
  const BinSample* samples;
  const BinAxis* axes;
  const double* edges;
  const int32_t* grid_off;   // BinGridMapping as CSR: candidates of mega-bin c = grid_list[grid_off[c] .. grid_off[c+1])
  const int32_t* grid_list;  // box numbers, ascending within a cell (the reference's scan order)
  const double* boxes;       // BinInfo::Extent, [box][dim][2]
  const int32_t* sample_id;  // [E_pad]
  const double* kin;         // [E_pad][kin_stride] nominal kinematic variables
  int32_t* nom_bin;          // [E_pad][kin_stride] nominal bin per axis (FindNominalBin)
  int32_t* bin;              // [E_pad] out: global bin
  const KinShift* shifts;
  int32_t n_shifts;
  const KinCut* cuts;
  int32_t n_cuts;
  const double* params;      // device parameter vector (needed when n_shifts > 0)
  int32_t kin_stride;
  int32_t compute_nominal;   // 1: set-up pass, fill nom_bin with upper_bound; 0: per-step pass using nom_bin
  uint64_t n_events;
  uint64_t word_bin_mask;    // re-binning into the packed words: the bits of a word that hold bin + 1 (0xFFFFF, or 0x3FFF when the words
                             // carry the oscillation weight: ReweightArgs::evw_weight)
};
void launch_find_bins(const BinArgs& a, hipStream_t s);
// per-step re-binning straight into the bin field of the events' packed sample words (ReweightArgs::evw)
void launch_rebin_pack(const BinArgs& a, uint64_t* evw, hipStream_t s);
// ... with the binning tables in LDS (uniform samples whose tables fit: rebin_tables_fit_lds); evw null: into a.bin instead
bool rebin_tables_fit_lds(int n_samples, int n_axes, int n_edges, int n_shifts, int n_cuts, bool nonuniform);
void launch_rebin_lds(const BinArgs& a, uint64_t* evw, int n_samples, int n_axes, int n_edges, hipStream_t s);
// ... one sample binned in one variable (rebin_fits_1d): everything but the edges in registers (k_rebin_1d)
bool rebin_fits_1d(const BinSample* samples, int n_samples, const BinAxis* axes, const KinShift* shifts, int n_shifts, const KinCut* cuts, int n_cuts, int kin_stride);
void launch_rebin_1d(const BinArgs& b, const BinSample& sm, const BinAxis& ax, const KinShift* shifts, const KinCut* cuts, uint64_t* evw, hipStream_t s);
void launch_osc_expand(const int32_t* osc, const float* table, float* out, uint64_t n, hipStream_t s);
void launch_unpack_bins(const uint64_t* evw, int32_t* bin, uint64_t n, uint64_t bin_mask, hipStream_t s);

// One experiment weight beyond oscillation and splines (SampleHandlerFD::AddAdditionalWeightPointers appends such pointers to
// total_weight_pointers, Samples/SampleHandlerFD.cpp:196; CalcWeightFunc refreshes what they point at, SampleHandlerFD.h:289):
// either a per-event value (col) or an index per event into a table that may change every step (idx, tab; -1 = no pointer).
struct ExtraSlot {
  const float* col;
  const int32_t* idx;
  const float* tab;
  uint64_t pad_;
};
constexpr int M3_MAX_EXTRA = 8;

// norm-parameter indices in the packed sample word (ReweightArgs::evw): an event's first three as 6-bit codes; events with
// more read the rest from the index planes in the kernel's tail
#define M3_EVW_NORM_SLOTS 3

struct ReweightArgs {
  const TileDev* tiles;
  const BrickDev* bricks;
  const PTab* ptab;          // [P]
  const float* pval;         // [P] ParamValues
  int32_t n_params;          // P
  float* spline_w;           // [E_pad] monolith total weight per event (may be null)
  float* event_w;            // [E_pad] CalcWeightTotal per event (may be null)
  // fused fill
  const int32_t* bin;        // [E_pad]
  const int32_t* osc;        // [E_pad]
  const int16_t* nidx;       // [M][E_pad] ELL of global parameter indices, padded with n_par (-> 1.0)
  // The three arrays above as ONE 8-byte word per event (brick kernel; null when a sample does not fit the fields):
  // bin + 1 in bits 0..19, osc + 1 in bits 20..43, up to three norm indices in bits 44..61 as 6-bit codes idx - evw_norm_base + 1
  // (0: none).  14 B -> 8 B per event; rebuilt on the device (k_pack_events) whenever one of the arrays changes.
  // evw_weight = 1 (round 5): the word carries the event's oscillation WEIGHT instead of its index -- bin + 1 in bits 0..13, the
  // three norm codes in bits 14..31, the weight's float bits in the high dword (1.0f for an event without an oscillation pointer:
  // multiplying by it leaves the product's bits alone).  What the engine packs once a pre-staged table has stopped changing
  // (m3hip_step.hip): no gather from the table per event, and no separate 4-byte column beside the word for a table too large
  // to stay in L2 (BASELINE configs[4]: 204 000 entries).  Needs a bin space below 2^14.
  const uint64_t* evw;
  uint32_t evw_weight;
  // brick kernel: 1 = the TF1 codes are loaded non-temporally, around the Infinity Cache -- the engine's choice when the small
  // streams a step re-reads do not all fit its budget there (m3hip_step.hip: choose_nt_codes); the large streams always are
  uint32_t nt_codes;
  int32_t evw_norm_base;
  uint32_t evw_norm_slots;  // norm slots held in the word (M3_EVW_NORM_SLOTS or 0); an event's further slots are read from nidx
  const double* params;      // [n_par + 1], last entry 1.0
  int32_t n_par_total;       // n_par of this step's parameter vector (the brick kernel bounds its window of norm values by it)
  const float* params_f32;   // [n_par + 1] float(params): what the window is read from
  const int32_t* bwidx;      // [n_bw_slots][E_pad] ELL of indices into the binned-spline weights, -1 = empty
  // ... or, when neighbouring events share their lists (MC ordered by spline bin), one list id per event into a
  // store of the distinct lists: the index traffic drops from 4 B per weight to 4 B per event
  const uint32_t* perm;      // [E_pad] or null: lane `pos` of a tile works on event perm[pos] (binned-only event
                             // spaces: events that share a pointer list are processed next to each other whatever
                             // order they were registered in; every per-event array stays in the caller's order)
  const int32_t* bl_id;      // [E_pad] list of the event, -1 = none; null = the ELL form above is in use
  const int32_t* bl_off;     // [n_lists + 1]
  const int32_t* bl_idx;     // concatenated lists, each in pointer order
  const float* bw;           // weightvec_Monolith of the binned spline handler
  uint32_t n_bw_slots;
  const float* osc_table;
  const float* oscw;         // [E_pad] or null: the events' oscillation weights themselves (brick kernel; the engine expands a large
                             // table that has stopped changing: k_osc_expand), 1.0 where an event has no oscillation pointer
  const ExtraSlot* extras;   // [n_extra] device array, multiplied after the spline weight in slot order
  uint32_t n_extra;
  // M3::float_t = double build of the sample side (Manager/Core.h:27-51: the default, non-low-memory
  // build): bin-spline weights, oscillation weights and the event weight product are double
  const double* bw_d;
  const double* osc_table_d;
  double* event_w_d;         // [E_pad] (may be null)
  double* hist;              // [2][hist_rep][n_bins]: mc replicas, then w2 replicas
  int32_t hist_rep;          // replicas the workgroups spread their flushes over (contention)
  double* partial;           // deterministic mode: [tiles + bricks][2 * n_bins], one row per workgroup (null: atomics)
  uint32_t partial_row0;     // brick launches: row of the launch's first brick (the bricks' rows follow the tiles')
  uint64_t e_pad;
  uint32_t n_norm_slots;     // M
  int32_t n_bins;
  int32_t fill_w2;
  int32_t hist_lds;          // 1: workgroup-private bin vector in LDS, flushed once (few bins); 0: every event adds
                             // straight into its workgroup's global replica (thousands of bins: no contention to hide,
                             // and the LDS would cost occupancy)
  uint32_t tile_base;        // index of the first tile handled by this launch
  uint32_t brick_groups;     // G of the engine's bricks (1, 2 or 4): events per brick = G x 1024
  BinArgs dyn;               // DYN variants of the uniform-tile kernel only: the event's bin is recomputed from its shifted
                             // kinematics in the epilogue (functional parameters), instead of being read from `bin`
};

enum ReweightMode : int { kModeEval = 1, kModeFill = 2, kModeEventW = 4 };

struct PrepArgs {
  // FindSplineSegment inputs
  const double* params_src;   // pinned host (device-mapped) or device: proposed values [n_par]
  double* params_dev;         // [n_par + 1] device copy, [n_par] = 1.0
  float* params_f32;          // [n_par + 1] or null: float(value) of the same (the brick kernel's window of norm parameters)
  int32_t n_par;
  const int32_t* par_map;     // [P] spline parameter -> global index
  const float* coeff_x;       // [P*K]
  const double* coeff_x_d;    // [P*K] or null: knots as double (float_t = double build: xPts are double and the
                              // float parameter value is compared against them, SplineBase.cpp:51-54)
  const int16_t* npts;        // [P]
  int16_t* curr_seg;          // [P] FastSplineInfo.CurrSegment (state)
  int32_t n_params, max_knots;
  // outputs
  PTab* ptab;
  float* pval;
  int16_t* seg_out;
  // host-supplied vals/segments path (RunGPU_SplineMonolith inputs); params_src unused then
  const float* vals_in;
  const int16_t* seg_in;
  // slot-list resolution (all unique lists, flattened): see RSlot
  const int32_t* u_param;   // [n_res] parameter of the slot
  const int32_t* u_plane0;  // [n_res] first plane of the slot, -1 for a TF1 slot
  RSlot* resolved;          // [n_res]
  int32_t n_res;            // 0: resolution is done by k_resolve instead
  uint64_t* dbg_clock;      // optional: [0] <- wall_clock64() at k_prep start (diagnostics)
  // histogram reset (ResetHistograms)
  double* hist;
  int32_t hist_rep;
  int32_t n_bins;
  int32_t reset_w2;
  int32_t reset_hist;
};

// ---- binned splines (ref: Splines/BinnedSplineHandler.cpp:295-341)
// BinnedSplineHandler::CalcSplineWeights on the engine's own layout of the handler's tables (m3hip_upload.hip: binned_planes).
// The reference keeps one AoS record {y,b,c,d}[knot] per bin-spline and a step reads ONE quad of it -- a 128-byte line for 16 bytes;
// read as it is (rounds 2-4) the pass moved 121 B per non-flat bin-spline.  Here the non-flat bin-splines are cut into blocks of
// `block` (a multiple of 256), sorted by parameter inside a block, and stored segment-major per block:
//     many[(b * n_seg + k) * block + j]    quad of segment k of the j-th bin-spline of block b (float4, or four doubles)
//     xk  [(b * n_seg + k) * block + j]    its knot x
// so the lanes of a wave -- a run of one parameter, i.e. of one segment -- read 1 KB of contiguous quads, and a bin-spline costs
// 2 (parameter) + 4 (target) + 16 + 4 B read and its weight written: 30 B (54 B in the double build).  The weight vector keeps
// the handler's order (the events' pointer lists index it): a block's targets lie in a window of a few KB that L2 merges.
struct BinnedArgs {
  const int16_t* par;        // [n_unique] unique spline parameter of the bin-spline at sorted position q
  const int32_t* target;     // [n_unique] its index in weightvec_Monolith
  const float4* many;        // float build: segment-major quads
  const float* xk;
  const double* many_d;      // double build: [..][4]
  const double* xk_d;
  const PTab* ptab;
  const float* pval;
  float* weightvec;          // [M]
  double* weightvec_d;
  const double* params;      // double build: xvar is the parameter's own double value (BinnedSplineHandler.cpp:327),
  const int32_t* par_map;    // read through par_map from the device parameter vector
  int32_t n_unique;
  int32_t block;             // bin-splines per block
  int32_t n_seg;             // segment planes per block (max_knots - 1)
};
void launch_binned_weights(const BinnedArgs& a, hipStream_t s);

enum LlhW2Mode : int { kW2Replica0 = 0, kW2Fold = 1, kW2FoldInPlace = 2 };
// bin vectors beyond this many bins are evaluated one lane per bin across workgroups (k_llh_terms) and summed by a
// second small kernel (k_llh_sum), instead of one workgroup walking all of them with its terms in LDS
constexpr int M3_LLH_SPLIT_BINS = 4096;
constexpr int M3_LLH_SPLIT_LANES = 1024;

struct LlhArgs {
  const double* data;
  const double* hist;  // [2][hist_rep][n_bins]
  int32_t hist_rep;
  int32_t n_bins;
  int32_t n_samples;
  const int32_t* sample_off;  // [n_samples + 1]
  int32_t kind;
  int32_t w2_mode;  // kW2Replica0: w2 sits folded in replica 0; kW2Fold: sum the replicas (read only);
                    // kW2FoldInPlace: sum them, store the sum in replica 0 and clear the others (at the freeze)
  double* out;  // [1 + n_samples] (device pointer, may be mapped pinned host memory)
  double* terms;  // [n_bins] per-bin terms of the split form (more than M3_LLH_SPLIT_BINS bins), else unused
  uint64_t* dbg_clock;           // optional: [1] <- wall_clock64() at k_llh end
  volatile uint64_t* done_flag;  // mapped pinned host word; set to `ticket` after `out` is written
  uint64_t ticket;
};

// osc_table non-null: the words carry the oscillation weight (ReweightArgs::evw_weight)
void launch_pack_events(const int32_t* bin, const int32_t* osc, const int16_t* nidx, uint32_t n_norm_slots, uint64_t e_pad, int32_t norm_base,
                        uint64_t* out, const float* osc_table, hipStream_t s);
constexpr uint64_t kEvwBinMask = 0xFFFFFull, kEvwBinMaskWeight = 0x3FFFull;
void launch_prep(const PrepArgs& a, hipStream_t s);
void launch_resolve(const PrepArgs& a, int32_t n_res, hipStream_t s);
// run-time tuning knobs of the reweight kernel (m3hip_debug_set_knobs)
struct ReweightKnobs {
  int unroll = 8;   // loads kept in flight per lane: 8 or 16
  int nt = 1;       // 1: non-temporal coefficient loads (+5% on C2: planes are read once per launch)
  int lds_pad = 0;  // > 0: dynamic LDS bytes per workgroup, to cap workgroups per CU
  // optional: events that receive the dispatch's own start / end timestamps (hipExtLaunchKernelGGL) --
  // the same clock readings a profiler reports, without extra packets on the stream
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
};

void launch_reweight_uniform(const ReweightArgs& a, uint32_t n_tiles, int mode, const ReweightKnobs& k, hipStream_t s, bool dyn_bins = false);
// float_t = double epilogue (event spaces without event-by-event splines: all tiles uniform)
// The event side of a binned-only sample in PROCESSING order (position p = the p-th event of the order the engine works in: events
// with equal pointer lists next to each other), packed by k_pack_positions whenever one of the per-event arrays changes; read by
// k_reweight_lists (m3hip_kernels.hip)
struct PosArgs {
  const uint32_t* ev;    // [n_pos] position -> event (= ReweightArgs::perm)
  const int32_t* bin;    // bin of the event at the position
  const int32_t* osc;    // its oscillation index (-1: none)
  const uint32_t* loff;  // first entry of its pointer list in ReweightArgs::bl_idx
  const uint16_t* llen;  // entries of the list (0: the event has none)
  uint64_t n_pos;        // = events
};
void launch_reweight_lists(const ReweightArgs& a, const PosArgs& p, int mode, bool f64, const ReweightKnobs& k, hipStream_t s);
void launch_pack_positions(const uint32_t* perm, const int32_t* bin, const int32_t* osc, const int32_t* bl_id, const int32_t* bl_off, uint64_t n, int32_t* p_bin,
                           int32_t* p_osc, uint32_t* p_loff, uint16_t* p_llen, hipStream_t s);
void launch_reweight_uniform_f64(const ReweightArgs& a, uint32_t n_tiles, int mode, const ReweightKnobs& k, hipStream_t s);
void launch_reweight_brick(const ReweightArgs& a, uint32_t n_bricks, int mode, const ReweightKnobs& k, hipStream_t s);
// dynamic LDS of a fill launch and whether the bin vector is privatised there (see ReweightArgs::hist_lds)
constexpr int M3_HIST_LDS_MAX_BYTES = 20 * 1024;
void launch_llh(const LlhArgs& a, hipStream_t s);
// prior (rows 0..n-1 of `inv`, lower triangle) and correlated throw (L z) of one Metropolis step: in = [d | z] (2n doubles),
// out = [prior | corr_throw[n]] in mapped host memory, then `ticket` into *done_flag
void launch_par_step(const double* inv, const double* chol, const double* in_host, double* in_dev, int n, double* rows, double* out,
                     volatile uint64_t* done_flag, uint64_t ticket, hipStream_t s);
void launch_osc_convert(const float* f32, const double* f64, void* dst, uint64_t n, bool widen, hipStream_t s);
// One-shot exchange of the folded bin vector (m3hip_xch_*): every rank stores its vector and then a ticket into slot
// [parity][rank] of every rank's buffer; a rank's bins are the rank-ordered sum of the N slots of its own buffer once all N
// tickets of the parity carry the step number.  `status` (mapped host word): set to the step number when a ticket did not
// arrive within the time-out, so that the host can report it instead of waiting for ever.
void launch_xch(const double* src, double* dst, int n_msg, double* const* targets, int n_ranks, const double* slots, int slot_doubles, uint64_t step,
                uint64_t* status, unsigned long long* pushed, uint64_t timeout_ticks, hipStream_t s);
void launch_fold_hist(double* hist, int hist_rep, int n_bins, int with_w2, int park_w2, hipStream_t s);
void launch_unpark_w2(double* hist, int hist_rep, int n_bins, hipStream_t s);
// deterministic mode: ordered sum of the workgroups' rows into replica 0, 32 rows at a time (scratch: ceil(n_rows / 32) rows)
void launch_fold_partials(const double* partial, int n_rows, int n_bins, int with_w2, double* scratch, double* hist, int hist_rep, hipStream_t s);

}  // namespace m3hip
