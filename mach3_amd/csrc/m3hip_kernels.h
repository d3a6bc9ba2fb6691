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
struct TileDev {
  const float4* spl;    // [(plane0[j] + k) * M3_TE + lane]
  const float2* tf1;    // [j * M3_TE + lane]
  const int32_t* meta;  // [2*J_s] {param, plane0}, then [J_t] {param}
  uint32_t n_spl_slots;
  uint32_t n_tf1_slots;
  uint32_t n_events;
  uint32_t first_event;  // low 32 bits unused here; events are addressed by tile index * M3_TE
};

struct ReweightArgs {
  const TileDev* tiles;
  const PTab* ptab;          // [P]
  const float* pval;         // [P] ParamValues
  float* spline_w;           // [E_pad] monolith total weight per event (may be null)
  float* event_w;            // [E_pad] CalcWeightTotal per event (may be null)
  // fused fill
  const int32_t* bin;        // [E_pad]
  const int32_t* osc;        // [E_pad]
  const int16_t* nidx;       // [M][E_pad] ELL of global parameter indices, padded with n_par (-> 1.0)
  const double* params;      // [n_par + 1], last entry 1.0
  const float* osc_table;
  double* hist;              // [2 * n_bins]: mc then w2
  uint64_t e_pad;
  uint32_t n_norm_slots;     // M
  int32_t n_bins;
  int32_t fill_w2;
  uint32_t tile_base;        // index of the first tile handled by this launch
};

enum ReweightMode : int { kModeEval = 1, kModeFill = 2, kModeEventW = 4 };

struct PrepArgs {
  // FindSplineSegment inputs
  const double* params_src;   // pinned host (device-mapped) or device: proposed values [n_par]
  double* params_dev;         // [n_par + 1] device copy, [n_par] = 1.0
  int32_t n_par;
  const int32_t* par_map;     // [P] spline parameter -> global index
  const float* coeff_x;       // [P*K]
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
  // histogram reset (ResetHistograms)
  double* hist;
  int32_t n_bins;
  int32_t reset_w2;
  int32_t reset_hist;
};

struct LlhArgs {
  const double* data;
  const double* hist;  // mc[n_bins], w2[n_bins]
  int32_t n_bins;
  int32_t n_samples;
  const int32_t* sample_off;  // [n_samples + 1]
  int32_t kind;
  double* out;  // [1 + n_samples] (device pointer, may be mapped pinned host memory)
};

void launch_prep(const PrepArgs& a, hipStream_t s);
void launch_reweight_uniform(const ReweightArgs& a, uint32_t n_tiles, int mode, hipStream_t s);
void launch_llh(const LlhArgs& a, hipStream_t s);

}  // namespace m3hip
