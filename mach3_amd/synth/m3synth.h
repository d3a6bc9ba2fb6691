/*
 * m3synth.h -- seeded synthetic far-detector MC in the REFERENCE's flat layout.
 *
 * Produces, for any event range [e0,e1) of a configuration, exactly the arrays that
 * SMonolith::PrepareForGPU (ref: Splines/SplineMonolith.cpp:53-250) and
 * SampleHandlerFD::Initialise (ref: Samples/SampleHandlerFD.cpp:169-202) would hand to
 * the per-step path, so that the HIP engine and the CPU oracle ingest identical bytes.
 * Every random draw is a pure function of (seed, stream, event, item): an event's data
 * does not depend on the chunking or on the rank that generates it, which is what makes
 * event sharding across GPUs reproducible.
 *
 * Distributions follow SURVEY.md section 8(d) (knots at integer sigma, quadratic knot
 * responses with a natural cubic through them, linear TF1s, Gamma(3,0.35) reconstructed
 * energy, 0-3 normalisation parameters per event, pre-staged oscillation table).
 */
#ifndef M3SYNTH_H
#define M3SYNTH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  uint64_t seed;
  int32_t n_params;   /* spline-handler parameters P (TSpline3 + TF1) */
  int32_t n_knots;    /* knots per TSpline3 parameter (K, the monolith's _max_knots) */
  int32_t mixed;      /* 0: every parameter is a TSpline3; 1: odd index = TF1, even = TSpline3 */
  int32_t n_norm;     /* normalisation parameters, stored after the P spline parameters */
  int32_t max_norm_per_event; /* each event gets 0..max (uniform) of them */
  int32_t n_samples;  /* far-detector samples; each has n_bins uniform bins on [bin_lo,bin_hi) */
  int32_t n_bins;
  int32_t osc_n;      /* entries in the pre-staged oscillation-weight table */
  double occupancy;   /* P(event has a response for a parameter) */
  double bin_lo, bin_hi;
  double nc_fraction; /* events with no oscillation pointer (weight 1) */
  uint64_t osc_sorted_events; /* 0: every event draws its table index independently (the configurations' default);
                               * N > 0: MC registered in oscillation-bin order -- event e of N points at entry
                               * floor(e * osc_n / N) (same number of events per entry, ascending with the event number) */
} m3synth_config;

/* coeff_x[P*K] (-999 where unused, as SplineMonolith.cpp:105 initialises it) and
 * FastSplineInfo.nPts per parameter (-999 for TF1 parameters). */
void m3synth_knots(const m3synth_config* c, float* coeff_x, int16_t* npts);

/* number of valid TSpline3 / TF1 responses of events [e0,e1) */
void m3synth_count(const m3synth_config* c, uint64_t e0, uint64_t e1, uint64_t* n_splines, uint64_t* n_tf1);

/* Reference-layout monolith arrays for events [e0,e1); offsets are chunk-local
 * (nknots_arr[0] == 0, n_param_per_event start of the first event == 0). */
void m3synth_monolith(const m3synth_config* c, uint64_t e0, uint64_t e1,
                      float* coeff_many, unsigned int* nknots_arr, int16_t* param_no,
                      float* coeff_tf1, int16_t* param_no_tf1,
                      unsigned int* n_param_per_event, unsigned int* n_param_per_event_tf1);

/* Sample-side per-event data for events [e0,e1): NominalSample, the binning variable,
 * oscillation-table index (-1: none), CSR of normalisation-parameter GLOBAL indices
 * (norm_off has n+1 entries; norm_idx capacity n*max_norm_per_event). */
void m3synth_events(const m3synth_config* c, uint64_t e0, uint64_t e1,
                    int32_t* sample_id, double* erec, int32_t* osc_idx,
                    uint64_t* norm_off, int16_t* norm_idx);

void m3synth_osc_table(const m3synth_config* c, float* table);
/* bin edges of one sample: n_bins+1 doubles */
void m3synth_bin_edges(const m3synth_config* c, double* edges);
/* global parameter vector [P + n_norm] for lap `lap`: lap 0 = nominal (theta=0, norm=1),
 * lap>0: theta = 0.3 z, norm = 1 + 0.05 z */
void m3synth_params(const m3synth_config* c, uint64_t lap, double* par);

#ifdef __cplusplus
}
#endif
#endif
