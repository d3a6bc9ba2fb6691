/* m3synth.c -- see m3synth.h.  Plain C + OpenMP; no dependence on the oracle or the engine. */
#include "m3synth.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define M3S_MAX_KNOTS 32

/* ---- counter-based random numbers: splitmix64 finaliser over (seed, stream, a, b) ---- */
static inline uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static inline uint64_t h3(uint64_t seed, uint64_t stream, uint64_t a, uint64_t b) {
  return mix64(mix64(mix64(seed ^ (stream * 0xD6E8FEB86659FD93ull)) + a) + b);
}
static inline double u01(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }
/* approximately N(0,1): centred sum of four 32-bit uniforms (variance 4/12), arithmetic only */
static inline double z4(uint64_t h0, uint64_t h1) {
  const double s = (double)(uint32_t)h0 + (double)(uint32_t)(h0 >> 32) + (double)(uint32_t)h1 + (double)(uint32_t)(h1 >> 32);
  return (s * (1.0 / 4294967296.0) - 2.0) * 1.7320508075688772;
}

enum { ST_PRESENT = 1, ST_RESP_A = 2, ST_RESP_B = 3, ST_TF1 = 4, ST_SAMPLE = 5, ST_EREC = 6, ST_OSC = 7, ST_NC = 8,
       ST_NNORM = 9, ST_NORM = 10, ST_OSCTAB = 11, ST_PAR = 12 };

static inline int is_tf1(const m3synth_config* c, int p) { return c->mixed && (p & 1); }

static inline int present(const m3synth_config* c, uint64_t e, int p) {
  if (c->occupancy >= 1.0) return 1;
  return u01(h3(c->seed, ST_PRESENT, e, (uint64_t)p)) < c->occupancy;
}

static inline double knot_x(const m3synth_config* c, int j) { return (double)j - 0.5 * (double)(c->n_knots - 1); }

void m3synth_knots(const m3synth_config* c, float* coeff_x, int16_t* npts) {
  for (int p = 0; p < c->n_params; ++p) {
    for (int j = 0; j < c->n_knots; ++j) coeff_x[(size_t)p * c->n_knots + j] = is_tf1(c, p) ? -999.f : (float)knot_x(c, j);
    npts[p] = is_tf1(c, p) ? (int16_t)-999 : (int16_t)c->n_knots;
  }
}

/* natural cubic spline through (x_j, y_j), unit knot spacing; writes {y,b,c,d} per knot
 * in double then casts to float (ref: SplineMonolith.cpp:673-677 casts the same way) */
static void build_spline(const m3synth_config* c, uint64_t e, int p, float* out /*[K][4]*/) {
  const int K = c->n_knots;
  double y[M3S_MAX_KNOTS], M[M3S_MAX_KNOTS], cp[M3S_MAX_KNOTS], dp[M3S_MAX_KNOTS];
  const double a = 0.08 * z4(h3(c->seed, ST_RESP_A, e, (uint64_t)p), h3(c->seed, ST_RESP_A, e, (uint64_t)p + 0x100000000ull));
  const double b = 0.01 * z4(h3(c->seed, ST_RESP_B, e, (uint64_t)p), h3(c->seed, ST_RESP_B, e, (uint64_t)p + 0x100000000ull));
  for (int j = 0; j < K; ++j) {
    const double x = knot_x(c, j);
    const double v = 1.0 + a * x + b * x * x;
    y[j] = v < 0.05 ? 0.05 : v;
  }
  M[0] = 0.0;
  M[K - 1] = 0.0;
  if (K > 2) { /* Thomas algorithm on M[i-1] + 4 M[i] + M[i+1] = 6 (y[i+1] - 2 y[i] + y[i-1]) */
    cp[1] = 1.0 / 4.0;
    dp[1] = 6.0 * (y[2] - 2.0 * y[1] + y[0]) / 4.0;
    for (int i = 2; i <= K - 2; ++i) {
      const double den = 4.0 - cp[i - 1];
      cp[i] = 1.0 / den;
      dp[i] = (6.0 * (y[i + 1] - 2.0 * y[i] + y[i - 1]) - dp[i - 1]) / den;
    }
    M[K - 2] = dp[K - 2];
    for (int i = K - 3; i >= 1; --i) M[i] = dp[i] - cp[i] * M[i + 1];
  }
  for (int j = 0; j < K; ++j) {
    double bb = 0.0, cc = 0.0, dd = 0.0;
    if (j < K - 1) {
      bb = (y[j + 1] - y[j]) - (2.0 * M[j] + M[j + 1]) / 6.0;
      cc = 0.5 * M[j];
      dd = (M[j + 1] - M[j]) / 6.0;
    }
    out[4 * j + 0] = (float)y[j];
    out[4 * j + 1] = (float)bb;
    out[4 * j + 2] = (float)cc;
    out[4 * j + 3] = (float)dd;
  }
}

static void count_event(const m3synth_config* c, uint64_t e, unsigned int* ns, unsigned int* nt) {
  unsigned int s = 0, t = 0;
  if (c->occupancy >= 1.0) {
    for (int p = 0; p < c->n_params; ++p) { if (is_tf1(c, p)) ++t; else ++s; }
  } else {
    for (int p = 0; p < c->n_params; ++p)
      if (present(c, e, p)) { if (is_tf1(c, p)) ++t; else ++s; }
  }
  *ns = s;
  *nt = t;
}

void m3synth_count(const m3synth_config* c, uint64_t e0, uint64_t e1, uint64_t* n_splines, uint64_t* n_tf1) {
  uint64_t S = 0, T = 0;
#pragma omp parallel for reduction(+ : S, T) schedule(static)
  for (uint64_t e = e0; e < e1; ++e) {
    unsigned int s, t;
    count_event(c, e, &s, &t);
    S += s;
    T += t;
  }
  *n_splines = S;
  *n_tf1 = T;
}

void m3synth_monolith(const m3synth_config* c, uint64_t e0, uint64_t e1, float* coeff_many, unsigned int* nknots_arr,
                      int16_t* param_no, float* coeff_tf1, int16_t* param_no_tf1, unsigned int* npe,
                      unsigned int* npe_tf1) {
  const uint64_t n = e1 - e0;
  const int K = c->n_knots;
  /* pass 1: {count,start} maps (ref: SplineMonolith.cpp:209-215) */
#pragma omp parallel for schedule(static)
  for (uint64_t i = 0; i < n; ++i) count_event(c, e0 + i, &npe[2 * i], &npe_tf1[2 * i]);
  unsigned int s0 = 0, t0 = 0;
  for (uint64_t i = 0; i < n; ++i) {
    npe[2 * i + 1] = s0;
    npe_tf1[2 * i + 1] = t0;
    s0 += npe[2 * i];
    t0 += npe_tf1[2 * i];
  }
  /* pass 2: coefficients, ascending parameter number within an event (ref: :142-219) */
#pragma omp parallel for schedule(dynamic, 256)
  for (uint64_t i = 0; i < n; ++i) {
    const uint64_t e = e0 + i;
    uint64_t s = npe[2 * i + 1], t = npe_tf1[2 * i + 1];
    for (int p = 0; p < c->n_params; ++p) {
      if (!present(c, e, p)) continue;
      if (is_tf1(c, p)) {
        const double a = 0.05 * z4(h3(c->seed, ST_TF1, e, (uint64_t)p), h3(c->seed, ST_TF1, e, (uint64_t)p + 0x100000000ull));
        coeff_tf1[2 * t + 0] = (float)a; /* w = a*x + b (ref: SplineMonolith.cpp:776-779) */
        coeff_tf1[2 * t + 1] = 1.0f;
        param_no_tf1[t] = (int16_t)p;
        ++t;
      } else {
        build_spline(c, e, p, &coeff_many[(size_t)s * K * 4]);
        nknots_arr[s] = (unsigned int)(s * (uint64_t)K);
        param_no[s] = (int16_t)p;
        ++s;
      }
    }
  }
}

void m3synth_events(const m3synth_config* c, uint64_t e0, uint64_t e1, int32_t* sample_id, double* erec,
                    int32_t* osc_idx, uint64_t* norm_off, int16_t* norm_idx) {
  const uint64_t n = e1 - e0;
  norm_off[0] = 0;
  for (uint64_t i = 0; i < n; ++i) {
    const uint64_t e = e0 + i;
    const int nn = c->n_norm > 0 ? (int)(h3(c->seed, ST_NNORM, e, 0) % (uint64_t)(c->max_norm_per_event + 1)) : 0;
    norm_off[i + 1] = norm_off[i] + (uint64_t)nn;
  }
#pragma omp parallel for schedule(static)
  for (uint64_t i = 0; i < n; ++i) {
    const uint64_t e = e0 + i;
    sample_id[i] = (int32_t)(h3(c->seed, ST_SAMPLE, e, 0) % (uint64_t)c->n_samples);
    /* Erec ~ Gamma(k=3, theta=0.35): sum of three exponentials */
    const double u1 = u01(h3(c->seed, ST_EREC, e, 0)), u2 = u01(h3(c->seed, ST_EREC, e, 1)), u3 = u01(h3(c->seed, ST_EREC, e, 2));
    erec[i] = -0.35 * log((1.0 - u1) * (1.0 - u2) * (1.0 - u3));
    const int nc = u01(h3(c->seed, ST_NC, e, 0)) < c->nc_fraction;
    if (nc || c->osc_n <= 0) osc_idx[i] = -1;
    else if (c->osc_sorted_events) {
      const uint64_t k = (uint64_t)(((unsigned __int128)e * (uint64_t)c->osc_n) / c->osc_sorted_events);
      osc_idx[i] = (int32_t)(k < (uint64_t)c->osc_n ? k : (uint64_t)c->osc_n - 1);
    } else osc_idx[i] = (int32_t)(h3(c->seed, ST_OSC, e, 0) % (uint64_t)c->osc_n);
    const int nn = (int)(norm_off[i + 1] - norm_off[i]);
    for (int k = 0; k < nn; ++k)
      norm_idx[norm_off[i] + k] = (int16_t)(c->n_params + (int)(h3(c->seed, ST_NORM, e, (uint64_t)k) % (uint64_t)c->n_norm));
  }
}

void m3synth_osc_table(const m3synth_config* c, float* table) {
  for (int i = 0; i < c->osc_n; ++i) table[i] = (float)u01(h3(c->seed, ST_OSCTAB, (uint64_t)i, 0));
}

void m3synth_bin_edges(const m3synth_config* c, double* edges) {
  for (int i = 0; i <= c->n_bins; ++i) edges[i] = c->bin_lo + (c->bin_hi - c->bin_lo) * (double)i / (double)c->n_bins;
}

void m3synth_params(const m3synth_config* c, uint64_t lap, double* par) {
  for (int p = 0; p < c->n_params; ++p)
    par[p] = lap == 0 ? 0.0 : 0.3 * z4(h3(c->seed, ST_PAR, lap, (uint64_t)p), h3(c->seed, ST_PAR, lap, (uint64_t)p + 0x100000000ull));
  for (int k = 0; k < c->n_norm; ++k) {
    const uint64_t q = (uint64_t)(c->n_params + k);
    par[c->n_params + k] = lap == 0 ? 1.0 : 1.0 + 0.05 * z4(h3(c->seed, ST_PAR, lap, q), h3(c->seed, ST_PAR, lap, q + 0x100000000ull));
  }
}
