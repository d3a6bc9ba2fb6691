// likelihood_fit_nm.cpp -- the minimisers' caller of the hot path: LikelihoodFit::CalcChi2 (Fitters/LikelihoodFit.cpp:39-123), what
// MinuitFit and PSO evaluate, over the GPU sample of the C++ facade.  No Minuit2 here: a Nelder-Mead simplex (with restarts)
// minimises CalcChi2 over every parameter of a small synthetic sample, starting half a prior sigma away from an Asimov data set's
// truth.  The minimum of an Asimov fit is the point the data were made at, with chi2 = 0: the fit must return it.
//
// How close a minimiser can get is set by the likelihood's own rounding: near the minimum chi2 = sum (delta_i / sigma_i)^2 (the prior
// dominates this sample's curvature), and a Poisson bin's mc - data + data log(data / mc) cancels to ~1e-16 x its content -- with
// 2e4 events chi2 is known to ~1e-13, i.e. the parameters to ~5e-7 sigma; with 1e6 events (fifty times the content per bin) to 3e-6.
//
//   usage: likelihood_fit_nm [n_events=20000] [n_spline_params=4] [n_norm=2]
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <numeric>

#include "../include/mach3_amd/Fitters.h"
#include "../include/mach3_amd/SampleHandlerFD.h"
#include "../mach3_amd/synth/m3synth.h"

// Nelder-Mead with the adaptive coefficients of Gao & Han (2012); returns the number of function evaluations
static int nelder_mead(const std::function<double(const double*)>& f, std::vector<double>& x, const std::vector<double>& step, double size_tol, int max_eval) {
  const int n = int(x.size());
  const double alpha = 1.0, beta = 1.0 + 2.0 / n, gamma = 0.75 - 0.5 / n, delta = 1.0 - 1.0 / n;
  std::vector<std::vector<double>> s(size_t(n) + 1, x);
  std::vector<double> fv(size_t(n) + 1);
  for (int i = 0; i < n; ++i) s[size_t(i) + 1][size_t(i)] += step[size_t(i)];
  int evals = 0;
  for (auto i = size_t(0); i <= size_t(n); ++i, ++evals) fv[i] = f(s[i].data());
  std::vector<size_t> ord(size_t(n) + 1);
  std::vector<double> c(static_cast<size_t>(n)), xr(static_cast<size_t>(n)), xe(static_cast<size_t>(n));
  while (evals < max_eval) {
    std::iota(ord.begin(), ord.end(), size_t(0));
    std::sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return fv[a] < fv[b]; });
    const size_t lo = ord[0], hi = ord[size_t(n)], nhi = ord[size_t(n) - 1];
    double size = 0.0;
    for (size_t i = 1; i <= size_t(n); ++i)
      for (int k = 0; k < n; ++k) size = std::max(size, std::fabs(s[ord[i]][size_t(k)] - s[lo][size_t(k)]) / step[size_t(k)]);
    if (size < size_tol) break;
    for (int k = 0; k < n; ++k) {
      c[size_t(k)] = 0.0;
      for (size_t i = 0; i <= size_t(n); ++i)
        if (i != hi) c[size_t(k)] += s[i][size_t(k)] / n;
      xr[size_t(k)] = c[size_t(k)] + alpha * (c[size_t(k)] - s[hi][size_t(k)]);
    }
    const double fr = f(xr.data());
    ++evals;
    if (fr < fv[lo]) {
      for (int k = 0; k < n; ++k) xe[size_t(k)] = c[size_t(k)] + beta * (xr[size_t(k)] - c[size_t(k)]);
      const double fe = f(xe.data());
      ++evals;
      if (fe < fr) { s[hi] = xe; fv[hi] = fe; } else { s[hi] = xr; fv[hi] = fr; }
    } else if (fr < fv[nhi]) {
      s[hi] = xr; fv[hi] = fr;
    } else {
      const bool outside = fr < fv[hi];
      for (int k = 0; k < n; ++k) xe[size_t(k)] = c[size_t(k)] + gamma * ((outside ? xr[size_t(k)] : s[hi][size_t(k)]) - c[size_t(k)]);
      const double fc = f(xe.data());
      ++evals;
      if (fc < (outside ? fr : fv[hi])) { s[hi] = xe; fv[hi] = fc; }
      else {  // shrink towards the best vertex
        for (size_t i = 0; i <= size_t(n); ++i)
          if (i != lo) {
            for (int k = 0; k < n; ++k) s[i][size_t(k)] = s[lo][size_t(k)] + delta * (s[i][size_t(k)] - s[lo][size_t(k)]);
            fv[i] = f(s[i].data());
            ++evals;
          }
      }
    }
  }
  x = s[size_t(std::min_element(fv.begin(), fv.end()) - fv.begin())];
  return evals;
}

int main(int argc, char** argv) {
  const uint64_t E = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 20000;
  const int P = argc > 2 ? std::atoi(argv[2]) : 4;
  const int NN = argc > 3 ? std::atoi(argv[3]) : 2;
  m3synth_config cfg{};
  cfg.seed = 20260501; cfg.n_params = P; cfg.n_knots = 7; cfg.mixed = 1; cfg.n_norm = NN; cfg.max_norm_per_event = std::min(NN, 2);
  cfg.n_samples = 1; cfg.n_bins = 100; cfg.osc_n = 4000; cfg.occupancy = 1.0; cfg.bin_lo = 0.0; cfg.bin_hi = 5.0; cfg.nc_fraction = 0.1;
  const int nPar = P + NN;
  try {
    uint64_t S = 0, T = 0;
    m3synth_count(&cfg, 0, E, &S, &T);
    std::vector<float> coeff_many(S * 7 * 4), coeff_tf1(T * 2), coeff_x(size_t(P) * 7);
    std::vector<unsigned> nknots(S), npe(2 * E), npe_t(2 * E);
    std::vector<int16_t> pno(S), pno_t(T), npts(P);
    m3synth_knots(&cfg, coeff_x.data(), npts.data());
    m3synth_monolith(&cfg, 0, E, coeff_many.data(), nknots.data(), pno.data(), coeff_tf1.data(), pno_t.data(), npe.data(), npe_t.data());
    std::vector<RespFuncType> SplineType(P);
    for (int p = 0; p < P; ++p) SplineType[p] = (p & 1) ? kTF1_red : kTSpline3_red;
    std::vector<std::vector<TResponseFunction_red*>> MasterSpline(E, std::vector<TResponseFunction_red*>(P, nullptr));
    for (uint64_t e = 0; e < E; ++e) {
      for (unsigned i = 0; i < npe[2 * e]; ++i) {
        const unsigned s = npe[2 * e + 1] + i;
        const float* c = &coeff_many[size_t(nknots[s]) * 4];
        float X[7], Y[7], Pf[21];
        for (int k = 0; k < 7; ++k) {
          X[k] = coeff_x[size_t(pno[s]) * 7 + k];
          Y[k] = c[4 * k];
          Pf[3 * k] = c[4 * k + 1]; Pf[3 * k + 1] = c[4 * k + 2]; Pf[3 * k + 2] = c[4 * k + 3];
        }
        MasterSpline[e][pno[s]] = new TSpline3_red(X, Y, 7, Pf);
      }
      for (unsigned i = 0; i < npe_t[2 * e]; ++i) {
        const unsigned t = npe_t[2 * e + 1] + i;
        MasterSpline[e][pno_t[t]] = new TF1_red(2, &coeff_tf1[2 * t]);
      }
    }
    std::vector<double> nominal(nPar), sigma(nPar), lo(nPar), hi(nPar), cov(size_t(nPar) * nPar, 0.0);
    m3synth_params(&cfg, 0, nominal.data());
    for (int i = 0; i < nPar; ++i) {
      sigma[i] = i < P ? 1.0 : 0.1;
      lo[i] = i < P ? -2.9 : 0.0;
      hi[i] = i < P ? 2.9 : 2.0;
      cov[size_t(i) * nPar + i] = sigma[i] * sigma[i];
    }
    ParameterHandlerBase ph("synthetic_systematics", nominal, cov, lo, hi, {}, 7);
    auto splines = std::make_unique<SMonolith>(MasterSpline, SplineType);
    std::vector<const double*> ptrs(P);
    for (int p = 0; p < P; ++p) ptrs[p] = ph.RetPointer(p);
    splines->setSplinePointers(ptrs);
    auto bins = std::make_unique<BinningHandler>();
    std::vector<double> edges(cfg.n_bins + 1);
    m3synth_bin_edges(&cfg, edges.data());
    bins->AddUniformSample({edges});
    SampleHandlerFD sample("SyntheticFD", std::move(splines), std::move(bins), ph.RetPointer(0), nPar);
    sample.SetTestStatistic(kPoisson);
    {
      std::vector<int32_t> sid(E), osc(E);
      std::vector<double> erec(E);
      std::vector<uint64_t> noff(E + 1);
      std::vector<int16_t> nidx(E * std::max(1, cfg.max_norm_per_event) + 1);
      m3synth_events(&cfg, 0, E, sid.data(), erec.data(), osc.data(), noff.data(), nidx.data());
      for (uint64_t e = 0; e < E; ++e)
        sample.AddEvent(unsigned(e), 0, {erec[e]}, std::vector<int16_t>(nidx.begin() + noff[e], nidx.begin() + noff[e + 1]), osc[e]);
      sample.FinaliseEvents();
      std::vector<float> tbl(cfg.osc_n);
      m3synth_osc_table(&cfg, tbl.data());
      sample.SetOscillationWeights(tbl);
    }
    sample.SetDeterministicFill(true);  // chi2 is then a FUNCTION of the parameters: the same point gives the same bits
    sample.Reweight();
    sample.AddData(0, sample.GetMCArray(0));  // Asimov data at the nominal point (ref: Fitters/MaCh3Factory.h:142-153)

    LikelihoodFit fit;
    fit.AddSampleHandler(&sample);
    fit.AddSystObj(&ph);
    fit.SetMirroring(true);
    fit.PrepareFit();
    if (fit.GetNPars() != nPar) throw MaCh3Exception(__FILE__, __LINE__, "PrepareFit: parameter count");
    const double at_truth = fit.CalcChi2(nominal.data());
    if (std::fabs(at_truth) > 1e-9) throw MaCh3Exception(__FILE__, __LINE__, "chi2 at the Asimov point is not zero");
    // mirroring: a value beyond a bound evaluates as its reflection
    {
      std::vector<double> a = nominal, b = nominal;
      a[0] = hi[0] + 0.25;
      b[0] = hi[0] - 0.25;
      if (fit.CalcChi2(a.data()) != fit.CalcChi2(b.data())) throw MaCh3Exception(__FILE__, __LINE__, "mirroring at the upper bound");
    }
    std::vector<double> x = nominal, step0(static_cast<size_t>(nPar));
    for (int i = 0; i < nPar; ++i) {
      x[size_t(i)] += (i % 2 ? -0.5 : 0.5) * sigma[size_t(i)];
      step0[size_t(i)] = 0.25 * sigma[size_t(i)];
    }
    const double chi2_start = fit.CalcChi2(x.data());
    const auto t0 = std::chrono::steady_clock::now();
    int evals = 0;
    auto f = [&](const double* p) { return fit.CalcChi2(p); };
    std::vector<double> st = step0;
    for (int restart = 0; restart < 6; ++restart) {  // a fresh simplex around the best point, a tenth of the size each time
      evals += nelder_mead(f, x, st, 1e-4, 4000);
      for (auto& v : st) v *= 0.1;
    }
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double chi2_min = fit.CalcChi2(x.data());
    double worst = 0.0;
    for (int i = 0; i < nPar; ++i) worst = std::max(worst, std::fabs(x[size_t(i)] - nominal[size_t(i)]));
    std::printf("LikelihoodFit::CalcChi2 under Nelder-Mead: %d parameters, %llu events, chi2 %.4f at the start -> %.3e after %d evaluations (%.0f evaluations/s); "
                "largest |fitted - true| = %.2e\n", nPar, (unsigned long long)E, chi2_start, chi2_min, evals, evals / el, worst);
    if (!(worst <= 1e-6)) throw MaCh3Exception(__FILE__, __LINE__, "the Asimov fit did not return the point the data were made at (1e-6)");
    if (fit.GetStepCount() < evals) throw MaCh3Exception(__FILE__, __LINE__, "step bookkeeping");
    std::printf("OK\n");
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "likelihood_fit_nm: %s\n", e.what());
    return 1;
  }
}
