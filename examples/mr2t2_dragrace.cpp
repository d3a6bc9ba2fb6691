// mr2t2_dragrace.cpp -- a ROOT-free host driver over the C++ facade (include/mach3_amd/):
//   1. builds a MasterSpline of TSpline3_red / TF1_red objects and hands it to SMonolith
//      (the reference's constructor signature, Splines/SplineMonolith.h:18-26);
//   2. FitterBase::DragRace: times Reweight() and GetLikelihood() over NLaps
//      (ref: Fitters/FitterBase.cpp:461-520);
//   3. a short Metropolis chain in MR2T2::ProposeStep order: propose -> prior / bounds ->
//      skip reweight when out of bounds -> Reweight -> GetLikelihood -> accept/reject
//      (ref: Fitters/MR2T2.cpp:25-115, Fitters/MCMCBase.cpp:188-216).
// Synthetic MC comes from libm3synth (same generator as the tests and bench.py).
//
//   usage: mr2t2_dragrace [n_events=50000] [n_params=32] [n_laps=100] [n_steps=2000]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../include/mach3_amd/SampleHandlerFD.h"
#include "../mach3_amd/synth/m3synth.h"

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const uint64_t E = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 50000;
  const int P = argc > 2 ? std::atoi(argv[2]) : 32;
  const int NLaps = argc > 3 ? std::atoi(argv[3]) : 100;
  const int NSteps = argc > 4 ? std::atoi(argv[4]) : 2000;

  m3synth_config cfg{};
  cfg.seed = 20260314;
  cfg.n_params = P;
  cfg.n_knots = 7;
  cfg.mixed = 1;  // odd parameters are TF1
  cfg.n_norm = 16;
  cfg.max_norm_per_event = 3;
  cfg.n_samples = 2;
  cfg.n_bins = 100;
  cfg.osc_n = 4000;
  cfg.occupancy = 1.0;
  cfg.bin_lo = 0.0;
  cfg.bin_hi = 5.0;
  cfg.nc_fraction = 0.1;
  const int nPar = P + cfg.n_norm;

  try {
    // ---- reference-layout arrays from the generator, turned back into response-function objects
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
    std::vector<TSpline3_red> keep;  // double-precision cross-check of a few events
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
        if (e < 4) keep.emplace_back(X, Y, 7, Pf);
      }
      for (unsigned i = 0; i < npe_t[2 * e]; ++i) {
        const unsigned t = npe_t[2 * e + 1] + i;
        MasterSpline[e][pno_t[t]] = new TF1_red(2, &coeff_tf1[2 * t]);
      }
    }
    std::vector<float>().swap(coeff_many);

    std::vector<double> par(nPar), nominal(nPar), sigma(nPar), lo(nPar), hi(nPar);
    m3synth_params(&cfg, 0, par.data());
    nominal = par;
    for (int i = 0; i < nPar; ++i) {
      sigma[i] = i < P ? 1.0 : 0.1;
      lo[i] = i < P ? -5.0 : 0.0;
      hi[i] = i < P ? 5.0 : 2.0;
    }

    double t0 = now_s();
    auto splines = std::make_unique<SMonolith>(MasterSpline, SplineType);
    std::vector<const double*> ptrs(P);
    for (int p = 0; p < P; ++p) ptrs[p] = &par[p];
    splines->setSplinePointers(ptrs);
    std::printf("SMonolith: %llu events, %d parameters, %llu TSpline3 + %llu TF1 responses, upload %.2f s\n",
                (unsigned long long)E, P, (unsigned long long)S, (unsigned long long)T, now_s() - t0);

    // seam 1 check: SMonolith::Evaluate against TSpline3_red::Eval / TF1_red::Eval in double
    for (int p = 0; p < P; ++p) par[p] = 0.37 * ((p % 5) - 2);
    splines->Evaluate();
    splines->SynchroniseMemTransfer();
    {
      size_t k = 0;
      for (uint64_t e = 0; e < 4 && e < E; ++e) {
        double w = 1.0;
        for (int p = 0; p < P; p += 2) w *= keep[k++].Eval(float(par[p]));
        for (unsigned i = 0; i < npe_t[2 * e]; ++i) {
          const unsigned t = npe_t[2 * e + 1] + i;
          w *= double(coeff_tf1[2 * t + 1]) + double(coeff_tf1[2 * t]) * double(float(par[pno_t[t]]));
        }
        const double got = *splines->retPointer(int(e));
        if (std::fabs(got - w) > 2e-5 * std::fabs(w)) throw MaCh3Exception(__FILE__, __LINE__, "Evaluate() disagrees with Eval() in double");
      }
      std::printf("Evaluate(): event weights agree with TSpline3_red::Eval (double) to 2e-5\n");
    }
    par = nominal;

    auto bins = std::make_unique<BinningHandler>();
    std::vector<double> edges(cfg.n_bins + 1);
    m3synth_bin_edges(&cfg, edges.data());
    for (int s = 0; s < cfg.n_samples; ++s) bins->AddUniformSample({edges});
    SampleHandlerFD sample("SyntheticFD", std::move(splines), std::move(bins), par.data(), nPar);
    sample.SetTestStatistic(kBarlowBeeston);
    {
      std::vector<int32_t> sid(E), osc(E);
      std::vector<double> erec(E);
      std::vector<uint64_t> noff(E + 1);
      std::vector<int16_t> nidx(E * cfg.max_norm_per_event + 1);
      m3synth_events(&cfg, 0, E, sid.data(), erec.data(), osc.data(), noff.data(), nidx.data());
      for (uint64_t e = 0; e < E; ++e)
        sample.AddEvent(unsigned(e), sid[e], {erec[e]}, std::vector<int16_t>(nidx.begin() + noff[e], nidx.begin() + noff[e + 1]), osc[e]);
      sample.FinaliseEvents();
      std::vector<float> tbl(cfg.osc_n);
      m3synth_osc_table(&cfg, tbl.data());
      sample.SetOscillationWeights(tbl);
    }
    // Asimov data = prediction at nominal (ref: Fitters/MaCh3Factory.h:142-153)
    sample.Reweight();
    for (int s = 0; s < cfg.n_samples; ++s) sample.AddData(s, sample.GetMCArray(s));
    sample.PrintRates();
    const double llh0 = sample.GetLikelihood();
    if (std::fabs(llh0) > 1e-9) throw MaCh3Exception(__FILE__, __LINE__, "Asimov likelihood is not zero");

    // ---- DragRace
    std::mt19937_64 rng(1234);
    std::normal_distribution<double> gaus(0.0, 1.0);
    double t_rw = 0, t_llh = 0;
    // spin-up: a fresh process shows one-off 50-100 ms stalls in its first second of GPU work
    for (const double t_s = now_s(); now_s() - t_s < 1.0;) {
      for (int i = 0; i < nPar; ++i) par[i] = nominal[i] + 0.3 * sigma[i] * gaus(rng);
      sample.Reweight();
      (void)sample.GetLikelihood();
    }
    for (int lap = 0; lap < NLaps + 10; ++lap) {
      for (int i = 0; i < nPar; ++i) par[i] = nominal[i] + 0.3 * sigma[i] * gaus(rng);
      const double a = now_s();
      sample.Reweight();
      const double b = now_s();
      const double llh = sample.GetLikelihood();
      const double c = now_s();
      (void)llh;
      if (lap >= 10) { t_rw += b - a; t_llh += c - b; }
    }
    std::printf("DragRace: %d laps, Reweight() launch %.1f us/lap, GetLikelihood() (incl. wait) %.1f us/lap -> %.0f steps/s\n", NLaps,
                1e6 * t_rw / NLaps, 1e6 * t_llh / NLaps, NLaps / (t_rw + t_llh));

    // ---- Metropolis chain, MR2T2 order
    std::uniform_real_distribution<double> unif(0.0, 1.0);
    par = nominal;
    std::vector<double> curr = par, prop(nPar);
    auto prior = [&](const std::vector<double>& x) {
      double l = 0;
      for (int i = 0; i < nPar; ++i) {
        if (x[i] < lo[i] || x[i] > hi[i]) return M3::_LARGE_LOGL_;  // ref: ParameterHandlerBase.cpp:859-867
        const double d = (x[i] - nominal[i]) / sigma[i];
        l += 0.5 * d * d;
      }
      return l;
    };
    sample.Reweight();
    double logLCurr = prior(curr) + sample.GetLikelihood();
    int accepted = 0, skipped = 0;
    const double scale = 0.5 / std::sqrt(double(nPar));
    t0 = now_s();
    for (int step = 0; step < NSteps; ++step) {
      for (int i = 0; i < nPar; ++i) prop[i] = curr[i] + scale * sigma[i] * gaus(rng);
      double logLProp = prior(prop);
      bool out_of_bounds = logLProp >= M3::_LARGE_LOGL_;
      if (!out_of_bounds) {  // ref: MR2T2.cpp:51-76: out-of-bounds proposals never reach Reweight()
        std::copy(prop.begin(), prop.end(), par.begin());
        sample.Reweight();
        logLProp += sample.GetLikelihood();
      } else {
        ++skipped;
      }
      const double accProb = out_of_bounds ? 0.0 : std::min(1.0, std::exp(logLCurr - logLProp));  // ref: MR2T2.cpp:103-115
      if (unif(rng) <= accProb) {
        curr = prop;
        logLCurr = logLProp;
        ++accepted;
      }
    }
    const double el = now_s() - t0;
    std::printf("MCMC: %d steps in %.3f s = %.0f steps/s, acceptance %.3f, %d out-of-bounds proposals skipped, final -logL %.4f\n", NSteps,
                el, NSteps / el, double(accepted) / NSteps, skipped, logLCurr);
    if (accepted == 0 || accepted == NSteps) throw MaCh3Exception(__FILE__, __LINE__, "degenerate acceptance rate");
  } catch (const std::exception& e) {
    std::fprintf(stderr, "FAILED: %s\n", e.what());
    return 1;
  }
  std::printf("OK\n");
  return 0;
}
