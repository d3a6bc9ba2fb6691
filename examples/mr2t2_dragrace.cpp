// mr2t2_dragrace.cpp -- a ROOT-free host driver over the C++ facade (include/mach3_amd/):
//   1. builds a MasterSpline of TSpline3_red / TF1_red objects and hands it to SMonolith
//      (the reference's constructor signature, Splines/SplineMonolith.h:18-26);
//   2. FitterBase::DragRace: times Reweight() and GetLikelihood() over NLaps
//      (ref: Fitters/FitterBase.cpp:461-520);
//   3. Markov chains with the facade's ROOT-free fitters (include/mach3_amd/Fitters.h): MR2T2
//      (propose -> prior / bounds -> skip reweight when out of bounds -> Reweight -> GetLikelihood ->
//      accept/reject; ref: Fitters/MR2T2.cpp:25-115), the same with the adaptive proposal
//      (Parameters/AdaptiveMCMCHandler.cpp:332-397) and DelayedMR2T2 (Fitters/DelayedMR2T2.cpp:95-235).
// Synthetic MC comes from libm3synth (same generator as the tests and bench.py).
//
//   usage: mr2t2_dragrace [n_events=50000] [n_params=32] [n_laps=100] [n_steps=2000]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <random>

#include "../include/mach3_amd/Fitters.h"
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
  const int nPar = P + cfg.n_norm + 1;  // spline parameters, normalisations, one energy-scale functional parameter
  const int iEScale = nPar - 1;

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

    std::vector<double> nominal(nPar), sigma(nPar), lo(nPar), hi(nPar), cov(size_t(nPar) * nPar, 0.0);
    m3synth_params(&cfg, 0, nominal.data());  // fills the first nPar - 1
    for (int i = 0; i < nPar; ++i) {
      sigma[i] = i < P ? 1.0 : 0.1;
      lo[i] = i < P ? -5.0 : 0.0;
      hi[i] = i < P ? 5.0 : 2.0;
      cov[size_t(i) * nPar + i] = sigma[i] * sigma[i];
    }
    nominal[iEScale] = 0.0; sigma[iEScale] = 0.02; lo[iEScale] = -0.2; hi[iEScale] = 0.2;
    cov[size_t(iEScale) * nPar + iEScale] = sigma[iEScale] * sigma[iEScale];
    // the parameter handler owns the proposed values; splines and the sample read them through pointers
    ParameterHandlerBase ph("synthetic_systematics", nominal, cov, lo, hi, {}, 4242);
    auto setpar = [&](int i, double v) { ph.SetParCurrProp(i, v); };

    double t0 = now_s();
    const std::string dump = "/tmp/m3_dragrace_monolith_" + std::to_string(uint64_t(now_s() * 1e6)) + ".bin";
    auto splines = std::make_unique<SMonolith>(MasterSpline, SplineType, /*SaveFlatTree=*/true, dump);
    std::vector<const double*> ptrs(P);
    for (int p = 0; p < P; ++p) ptrs[p] = ph.RetPointer(p);
    splines->setSplinePointers(ptrs);
    std::printf("SMonolith: %llu events, %d parameters, %llu TSpline3 + %llu TF1 responses, upload %.2f s\n",
                (unsigned long long)E, P, (unsigned long long)S, (unsigned long long)T, now_s() - t0);

    // seam 1 check: SMonolith::Evaluate against TSpline3_red::Eval / TF1_red::Eval in double
    for (int p = 0; p < P; ++p) setpar(p, 0.37 * ((p % 5) - 2));
    const std::vector<double> par = ph.GetParPropVec();
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
    {
      // PrepareSplineFile / LoadSplineFile round trip (ref: SplineMonolith.cpp:454-614): a monolith rebuilt from the
      // flat dump evaluates to the same bits
      SMonolith reloaded(dump);
      std::remove(dump.c_str());
      reloaded.setSplinePointers(ptrs);
      reloaded.Evaluate();
      reloaded.SynchroniseMemTransfer();
      for (uint64_t e = 0; e < E; ++e)
        if (std::memcmp(reloaded.retPointer(int(e)), splines->retPointer(int(e)), sizeof(float)) != 0)
          throw MaCh3Exception(__FILE__, __LINE__, "monolith reloaded from its dump evaluates differently");
      std::printf("LoadSplineFile(): %llu event weights of the reloaded monolith are bit-identical\n", (unsigned long long)E);
    }
    for (int i = 0; i < nPar; ++i) setpar(i, nominal[i]);

    auto bins = std::make_unique<BinningHandler>();
    std::vector<double> edges(cfg.n_bins + 1);
    m3synth_bin_edges(&cfg, edges.data());
    for (int s = 0; s < cfg.n_samples; ++s) bins->AddUniformSample({edges});
    SampleHandlerFD sample("SyntheticFD", std::move(splines), std::move(bins), ph.RetPointer(0), nPar);
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

    // ---- experiment weights behind the splines (AddAdditionalWeightPointers / CalcWeightFunc): a per-event column of
    // 2.0 doubles every bin exactly, a shared weight of 0.5 behind it restores the prediction bit for bit, and a
    // repeated step in deterministic mode reproduces its bins exactly
    {
      std::vector<double> mc0 = sample.GetMCArray();
      sample.AddAdditionalWeightPointers(0, std::vector<float>(E, 2.0f));
      sample.Reweight();
      std::vector<double> mc2 = sample.GetMCArray();
      for (size_t b = 0; b < mc0.size(); ++b)
        if (std::fabs(mc2[b] - 2.0 * mc0[b]) > 1e-12 * std::max(1.0, mc2[b])) throw MaCh3Exception(__FILE__, __LINE__, "weight slot 0 did not double bin " + std::to_string(b));
      sample.AddAdditionalWeightPointers(1, std::vector<int32_t>(E, 0));
      sample.SetAdditionalWeights(1, {0.5f});
      sample.SetDeterministicFill(true);
      sample.Reweight();
      const std::vector<double> mc1 = sample.GetMCArray();
      sample.Reweight();
      if (sample.GetMCArray() != mc1) throw MaCh3Exception(__FILE__, __LINE__, "deterministic fill is not reproducible");
      for (size_t b = 0; b < mc0.size(); ++b)
        if (std::fabs(mc1[b] - mc0[b]) > 1e-12 * std::max(1.0, mc0[b])) throw MaCh3Exception(__FILE__, __LINE__, "weight slots 0 x 1 did not restore bin " + std::to_string(b));
      sample.SetDeterministicFill(false);
      sample.SetAdditionalWeights(1, {1.0f});
      sample.AddAdditionalWeightPointers(0, std::vector<float>(E, 1.0f));
      std::printf("experiment weight slots: column x2 doubles the bins, shared x0.5 restores them; deterministic fill reproduces its bins exactly\n");
    }

    // ---- DragRace (ref: Fitters/FitterBase.cpp:461-520)
    MR2T2 fitter(1234);
    fitter.AddSampleHandler(&sample);
    fitter.AddSystObj(&ph);
    ph.SetStepScale(0.5 / std::sqrt(double(nPar)));
    double t_rw = 0, t_llh = 0;
    fitter.DragRace(std::max(NLaps, 2000), &t_rw, &t_llh);  // also the spin-up: a fresh process shows one-off 50-100 ms stalls
    fitter.DragRace(NLaps, &t_rw, &t_llh);
    std::printf("DragRace: %d laps, Reweight() launch %.1f us/lap, GetLikelihood() (incl. wait) %.1f us/lap -> %.0f steps/s\n", NLaps,
                1e6 * t_rw, 1e6 * t_llh, 1.0 / (t_rw + t_llh));

    // ---- MR2T2 chain
    double t0c = now_s();
    fitter.RunMCMC(NSteps);
    double el = now_s() - t0c;
    const double acc_plain = double(fitter.GetAccCount()) / fitter.GetStepCount();
    std::printf("MCMC: %d steps in %.3f s = %.0f steps/s, acceptance %.3f, %d out-of-bounds proposals skipped, final -logL %.4f\n", NSteps, el,
                NSteps / el, acc_plain, fitter.GetOutOfBoundsCount(), fitter.GetLogLCurr());
    if (fitter.GetAccCount() == 0 || fitter.GetAccCount() == NSteps) throw MaCh3Exception(__FILE__, __LINE__, "degenerate acceptance rate");

    // ---- adaptive proposal: the throw matrix follows the chain's own covariance
    ph.InitialiseAdaption(0, 100 * NSteps, NSteps / 4, std::max(50, NSteps / 10));
    t0c = now_s();
    fitter.RunMCMC(NSteps);
    el = now_s() - t0c;
    std::printf("MCMC + adaptive proposal: %d more steps, %.0f steps/s, cumulative acceptance %.3f\n", NSteps, NSteps / el,
                double(fitter.GetAccCount()) / fitter.GetStepCount());

    // ---- delayed rejection on the same sample and parameters
    DelayedMR2T2 delayed(99, 1.0, 0.5, 2, 0.1);
    delayed.AddSampleHandler(&sample);
    delayed.AddSystObj(&ph);
    t0c = now_s();
    delayed.RunMCMC(NSteps / 2);
    el = now_s() - t0c;
    std::printf("DelayedMR2T2: %d steps, %.0f steps/s, acceptance %.3f (%d accepted at a delayed stage)\n", NSteps / 2, (NSteps / 2) / el,
                double(delayed.GetAccCount()) / delayed.GetStepCount(), delayed.GetDelayedAcceptCount());

    // ---- likelihood scans (ref: FitterBase::RunLLHScan, Fitters/FitterBase.cpp:622-830): every parameter but the first three
    // skipped; each point against a step made by hand, the per-sample split against the total, the scanned parameter put back
    {
      for (int i = 0; i < nPar; ++i) setpar(i, nominal[i]);
      std::vector<std::pair<int, int>> skip;
      for (int i = 3; i < nPar; ++i) skip.emplace_back(0, i);
      const auto scans = fitter.RunLLHScan(9, 1.0, /*BySample=*/true, skip);
      if (scans.size() != 3) throw MaCh3Exception(__FILE__, __LINE__, "RunLLHScan: expected three scans");
      for (const auto& sc : scans) {
        if (ph.GetParProp(sc.iParam) != sc.CentralValue) throw MaCh3Exception(__FILE__, __LINE__, "RunLLHScan left a parameter moved");
        for (size_t j = 0; j < sc.x.size(); ++j) {
          ph.SetParProp(sc.iParam, sc.x[j]);
          sample.Reweight();
          const double l = sample.GetLikelihood(), pr = ph.GetLikelihood();
          const double s0 = sample.GetSampleLikelihood(0), s1 = sample.GetSampleLikelihood(1);
          auto close = [](double a, double b) { return std::fabs(a - b) <= 1e-9 * std::max(1.0, std::fabs(b)); };
          if (!close(sc.sam[j], 2 * l) || !close(sc.cov[0][j], 2 * pr) || !close(sc.full[j], 2 * (l + pr)) || !close(sc.sample[0][j], 2 * l) ||
              !close(sc.split[0][j], 2 * s0) || !close(sc.split[1][j], 2 * s1) || !close(sc.split[0][j] + sc.split[1][j], sc.sam[j]))
            throw MaCh3Exception(__FILE__, __LINE__, "RunLLHScan: a scan point differs from the step made by hand");
        }
        ph.SetParProp(sc.iParam, sc.CentralValue);
        // Asimov data at nominal: the sample term has its minimum (zero) at the central value, in the middle of the scan
        const size_t mid = sc.x.size() / 2;
        if (!(sc.sam[mid] < 1e-6) || !(sc.sam.front() > sc.sam[mid]) || !(sc.sam.back() > sc.sam[mid]))
          throw MaCh3Exception(__FILE__, __LINE__, "RunLLHScan: the Asimov scan has no minimum at the central value");
      }
      // ... and in two dimensions (ref: FitterBase::Run2DLLHScan, :936-1035): parameters 0 and 2, 5 x 5 points, the diagonal
      // entry of the grid's centre is the Asimov minimum, every entry a step made by hand
      std::vector<std::pair<int, int>> skip2;
      for (int i = 0; i < nPar; ++i)
        if (i != 0 && i != 2) skip2.emplace_back(0, i);
      const auto scans2 = fitter.Run2DLLHScan(5, 1.0, skip2);
      if (scans2.size() != 1 || scans2[0].iParamX != 2 || scans2[0].iParamY != 0 || scans2[0].sam.size() != 25)
        throw MaCh3Exception(__FILE__, __LINE__, "Run2DLLHScan: expected one 5 x 5 scan of parameters (2, 0)");
      for (int x = 0; x < 5; ++x)
        for (int y = 0; y < 5; ++y) {
          ph.SetParProp(2, scans2[0].x[size_t(x)]);
          ph.SetParProp(0, scans2[0].y[size_t(y)]);
          sample.Reweight();
          const double l = 2 * sample.GetLikelihood();
          if (std::fabs(l - scans2[0].sam[size_t(x) * 5 + size_t(y)]) > 1e-9 * std::max(1.0, l)) throw MaCh3Exception(__FILE__, __LINE__, "Run2DLLHScan: a grid point differs from the step made by hand");
        }
      ph.SetParProp(2, nominal[2]);
      ph.SetParProp(0, nominal[0]);
      if (!(scans2[0].sam[12] < 1e-6) || !(scans2[0].sam[0] > scans2[0].sam[12])) throw MaCh3Exception(__FILE__, __LINE__, "Run2DLLHScan: no Asimov minimum at the grid's centre");
      sample.Reweight();
      std::printf("LLH scan: 3 parameters x 9 points equal steps made by hand; per-sample split adds up; Asimov minimum at the central value (2LLH at -1 sigma: %.3f); 2-D scan 5 x 5 equal (corner %.3f)\n",
                  scans[0].full.front(), scans2[0].sam[0]);
    }

    // ---- functional parameter: E_rec *= 1 + theta, events re-binned on the device every step
    sample.EnableDeviceBinning();  // also checks device FindGlobalBin == host FindGlobalBin for every event
    sample.RegisterFunctionalParameter(-1, 0, 1, iEScale);
    MR2T2 fitter_fp(777);  // a fresh chain from the current point
    fitter_fp.AddSampleHandler(&sample);
    fitter_fp.AddSystObj(&ph);
    t0c = now_s();
    fitter_fp.RunMCMC(NSteps / 2);
    const double el_fp = now_s() - t0c;
    std::printf("MCMC + energy-scale functional parameter (device re-binning): %d steps, %.0f steps/s, acceptance %.3f\n", NSteps / 2,
                (NSteps / 2) / el_fp, double(fitter_fp.GetAccCount()) / fitter_fp.GetStepCount());
  } catch (const std::exception& e) {
    std::fprintf(stderr, "FAILED: %s\n", e.what());
    return 1;
  }
  std::printf("OK\n");
  return 0;
}
