// Fitters.h -- a ROOT-free step driver for the engine: the callers of the hot
// path, reduced to what a Markov chain needs.  Host-only, O(nPar^2) per step.
//
//   ParameterHandlerBase   Parameters/ParameterHandlerBase.{h,cpp}: ProposeStep (:652-664) =
//                          Randomize (:690-707) + CorrelateSteps (:726-745), AcceptStep (:748-761),
//                          CalcLikelihood (:816-841), CheckBounds / GetLikelihood (:844-867),
//                          UpdateAdaptiveCovariance (:1306-1345) with the Haario update of
//                          Parameters/AdaptiveMCMCHandler.cpp:332-397
//   MCMCBase / MR2T2       Fitters/MCMCBase.cpp:32-216, Fitters/MR2T2.cpp:11-115
//   DelayedMR2T2           Fitters/DelayedMR2T2.cpp:95-235 (delayed rejection, DRAM acceptance)
//
// FitterBase::RunLLHScan / Run2DLLHScan (Fitters/FitterBase.cpp:559-830, 936-1035) as MCMCBase methods: the scans returned as vectors.
// Also here: the two special proposals (circular bounds and parameter flips, EnableCircularBounds / EnableFlip), fixed
// parameters, flat priors, simulated annealing.
// Not here: YAML configuration, ROOT output trees (steps are kept in memory), PCA, tunes, Robbins-Monro scaling,
// block-wise adaption, multi-chain tools.
#pragma once

#include <algorithm>
#include <chrono>
#include <cmath>
#include <random>
#include <string>
#include <vector>

#include "Core.h"
#include "SampleHandlerFD.h"

// OpenMP is optional for the host driver: with -fopenmp the quadratic loops are vectorised and the Haario update runs on a
// small team (SetAdaptionThreads; at most 8 -- a team of all of a 256-thread host costs more in its fork than the loop saves)
#if defined(_OPENMP)
#include <omp.h>
#define M3_OMP_STR(x) #x
#define M3_OMP(x) _Pragma(M3_OMP_STR(x))
#else
#define M3_OMP(x)
#endif

class ParameterHandlerBase {
 public:
  /// prior central values, prior covariance (row-major nPar x nPar), bounds; flat_prior[i] drops
  /// parameter i from the prior term (ref: ParameterHandlerBase.cpp:816-841)
  ParameterHandlerBase(std::string name, const std::vector<double>& prefit, const std::vector<double>& covariance, const std::vector<double>& low,
                       const std::vector<double>& up, const std::vector<bool>& flat_prior = {}, uint64_t seed = 0)
      : matrixName(std::move(name)), _fNumPar(int(prefit.size())), _fPreFitValue(prefit), _fCurrVal(prefit), _fPropVal(prefit), _fLowBound(low),
        _fUpBound(up), _fFlatPrior(flat_prior.empty() ? std::vector<bool>(prefit.size(), false) : flat_prior), _fIndivStepScale(prefit.size(), 1.0),
        fixed(prefit.size(), false), randParams(prefit.size(), 0.0), corr_throw(prefit.size(), 0.0), covMatrix(covariance), rng(seed ? seed : 0x5EEDull) {
    if (int(covariance.size()) != _fNumPar * _fNumPar || int(low.size()) != _fNumPar || int(up.size()) != _fNumPar)
      throw MaCh3Exception(__FILE__, __LINE__, "ParameterHandlerBase: inconsistent sizes");
    InvertCovMatrix = Invert(covMatrix, _fNumPar);
    SetThrowMatrix(covMatrix);
    _fGlobalStepScale = 2.38 / std::sqrt(double(_fNumPar));  // the usual optimal scale for a Gaussian target
  }
  virtual ~ParameterHandlerBase() {}
  // An attached handler owns a context on the engine (AttachDevice: dev).  A copy would share it -- detaching either one
  // frees the slot under the other, and the engine may hand it to another handler -- so there are no copies; a move takes the
  // attachment along and leaves the source detached.  (Samples hold RetPointer() addresses: they stay valid across a move, the
  // vectors' buffers move with the object.)
  ParameterHandlerBase(const ParameterHandlerBase&) = delete;
  ParameterHandlerBase& operator=(const ParameterHandlerBase&) = delete;
  ParameterHandlerBase(ParameterHandlerBase&& o) = default;
  ParameterHandlerBase& operator=(ParameterHandlerBase&&) = delete;

  std::string GetName() const { return matrixName; }
  int GetNumParams() const { return _fNumPar; }
  /// address of the PROPOSED value: what samples read at Reweight() (ref: ParameterHandlerBase.h:205)
  const double* RetPointer(const int iParam) { return &_fPropVal[size_t(iParam)]; }
  const std::vector<double>& GetParCurrVec() const { return _fCurrVal; }
  const std::vector<double>& GetParPropVec() const { return _fPropVal; }
  double GetParProp(int i) const { return _fPropVal[size_t(i)]; }
  double GetParInit(int i) const { return _fPreFitValue[size_t(i)]; }
  void SetParProp(const int i, const double val) { _fPropVal[size_t(i)] = val; }
  double GetDiagonalError(const int i) const { return std::sqrt(covMatrix[size_t(i) * size_t(_fNumPar) + size_t(i)]); }
  double GetLowerBound(const int i) const { return _fLowBound[size_t(i)]; }
  double GetUpperBound(const int i) const { return _fUpBound[size_t(i)]; }
  double GetParCurr(int i) const { return _fCurrVal[size_t(i)]; }
  void SetParCurrProp(const int parNo, const double parVal) { _fPropVal[size_t(parNo)] = _fCurrVal[size_t(parNo)] = parVal; }
  /// all proposed values at once (ref: ParameterHandlerBase::SetParameters, Parameters/ParameterHandlerBase.cpp:871-890)
  void SetParameters(const std::vector<double>& pars) {
    if (int(pars.size()) != _fNumPar) throw MaCh3Exception(__FILE__, __LINE__, "SetParameters: " + std::to_string(pars.size()) + " values for " + std::to_string(_fNumPar) + " parameters");
    _fPropVal = pars;
  }
  void SetStepScale(const double scale, bool = false) { _fGlobalStepScale = scale; }
  double GetGlobalStepScale() const { return _fGlobalStepScale; }
  void SetIndivStepScale(int i, double s) { _fIndivStepScale[size_t(i)] = s; }
  bool GetDoAdaption() const { return use_adaptive; }

  void SetSingleParameter(const int parNo, const double parVal) { SetParCurrProp(parNo, parVal); }  // ref: :629-636
  /// flat prior: the parameter drops out of the penalty term (ref: :1063-1080)
  void SetFlatPrior(const int i, const bool eL) {
    if (i < 0 || i >= _fNumPar) throw MaCh3Exception(__FILE__, __LINE__, "SetFlatPrior: parameter index out of range");
    _fFlatPrior[size_t(i)] = eL;
  }
  /// fixed parameters keep their value: no random throw, no correlated step (ref: :690-745, 960-1029; the reference
  /// flags them through the sign of _fError)
  bool IsParameterFixed(const int i) const { return fixed[size_t(i)]; }
  void ToggleFixParameter(const int i) {
    throw_ready = throw_drawn = false;
    ++throw_epoch;
    if (i < 0 || i >= _fNumPar) throw MaCh3Exception(__FILE__, __LINE__, "ToggleFixParameter: parameter index out of range");
    fixed[size_t(i)] = !fixed[size_t(i)];
  }
  void SetFixParameter(const int i) { if (!IsParameterFixed(i)) ToggleFixParameter(i); }
  void SetFreeParameter(const int i) { if (IsParameterFixed(i)) ToggleFixParameter(i); }
  void SetFixAllParameters() { for (int i = 0; i < _fNumPar; ++i) SetFixParameter(i); }
  void SetFreeAllParameters() { for (int i = 0; i < _fNumPar; ++i) SetFreeParameter(i); }
  /// the two special proposals of EnableSpecialProposal (ref: :381-428; the YAML keys "CircularBounds" and
  /// "FlipParameter" become two calls): wrap a cyclic parameter into [lo, up], mirror a parameter about a point
  void EnableCircularBounds(const int index, const double LowBound, const double UpBound) {
    if (index < 0 || index >= _fNumPar || !(LowBound < UpBound)) throw MaCh3Exception(__FILE__, __LINE__, "EnableCircularBounds: bad arguments");
    // ref: :407-415 -- outside the physical bounds the chain never goes, so the circular range must lie inside them
    if (LowBound < _fLowBound[size_t(index)] || UpBound > _fUpBound[size_t(index)])
      throw MaCh3Exception(__FILE__, __LINE__, "EnableCircularBounds: the circular range exceeds the parameter's physical bounds");
    doSpecialStepProposal = true;
    CircularBoundsIndex.push_back(index);
    CircularBoundsValues.emplace_back(LowBound, UpBound);
  }
  void EnableFlip(const int index, const double FlipPoint) {
    if (index < 0 || index >= _fNumPar) throw MaCh3Exception(__FILE__, __LINE__, "EnableFlip: parameter index out of range");
    doSpecialStepProposal = true;
    FlipParameterIndex.push_back(index);
    FlipParameterPoint.push_back(FlipPoint);
  }
  void CircularParBounds(const int index, const double LowBound, const double UpBound) {  // ref: :764-772
    double& v = _fPropVal[size_t(index)];
    if (v > UpBound) v = LowBound + std::fmod(v - UpBound, UpBound - LowBound);
    else if (v < LowBound) v = UpBound - std::fmod(LowBound - v, UpBound - LowBound);
  }
  void FlipParameterValue(const int index, const double FlipPoint) {  // ref: :775-780
    if (unif(rng) < 0.5) _fPropVal[size_t(index)] = 2 * FlipPoint - _fPropVal[size_t(index)];
  }
  void SpecialStepProposal() {  // ref: :667-684 "Step -> Circular Bounds -> Flip"
    for (size_t i = 0; i < CircularBoundsIndex.size(); ++i)
      if (!IsParameterFixed(CircularBoundsIndex[i])) CircularParBounds(CircularBoundsIndex[i], CircularBoundsValues[i].first, CircularBoundsValues[i].second);
    for (size_t i = 0; i < FlipParameterIndex.size(); ++i)
      if (!IsParameterFixed(FlipParameterIndex[i])) FlipParameterValue(FlipParameterIndex[i], FlipParameterPoint[i]);
  }
  /// start the chain somewhere else: every free parameter thrown within one prior sigma (or its range, if narrower)
  /// of its prior value, inside its bounds (ref: :591-626)
  void RandomConfiguration() {
    for (int i = 0; i < _fNumPar; ++i) {
      if (IsParameterFixed(i)) continue;
      const double paramrange = _fUpBound[size_t(i)] - _fLowBound[size_t(i)];
      const double sigma = std::sqrt(covMatrix[size_t(i) * _fNumPar + i]);
      const double throwrange = paramrange < sigma ? paramrange : sigma;
      int throws = 0;
      do {
        if (throws++ > 1000) throw MaCh3Exception(__FILE__, __LINE__, "RandomConfiguration: parameter " + std::to_string(i) + " cannot be thrown inside its bounds");
        _fPropVal[size_t(i)] = _fPreFitValue[size_t(i)] + gaus(rng) * throwrange;
      } while (_fPropVal[size_t(i)] > _fUpBound[size_t(i)] || _fPropVal[size_t(i)] < _fLowBound[size_t(i)]);
      _fCurrVal[size_t(i)] = _fPropVal[size_t(i)];
    }
    SpecialStepProposal();
  }

  /// The two O(nPar^2) sweeps of a step -- the prior term (CalcLikelihood, ref: :816-841) and the next proposal's correlated
  /// throw (CorrelateSteps, ref: :726-745) -- on the engine's device (m3hip_par_*), beside the sample's reweight kernel.
  /// The host keeps the random numbers (Randomize, :690-707), the bounds, the special proposals and the Haario update.
  /// At the 1 040 parameters of BASELINE configs[4] the sweeps take 0.4 ms on one host core and a few microseconds there.
  /// Same numbers up to fp64 rounding (fixed summation order on the device; tests/test_cpp_facade.py).
  /// Every attached handler owns a context of its own on the engine (m3hip_par_create): several handlers of one fit -- MCMCBase
  /// loops over all systematics objects -- may share an engine, with equal or different numbers of parameters.
  void AttachDevice(m3hip_handle h) {
    if (!h) throw MaCh3Exception(__FILE__, __LINE__, "AttachDevice: null engine handle");
    DetachDevice();
    int slot = -1;
    if (m3hip_par_create(h, &slot) != M3HIP_OK) throw MaCh3Exception(__FILE__, __LINE__, std::string("AttachDevice: ") + m3hip_last_error(h));
    if (m3hip_par_upload(h, slot, 0, InvertCovMatrix.data(), _fNumPar) != M3HIP_OK ||
        m3hip_par_upload(h, slot, 1, throwMatrixCholDecomp.data(), _fNumPar) != M3HIP_OK) {
      const std::string why = m3hip_last_error(h);
      m3hip_par_destroy(h, slot);
      throw MaCh3Exception(__FILE__, __LINE__, "AttachDevice: " + why);
    }
    dev = h;
    dev.slot = slot;
  }
  /// back to the host loops (call before the engine behind the handle is destroyed)
  void DetachDevice() {
    if (dev) m3hip_par_destroy(dev, dev.slot);
    dev = nullptr;
    dev.slot = -1;
  }
  bool OnDevice() const { return dev != nullptr; }
  /// enqueue (nothing waits): the prior of the CURRENT proposal and the correlated throw of the NEXT one
  void LaunchDeviceStep() {
    std::vector<double>& d = diff_scratch;
    d.resize(size_t(_fNumPar));
    for (int i = 0; i < _fNumPar; ++i) d[size_t(i)] = _fFlatPrior[size_t(i)] ? 0.0 : _fPropVal[size_t(i)] - _fPreFitValue[size_t(i)];
    if (!throw_drawn) {
      for (int i = 0; i < _fNumPar; ++i) randParams[size_t(i)] = IsParameterFixed(i) ? 0.0 : gaus(rng);
      throw_drawn = true;
    }
    if (m3hip_par_launch(dev, dev.slot, d.data(), randParams.data(), _fNumPar) != M3HIP_OK)
      throw MaCh3Exception(__FILE__, __LINE__, std::string("LaunchDeviceStep: ") + m3hip_last_error(dev));
    launched_epoch = throw_epoch;
  }
  /// the prior term of the proposal LaunchDeviceStep was called for; the next proposal's throw is in place, unless the
  /// throw matrix or the set of fixed parameters changed since (ProposeStep then recomputes it on the host, same random vector)
  double CollectDeviceStep() {
    double prior = 0.0;
    std::vector<double>& t = throw_scratch;
    t.resize(size_t(_fNumPar));
    if (m3hip_par_result(dev, dev.slot, &prior, t.data(), _fNumPar) != M3HIP_OK)
      throw MaCh3Exception(__FILE__, __LINE__, std::string("CollectDeviceStep: ") + m3hip_last_error(dev));
    if (launched_epoch == throw_epoch && throw_drawn) {
      corr_throw.swap(t);
      throw_ready = true;
    }
    return prior;
  }
  /// the lower Cholesky factor of the throw matrix, row-major (what CorrelateSteps multiplies with)
  const std::vector<double>& GetThrowMatrixCholesky() const { return throwMatrixCholDecomp; }
  /// threads of the Haario update's team (with -fopenmp; 1 = serial)
  void SetAdaptionThreads(int n) { adaption_threads = std::max(1, std::min(n, 8)); }

  /// ref: ParameterHandlerBase.cpp:652-664 (ProposeStep), 690-707 (Randomize), 726-745 (CorrelateSteps)
  void ProposeStep() {
    if (!throw_ready) PrepareNextThrow();
    throw_ready = throw_drawn = false;
    for (int i = 0; i < _fNumPar; ++i)
      if (!IsParameterFixed(i)) _fPropVal[size_t(i)] = _fCurrVal[size_t(i)] + corr_throw[size_t(i)] * _fGlobalStepScale * _fIndivStepScale[size_t(i)];
    if (doSpecialStepProposal) SpecialStepProposal();
  }
  /// The correlated throw of the NEXT proposal: Randomize (:690-707) + CorrelateSteps (:726-745).  It depends on the throw
  /// matrix only, not on where the chain stands, so a fitter may compute it ahead of time under the GPU's work on the
  /// current step (MCMCBase::EvaluateProposal); ProposeStep computes it itself when nobody did, and again (same random
  /// vector) when the throw matrix or the set of fixed parameters changed in between.
  void PrepareNextThrow() {
    if (!throw_drawn) {
      for (int i = 0; i < _fNumPar; ++i) randParams[size_t(i)] = IsParameterFixed(i) ? 0.0 : gaus(rng);
      throw_drawn = true;
    }
    // corr_throw = L z, L lower triangular.  O(nPar^2) per step: at the 1 040 parameters of BASELINE configs[4] the three
    // quadratic loops of this class (this one, the prior, the Haario update) would cost more than the GPU step if left
    // scalar and serial to it, so their inner loops are contiguous row sweeps the compiler vectorises (`omp simd` with
    // -fopenmp, plain loops without), and the fitter runs all three while the device works.
    const double* L = throwMatrixCholDecomp.data();
    const double* z = randParams.data();
    for (int i = 0; i < _fNumPar; ++i) {
      const double* row = L + size_t(i) * _fNumPar;
      double s = 0.0;
M3_OMP(omp simd reduction(+ : s))
      for (int j = 0; j <= i; ++j) s += row[j] * z[j];
      corr_throw[size_t(i)] = s;
    }
    throw_ready = true;
  }
  void AcceptStep() {  // ref: :748-761
    FinishPendingAdaption();  // the pending Haario update belongs to the point the chain is about to leave
    _fCurrVal = _fPropVal;
  }

  /// ref: :816-841 -- 0.5 * d^T V^-1 d over the parameters with a Gaussian prior
  double CalcLikelihood() const {
    // sum over i, j <= i of (i == j ? 0.5 : 1) d_i d_j V^-1_ij with d = 0 for flat priors, one row sweep per parameter
    std::vector<double>& d = diff_scratch;
    d.resize(size_t(_fNumPar));
    for (int i = 0; i < _fNumPar; ++i) d[size_t(i)] = _fFlatPrior[size_t(i)] ? 0.0 : _fPropVal[size_t(i)] - _fPreFitValue[size_t(i)];
    const double* dv = d.data();
    double logL = 0.0;
    for (int i = 0; i < _fNumPar; ++i) {
      if (_fFlatPrior[size_t(i)]) continue;
      const double* row = InvertCovMatrix.data() + size_t(i) * _fNumPar;
      double s = 0.0;
M3_OMP(omp simd reduction(+ : s))
      for (int j = 0; j < i; ++j) s += dv[j] * row[j];
      logL += dv[i] * (s + 0.5 * dv[i] * row[i]);
    }
    return logL;
  }
  int CheckBounds() const {  // ref: :844-856
    int NOutside = 0;
    for (int i = 0; i < _fNumPar; ++i)
      if (_fPropVal[size_t(i)] > _fUpBound[size_t(i)] || _fPropVal[size_t(i)] < _fLowBound[size_t(i)]) NOutside++;
    return NOutside;
  }
  double GetLikelihood() {  // ref: :859-867
    const int NOutside = CheckBounds();
    if (NOutside > 0) return NOutside * M3::_LARGE_LOGL_;
    return CalcLikelihood();
  }

  /// Adaptive MCMC: between start_update and end_update every step feeds the running mean and
  /// covariance of the chain; from start_throw on, every update_step steps the proposal's throw
  /// matrix becomes that covariance (ref: AdaptiveMCMCHandler.cpp:398-432, ParameterHandlerBase.cpp:1306-1345)
  void InitialiseAdaption(int start_update, int end_update, int start_throw, int update_step) {
    use_adaptive = true;
    start_adaptive_update = start_update;
    end_adaptive_update = end_update;
    start_adaptive_throw = start_throw;
    adaptive_update_step = update_step;
    total_steps = 0;
    par_means.assign(size_t(_fNumPar), 0.0);
    adaptive_covariance.assign(size_t(_fNumPar) * _fNumPar, 0.0);
  }
  void UpdateAdaptiveCovariance() {
    FinishPendingAdaption();  // (a driver that did not finish the previous step's update under its GPU work)
    if (total_steps > end_adaptive_update || total_steps < start_adaptive_update) {  // SkipAdaption
      ++total_steps;
      return;
    }
    // The running mean / covariance only matter to the chain when they are turned into a new throw matrix.  On the steps
    // in between, the O(nPar^2) update is left pending: the fitter finishes it under the next step's GPU work
    // (FinishPendingAdaption); AcceptStep -- the next thing that would change the current values -- finishes it itself if
    // nobody did.  Same sequence of updates, same numbers.
    const bool refresh = total_steps >= start_adaptive_throw && (total_steps - start_adaptive_throw) % adaptive_update_step == 0;
    adaption_pending = true;
    if (refresh || total_steps == start_adaptive_throw) FinishPendingAdaption();
  }
  void FinishPendingAdaption() {
    if (!adaption_pending) return;
    adaption_pending = false;
    // Haario update (ref: AdaptiveMCMCHandler.cpp:332-397), no blocks / skipped parameters.  Only the lower triangle is kept
    // (SetThrowMatrix reads nothing else; GetAdaptiveCovariance mirrors it): the matrix is 8.7 MB at 1 040 parameters, and a
    // mirrored write per element would stride through all of it.
    par_means_prev = par_means;
    const int steps_post_burn = total_steps - start_adaptive_update;
    for (int i = 0; i < _fNumPar; ++i) par_means[size_t(i)] = (_fCurrVal[size_t(i)] + par_means_prev[size_t(i)] * steps_post_burn) / (steps_post_burn + 1);
    const int n = _fNumPar;
    double* cov = adaptive_covariance.data();
    const double* mp = par_means_prev.data();
    const double* mc = par_means.data();
    const double* x = _fCurrVal.data();
    if (steps_post_burn > 0) {
      const double keep = double(steps_post_burn - 1) / steps_post_burn, nb = double(steps_post_burn), nb1 = double(steps_post_burn + 1);
      // rows are independent; chunks of 8 rows dealt round-robin balance the triangle over the team
M3_OMP(omp parallel for schedule(static, 8) num_threads(adaption_threads) if (n >= 256 && adaption_threads > 1))
      for (int i = 0; i < n; ++i) {
        double* row = cov + size_t(i) * n;
        const double pi = nb * mp[i], ci = nb1 * mc[i], xi = x[i];
M3_OMP(omp simd)
        for (int j = 0; j <= i; ++j) row[j] = row[j] * keep + (pi * mp[j] - ci * mc[j] + xi * x[j]) / nb;
      }
    } else {
      for (int i = 0; i < n; ++i) std::fill(cov + size_t(i) * n, cov + size_t(i) * n + i + 1, 0.0);
    }
    if (total_steps == start_adaptive_throw) std::fill(_fIndivStepScale.begin(), _fIndivStepScale.end(), 1.0);  // IndivStepScaleAdapt
    if (total_steps >= start_adaptive_throw && (total_steps - start_adaptive_throw) % adaptive_update_step == 0) {  // UpdateMatrixAdapt
      try {
        SetThrowMatrix(adaptive_covariance);
      } catch (const MaCh3Exception&) {
        // could not be made positive definite: keep the previous throw matrix
      }
    }
    ++total_steps;
  }
  /// the running covariance as a full symmetric matrix (only its lower triangle is maintained)
  std::vector<double> GetAdaptiveCovariance() {
    FinishPendingAdaption();
    std::vector<double> full(adaptive_covariance);
    for (int i = 0; i < _fNumPar; ++i)
      for (int j = 0; j < i; ++j) full[size_t(j) * _fNumPar + i] = full[size_t(i) * _fNumPar + j];
    return full;
  }

  /// throw matrix -> its lower Cholesky factor (ref: SetThrowMatrix, ParameterHandlerBase.cpp:1140-1200).  A matrix that
  /// cannot be decomposed gets its diagonal shifted until it can, as M3::MakeMatrixPosDef does
  /// (Parameters/ParameterHandlerUtils.h:512-545; an adaptive covariance from fewer steps than parameters is singular):
  /// the reference's steps of 1e-9 for the first 64 attempts, then doubling, up to the reference's total of 1e-4
  /// (1e5 steps of 1e-9 there; at 1 040 parameters one attempt is an O(n^3) factorisation of ~0.2 s, so the linear walk
  /// could take hours).  A non-finite entry can never be repaired by a shift: refused at once.
  void SetThrowMatrix(const std::vector<double>& cov) {
    std::vector<double> L(size_t(_fNumPar) * _fNumPar, 0.0);
    for (int i = 0; i < _fNumPar; ++i)
      for (int j = 0; j <= i; ++j)
        if (!std::isfinite(cov[size_t(i) * _fNumPar + j])) throw MaCh3Exception(__FILE__, __LINE__, "throw matrix has a non-finite entry");
    constexpr double MaxShift = 1e-4;
    double shift = 0.0;
    const int n = _fNumPar;
    // L_ij = (A_ij - sum_{k<j} L_ik L_jk) / L_jj in panels of 32 columns.  One factorisation of 1 040 parameters is 0.37 GFLOP
    // -- 35 ms as a plain row sweep on one core -- and it sits on the step's critical path whenever the adaptive proposal
    // refreshes its matrix, so the team of the Haario update shares it: (1) every row's sums over the columns left of the
    // panel, the bulk of the work, rows independent; (2a) the panel's diagonal block, one thread; (2b) the rows below it,
    // independent again.  Three barriers per panel.  Every entry is the same sum in the same order whoever computes it: the
    // factor does not depend on the number of threads.
    constexpr int kPanel = 32;
    for (int attempt = 0;; ++attempt, shift = attempt <= 64 ? shift + 1e-9 : 2 * shift) {
      bool ok = true;
M3_OMP(omp parallel num_threads(adaption_threads) if (n >= 256 && adaption_threads > 1))
      for (int j0 = 0; j0 < n; j0 += kPanel) {
        if (!ok) continue;  // (every thread sees the same `ok`: it only changes between two barriers)
        const int j1 = std::min(n, j0 + kPanel);
M3_OMP(omp for schedule(static, 4))
        for (int i = j0; i < n; ++i) {  // (1)
          double* li = L.data() + size_t(i) * n;
          const int jmax = std::min(i, j1 - 1);
          for (int j = j0; j <= jmax; ++j) {
            const double* lj = L.data() + size_t(j) * n;
            double dot = 0.0;
M3_OMP(omp simd reduction(+ : dot))
            for (int k = 0; k < j0; ++k) dot += li[k] * lj[k];
            li[j] = cov[size_t(i) * n + j] + (i == j ? shift : 0.0) - dot;
          }
        }
M3_OMP(omp single)
        for (int i = j0; i < j1 && ok; ++i) {  // (2a)
          double* li = L.data() + size_t(i) * n;
          for (int j = j0; j <= i; ++j) {
            const double* lj = L.data() + size_t(j) * n;
            double s = li[j];
            for (int k = j0; k < j; ++k) s -= li[k] * lj[k];
            if (i == j) {
              if (!(s > 0.0)) {
                ok = false;
                break;
              }
              li[i] = std::sqrt(s);
            } else {
              li[j] = s / lj[j];
            }
          }
        }
        if (ok) {
M3_OMP(omp for schedule(static, 4))
          for (int i = j1; i < n; ++i) {  // (2b)
            double* li = L.data() + size_t(i) * n;
            for (int j = j0; j < j1; ++j) {
              const double* lj = L.data() + size_t(j) * n;
              double s = li[j];
              for (int k = j0; k < j; ++k) s -= li[k] * lj[k];
              li[j] = s / lj[j];
            }
          }
        }
      }
      if (ok) break;
      if (shift >= MaxShift) throw MaCh3Exception(__FILE__, __LINE__, "throw matrix is not positive definite");
      for (int i = 0; i < n; ++i) std::fill(L.begin() + size_t(i) * n, L.begin() + size_t(i) * n + i + 1, 0.0);
    }
    throwMatrixCholDecomp = L;
    throw_ready = false;  // a throw prepared with the old matrix is recomputed (same random vector)
    ++throw_epoch;
    if (dev && m3hip_par_upload(dev, dev.slot, 1, throwMatrixCholDecomp.data(), _fNumPar) != M3HIP_OK)
      throw MaCh3Exception(__FILE__, __LINE__, std::string("SetThrowMatrix: ") + m3hip_last_error(dev));
  }

 private:
  static std::vector<double> Invert(const std::vector<double>& a, int n) {  // Gauss-Jordan with partial pivoting
    std::vector<double> m(a), inv(size_t(n) * n, 0.0);
    for (int i = 0; i < n; ++i) inv[size_t(i) * n + i] = 1.0;
    for (int c = 0; c < n; ++c) {
      int p = c;
      for (int r = c + 1; r < n; ++r)
        if (std::fabs(m[size_t(r) * n + c]) > std::fabs(m[size_t(p) * n + c])) p = r;
      if (std::fabs(m[size_t(p) * n + c]) < 1e-300) throw MaCh3Exception(__FILE__, __LINE__, "singular covariance matrix");
      for (int k = 0; k < n; ++k) {
        std::swap(m[size_t(c) * n + k], m[size_t(p) * n + k]);
        std::swap(inv[size_t(c) * n + k], inv[size_t(p) * n + k]);
      }
      const double d = m[size_t(c) * n + c];
      for (int k = 0; k < n; ++k) { m[size_t(c) * n + k] /= d; inv[size_t(c) * n + k] /= d; }
      for (int r = 0; r < n; ++r) {
        if (r == c) continue;
        const double f = m[size_t(r) * n + c];
        if (f == 0.0) continue;
        for (int k = 0; k < n; ++k) { m[size_t(r) * n + k] -= f * m[size_t(c) * n + k]; inv[size_t(r) * n + k] -= f * inv[size_t(c) * n + k]; }
      }
    }
    // entries of an inverse that decay geometrically away from the diagonal end up denormal, and every multiplication with a
    // denormal costs ~100 cycles: the prior's row sweeps then take ten times longer for nothing
    for (double& v : inv)
      if (std::fabs(v) < 1e-290) v = 0.0;
    return inv;
  }

  std::string matrixName;
  int _fNumPar;
  std::vector<double> _fPreFitValue, _fCurrVal, _fPropVal, _fLowBound, _fUpBound;
  std::vector<bool> _fFlatPrior;
  std::vector<double> _fIndivStepScale;
  std::vector<bool> fixed;
  bool doSpecialStepProposal = false;
  std::vector<int> CircularBoundsIndex, FlipParameterIndex;
  std::vector<std::pair<double, double>> CircularBoundsValues;
  std::vector<double> FlipParameterPoint;
  double _fGlobalStepScale = 1.0;
  std::vector<double> randParams, corr_throw, covMatrix, InvertCovMatrix, throwMatrixCholDecomp;
  std::mt19937_64 rng;
  std::normal_distribution<double> gaus{0.0, 1.0};
  std::uniform_real_distribution<double> unif{0.0, 1.0};
  // adaption
  bool use_adaptive = false;
  int start_adaptive_update = 0, end_adaptive_update = 0, start_adaptive_throw = 0, adaptive_update_step = 1, total_steps = 0;
  std::vector<double> par_means, par_means_prev, adaptive_covariance;
  mutable std::vector<double> diff_scratch;
  std::vector<double> throw_scratch;
  bool adaption_pending = false, throw_ready = false, throw_drawn = false;
  // device side (AttachDevice): the engine and this handler's context on it (m3hip_par_create); moved-from = detached
  struct DeviceSlot {
    m3hip_handle h = nullptr;
    int slot = -1;
    DeviceSlot() = default;
    DeviceSlot(DeviceSlot&& o) noexcept : h(o.h), slot(o.slot) { o.h = nullptr; o.slot = -1; }
    DeviceSlot& operator=(DeviceSlot&& o) noexcept { h = o.h; slot = o.slot; o.h = nullptr; o.slot = -1; return *this; }
    DeviceSlot& operator=(m3hip_handle x) { h = x; return *this; }
    operator m3hip_handle() const { return h; }
  } dev;
  uint64_t throw_epoch = 0, launched_epoch = 0;  // bumped whenever a throw in flight would be stale
#if defined(_OPENMP)
  int adaption_threads = std::max(1, std::min(8, omp_get_max_threads()));
#else
  int adaption_threads = 1;
#endif
};

/// ref: Fitters/FitterBase.h:36-61 + Fitters/MCMCBase.cpp:32-216 (chain bookkeeping kept in memory)
class MCMCBase {
 public:
  explicit MCMCBase(uint64_t seed = 12345) : random(seed) {}
  virtual ~MCMCBase() {}
  /// non-owning; a sub-sample title may be registered once only (ref: Fitters/FitterBase.cpp:262-294)
  void AddSampleHandler(SampleHandlerBase* sample) {
    if (!sample) throw MaCh3Exception(__FILE__, __LINE__, "AddSampleHandler: null handler");
    for (SampleHandlerBase* s : samples)
      for (int i = 0; i < s->GetNsamples(); ++i)
        for (int j = 0; j < sample->GetNsamples(); ++j)
          if (s->GetSampleTitle(i) == sample->GetSampleTitle(j))
            throw MaCh3Exception(__FILE__, __LINE__, "Duplicate sample title '" + sample->GetSampleTitle(j) + "' in handler " + sample->GetName() +
                                                         ": same title exists in handler " + s->GetName());
    samples.push_back(sample);
  }
  /// simulated annealing: the likelihood difference is divided by exp(-step / T); T < 0 switches it off
  /// (ref: Fitters/MCMCBase.cpp:19-26, the "AnnealTemp" setting)
  void SetAnnealTemp(const double T) { AnnealTemp = T; anneal = !(T < 0); }
  void AddSystObj(ParameterHandlerBase* cov) { systematics.push_back(cov); }

  /// ref: MCMCBase.cpp:32-77
  void RunMCMC(int n_steps) {
    syst_llh.assign(systematics.size(), 0.0);
    sample_llh.assign(samples.size(), 0.0);
    if (!started) {  // the chain starts from the current values: evaluate them once (FitterBase.cpp:348-410 restart logic is out of scope)
      ProposeStepAtCurrent();
      logLCurr = logLProp;
      started = true;
    }
    for (int i = 0; i < n_steps; ++i) {
      const auto t0 = std::chrono::steady_clock::now();
      const int acc_before = accCount;
      DoStep();
      AccStore.push_back(char(accCount != acc_before));
      for (auto* s : systematics)
        if (s->GetDoAdaption()) s->UpdateAdaptiveCovariance();  // MCMCBase::AdaptiveStep
      ++step;
      stepTime.push_back(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
      LogLStore.push_back(logLCurr);
      for (size_t k = 0; k < systematics.size(); ++k) StepStore.insert(StepStore.end(), systematics[k]->GetParCurrVec().begin(), systematics[k]->GetParCurrVec().end());
    }
  }
  int GetAccCount() const { return accCount; }
  int GetStepCount() const { return step; }
  double GetLogLCurr() const { return logLCurr; }
  int GetOutOfBoundsCount() const { return oobCount; }
  const std::vector<double>& GetStepStore() const { return StepStore; }  ///< [step][all parameters of all handlers]
  const std::vector<double>& GetLogLStore() const { return LogLStore; }
  const std::vector<char>& GetAcceptStore() const { return AccStore; }  ///< [step] 1: the step was accepted
  /// the last proposal's likelihood per sample handler / per systematics object (the reference writes both to its output
  /// tree every step: Fitters/FitterBase.cpp:195-215)
  const std::vector<double>& GetSampleLikelihoods() const { return sample_llh; }
  const std::vector<double>& GetSystLikelihoods() const { return syst_llh; }
  double GetLogLProp() const { return logLProp; }
  const std::vector<double>& GetStepTimes() const { return stepTime; }

  /// One parameter's likelihood scan (ref: FitterBase::RunLLHScan, Fitters/FitterBase.cpp:622-830; what the reference writes into
  /// TH1Ds is returned here): n_points bin centres between CentralValue -+ nSigma * prior error, clipped to the parameter's
  /// bounds (GetParameterScanRange, :559-617); every entry is 2 x the negative log-likelihood at that point.
  struct LLHScanResult {
    int iHandler = 0, iParam = 0;
    double CentralValue = 0, lower = 0, upper = 0;
    std::vector<double> x;                        ///< the scan points (TH1D bin centres)
    std::vector<double> full, sam;                ///< hScan: samples + all systematics objects; hScanSam: the samples
    std::vector<std::vector<double>> sample;      ///< hScanSample: [sample handler][point]
    std::vector<std::vector<double>> cov;         ///< hScanCov: [systematics object][point]
    std::vector<std::vector<double>> split;       ///< hScanSamSplit (LLHScanBySample): [sub-sample over all handlers][point]
  };
  /// Scans every parameter of every systematics object (Skip: {handler, parameter} pairs left out, the reference's
  /// LLHScanSkipVector); all other parameters stay at their proposed values, the scanned one is put back afterwards.
  /// As in the reference the samples are reweighted for every point, all of them before the first likelihood is asked for.
  std::vector<LLHScanResult> RunLLHScan(const int n_points = 100, const double nSigma = 1.0, const bool BySample = false,
                                        const std::vector<std::pair<int, int>>& Skip = {}) {
    if (n_points < 1) throw MaCh3Exception(__FILE__, __LINE__, "RunLLHScan: no points");
    std::vector<LLHScanResult> out;
    for (size_t ic = 0; ic < systematics.size(); ++ic) {
      ParameterHandlerBase* cov = systematics[ic];
      for (int i = 0; i < cov->GetNumParams(); ++i) {
        if (std::find(Skip.begin(), Skip.end(), std::make_pair(int(ic), i)) != Skip.end()) continue;
        LLHScanResult r;
        r.iHandler = int(ic);
        r.iParam = i;
        r.CentralValue = cov->GetParProp(i);
        r.lower = std::max(r.CentralValue - nSigma * cov->GetDiagonalError(i), cov->GetLowerBound(i));
        r.upper = std::min(r.CentralValue + nSigma * cov->GetDiagonalError(i), cov->GetUpperBound(i));
        r.sample.assign(samples.size(), {});
        r.cov.assign(systematics.size(), {});
        size_t n_split = 0;
        if (BySample)
          for (auto* smp : samples) n_split += size_t(smp->GetNsamples());
        r.split.assign(n_split, {});
        for (int j = 0; j < n_points; ++j) {
          const double x = r.lower + (double(j) + 0.5) * (r.upper - r.lower) / double(n_points);  // TH1D::GetBinCenter(j + 1)
          cov->SetParProp(i, x);
          for (auto* smp : samples) smp->Reweight();
          double samplellh = 0.0, totalllh = 0.0;
          for (size_t ivs = 0; ivs < samples.size(); ++ivs) {
            const double l = samples[ivs]->GetLikelihood();
            r.sample[ivs].push_back(2 * l);
            samplellh += l;
          }
          for (size_t ivc = 0; ivc < systematics.size(); ++ivc) {
            const double l = systematics[ivc]->GetLikelihood();
            r.cov[ivc].push_back(2 * l);
            totalllh += l;
          }
          totalllh += samplellh;
          if (BySample) {
            size_t it = 0;
            for (auto* smp : samples)
              for (int is = 0; is < smp->GetNsamples(); ++is) r.split[it++].push_back(2 * smp->GetSampleLikelihood(is));
          }
          r.x.push_back(x);
          r.sam.push_back(2 * samplellh);
          r.full.push_back(2 * totalllh);
        }
        cov->SetParProp(i, r.CentralValue);
        out.push_back(std::move(r));
      }
    }
    for (auto* smp : samples) smp->Reweight();  // the samples back at the point the scan started from
    return out;
  }

  /// Two parameters' scan of the samples' likelihood (ref: FitterBase::Run2DLLHScan, Fitters/FitterBase.cpp:936-1035; the TH2D's
  /// contents returned row-major, sam[x * n_points + y] = 2 x the samples' -log L): parameter pairs (i, j < i) of one systematics
  /// object, both ranges as in RunLLHScan.
  struct LLHScan2DResult {
    int iHandler = 0, iParamX = 0, iParamY = 0;
    double lowerX = 0, upperX = 0, lowerY = 0, upperY = 0;
    std::vector<double> x, y, sam;
  };
  std::vector<LLHScan2DResult> Run2DLLHScan(const int n_points = 20, const double nSigma = 1.0, const std::vector<std::pair<int, int>>& Skip = {}) {
    if (n_points < 1) throw MaCh3Exception(__FILE__, __LINE__, "Run2DLLHScan: no points");
    auto skipped = [&](size_t ic, int i) { return std::find(Skip.begin(), Skip.end(), std::make_pair(int(ic), i)) != Skip.end(); };
    auto range = [&](ParameterHandlerBase* cov, int i, double& lo, double& up) {
      const double cv = cov->GetParProp(i);
      lo = std::max(cv - nSigma * cov->GetDiagonalError(i), cov->GetLowerBound(i));
      up = std::min(cv + nSigma * cov->GetDiagonalError(i), cov->GetUpperBound(i));
      return cv;
    };
    std::vector<LLHScan2DResult> out;
    for (size_t ic = 0; ic < systematics.size(); ++ic) {
      ParameterHandlerBase* cov = systematics[ic];
      for (int i = 0; i < cov->GetNumParams(); ++i) {
        if (skipped(ic, i)) continue;
        for (int j = 0; j < i; ++j) {
          if (skipped(ic, j)) continue;
          LLHScan2DResult r;
          r.iHandler = int(ic);
          r.iParamX = i;
          r.iParamY = j;
          const double central_x = range(cov, i, r.lowerX, r.upperX), central_y = range(cov, j, r.lowerY, r.upperY);
          for (int k = 0; k < n_points; ++k) {
            r.x.push_back(r.lowerX + (double(k) + 0.5) * (r.upperX - r.lowerX) / double(n_points));
            r.y.push_back(r.lowerY + (double(k) + 0.5) * (r.upperY - r.lowerY) / double(n_points));
          }
          r.sam.reserve(size_t(n_points) * size_t(n_points));
          for (int x = 0; x < n_points; ++x)
            for (int y = 0; y < n_points; ++y) {
              cov->SetParProp(i, r.x[size_t(x)]);
              cov->SetParProp(j, r.y[size_t(y)]);
              for (auto* smp : samples) smp->Reweight();
              double samplellh = 0.0;
              for (auto* smp : samples) samplellh += smp->GetLikelihood();
              r.sam.push_back(2 * samplellh);
            }
          cov->SetParProp(i, central_x);
          cov->SetParProp(j, central_y);
          out.push_back(std::move(r));
        }
      }
    }
    for (auto* smp : samples) smp->Reweight();
    return out;
  }

  /// ref: Fitters/FitterBase.cpp:461-520
  void DragRace(int NLaps, double* reweight_s_per_lap, double* llh_s_per_lap) {
    double tr = 0, tl = 0;
    for (int lap = 0; lap < NLaps; ++lap) {
      for (auto* s : systematics) s->ProposeStep();
      const auto a = std::chrono::steady_clock::now();
      for (auto* s : samples) s->Reweight();
      const auto b = std::chrono::steady_clock::now();
      for (auto* s : samples) (void)s->GetLikelihood();
      const auto c = std::chrono::steady_clock::now();
      tr += std::chrono::duration<double>(b - a).count();
      tl += std::chrono::duration<double>(c - b).count();
    }
    *reweight_s_per_lap = tr / NLaps;
    *llh_s_per_lap = tl / NLaps;
    for (auto* s : systematics)  // back to the current point
      for (int i = 0; i < s->GetNumParams(); ++i) s->SetParCurrProp(i, s->GetParCurr(i));
  }

 protected:
  virtual void DoStep() = 0;
  bool IsStepAccepted(const double acc_prob) { return !(unif(random) > acc_prob); }  // ref: MCMCBase.cpp:188-203
  void AcceptStep() {                                                                  // ref: :206-216
    ++accCount;
    logLCurr = logLProp;
    for (auto* s : systematics) s->AcceptStep();
  }
  /// everything MR2T2::ProposeStep does after the proposal itself (ref: MR2T2.cpp:33-98)
  void EvaluateProposal() {
    // Same decisions as the reference (bounds first: a proposal outside them never reaches the samples, MR2T2.cpp:62),
    // but the O(nPar^2) host work rides under the GPU: Reweight() only LAUNCHES the step; the prior term and the Haario
    // update left over from the previous step are computed while the device works, and GetLikelihood() then waits.
    out_of_bounds = false;
    for (auto* s : systematics)
      if (s->CheckBounds() > 0) out_of_bounds = true;
    if (!out_of_bounds)
      for (auto* s : samples) s->Reweight();
    // handlers attached to the device (ParameterHandlerBase::AttachDevice) enqueue their prior and their next throw beside
    // the samples' kernels; the others compute both here, under the GPU's work
    for (auto* s : systematics)
      if (s->OnDevice() && !out_of_bounds) s->LaunchDeviceStep();
    for (size_t s = 0; s < systematics.size(); ++s)
      if (!systematics[s]->OnDevice() || out_of_bounds) syst_llh[s] = systematics[s]->GetLikelihood();
    for (auto* s : systematics) {
      s->FinishPendingAdaption();
      if (!s->OnDevice() || out_of_bounds) s->PrepareNextThrow();
    }
    if (!out_of_bounds) {
      for (size_t i = 0; i < samples.size(); ++i) sample_llh[i] = samples[i]->GetLikelihood();
    } else {
      ++oobCount;
      for (size_t i = 0; i < samples.size(); ++i) sample_llh[i] = M3::_LARGE_LOGL_;
    }
    for (size_t s = 0; s < systematics.size(); ++s)
      if (systematics[s]->OnDevice() && !out_of_bounds) syst_llh[s] = systematics[s]->CollectDeviceStep();
    // the sum in the reference's order whatever order the terms became available in: systematics first, then the samples
    // (MR2T2.cpp:33-84) -- logLProp is then the same double as the reference's, and an accept / reject on the threshold
    // cannot flip between a device-attached and a host-only chain
    double llh = 0.0;
    for (size_t s = 0; s < systematics.size(); ++s) llh += syst_llh[s];
    if (!out_of_bounds)
      for (size_t i = 0; i < samples.size(); ++i) llh += sample_llh[i];
    logLProp = llh;
  }
  void ProposeStepAtCurrent() {
    for (auto* s : systematics)
      for (int i = 0; i < s->GetNumParams(); ++i) s->SetParCurrProp(i, s->GetParCurr(i));
    EvaluateProposal();
  }

  std::vector<SampleHandlerBase*> samples;
  std::vector<ParameterHandlerBase*> systematics;
  std::vector<double> syst_llh, sample_llh;
  double logLCurr = 0.0, logLProp = 0.0;
  bool out_of_bounds = false, started = false;
  int step = 0, accCount = 0, oobCount = 0;
  bool anneal = false;
  double AnnealTemp = -999;
  std::mt19937_64 random;
  std::uniform_real_distribution<double> unif{0.0, 1.0};
  std::vector<double> StepStore, LogLStore, stepTime;
  std::vector<char> AccStore;
};

/// The minimisers' caller of the hot path (ref: Fitters/LikelihoodFit.h:6-36, Fitters/LikelihoodFit.cpp:20-123): MinuitFit and PSO
/// hand a flat parameter vector to CalcChi2, which sets it on the systematics objects (mirrored at the bounds if asked), sums their
/// prior terms, reweights every sample, adds the samples' likelihoods and returns 2 x the total -- the same Reweight() +
/// GetLikelihood() pair as a chain's step, with every evaluation accepted.  The minimiser itself (Minuit2 in the reference) is the
/// derived class's; examples/likelihood_fit_nm.cpp drives this with a Nelder-Mead simplex.
class LikelihoodFit : public MCMCBase {
 public:
  explicit LikelihoodFit(uint64_t seed = 12345) : MCMCBase(seed) {}
  void SetMirroring(const bool on) { fMirroring = on; }
  /// ref: LikelihoodFit::PrepareFit (:20-36): the length of CalcChi2's argument
  void PrepareFit() {
    NPars = 0;
    for (auto* s : systematics) NPars += s->GetNumParams();
    syst_llh.assign(systematics.size(), 0.0);
    sample_llh.assign(samples.size(), 0.0);
  }
  int GetNPars() const { return NPars; }
  /// ref: LikelihoodFit::CalcChi2 (:39-123), statement for statement without the output tree
  double CalcChi2(const double* x) {
    const auto t0 = std::chrono::steady_clock::now();
    if (syst_llh.size() != systematics.size() || sample_llh.size() != samples.size()) PrepareFit();
    int ParCounter = 0;
    double llh = 0.0;
    for (auto* it : systematics) {
      const int NumPar = it->GetNumParams();
      std::vector<double> pars(static_cast<size_t>(NumPar));
      for (int i = 0; i < NumPar; ++i, ++ParCounter) {
        double ParVal = x[ParCounter];
        if (fMirroring) {  // a value beyond a bound is reflected back inside
          if (ParVal < it->GetLowerBound(i)) ParVal = it->GetLowerBound(i) + (it->GetLowerBound(i) - ParVal);
          else if (ParVal > it->GetUpperBound(i)) ParVal = it->GetUpperBound(i) - (ParVal - it->GetUpperBound(i));
        }
        pars[size_t(i)] = ParVal;
      }
      it->SetParameters(pars);
      it->AcceptStep();
    }
    for (size_t k = 0; k < systematics.size(); ++k) {
      // (CalcLikelihood, not GetLikelihood: out of bounds is the minimiser's business, ref :84-86)
      syst_llh[k] = systematics[k]->CalcLikelihood();
      llh += syst_llh[k];
    }
    for (auto* s : samples) s->Reweight();
    for (size_t i = 0; i < samples.size(); ++i) {
      sample_llh[i] = samples[i]->GetLikelihood();
      llh += sample_llh[i];
    }
    logLProp = logLCurr = llh;  // (accProb = 1: every evaluation is kept)
    ++step;
    ++accCount;
    stepTime.push_back(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return 2.0 * llh;
  }

 protected:
  void DoStep() override { throw MaCh3Exception(__FILE__, __LINE__, "LikelihoodFit is driven through CalcChi2 by its minimiser, not through RunMCMC"); }
  int NPars = 0;
  bool fMirroring = false;
};

/// ref: Fitters/MR2T2.cpp:11-115
class MR2T2 : public MCMCBase {
 public:
  using MCMCBase::MCMCBase;

 protected:
  void DoStep() override {  // :11-21
    ProposeStep();
    accProb = AcceptanceProbability();
    if (IsStepAccepted(accProb) && !out_of_bounds) AcceptStep();
  }
  void ProposeStep() {  // :25-99
    for (auto* s : systematics) s->ProposeStep();
    EvaluateProposal();
  }
  double AcceptanceProbability() const {  // :103-115
    if (anneal) return std::min(1., std::exp(-(logLProp - logLCurr) / (std::exp(-step / AnnealTemp))));
    return std::min(1., std::exp(logLCurr - logLProp));
  }
  double accProb = 0.0;
};

/// Delayed rejection: after a rejection, retry from the same point with a shrunken proposal
/// (ref: Fitters/DelayedMR2T2.cpp:95-235; acceptance of the later tries per arXiv:2010.04190)
class DelayedMR2T2 : public MR2T2 {
 public:
  /// delay_probability follows the reference literally: after a rejected try the loop STOPS when
  /// Rndm() <= delay_probability (DelayedMR2T2.cpp:148-150, 167-171), so small values delay often and
  /// 1 never delays; <= 0 switches delayed rejection off (DelayedMR2T2.cpp:24-29).
  DelayedMR2T2(uint64_t seed, double initial_scale_, double decay_rate_, int max_rejections_, double delay_probability_,
               bool delay_on_oob_only_ = false)
      : MR2T2(seed), initial_scale(initial_scale_), decay_rate(decay_rate_), max_rejections(max_rejections_),
        delay_probability(std::min(1.0, std::max(0.0, delay_probability_))), delay_on_oob_only(delay_on_oob_only_) {
    if (delay_probability <= 0.0) max_rejections = 0;
  }
  int GetDelayedAcceptCount() const { return delayedAccCount; }

 protected:
  void DoStep() override {  // :95-165
    std::vector<std::vector<double>> current_step_vals;
    std::vector<double> start_step_scale;
    for (auto* s : systematics) {  // StoreCurrentStep
      current_step_vals.push_back(s->GetParCurrVec());
      start_step_scale.push_back(s->GetGlobalStepScale());
    }
    for (auto* s : systematics) s->SetStepScale(initial_scale * s->GetGlobalStepScale());
    bool accepted = false, delay = false;
    MinLogLikelihood = M3::_LARGE_LOGL_;
    for (int i = 0; i <= max_rejections; ++i) {
      ProposeStep();
      for (auto* s : systematics) s->AcceptStep();  // "leapfrog": the next try starts from this proposal
      if (out_of_bounds || logLProp > MinLogLikelihood) continue;
      if (i == 0) {
        accProb = MR2T2::AcceptanceProbability();
      } else {
        accProb = DelayedAcceptanceProbability();
        delay = true;
      }
      if (IsStepAccepted(accProb)) {
        AcceptStep();
        accepted = true;
        if (delay) ++delayedAccCount;
        break;
      }
      if (!out_of_bounds && delay_on_oob_only) break;
      if (!(unif(random) > delay_probability)) break;  // `if (not ProbabilisticDelay()) break;`
      for (auto* s : systematics) s->SetStepScale(decay_rate * s->GetGlobalStepScale());
      MinLogLikelihood = logLProp;
    }
    if (!accepted)
      for (size_t k = 0; k < systematics.size(); ++k)
        for (int j = 0; j < systematics[k]->GetNumParams(); ++j) systematics[k]->SetParCurrProp(j, current_step_vals[k][size_t(j)]);
    for (size_t k = 0; k < systematics.size(); ++k) systematics[k]->SetStepScale(start_step_scale[k]);
  }
  double DelayedAcceptanceProbability() const {  // :176-190
    const double numerator = std::max(0.0, std::exp(MinLogLikelihood - logLProp) - 1.0);
    const double denominator = std::exp(MinLogLikelihood - logLCurr) - 1.0;
    if (denominator <= 0.0) return 1.0;
    if (std::isinf(numerator) || std::isinf(denominator)) return MR2T2::AcceptanceProbability();
    return std::min(numerator / denominator, 1.0);
  }

  double initial_scale, decay_rate;
  int max_rejections;
  double delay_probability;
  bool delay_on_oob_only;
  double MinLogLikelihood = M3::_LARGE_LOGL_;
  int delayedAccCount = 0;
};
