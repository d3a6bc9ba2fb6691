// SampleHandlerFD.h -- the sample operator of the per-step path on the GPU.
//
//   SampleHandlerBase  Samples/SampleHandlerBase.h:37-56, 181-185 (what a fitter calls per step)
//   BinningHandler     Samples/BinningHandler.h (uniform N-D binning: FindGlobalBin /
//                      FindNominalBin, Samples/BinningHandler.cpp:257-355, SampleStructs.h:577-613)
//   SampleHandlerFD    Samples/SampleHandlerFD.h:21-413 -- only the per-step surface: Reweight,
//                      GetLikelihood, GetSampleLikelihood, GetMCArray / GetW2Array / GetDataArray,
//                      AddData, SetTestStatistic.  Configuration (YAML), ROOT I/O and plotting
//                      projections are out of scope; events are registered through
//                      AddEvent/FinaliseEvents instead of SetupExperimentMC.  Functional parameters
//                      (SampleHandlerFD.cpp:545-564) and selection cuts (:281-294) are available
//                      from a closed set once EnableDeviceBinning() moved FindGlobalBin to the device.
//
// Reweight() runs spline evaluation, weight product, bin fill and (on GetLikelihood) the bin
// likelihood on the device; per step only the proposed parameter values cross PCIe.
#pragma once

#include <algorithm>
#include <array>
#include <cstdio>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "../m3hip.h"
#include "Core.h"
#include "SplineMonolith.h"

/// ref: Samples/SampleHandlerBase.h:37-56
class SampleHandlerBase {
 public:
  virtual ~SampleHandlerBase() {}
  virtual M3::int_t GetNsamples() { return nSamples; }
  virtual std::string GetName() const = 0;
  virtual std::string GetSampleTitle(const int Sample) const = 0;
  virtual void Reweight() = 0;
  virtual double GetLikelihood() const = 0;
  virtual double GetSampleLikelihood(const int isample) const = 0;
  virtual void CleanMemoryBeforeFit() = 0;
  virtual void PrintRates(const bool DataOnly = false) = 0;
  void SetTestStatistic(TestStatistic testStat) { fTestStatistic = testStat; TestStatisticChanged(); }

 protected:
  virtual void TestStatisticChanged() {}
  TestStatistic fTestStatistic = kPoisson;
  M3::int_t nSamples = 0;
  unsigned int nEvents = 0;
};

/// Uniform N-dimensional binning of several samples in one global bin space.
class BinningHandler {
 public:
  /// edges[dim] = increasing bin edges of that axis; returns the sample index
  int AddUniformSample(const std::vector<std::vector<double>>& edges) {
    for (const auto& e : edges)
      if (e.size() < 2 || !std::is_sorted(e.begin(), e.end())) throw MaCh3Exception(__FILE__, __LINE__, "bin edges must be increasing");
    Sample s;
    s.Edges = edges;
    int stride = 1;
    for (const auto& e : edges) {
      s.Strides.push_back(stride);
      stride *= int(e.size()) - 1;
    }
    s.nBins = stride;
    s.GlobalOffset = TotalBins;
    TotalBins += s.nBins;
    Samples.push_back(std::move(s));
    return int(Samples.size()) - 1;
  }
  /// Non-uniform binning: every bin is an N-dimensional box {lo, hi} per axis, a point belongs to it
  /// when lo < x <= hi on every axis (ref: Samples/SampleStructs.h:186-220).  Bins are found through
  /// a coarse 10-cells-per-axis grid over their bounding box whose cells list the boxes they overlap
  /// (SampleStructs.h:397-530); the grid itself is searched like uniform bins, [lo, hi).
  int AddNonUniformSample(const std::vector<std::vector<std::array<double, 2>>>& boxes) {
    if (boxes.empty() || boxes[0].size() < 2) throw MaCh3Exception(__FILE__, __LINE__, "non-uniform binning needs >= 2 dimensions");
    const size_t dim = boxes[0].size();
    Sample s;
    s.Uniform = false;
    s.Boxes = boxes;
    constexpr int kCells = 10;
    for (size_t d = 0; d < dim; ++d) {
      double lo = std::numeric_limits<double>::max(), hi = std::numeric_limits<double>::lowest();
      for (const auto& b : boxes) {
        if (b.size() != dim) throw MaCh3Exception(__FILE__, __LINE__, "bins of different dimension");
        lo = std::min(lo, b[d][0]);
        hi = std::max(hi, b[d][1]);
      }
      std::vector<double> e(kCells + 1);
      const double w = (hi - lo) / double(kCells);
      for (int i = 0; i <= kCells; ++i) e[size_t(i)] = lo + double(i) * w;
      s.Edges.push_back(e);
    }
    int stride = 1;
    for (size_t d = 0; d < dim; ++d) {
      s.Strides.push_back(stride);
      stride *= kCells;
    }
    s.CellBoxes.assign(size_t(stride), {});
    std::vector<int> idx(dim, 0);
    for (int cell = 0; cell < stride; ++cell) {  // cell = sum idx[d] * 10^d
      int r = cell;
      for (size_t d = 0; d < dim; ++d) { idx[d] = r % kCells; r /= kCells; }
      for (size_t ib = 0; ib < boxes.size(); ++ib) {
        bool overlap = true;
        for (size_t d = 0; d < dim && overlap; ++d)
          overlap = boxes[ib][d][1] > s.Edges[d][size_t(idx[d])] && boxes[ib][d][0] < s.Edges[d][size_t(idx[d]) + 1];
        if (overlap) s.CellBoxes[size_t(cell)].push_back(int(ib));
      }
    }
    s.nBins = int(boxes.size());
    s.GlobalOffset = TotalBins;
    TotalBins += s.nBins;
    Samples.push_back(std::move(s));
    return int(Samples.size()) - 1;
  }
  int GetNBins() const { return TotalBins; }
  int GetNBins(int Sample) const { return Samples[Sample].nBins; }
  int GetNSamples() const { return int(Samples.size()); }
  int GetNDim(int Sample) const { return int(Samples[Sample].Edges.size()); }
  int GetSampleStartBin(int Sample) const { return Samples[Sample].GlobalOffset; }
  int GetSampleEndBin(int Sample) const { return Samples[Sample].GlobalOffset + Samples[Sample].nBins; }
  /// ref: Samples/BinningHandler.cpp:294-307
  int FindNominalBin(int iSample, int iDim, double Var) const {
    const auto& e = Samples[iSample].Edges[iDim];
    if (!(Var >= e.front() && Var < e.back())) return M3::UnderOverFlowBin;  // written so that a NaN is out of range (the device does the same)
    return int(std::upper_bound(e.begin(), e.end(), Var) - e.begin()) - 1;
  }
  /// Bins are [lo, hi); a value outside any axis gives UnderOverFlowBin.  NomBin is the cached
  /// nominal bin per axis (ref: Samples/SampleStructs.h:577-613: nominal, +-1 neighbour, then
  /// upper_bound; all four routes return the same index, the fast ones just skip the search).
  int FindGlobalBin(int NomSample, const std::vector<const double*>& KinVar, const std::vector<int>& NomBin) const {
    const Sample& s = Samples[NomSample];
    int Global = 0;
    for (size_t d = 0; d < KinVar.size(); ++d) {
      const auto& e = s.Edges[d];
      const double v = *KinVar[d];
      const int n = int(e.size()) - 1;
      if (!(v >= e[0] && v < e[n])) return M3::UnderOverFlowBin;  // NaN: out of range, not bin n
      int b = NomBin[d];
      if (!(b > M3::UnderOverFlowBin && v >= e[b] && v < e[b + 1])) {
        if (b > 0 && v < e[b] && v >= e[b - 1]) b = b - 1;
        else if (b > M3::UnderOverFlowBin && b + 2 <= n && v >= e[b + 1] && v < e[b + 2]) b = b + 1;
        else b = int(std::upper_bound(e.begin(), e.end(), v) - e.begin()) - 1;
      }
      Global += b * s.Strides[d];
    }
    if (s.Uniform) return Global + s.GlobalOffset;
    for (int ib : s.CellBoxes[size_t(Global)]) {  // ref: Samples/BinningHandler.cpp:277-289
      bool inside = true;
      for (size_t d = 0; d < KinVar.size(); ++d) inside &= (*KinVar[d] > s.Boxes[size_t(ib)][d][0]) & (*KinVar[d] <= s.Boxes[size_t(ib)][d][1]);
      if (inside) return ib + s.GlobalOffset;
    }
    return M3::UnderOverFlowBin;
  }
  /// hand the binning to the device (m3hip_binning_*): defines the global bin space there and lets
  /// the device run FindGlobalBin itself once kinematics are uploaded
  void UploadTo(m3hip_handle h) const {
    M3HIP_CALL(h, m3hip_binning_begin(h));
    for (const Sample& s : Samples) {
      if (s.Uniform) {
        std::vector<int32_t> ne;
        std::vector<double> flat;
        for (const auto& e : s.Edges) {
          ne.push_back(int32_t(e.size()));
          flat.insert(flat.end(), e.begin(), e.end());
        }
        M3HIP_CALL(h, m3hip_binning_add_uniform(h, int(s.Edges.size()), ne.data(), flat.data()));
      } else {
        std::vector<double> ext;
        for (const auto& b : s.Boxes)
          for (const auto& lohi : b) { ext.push_back(lohi[0]); ext.push_back(lohi[1]); }
        M3HIP_CALL(h, m3hip_binning_add_nonuniform(h, int(s.Boxes.size()), int(s.Boxes[0].size()), ext.data()));
      }
    }
    M3HIP_CALL(h, m3hip_binning_end(h));
  }
  std::vector<int32_t> SampleBinOffsets() const {
    std::vector<int32_t> off;
    for (const auto& s : Samples) off.push_back(s.GlobalOffset);
    off.push_back(TotalBins);
    return off;
  }

 private:
  struct Sample {
    std::vector<std::vector<double>> Edges;
    std::vector<int> Strides;
    int nBins = 0, GlobalOffset = 0;
    bool Uniform = true;
    std::vector<std::vector<std::array<double, 2>>> Boxes;  // non-uniform bins
    std::vector<std::vector<int>> CellBoxes;                // grid cell -> candidate bins
  };
  std::vector<Sample> Samples;
  int TotalBins = 0;
};

/// A normalisation parameter's applicability (ref: Parameters/ParameterStructs.h:111-131).  The reference names kinematic
/// variables by strings resolved through the experiment's ReturnKinematicParameter; here they are columns of the event's KinVars.
struct NormParameter {
  std::string name;
  int index = 0;                       ///< position in the parameter vector (what becomes a norm_pointer)
  std::vector<int> modes, pdgs, preoscpdgs, targets;  ///< empty: applies to all (SampleHandlerBase::MatchCondition)
  bool hasKinBounds = false;
  std::vector<int> KinematicColumn;    ///< per bounded variable: column of KinVars (KinematicVarStr in the reference)
  std::vector<std::vector<std::vector<double>>> Selection;  ///< [variable][alternative] = {lo, hi}: lo < x <= hi, any alternative
};
/// What the applicability test reads of an event (ref: Samples/FarDetectorCoreInfoStruct.h: mode, Target, nupdg, nupdgUnosc, isNC)
struct EventTruth {
  int mode = 0, target = 0, nupdg = 0, nupdgUnosc = 0;
  bool isNC = false;
};

class SampleHandlerFD : public SampleHandlerBase {
 public:
  /// SampleHandlerFD::CalcNormsBins for one event (ref: Samples/SampleHandlerFD.cpp:667-731, PassesSelection :2101-2130,
  /// MatchCondition Samples/SampleHandlerBase.h:196-202): the indices of the normalisation parameters that apply, in the order of
  /// the parameter list -- what AddEvent takes as NormParams.  Oscillated NC events get none (:678-681); target, flavour,
  /// unoscillated flavour and mode must each be listed (an empty list admits all); every bounded kinematic variable must lie in
  /// one of its (lo, hi] ranges.
  static std::vector<int16_t> CalcNormsBins(const std::vector<NormParameter>& norm_parameters, const EventTruth& ev, const std::vector<double>& KinVars) {
    std::vector<int16_t> NormBins;
    if (ev.isNC && ev.nupdg != ev.nupdgUnosc) return NormBins;
    auto MatchCondition = [](const std::vector<int>& allowed, int value) {
      return allowed.empty() || std::find(allowed.begin(), allowed.end(), value) != allowed.end();
    };
    for (const NormParameter& par : norm_parameters) {
      if (!MatchCondition(par.targets, ev.target)) continue;
      if (!MatchCondition(par.pdgs, ev.nupdg)) continue;
      if (!MatchCondition(par.preoscpdgs, ev.nupdgUnosc)) continue;
      if (!MatchCondition(par.modes, ev.mode)) continue;
      bool IsSelected = true;
      if (par.hasKinBounds) {
        if (par.KinematicColumn.size() != par.Selection.size()) throw MaCh3Exception(__FILE__, __LINE__, "NormParameter " + par.name + ": one Selection entry per bounded variable");
        for (size_t iKinPar = 0; iKinPar < par.KinematicColumn.size() && IsSelected; ++iKinPar) {
          const size_t col = size_t(par.KinematicColumn[iKinPar]);
          if (col >= KinVars.size()) throw MaCh3Exception(__FILE__, __LINE__, "NormParameter " + par.name + ": kinematic column beyond the event's variables");
          const double kinVal = KinVars[col];
          bool passedAnyBound = false;
          for (const auto& bounds : par.Selection[iKinPar])
            if (kinVal > bounds[0] && kinVal <= bounds[1]) { passedAnyBound = true; break; }
          if (!passedAnyBound) IsSelected = false;
        }
      }
      if (!IsSelected) continue;
      NormBins.push_back(int16_t(par.index));
    }
    return NormBins;
  }

  /// `Splines` keeps ownership of the device handle; `ParameterValues` plays
  /// ParameterHandlerBase::_fPropVal (read at every Reweight, like the reference's pointers).
  SampleHandlerFD(std::string Name, std::unique_ptr<SplineBase> Splines, std::unique_ptr<BinningHandler> Bins,
                  const double* ParameterValues, int nParameters, bool UpdateW2_ = false)
      : SampleName(std::move(Name)), SplineHandler(std::move(Splines)), Binning(std::move(Bins)), fPropVal(ParameterValues),
        nPar(nParameters), UpdateW2(UpdateW2_) {
    h = SplineHandler->Handle();
    nSamples = M3::int_t(Binning->GetNSamples());
    nEvents = SplineHandler->GetNEvents();
    Binning->UploadTo(h);
    M3HIP_CALL(h, m3hip_set_test_statistic(h, int(fTestStatistic), UpdateW2 ? 1 : 0));
    bin_idx.assign(nEvents, M3::UnderOverFlowBin);
    osc_idx.assign(nEvents, -1);
    norm_off.assign(size_t(nEvents) + 1, 0);
  }

  /// Which entry of the parameter vector drives spline parameter i (SMonolith::setSplinePointers)
  void SetSplineParameterIndices(const std::vector<int32_t>& idx) { M3HIP_CALL(h, m3hip_set_spline_param_map(h, idx.data(), int(idx.size()))); }

  /// Per-event set-up, in event order: the binning variables, the normalisation parameters
  /// that apply (indices into the parameter vector -> norm_pointers) and the event's entry in
  /// the oscillation-weight table (-1: no oscillation weight, e.g. NC; ref:
  /// Samples/SampleHandlerFD.cpp:637-663, 1108-1133).
  void AddEvent(unsigned int iEvent, int Sample, const std::vector<double>& KinVars, const std::vector<int16_t>& NormParams, int OscIndex) {
    if (iEvent != n_added || iEvent >= nEvents) throw MaCh3Exception(__FILE__, __LINE__, "events must be added in order, one per event of the spline handler");
    if (Sample < 0 || Sample >= Binning->GetNSamples()) throw MaCh3Exception(__FILE__, __LINE__, "AddEvent: sample index out of range");
    // the first GetNDim(Sample) entries are the binning axes; further entries are cut-only columns (AddSelection)
    const size_t nd = size_t(Binning->GetNDim(Sample));
    if (KinVars.size() < nd) throw MaCh3Exception(__FILE__, __LINE__, "fewer kinematic variables than binning axes");
    std::vector<const double*> kv;
    std::vector<int> nom;
    for (size_t d = 0; d < nd; ++d) {
      kv.push_back(&KinVars[d]);
      nom.push_back(Binning->FindNominalBin(Sample, int(d), KinVars[d]));
    }
    bin_idx[iEvent] = Binning->FindGlobalBin(Sample, kv, nom);
    if (KinVars.size() > size_t(kKinStride)) throw MaCh3Exception(__FILE__, __LINE__, "at most 4 kinematic columns per event");
    sample_id.push_back(Sample);
    for (int d = 0; d < kKinStride; ++d) kin.push_back(size_t(d) < KinVars.size() ? KinVars[size_t(d)] : 0.0);
    osc_idx[iEvent] = OscIndex;
    norm_idx.insert(norm_idx.end(), NormParams.begin(), NormParams.end());
    norm_off[iEvent + 1] = norm_idx.size();
    ++n_added;
  }
  void FinaliseEvents() {
    if (n_added != nEvents) throw MaCh3Exception(__FILE__, __LINE__, "not every event was added");
    M3HIP_CALL(h, m3hip_upload_events(h, 0, nEvents, bin_idx.data(), osc_idx.data(), norm_off.data(), norm_idx.data()));
  }
  /// Move FindGlobalBin to the device: uploads the nominal kinematics, lets the device compute nominal and
  /// global bins, and checks them against the host's (integers: must agree exactly).  Needed before
  /// RegisterFunctionalParameter / AddSelection; KinVars beyond the sample's binning axes are cut variables.
  void EnableDeviceBinning() {
    if (n_added != nEvents || sample_id.size() != nEvents) throw MaCh3Exception(__FILE__, __LINE__, "EnableDeviceBinning: call after FinaliseEvents, before CleanMemoryBeforeFit");
    M3HIP_CALL(h, m3hip_upload_kinematics(h, 0, nEvents, sample_id.data(), kin.data(), kKinStride));
    M3HIP_CALL(h, m3hip_compute_bins(h));
    std::vector<int32_t> dev(nEvents);
    M3HIP_CALL(h, m3hip_get_event_bins(h, dev.data(), 0, nEvents));
    for (unsigned int i = 0; i < nEvents; ++i)
      if (dev[i] != bin_idx[i]) throw MaCh3Exception(__FILE__, __LINE__, "device and host FindGlobalBin disagree for event " + std::to_string(i));
    device_binning = true;
  }
  /// Functional parameter from the closed set the device supports (the reference registers arbitrary
  /// std::function callbacks, SampleHandlerFD.cpp:545-564): Kind 1: x *= 1 + theta, Kind 2: x += theta, where x
  /// is kinematic column `Column` and theta = ParameterValues[ParameterIndex].  Sample -1 = every sample.
  /// With one registered, every Reweight() re-bins all events.
  void RegisterFunctionalParameter(int Sample, int Column, int Kind, int ParameterIndex) {
    if (!device_binning) throw MaCh3Exception(__FILE__, __LINE__, "RegisterFunctionalParameter needs EnableDeviceBinning()");
    M3HIP_CALL(h, m3hip_add_kin_shift(h, Sample, Column, Kind, ParameterIndex));
  }
  /// Kind 3 of that closed set: x[Column] += theta * (nominal value of column AuxColumn)
  void RegisterFunctionalParameterAux(int Sample, int Column, int AuxColumn, int ParameterIndex) {
    if (!device_binning) throw MaCh3Exception(__FILE__, __LINE__, "RegisterFunctionalParameterAux needs EnableDeviceBinning()");
    M3HIP_CALL(h, m3hip_add_kin_shift_aux(h, Sample, Column, AuxColumn, ParameterIndex));
  }
  /// Selection lo <= x < hi on kinematic column `Column`, evaluated on the shifted values
  /// (SampleHandlerFD::IsEventSelected, SampleHandlerFD.cpp:281-294)
  void AddSelection(int Sample, int Column, double Lo, double Hi) {
    if (!device_binning) throw MaCh3Exception(__FILE__, __LINE__, "AddSelection needs EnableDeviceBinning()");
    M3HIP_CALL(h, m3hip_add_kin_cut(h, Sample, Column, Lo, Hi));
  }
  std::vector<int32_t> GetEventBins() const {
    std::vector<int32_t> dev(nEvents);
    M3HIP_CALL(h, m3hip_get_event_bins(h, dev.data(), 0, nEvents));
    return dev;
  }
  /// weights OscillationHandler::Evaluate would have left behind the event pointers
  void SetOscillationWeights(const std::vector<float>& w) { M3HIP_CALL(h, m3hip_set_osc_table(h, w.data(), w.size())); }
  /// M3::float_t = double build (with a BinnedSplineHandlerT<double> behind the handle)
  void SetOscillationWeights(const std::vector<double>& w) { M3HIP_CALL(h, m3hip_set_osc_table_f64(h, w.data(), w.size())); }
  /// The order in which to REGISTER the MC (events, their splines, everything per event) so that events of one oscillation bin lie
  /// next to each other: stable by oscillation index, events without a pointer (NC: index -1) last.  Any order gives the same
  /// bins up to fp64 summation order; this one makes the 64 look-ups of a wave fall into one or two 128-byte lines of the weight
  /// table instead of 64.  It matters only for a LARGE table (atmospheric: 200 000 entries) that a fit MOVES every step: one GPU's
  /// share of BASELINE configs[4], table replaced every step, 486 us (random order) -> 438 us (this order) = the resting table's
  /// 436 us (profiles/r06_c5_moving_osc_*).  m3hip_info::osc_lines_per_kevent says which case a sample is in.
  static std::vector<uint32_t> EventOrderByOscillationBin(const std::vector<int32_t>& osc_idx) {
    std::vector<uint32_t> order(osc_idx.size());
    for (size_t e = 0; e < order.size(); ++e) order[e] = uint32_t(e);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return uint32_t(osc_idx[a]) < uint32_t(osc_idx[b]); });
    return order;
  }
  /// The weight array of a GPU oscillation calculator, read where it lies: the reference's events alias the calculator's array
  /// (SetupNuOscillatorPointers, SampleHandlerFD.cpp:1108-1133) and OscillationHandler::Evaluate refills it inside every Reweight
  /// (SampleHandlerFD.cpp:323).  Call after every refill; ordering and lifetime as m3hip_set_osc_table_device states (m3hip.h).
  void SetOscillationWeightsDevice(const float* d_weights, size_t n) { M3HIP_CALL(h, m3hip_set_osc_table_device(h, d_weights, n)); }
  void SetOscillationWeightsDevice(const double* d_weights, size_t n) { M3HIP_CALL(h, m3hip_set_osc_table_device_f64(h, d_weights, n)); }

  /// Experiment weights behind the spline weight.  The reference lets an experiment append pointers to an event's
  /// total_weight_pointers (AddAdditionalWeightPointers, SampleHandlerFD.cpp:196; multiplied in that order at :584-591)
  /// and refresh what they point at in the virtual CalcWeightFunc (SampleHandlerFD.h:289, called at :428).  Here a
  /// pointer is a slot 0..7, multiplied in ascending order behind the spline weight.  Form 1: the target is a
  /// per-event variable (flux, POT, detector weights; or whatever CalcWeightFunc computed, uploaded again when it
  /// changes) ...
  void AddAdditionalWeightPointers(int Slot, const std::vector<float>& PerEventWeight, unsigned int FirstEvent = 0) {
    M3HIP_CALL(h, m3hip_upload_event_weights(h, Slot, FirstEvent, PerEventWeight.size(), PerEventWeight.data()));
  }
  /// ... form 2: the target is a variable shared by a class of events: an index per event (-1: the event has no such
  /// pointer) into a table set with SetAdditionalWeights, also every step.
  void AddAdditionalWeightPointers(int Slot, const std::vector<int32_t>& IndexPerEvent, unsigned int FirstEvent = 0) {
    M3HIP_CALL(h, m3hip_upload_weight_index(h, Slot, FirstEvent, IndexPerEvent.size(), IndexPerEvent.data()));
  }
  void SetAdditionalWeights(int Slot, const std::vector<float>& Table) { M3HIP_CALL(h, m3hip_set_weight_table(h, Slot, Table.data(), Table.size())); }
  /// bins, W2 and likelihood bit-reproducible from call to call (fixed-order fill instead of fp64 atomics)
  void SetDeterministicFill(bool On) { M3HIP_CALL(h, m3hip_set_deterministic(h, On ? 1 : 0)); }

  // ---- per step (ref: Samples/SampleHandlerFD.cpp:316-343, 1262-1300)
  void Reweight() override { M3HIP_CALL(h, m3hip_reweight(h, fPropVal, nPar)); }
  double GetLikelihood() const override {
    double llh = 0;
    per_sample.assign(size_t(nSamples), 0.0);
    M3HIP_CALL(h, m3hip_get_likelihood(h, &llh, per_sample.data()));
    return llh;
  }
  double GetSampleLikelihood(const int isample) const override {
    GetLikelihood();
    return per_sample.at(size_t(isample));
  }

  // ---- data and predictions (ref: Samples/SampleHandlerFD.h:162-187, .cpp:1027-1044)
  void AddData(const int Sample, const std::vector<double>& Data) {
    if (int(Data.size()) != Binning->GetNBins(Sample)) throw MaCh3Exception(__FILE__, __LINE__, "AddData: wrong number of bins");
    if (data.empty()) data.assign(size_t(Binning->GetNBins()), 0.0);
    std::copy(Data.begin(), Data.end(), data.begin() + Binning->GetSampleStartBin(Sample));
    M3HIP_CALL(h, m3hip_set_data(h, data.data(), int(data.size())));
  }
  std::vector<double> GetMCArray(const int Sample) const { return Slice(Bins(true), Sample); }
  std::vector<double> GetW2Array(const int Sample) const { return Slice(Bins(false), Sample); }
  std::vector<double> GetDataArray(const int Sample) const { return Slice(data, Sample); }
  std::vector<double> GetMCArray() const { return Bins(true); }
  std::vector<double> GetW2Array() const { return Bins(false); }

  std::string GetName() const override { return SampleName; }
  std::string GetSampleTitle(const int Sample) const override { return SampleName + "_" + std::to_string(Sample); }
  void CleanMemoryBeforeFit() override {
    std::vector<int32_t>().swap(bin_idx);
    std::vector<int32_t>().swap(osc_idx);
    std::vector<uint64_t>().swap(norm_off);
    std::vector<int16_t>().swap(norm_idx);
    std::vector<int32_t>().swap(sample_id);
    std::vector<double>().swap(kin);
  }
  void PrintRates(const bool DataOnly = false) override {
    const std::vector<double> mc = Bins(true);
    for (int s = 0; s < nSamples; ++s) {
      double d = 0, m = 0;
      for (int b = Binning->GetSampleStartBin(s); b < Binning->GetSampleEndBin(s); ++b) {
        if (!data.empty()) d += data[size_t(b)];
        m += mc[size_t(b)];
      }
      if (DataOnly) std::printf("%s: data %.3f\n", GetSampleTitle(s).c_str(), d);
      else std::printf("%s: data %.3f  MC %.3f\n", GetSampleTitle(s).c_str(), d, m);
    }
  }
  // ---- one sample over several GPUs (new: the reference is one device, Manager/gpuUtils.cu:71).  Every rank -- one process
  // per GPU -- builds this object over ITS events only (M3::ShardRanges, Sharding.h); from InitComm / OpenExchange on,
  // Reweight() ends in one exchange of the bin vector, and GetLikelihood(), GetMCArray() and GetW2Array() return the whole
  // sample's numbers on every rank.  Every rank must then make the same calls in the same order (each Reweight is a
  // collective); a fitter seeded alike on all ranks does, because it sees the same likelihoods everywhere.  The transport of
  // the two small blobs (128 / 64 bytes) between the ranks is the caller's: MPI_Bcast / MPI_Allgather, a file, a socket.

  /// rank 0 creates the RCCL unique id and hands it to all ranks (ncclGetUniqueId)
  static std::array<char, M3HIP_UNIQUE_ID_BYTES> GetCommUniqueId() {
    std::array<char, M3HIP_UNIQUE_ID_BYTES> id{};
    if (m3hip_comm_get_unique_id(id.data()) != M3HIP_OK) throw MaCh3Exception(__FILE__, __LINE__, m3hip_last_error(nullptr));
    return id;
  }
  static bool CommAvailable() { return m3hip_comm_available() != 0; }
  /// collective over all ranks (ncclCommInitRank): Reweight() then all-reduces [mc | w2] in place over RCCL / xGMI.
  /// Validate: every step also gathers the ranks' partial vectors and checks the reduced one against their host-side sum
  /// (first runs on a new node; costs a synchronisation per step).
  void InitComm(const void* UniqueId, int Rank, int nRanks, bool Validate = false) {
    M3HIP_CALL(h, m3hip_comm_init(h, UniqueId, Rank, nRanks));
    M3HIP_CALL(h, m3hip_comm_set_validate(h, Validate ? 1 : 0));
    comm_rank = Rank;
    comm_ranks = nRanks;
  }
  /// The one-shot form of the same exchange (one xGMI hop instead of a ring's 2 (N - 1); bit-identical bins on every rank by
  /// construction): CreateExchange allocates this rank's slot buffer and returns its 64-byte IPC handle; once every rank has
  /// everybody's, OpenExchange(handles of all ranks in rank order) switches Reweight() over.  TimeoutSeconds: how long a rank
  /// waits on the device for its peers' vectors of a step before the next GetLikelihood / GetMCArray throws (<= 50).
  std::array<char, M3HIP_XCH_HANDLE_BYTES> CreateExchange(int Rank, int nRanks) {
    std::array<char, M3HIP_XCH_HANDLE_BYTES> hdl{};
    M3HIP_CALL(h, m3hip_xch_create(h, Rank, nRanks, hdl.data()));
    comm_rank = Rank;
    comm_ranks = nRanks;
    return hdl;
  }
  void OpenExchange(const std::vector<std::array<char, M3HIP_XCH_HANDLE_BYTES>>& Handles, double TimeoutSeconds = 0.0) {
    if (int(Handles.size()) != comm_ranks) throw MaCh3Exception(__FILE__, __LINE__, "OpenExchange: one handle per rank, in rank order");
    M3HIP_CALL(h, m3hip_xch_open(h, Handles.data()));
    if (TimeoutSeconds > 0.0) M3HIP_CALL(h, m3hip_xch_set_timeout(h, TimeoutSeconds));
  }
  /// back to one device (collective for RCCL: every rank calls it)
  void CloseComm() {
    M3HIP_CALL(h, m3hip_xch_destroy(h));
    M3HIP_CALL(h, m3hip_comm_destroy(h));
    comm_rank = 0;
    comm_ranks = 1;
  }
  int GetCommRank() const { return comm_rank; }
  int GetCommSize() const { return comm_ranks; }

  const BinningHandler* GetBinning() const { return Binning.get(); }
  SplineBase* GetSplineHandler() { return SplineHandler.get(); }

 protected:
  void TestStatisticChanged() override { M3HIP_CALL(h, m3hip_set_test_statistic(h, int(fTestStatistic), UpdateW2 ? 1 : 0)); }

 private:
  std::vector<double> Bins(bool mc) const {
    std::vector<double> a(size_t(Binning->GetNBins())), b(a.size());
    M3HIP_CALL(h, m3hip_get_bins(h, a.data(), b.data(), int(a.size())));
    return mc ? a : b;
  }
  std::vector<double> Slice(const std::vector<double>& all, int Sample) const {
    if (all.empty()) return {};
    return std::vector<double>(all.begin() + Binning->GetSampleStartBin(Sample), all.begin() + Binning->GetSampleEndBin(Sample));
  }

  std::string SampleName;
  std::unique_ptr<SplineBase> SplineHandler;
  std::unique_ptr<BinningHandler> Binning;
  const double* fPropVal;
  int nPar;
  bool UpdateW2;
  m3hip_handle h = nullptr;
  static constexpr int kKinStride = 4;
  std::vector<int32_t> bin_idx, osc_idx, sample_id;
  std::vector<double> kin;  // [event][kKinStride]
  bool device_binning = false;
  std::vector<uint64_t> norm_off;
  std::vector<int16_t> norm_idx;
  std::vector<double> data;
  mutable std::vector<double> per_sample;
  unsigned int n_added = 0;
  int comm_rank = 0, comm_ranks = 1;
};
