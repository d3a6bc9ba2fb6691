// BinnedSplineHandler.h -- the per-step part of the reference's bin-by-bin spline handler on the
// GPU (ref: Splines/BinnedSplineHandler.h:11-165, .cpp:295-341).
//
// Only Evaluate() and the weight look-up exist here.  Loading splines from ROOT files, the
// 7-dimensional indexvec and the mode / oscillation-channel bookkeeping
// (BinnedSplineHandler.cpp:69-292, 343-1090) are set-up code outside the per-step path: the
// caller hands over the flat members that code produces.
//
// M3::float_t: BinnedSplineHandlerT<float> is the MaCh3_LOW_MEMORY_STRUCTS build (the name
// BinnedSplineHandler), BinnedSplineHandlerT<double> the reference's default build (Manager/Core.h:27-51):
// double tables, std::fma cubic, and a double weight product in the sample that uses it.
#pragma once

#include <algorithm>
#include <array>
#include <limits>
#include <type_traits>
#include <vector>

#include "../m3hip.h"
#include "Core.h"
#include "SplineMonolith.h"

template <class FloatT>
class BinnedSplineHandlerT : public SplineBase {
  static_assert(std::is_same<FloatT, float>::value || std::is_same<FloatT, double>::value, "M3::float_t is float or double");

 public:
  /// the handler's flat members (BinnedSplineHandler.h:116-140) + the knots of every unique
  /// spline parameter (SplineInfoArray) and the number of MC events of the sample
  BinnedSplineHandlerT(unsigned int n_events, short n_unique_params, short max_knots, const std::vector<FloatT>& xpts,
                       const std::vector<short>& npts, const std::vector<int>& coeffindexvec_,
                       const std::vector<int>& uniquesplinevec_Monolith_, const std::vector<int>& uniquecoeffindices_,
                       const std::vector<FloatT>& xcoeff_arr, const std::vector<FloatT>& manycoeff_arr, int device_id = 0)
      : gpu(SMonolithGPU::make_handler(device_id)), MonolithSize(int(coeffindexvec_.size())), nEvents(n_events) {
    nParams = n_unique_params;
    M3HIP_CALL(gpu->Handle(), Upload(gpu->Handle(), n_events, nParams, max_knots, xpts.data(), npts.data(), MonolithSize, coeffindexvec_.data(),
                                     uniquesplinevec_Monolith_.data(), int(uniquecoeffindices_.size()), uniquecoeffindices_.data(),
                                     int(xcoeff_arr.size()), xcoeff_arr.data(), manycoeff_arr.data()));
    weightvec_Monolith.assign(size_t(MonolithSize), FloatT(1));
  }
  /// In the fused sample operator the weights never leave the device; Evaluate() is what
  /// SampleHandlerFD::Reweight() triggers.  Stand-alone it refreshes the host copy.
  void Evaluate() override {}
  std::string GetName() const override { return "BinnedSplineHandler"; }
  void SynchroniseMemTransfer() const override {}
  void PrepareSplineFile(std::string) override { throw MaCh3Exception(__FILE__, __LINE__, "not part of the per-step path"); }
  void LoadSplineFile(std::string) override { throw MaCh3Exception(__FILE__, __LINE__, "not part of the per-step path"); }
  /// events' pointers into weightvec_Monolith, as indices (SampleHandlerFD::SetSplinePointers,
  /// Samples/SampleHandlerFD.cpp:1196-1243): CSR over events
  void SetEventSplines(const std::vector<uint64_t>& w_off, const std::vector<int32_t>& w_idx) {
    M3HIP_CALL(gpu->Handle(), m3hip_binned_upload_events(gpu->Handle(), 0, w_off.size() - 1, w_off.data(), w_idx.data()));
  }
  /// Event order that makes the per-event gathers of bin-spline weights cache-friendly: events that use the
  /// same bin-splines next to each other (sorted by their first weight index, stable).  Registering the MC in
  /// this order -- SampleHandlerFD::AddEvent and SetEventSplines alike -- made the event kernel 7.5x faster on
  /// 1e6 events x 30 weights (164 -> 22 us); any order gives the same bins up to fp64 summation order.
  static std::vector<uint32_t> EventOrderByCell(const std::vector<uint64_t>& w_off, const std::vector<int32_t>& w_idx) {
    std::vector<uint32_t> order(w_off.size() - 1);
    for (size_t e = 0; e < order.size(); ++e) order[e] = uint32_t(e);
    auto key = [&](uint32_t e) { return w_off[e + 1] > w_off[e] ? w_idx[w_off[e]] : std::numeric_limits<int32_t>::max(); };
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key(a) < key(b); });
    return order;
  }
  /// weightvec_Monolith after the last Reweight() (device -> host)
  const std::vector<FloatT>& GetWeights() {
    M3HIP_CALL(gpu->Handle(), Download(gpu->Handle(), weightvec_Monolith.data(), MonolithSize));
    return weightvec_Monolith;
  }
  m3hip_handle Handle() const override { return gpu->Handle(); }
  unsigned int GetNEvents() const override { return nEvents; }
  /// where the handler's unique spline parameters live in the parameter vector (setSplinePointers)
  void SetSplineParameterIndices(const std::vector<int32_t>& idx) { M3HIP_CALL(gpu->Handle(), m3hip_set_spline_param_map(gpu->Handle(), idx.data(), int(idx.size()))); }

 private:
  template <class... A>
  static int Upload(m3hip_handle h, uint64_t n, int np, int mk, const float* x, A... rest) { return m3hip_binned_upload(h, n, np, mk, x, rest...); }
  template <class... A>
  static int Upload(m3hip_handle h, uint64_t n, int np, int mk, const double* x, A... rest) { return m3hip_binned_upload_f64(h, n, np, mk, x, rest...); }
  static int Download(m3hip_handle h, float* out, int n) { return m3hip_get_binned_weights(h, out, n); }
  static int Download(m3hip_handle h, double* out, int n) { return m3hip_get_binned_weights_f64(h, out, n); }

  std::unique_ptr<SMonolithGPU> gpu;
  int MonolithSize;
  unsigned int nEvents;
  std::vector<FloatT> weightvec_Monolith;
};

using BinnedSplineHandler = BinnedSplineHandlerT<float>;

/// Which bin-splines an event points at: the handler's spline bookkeeping of ONE (sample, oscillation channel) and the look-up
/// SampleHandlerFD::SetSplinePointers makes per event (ref: BinnedSplineHandler::GetEventSplines, Splines/BinnedSplineHandler.cpp:832-884;
/// the fan-out to total_weight_pointers, Samples/SampleHandlerFD.cpp:1196-1243).  The reference keeps a seven-dimensional
/// indexvec[sample][channel][syst][mode][b1][b2][b3] filled from ROOT files; here the caller registers, per systematic, its spline
/// modes and the index in weightvec_Monolith of every (mode, bin1, bin2, bin3), and gets the event's CSR row for SetEventSplines.
class BinnedSplineIndex {
 public:
  /// the three spline axes (true energy, first and second kinematic variable; a 1-D sample's third axis is one bin around 0):
  /// ascending edges, bins [lo, hi) as TAxis::FindBin counts them
  BinnedSplineIndex(std::vector<double> Axis1, std::vector<double> Axis2, std::vector<double> Axis3) : Axis{std::move(Axis1), std::move(Axis2), std::move(Axis3)} {
    for (const auto& a : Axis)
      if (a.size() < 2 || !std::is_sorted(a.begin(), a.end())) throw MaCh3Exception(__FILE__, __LINE__, "BinnedSplineIndex: axis edges must ascend");
  }
  int GetNBins(int axis) const { return int(Axis[size_t(axis)].size()) - 1; }
  /// a systematic of this (sample, channel): the spline modes it has splines for (SplineModeVecs) and, mode-major, the
  /// weightvec_Monolith index of every bin -- SplineID[((iMode * n1 + b1) * n2 + b2) * n3 + b3]
  int AddSystematic(std::vector<int> SplineModes, std::vector<int32_t> SplineID) {
    const size_t cells = size_t(GetNBins(0)) * size_t(GetNBins(1)) * size_t(GetNBins(2));
    if (SplineID.size() != SplineModes.size() * cells) throw MaCh3Exception(__FILE__, __LINE__, "BinnedSplineIndex: one spline index per (mode, bin1, bin2, bin3)");
    // the reference's indexvec holds valid monolith indices only (BinnedSplineHandler.cpp:1100-1140): enforce it here, where the
    // numbers come in, so that GetEventSplines can index the handler's arrays with them
    for (const int32_t id : SplineID)
      if (id < 0) throw MaCh3Exception(__FILE__, __LINE__, "BinnedSplineIndex: negative spline index (every (mode, bin) of a registered systematic needs its weightvec_Monolith index)");
    Modes.push_back(std::move(SplineModes));
    Index.push_back(std::move(SplineID));
    return int(Modes.size()) - 1;
  }
  /// TAxis::FindBin(v) - 1: -1 below the first edge, GetNBins at or above the last
  int FindBin(int axis, double v) const {
    const auto& e = Axis[size_t(axis)];
    return int(std::upper_bound(e.begin(), e.end(), v) - e.begin()) - 1;
  }
  /// The event's bin-splines, systematic after systematic (the order total_weight_pointers gets them in): for every systematic
  /// whose spline modes contain `SplineMode`, the spline of the event's (bin1, bin2, bin3), unless it is flat.  An event outside
  /// any of the three axes gets none (:851-864).  `isflatarray` is the handler's, indexed like weightvec_Monolith (may be null).
  std::vector<int32_t> GetEventSplines(int SplineMode, double Var1Val, double Var2Val, double Var3Val, const std::vector<char>* isflatarray = nullptr) const {
    std::vector<int32_t> out;
    if (SplineMode < 0) return out;
    const int b1 = FindBin(0, Var1Val), b2 = FindBin(1, Var2Val), b3 = FindBin(2, Var3Val);
    if (b1 < 0 || b1 >= GetNBins(0) || b2 < 0 || b2 >= GetNBins(1) || b3 < 0 || b3 >= GetNBins(2)) return out;
    for (size_t iSyst = 0; iSyst < Modes.size(); ++iSyst)
      for (size_t iMode = 0; iMode < Modes[iSyst].size(); ++iMode)
        if (Modes[iSyst][iMode] == SplineMode) {
          const int32_t id = Index[iSyst][((iMode * size_t(GetNBins(0)) + size_t(b1)) * size_t(GetNBins(1)) + size_t(b2)) * size_t(GetNBins(2)) + size_t(b3)];
          if (isflatarray && size_t(id) >= isflatarray->size())
            throw MaCh3Exception(__FILE__, __LINE__, "BinnedSplineIndex::GetEventSplines: spline index " + std::to_string(id) + " beyond the handler's " +
                                                         std::to_string(isflatarray->size()) + " bin-splines");
          if (!(isflatarray && (*isflatarray)[size_t(id)])) out.push_back(id);
        }
    return out;
  }

 private:
  std::array<std::vector<double>, 3> Axis;
  std::vector<std::vector<int>> Modes;     // [syst] spline modes
  std::vector<std::vector<int32_t>> Index;  // [syst][mode][b1][b2][b3]
};
