// BinnedSplineHandler.h -- the per-step part of the reference's bin-by-bin spline handler on the
// GPU (ref: Splines/BinnedSplineHandler.h:11-165, .cpp:295-341).
//
// Only Evaluate() and the weight look-up exist here.  Loading splines from ROOT files, the
// 7-dimensional indexvec and the mode / oscillation-channel bookkeeping
// (BinnedSplineHandler.cpp:69-292, 343-1090) are set-up code outside the per-step path: the
// caller hands over the flat members that code produces.  float_t = float build.
#pragma once

#include <vector>

#include "../m3hip.h"
#include "Core.h"
#include "SplineMonolith.h"

class BinnedSplineHandler : public SplineBase {
 public:
  /// the handler's flat members (BinnedSplineHandler.h:116-140) + the knots of every unique
  /// spline parameter (SplineInfoArray) and the number of MC events of the sample
  BinnedSplineHandler(unsigned int n_events, short n_unique_params, short max_knots, const std::vector<float>& xpts,
                      const std::vector<short>& npts, const std::vector<int>& coeffindexvec_,
                      const std::vector<int>& uniquesplinevec_Monolith_, const std::vector<int>& uniquecoeffindices_,
                      const std::vector<float>& xcoeff_arr, const std::vector<float>& manycoeff_arr, int device_id = 0)
      : gpu(new SMonolithGPU(device_id)), MonolithSize(int(coeffindexvec_.size())) {
    nParams = n_unique_params;
    M3HIP_CALL(gpu->Handle(), m3hip_binned_upload(gpu->Handle(), n_events, nParams, max_knots, xpts.data(), npts.data(), MonolithSize,
                                                  coeffindexvec_.data(), uniquesplinevec_Monolith_.data(), int(uniquecoeffindices_.size()),
                                                  uniquecoeffindices_.data(), int(xcoeff_arr.size()), xcoeff_arr.data(), manycoeff_arr.data()));
    weightvec_Monolith.assign(size_t(MonolithSize), 1.f);
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
  /// weightvec_Monolith after the last Reweight() (device -> host)
  const std::vector<float>& GetWeights() {
    M3HIP_CALL(gpu->Handle(), m3hip_get_binned_weights(gpu->Handle(), weightvec_Monolith.data(), MonolithSize));
    return weightvec_Monolith;
  }
  m3hip_handle Handle() const { return gpu->Handle(); }

 private:
  std::unique_ptr<SMonolithGPU> gpu;
  int MonolithSize;
  std::vector<float> weightvec_Monolith;
};
