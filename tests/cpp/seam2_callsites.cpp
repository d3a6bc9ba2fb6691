// Seam 2 drop-in check: the reference's OWN call sites of class SMonolithGPU, in the reference's order and with the
// reference's argument lists, compiled against include/mach3_amd/SplineMonolith.h.  Only the calls are reproduced (they are
// the interface under test); everything around them is scaffolding.  Call sites, Splines/SplineMonolith.cpp:
//   :82-83    InitGPU_Segments / InitGPU_Vals -- through a gpu_spline_handler that is still null
//   :271      gpu_spline_handler = new SMonolithGPU();
//   :276-283  InitGPU_SplineMonolith(&cpu_total_weights, NEvents, nKnots, NSplines_valid, NTF1_valid, event_size_max)
//   :289-303  CopyToGPU_SplineMonolith(cpu_spline_handler, vectors by value ..., nParams, NSplines_valid, _max_knots, nKnots, NTF1_valid)
//   :702-707  RunGPU_SplineMonolith(cpu_total_weights, ParamValues, SplineSegments, NSplines_valid, NTF1_valid)
//   :854      SynchroniseSplines()
//   :621-625  CleanupGPU_SplineMonolith / CleanupGPU_Segments / delete
// Built into build/libm3seam2_test.so and driven from tests/test_cpp_facade.py with arrays from the synthetic generator.
#define MaCh3_CUDA
#include "../../include/mach3_amd/SplineMonolith.h"

namespace {
struct Members {  // the members of SMonolith the call sites use (Splines/SplineMonolith.h:53-137), same names
  SMonolithGPU* gpu_spline_handler = nullptr;
  SplineMonoStruct* cpu_spline_handler = nullptr;
  short int* SplineSegments = nullptr;
  float* ParamValues = nullptr;
  float* cpu_total_weights = nullptr;
  std::vector<float> cpu_coeff_TF1_many;
  std::vector<short int> cpu_paramNo_TF1_arr;
  std::vector<unsigned int> cpu_nParamPerEvent, cpu_nParamPerEvent_tf1;
  unsigned int NEvents = 0, nKnots = 0, NSplines_valid = 0, NTF1_valid = 0;
  short int nParams = 0, _max_knots = 0;
};
}  // namespace

extern "C" int seam2_callsites(unsigned int NEvents, short nParams, short max_knots, const float* coeff_x, const float* coeff_many,
                               unsigned int nKnots, const unsigned int* nKnots_arr, const short* paramNo_arr, unsigned int NSplines_valid,
                               const float* coeff_tf1, const short* paramNo_tf1, unsigned int NTF1_valid, const unsigned int* npe,
                               const unsigned int* npe_tf1, int n_laps, const float* vals, const short* segments, float* weights_out,
                               char* err, int err_len) {
  try {
    Members m;
    m.NEvents = NEvents; m.nParams = nParams; m._max_knots = max_knots; m.nKnots = nKnots;
    m.NSplines_valid = NSplines_valid; m.NTF1_valid = NTF1_valid;
    m.cpu_spline_handler = new SplineMonoStruct();
    m.cpu_spline_handler->coeff_x.assign(coeff_x, coeff_x + size_t(nParams) * max_knots);
    m.cpu_spline_handler->coeff_many.assign(coeff_many, coeff_many + size_t(nKnots) * _nCoeff_);
    m.cpu_spline_handler->nKnots_arr.assign(nKnots_arr, nKnots_arr + NSplines_valid);
    m.cpu_spline_handler->paramNo_arr.assign(paramNo_arr, paramNo_arr + NSplines_valid);
    m.cpu_coeff_TF1_many.assign(coeff_tf1, coeff_tf1 + size_t(NTF1_valid) * _nTF1Coeff_);
    m.cpu_paramNo_TF1_arr.assign(paramNo_tf1, paramNo_tf1 + NTF1_valid);
    m.cpu_nParamPerEvent.assign(npe, npe + 2 * size_t(NEvents));
    m.cpu_nParamPerEvent_tf1.assign(npe_tf1, npe_tf1 + 2 * size_t(NEvents));
    // (aliases so that the statements below read as in the reference)
    SMonolithGPU*& gpu_spline_handler = m.gpu_spline_handler;
    SplineMonoStruct*& cpu_spline_handler = m.cpu_spline_handler;
    short int*& SplineSegments = m.SplineSegments;
    float*& ParamValues = m.ParamValues;
    float*& cpu_total_weights = m.cpu_total_weights;
    std::vector<float>& cpu_coeff_TF1_many = m.cpu_coeff_TF1_many;
    std::vector<short int>& cpu_paramNo_TF1_arr = m.cpu_paramNo_TF1_arr;
    std::vector<unsigned int>&cpu_nParamPerEvent = m.cpu_nParamPerEvent, &cpu_nParamPerEvent_tf1 = m.cpu_nParamPerEvent_tf1;
    const short _max_knots = max_knots;

    // ---- Splines/SplineMonolith.cpp:82-83 (PrepareForGPU; gpu_spline_handler is still nullptr here)
    gpu_spline_handler->InitGPU_Segments(&SplineSegments);
    gpu_spline_handler->InitGPU_Vals(&ParamValues);
    for (M3::int_t j = 0; j < nParams; j++) {  // :89-93
      SplineSegments[j] = 0;
      ParamValues[j] = -999;
    }
    SMonolithGPU::upload_slice_events() = 1024;  // scaffolding: three upload slices for this small sample
    // ---- :262-303 (MoveToGPU)
    unsigned int event_size_max = _max_knots * nParams;
    gpu_spline_handler = new SMonolithGPU();
    gpu_spline_handler->InitGPU_SplineMonolith(
            &cpu_total_weights,
            NEvents,
            nKnots,
            NSplines_valid,
            NTF1_valid,
            event_size_max);
    gpu_spline_handler->CopyToGPU_SplineMonolith(
            cpu_spline_handler,
            cpu_coeff_TF1_many,
            cpu_paramNo_TF1_arr,
            NEvents,
            cpu_nParamPerEvent,
            cpu_nParamPerEvent_tf1,
            nParams,
            NSplines_valid,
            _max_knots,
            nKnots,
            NTF1_valid);
    // :305-310
    std::vector<float>().swap(cpu_coeff_TF1_many);
    std::vector<short int>().swap(cpu_paramNo_TF1_arr);
    std::vector<unsigned int>().swap(cpu_nParamPerEvent);
    std::vector<unsigned int>().swap(cpu_nParamPerEvent_tf1);
    delete cpu_spline_handler;
    cpu_spline_handler = nullptr;

    for (int lap = 0; lap < n_laps; ++lap) {
      // (FindSplineSegment's result is handed in by the test: it is checked on its own elsewhere)
      for (int j = 0; j < nParams; ++j) {
        ParamValues[j] = vals[size_t(lap) * nParams + j];
        SplineSegments[j] = segments[size_t(lap) * nParams + j];
      }
      // ---- :702-707 (Evaluate)
      gpu_spline_handler->RunGPU_SplineMonolith(
              cpu_total_weights,
              ParamValues,
              SplineSegments,
              NSplines_valid,
              NTF1_valid);
      // ---- :854 (SynchroniseMemTransfer)
      SynchroniseSplines();
      std::memcpy(weights_out + size_t(lap) * NEvents, cpu_total_weights, sizeof(float) * NEvents);
    }
    // ---- :621-625 (~SMonolith)
    gpu_spline_handler->CleanupGPU_SplineMonolith(cpu_total_weights);
    gpu_spline_handler->CleanupGPU_Segments(SplineSegments, ParamValues);
    delete gpu_spline_handler;
    return 0;
  } catch (const std::exception& e) {
    std::snprintf(err, size_t(err_len), "%s", e.what());
    return 1;
  }
}
