// SplineMonolith.h -- SplineBase / SMonolith / SMonolithGPU on top of the C ABI (m3hip.h).
//
// Class and method names follow the reference so SMonolith drops into
// SampleHandlerFD::SplineHandler (ref: Samples/SampleHandlerFD.h:214) unchanged:
//   SplineBase    Splines/SplineBase.h:27-93
//   SMonolith     Splines/SplineMonolith.h:11-149
//   SMonolithGPU  Splines/gpuSplineUtils.cuh:63-218 (the backend class; here every method is
//                 a thin call into libm3hip.so instead of CUDA)
// Differences, all forced by removing ROOT: PrepareSplineFile / LoadSplineFile read and write
// a raw binary dump of the flattened arrays (the reference stores a ROOT TTree,
// SplineMonolith.cpp:454-614), and n_params is a run-time value.
#pragma once

#include <cstdint>
#include <cstring>
#include <algorithm>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "../m3hip.h"
#include "Core.h"
#include "SplineStructs.h"

constexpr int _nCoeff_ = 4;     // ref: Splines/SplineCommon.h:13
constexpr int _nTF1Coeff_ = 2;  // ref: Splines/SplineCommon.h:15

/// ref: Splines/SplineCommon.h:30-50
struct SplineMonoStruct {
  std::vector<float> coeff_x;
  std::vector<float> coeff_many;
  std::vector<unsigned int> nKnots_arr;
  std::vector<short int> paramNo_arr;
};

#define M3HIP_CALL(handle, expr)                                                              \
  do {                                                                                        \
    if ((expr) != M3HIP_OK) throw MaCh3Exception(__FILE__, __LINE__, m3hip_last_error(handle)); \
  } while (0)

/// SynchroniseSplines(), a free function in the reference (ref: Splines/gpuSplineUtils.cu:71-73)
inline void SynchroniseSplines() { M3HIP_CALL(nullptr, m3hip_device_synchronize()); }

/// The GPU backend: same methods, same argument lists as the reference's class (ref: Splines/gpuSplineUtils.cuh:63-218),
/// so the call sites of Splines/SplineMonolith.cpp:82-83, 271-303, 706 and 619-637 compile against it unchanged
/// (tests/cpp/seam2_callsites.cpp).  Every method is a thin call into libm3hip.so.
class SMonolithGPU {
 public:
  SMonolithGPU() {
    if (m3hip_create(default_device(), &h) != M3HIP_OK) throw MaCh3Exception(__FILE__, __LINE__, m3hip_last_error(nullptr));
  }
  virtual ~SMonolithGPU() { m3hip_destroy(h); }
  SMonolithGPU(const SMonolithGPU&) = delete;
  SMonolithGPU& operator=(const SMonolithGPU&) = delete;
  /// device of the SMonolithGPU objects created from now on (the reference has one device, Manager/gpuUtils.cu:71)
  static int& default_device() {
    static int device = 0;
    return device;
  }
  static SMonolithGPU* make_handler(int device_id) {
    default_device() = device_id;
    return new SMonolithGPU();
  }

  /// pinned cpu_total_weights (ref: gpuSplineUtils.cu:116-171; the device buffers are sized at upload)
  void InitGPU_SplineMonolith(float** cpu_total_weights, int n_events, unsigned int /*total_nknots*/,
                              unsigned int /*n_splines*/, unsigned int /*n_tf1*/, int /*Eve_size*/) {
    M3HIP_CALL(h, m3hip_alloc_pinned(h, reinterpret_cast<void**>(cpu_total_weights), sizeof(float) * size_t(n_events)));
  }
  /// Pinned segment / value arrays.  The reference calls these two through a gpu_spline_handler that is still null
  /// (SplineMonolith.cpp:82-83, before `new SMonolithGPU()` at :271): they must not touch `this`, and they cannot
  /// know the number of parameters -- so they are sized for the most a `short` parameter index can address.
  void InitGPU_Segments(short int** segment) {
    M3HIP_CALL(nullptr, m3hip_alloc_pinned(nullptr, reinterpret_cast<void**>(segment), sizeof(short) * size_t(32767)));
  }
  void InitGPU_Vals(float** vals) {
    M3HIP_CALL(nullptr, m3hip_alloc_pinned(nullptr, reinterpret_cast<void**>(vals), sizeof(float) * size_t(32767)));
  }
  /// one-time upload (ref: gpuSplineUtils.cu:174-333), vectors by value as in the reference.  The knot count of a
  /// parameter (FastSplineInfo.nPts, which the engine's re-layout needs and the reference's argument list lacks) is
  /// recovered from the arrays themselves: spline s owns knots nKnots_arr[s] .. nKnots_arr[s+1], and all splines of
  /// a parameter have the same count (ref: SplineMonolith.cpp:162-164).  Events go up in slices (multiples of the
  /// engine's 1024-event groups), so the pinned staging stays bounded whatever the size of the sample; the
  /// {count,start} pairs index the shared arrays, so no data moves on the host.
  void CopyToGPU_SplineMonolith(SplineMonoStruct* cpu_spline_handler, std::vector<float> cpu_many_array_TF1,
                                std::vector<short int> cpu_paramNo_arr_TF1, int n_events, std::vector<unsigned int> cpu_nParamPerEvent,
                                std::vector<unsigned int> cpu_nParamPerEvent_TF1, int n_params, unsigned int n_splines,
                                short int spline_size, unsigned int total_nknots, unsigned int n_tf1) {
    std::vector<short int> npts(size_t(n_params), 0);
    for (unsigned int s = 0; s < n_splines; ++s) {
      const unsigned int next = s + 1 < n_splines ? cpu_spline_handler->nKnots_arr[s + 1] : total_nknots;
      const short int n = short(next - cpu_spline_handler->nKnots_arr[s]);
      short int& np = npts[size_t(cpu_spline_handler->paramNo_arr[s])];
      if (np != 0 && np != n) throw MaCh3Exception(__FILE__, __LINE__, "splines of one parameter differ in their number of knots");
      np = n;
    }
    M3HIP_CALL(h, m3hip_monolith_expect_events(h, uint64_t(n_events)));  // the total is known here; the engine's streamed upload cannot see it
    M3HIP_CALL(h, m3hip_monolith_begin(h, n_params, spline_size, cpu_spline_handler->coeff_x.data(), npts.data()));
    // an order key (next_event_order_key) is applied slice by slice: slices of at least 2^20 events, so that the events of a
    // 2e5-entry oscillation table's line end up next to each other (the slices index shared arrays: their size costs nothing)
    const std::vector<int32_t> key = std::move(next_event_order_key());
    next_event_order_key().clear();
    if (!key.empty() && key.size() != size_t(n_events)) throw MaCh3Exception(__FILE__, __LINE__, "the event order key is not one entry per event");
    const uint64_t slice = key.empty() ? upload_slice_events() : std::max<uint64_t>(upload_slice_events(), uint64_t(1) << 20);
    for (uint64_t e0 = 0; e0 < uint64_t(n_events); e0 += slice) {
      const uint64_t n = std::min<uint64_t>(slice, uint64_t(n_events) - e0);
      if (!key.empty()) M3HIP_CALL(h, m3hip_monolith_order_key(h, key.data() + e0, n));
      M3HIP_CALL(h, m3hip_monolith_append(h, n, cpu_spline_handler->coeff_many.data(), total_nknots, cpu_spline_handler->nKnots_arr.data(),
                                          cpu_spline_handler->paramNo_arr.data(), n_splines, cpu_many_array_TF1.data(),
                                          cpu_paramNo_arr_TF1.data(), n_tf1, cpu_nParamPerEvent.data() + 2 * e0,
                                          cpu_nParamPerEvent_TF1.data() + 2 * e0));
    }
    M3HIP_CALL(h, m3hip_monolith_end(h));
  }
  /// per step, asynchronous (ref: gpuSplineUtils.cu:444-512)
  void RunGPU_SplineMonolith(float* cpu_total_weights, float* vals, short int* segment, const unsigned int /*h_n_splines*/,
                             const unsigned int /*h_n_tf1*/) {
    M3HIP_CALL(h, m3hip_eval_splines(h, vals, segment, cpu_total_weights));
  }
  void CleanupGPU_SplineMonolith(float* cpu_total_weights) { m3hip_free_pinned(h, cpu_total_weights); }
  void CleanupGPU_Segments(short int* segment, float* vals) {
    m3hip_free_pinned(nullptr, segment);
    m3hip_free_pinned(nullptr, vals);
  }

  m3hip_handle Handle() const { return h; }
  /// Optional, before the SMonolith (or SMonolithGPU upload) it is meant for: a key per event -- the events' oscillation bins --
  /// by which the engine may ORDER the events in its own memory (m3hip_monolith_order_key; consumed by the next upload).  Nothing
  /// changes for the caller: AddEvent, cpu_total_weights and every per-event result keep the MC's numbering.  What it buys: a
  /// wavefront's look-ups into a large oscillation table fall into a few lines -- the step of a fit that MOVES oscillation
  /// parameters runs as fast as one whose table rests (one GPU's share of BASELINE configs[4]: 481 -> 440 us per kernel).
  static std::vector<int32_t>& next_event_order_key() {
    static std::vector<int32_t> key;
    return key;
  }
  /// events per upload slice (a multiple of 1024); settable for tests
  static uint64_t& upload_slice_events() {
    static uint64_t n = 131072;
    return n;
  }

 private:
  m3hip_handle h = nullptr;
};

/// ref: Splines/SplineBase.h:27-93
class SplineBase {
 public:
  SplineBase() {}
  virtual ~SplineBase() {}
  virtual void Evaluate() = 0;
  virtual std::string GetName() const { return "SplineBase"; }
  short int GetNParams() const { return nParams; }
  virtual void PrepareSplineFile(std::string FileName) = 0;
  virtual void LoadSplineFile(std::string FileName) = 0;
  virtual void SynchroniseMemTransfer() const = 0;
  /// the engine handle behind the operator and the number of MC events it spans: what SampleHandlerFD needs from
  /// whichever spline handler it was given (event-by-event SMonolith or BinnedSplineHandler, as in the reference
  /// where SampleHandlerFD holds a unique_ptr<SplineBase>, Samples/SampleHandlerFD.h:214)
  virtual m3hip_handle Handle() const = 0;
  virtual unsigned int GetNEvents() const = 0;

 protected:
  /// once per step and per PARAMETER, never per event (ref: Splines/SplineBase.cpp:44-109)
  void FindSplineSegment() {
    for (M3::int_t i = 0; i < nParams; ++i) {
      FastSplineInfo& info = SplineInfoArray[i];
      const float xvar = float(*info.splineParsPointer);
      ParamValues[i] = xvar;
      const std::vector<M3::float_t>& x = info.xPts;
      if (x.empty()) continue;
      const int n = info.nPts;
      int seg = 0, hi = n - 1;
      const int prev = info.CurrSegment;
      if (xvar <= x[0]) {
        seg = 0;
      } else if (xvar >= x[n - 1]) {
        seg = hi;
      } else if (x[prev + 1] > xvar && xvar >= x[prev]) {
        seg = prev;
      } else {
        while (hi - seg > 1) {
          const int mid = (seg + hi) / 2;
          if (xvar > x[mid]) seg = mid;
          else hi = mid;
        }
      }
      if (seg >= n - 1 && n > 1) seg = n - 2;
      info.CurrSegment = M3::int_t(seg);
      SplineSegments[i] = short(seg);
    }
  }

  std::vector<FastSplineInfo> SplineInfoArray;
  short int* SplineSegments = nullptr;  ///< pinned host memory
  float* ParamValues = nullptr;         ///< pinned host memory
  short int nParams = 0;
};

/// Event-by-event spline monolith on the GPU (ref: Splines/SplineMonolith.h:11-149).
class SMonolith : public SplineBase {
 public:
  /// Takes ownership of (and deletes) every TResponseFunction_red*, like the reference
  /// (ref: Splines/SplineMonolith.cpp:204-207).
  SMonolith(std::vector<std::vector<TResponseFunction_red*>>& MasterSpline, const std::vector<RespFuncType>& SplineType,
            const bool SaveFlatTree = false, const std::string& _FastSplineName = "SplineFile.bin", int device_id = 0)
      : gpu_spline_handler(SMonolithGPU::make_handler(device_id)) {
    PrepareForGPU(MasterSpline, SplineType);
    if (SaveFlatTree) PrepareSplineFile(_FastSplineName);
    MoveToGPU();
  }
  /// from a flat dump written by PrepareSplineFile
  explicit SMonolith(const std::string& FileName, int device_id = 0) : gpu_spline_handler(SMonolithGPU::make_handler(device_id)) {
    LoadSplineFile(FileName);
    MoveToGPU();
  }
  ~SMonolith() override {
    if (gpu_spline_handler) {
      gpu_spline_handler->CleanupGPU_SplineMonolith(cpu_total_weights);
      gpu_spline_handler->CleanupGPU_Segments(SplineSegments, ParamValues);
    }
  }

  /// host FindSplineSegment, then one asynchronous fused kernel (ref: SplineMonolith.cpp:695-708)
  void Evaluate() override {
    FindSplineSegment();
    gpu_spline_handler->RunGPU_SplineMonolith(cpu_total_weights, ParamValues, SplineSegments, NSplines_valid, NTF1_valid);
  }
  std::string GetName() const override { return "SplineMonolith"; }
  void SynchroniseMemTransfer() const override { M3HIP_CALL(gpu_spline_handler->Handle(), m3hip_sync(gpu_spline_handler->Handle())); }  // ref: :851-856
  const float* retPointer(const int event) const { return &cpu_total_weights[event]; }
  void setSplinePointers(std::vector<const double*> spline_ParsPointers) {
    for (M3::int_t i = 0; i < nParams; ++i) SplineInfoArray[i].splineParsPointer = spline_ParsPointers[i];
  }
  unsigned int GetNEvents() const override { return NEvents; }
  m3hip_handle Handle() const override { return gpu_spline_handler->Handle(); }

  /// raw binary dump / reload of the flattened monolith (the reference writes ROOT trees,
  /// ref: SplineMonolith.cpp:454-614)
  void PrepareSplineFile(std::string FileName) override {
    std::ofstream f(FileName, std::ios::binary);
    if (!f) throw MaCh3Exception(__FILE__, __LINE__, "cannot write " + FileName);
    const uint64_t hdr[8] = {0x4D334D4F4E4F4C31ull, NEvents, uint64_t(nParams), uint64_t(_max_knots), NSplines_valid, NTF1_valid, nKnots, 0};
    f.write(reinterpret_cast<const char*>(hdr), sizeof(hdr));
    auto put = [&](const void* p, size_t bytes) { f.write(reinterpret_cast<const char*>(p), std::streamsize(bytes)); };
    put(flat.coeff_x.data(), flat.coeff_x.size() * 4);
    put(npts.data(), npts.size() * 2);
    put(flat.coeff_many.data(), flat.coeff_many.size() * 4);
    put(flat.nKnots_arr.data(), flat.nKnots_arr.size() * 4);
    put(flat.paramNo_arr.data(), flat.paramNo_arr.size() * 2);
    put(cpu_coeff_TF1_many.data(), cpu_coeff_TF1_many.size() * 4);
    put(cpu_paramNo_TF1_arr.data(), cpu_paramNo_TF1_arr.size() * 2);
    put(cpu_nParamPerEvent.data(), cpu_nParamPerEvent.size() * 4);
    put(cpu_nParamPerEvent_tf1.data(), cpu_nParamPerEvent_tf1.size() * 4);
  }
  void LoadSplineFile(std::string FileName) override {
    std::ifstream f(FileName, std::ios::binary);
    uint64_t hdr[8];
    if (!f || !f.read(reinterpret_cast<char*>(hdr), sizeof(hdr)) || hdr[0] != 0x4D334D4F4E4F4C31ull)
      throw MaCh3Exception(__FILE__, __LINE__, "not a spline monolith dump: " + FileName);
    NEvents = unsigned(hdr[1]); nParams = short(hdr[2]); _max_knots = short(hdr[3]);
    NSplines_valid = unsigned(hdr[4]); NTF1_valid = unsigned(hdr[5]); nKnots = unsigned(hdr[6]);
    auto get = [&](void* p, size_t bytes) { if (!f.read(reinterpret_cast<char*>(p), std::streamsize(bytes))) throw MaCh3Exception(__FILE__, __LINE__, "truncated dump " + FileName); };
    flat.coeff_x.resize(size_t(nParams) * _max_knots); get(flat.coeff_x.data(), flat.coeff_x.size() * 4);
    npts.resize(nParams); get(npts.data(), npts.size() * 2);
    flat.coeff_many.resize(size_t(nKnots) * _nCoeff_); get(flat.coeff_many.data(), flat.coeff_many.size() * 4);
    flat.nKnots_arr.resize(NSplines_valid); get(flat.nKnots_arr.data(), flat.nKnots_arr.size() * 4);
    flat.paramNo_arr.resize(NSplines_valid); get(flat.paramNo_arr.data(), flat.paramNo_arr.size() * 2);
    cpu_coeff_TF1_many.resize(size_t(NTF1_valid) * _nTF1Coeff_); get(cpu_coeff_TF1_many.data(), cpu_coeff_TF1_many.size() * 4);
    cpu_paramNo_TF1_arr.resize(NTF1_valid); get(cpu_paramNo_TF1_arr.data(), cpu_paramNo_TF1_arr.size() * 2);
    cpu_nParamPerEvent.resize(2 * size_t(NEvents)); get(cpu_nParamPerEvent.data(), cpu_nParamPerEvent.size() * 4);
    cpu_nParamPerEvent_tf1.resize(2 * size_t(NEvents)); get(cpu_nParamPerEvent_tf1.data(), cpu_nParamPerEvent_tf1.size() * 4);
    SplineInfoArray.assign(nParams, FastSplineInfo());
    for (int p = 0; p < nParams; ++p)
      if (npts[p] > 0) {
        SplineInfoArray[p].nPts = npts[p];
        SplineInfoArray[p].xPts.assign(flat.coeff_x.begin() + size_t(p) * _max_knots, flat.coeff_x.begin() + size_t(p) * _max_knots + npts[p]);
      }
  }

  /// Event order that puts events with the same list of response objects next to each other (stable; events
  /// without any response last).  The key is the set of parameters for which the event carries a response object
  /// at all -- flat one-knot splines included: events of one interaction mode / channel share that set, and the
  /// flat ones, which PrepareForGPU drops, then only leave holes that the engine pads with identity responses.
  /// The engine keeps the order it is given and streams 256-event tiles with a common (padded) list at ~83 % of
  /// the HBM peak, ragged stretches at ~57 %: building the MasterSpline -- and registering the events with
  /// SampleHandlerFD -- in this order puts a sample on the fast path.  Bins and likelihood do not depend on the
  /// order beyond fp64 summation order.  Call before the constructor, which consumes the MasterSpline.
  static std::vector<uint32_t> EventOrderByParameterList(const std::vector<std::vector<TResponseFunction_red*>>& MasterSpline) {
    const size_t n = MasterSpline.size();
    std::vector<std::vector<short>> key(n);
    for (size_t e = 0; e < n; ++e)
      for (size_t p = 0; p < MasterSpline[e].size(); ++p)
        if (MasterSpline[e][p]) key[e].push_back(short(p));
    std::vector<uint32_t> order(n);
    for (size_t e = 0; e < n; ++e) order[e] = uint32_t(e);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
      if (key[a].empty() != key[b].empty()) return key[b].empty();
      return key[a] < key[b];
    });
    return order;
  }

  /// KS: total weights read by SampleHandlerFD (pinned host memory, filled asynchronously)
  float* cpu_total_weights = nullptr;

 private:
  /// Flatten the MasterSpline into the {x}, {y,b,c,d}, paramNo, knot-offset and {count,start}
  /// arrays (what the reference's PrepareForGPU + ScanMasterSpline produce,
  /// ref: Splines/SplineMonolith.cpp:53-250, 318-439).  Flat (one-knot) splines are dropped.
  void PrepareForGPU(std::vector<std::vector<TResponseFunction_red*>>& MasterSpline, const std::vector<RespFuncType>& SplineType) {
    NEvents = unsigned(MasterSpline.size());
    if (NEvents == 0) throw MaCh3Exception(__FILE__, __LINE__, "empty MasterSpline");
    nParams = short(MasterSpline[0].size());
    SplineInfoArray.assign(nParams, FastSplineInfo());
    npts.assign(nParams, short(-999));
    _max_knots = 0;
    for (auto& ev : MasterSpline) {
      if (short(ev.size()) != nParams) throw MaCh3Exception(__FILE__, __LINE__, "events carry different numbers of spline parameters");
      for (int p = 0; p < nParams; ++p)
        if (ev[p] && SplineType[p] == kTSpline3_red && ev[p]->GetNp() > _max_knots) _max_knots = ev[p]->GetNp();
    }
    if (_max_knots < 1) _max_knots = 1;
    flat.coeff_x.assign(size_t(nParams) * _max_knots, -999.f);
    cpu_nParamPerEvent.assign(2 * size_t(NEvents), 0u);
    cpu_nParamPerEvent_tf1.assign(2 * size_t(NEvents), 0u);
    unsigned knot = 0;
    for (unsigned e = 0; e < NEvents; ++e) {
      const unsigned s0 = unsigned(flat.paramNo_arr.size()), t0 = unsigned(cpu_paramNo_TF1_arr.size());
      for (int p = 0; p < nParams; ++p) {
        TResponseFunction_red* rf = MasterSpline[e][p];
        if (!rf) continue;
        if (SplineType[p] == kTSpline3_red) {
          TSpline3_red* spl = static_cast<TSpline3_red*>(rf);
          const int np = spl->GetNp();
          if (np > 1) {  // one knot = flat response, ignored (ref: SplineMonolith.cpp:161)
            FastSplineInfo& info = SplineInfoArray[p];
            if (info.xPts.empty()) {
              info.nPts = M3::int_t(np);
              info.xPts.resize(np);
              npts[p] = short(np);
            } else if (info.nPts != np) {
              throw MaCh3Exception(__FILE__, __LINE__, "splines of parameter " + std::to_string(p) + " have different knot counts");
            }
            for (int k = 0; k < np; ++k) {
              M3::float_t x, y, b, c, d;
              spl->GetCoeff(k, x, y, b, c, d);
              info.xPts[k] = x;
              flat.coeff_x[size_t(p) * _max_knots + k] = x;
              flat.coeff_many.insert(flat.coeff_many.end(), {float(y), float(b), float(c), float(d)});
            }
            for (int k = np; k < _max_knots; ++k) flat.coeff_x[size_t(p) * _max_knots + k] = 1.f;  // as GetSplineCoeff_SepMany leaves it
            flat.paramNo_arr.push_back(short(p));
            flat.nKnots_arr.push_back(knot);
            knot += unsigned(np);
          }
        } else {
          TF1_red* tf = static_cast<TF1_red*>(rf);
          if (tf->GetSize() != _nTF1Coeff_) throw MaCh3Exception(__FILE__, __LINE__, "TF1_red must carry exactly 2 coefficients");
          cpu_coeff_TF1_many.push_back(float(tf->GetParameter(0)));
          cpu_coeff_TF1_many.push_back(float(tf->GetParameter(1)));
          cpu_paramNo_TF1_arr.push_back(short(p));
        }
        delete rf;
        MasterSpline[e][p] = nullptr;
      }
      cpu_nParamPerEvent[2 * e] = unsigned(flat.paramNo_arr.size()) - s0;
      cpu_nParamPerEvent[2 * e + 1] = s0;
      cpu_nParamPerEvent_tf1[2 * e] = unsigned(cpu_paramNo_TF1_arr.size()) - t0;
      cpu_nParamPerEvent_tf1[2 * e + 1] = t0;
    }
    NSplines_valid = unsigned(flat.paramNo_arr.size());
    NTF1_valid = unsigned(cpu_paramNo_TF1_arr.size());
    nKnots = knot;
  }

  void MoveToGPU() {
    gpu_spline_handler->InitGPU_Segments(&SplineSegments);
    gpu_spline_handler->InitGPU_Vals(&ParamValues);
    for (int j = 0; j < nParams; ++j) {
      SplineSegments[j] = 0;
      ParamValues[j] = -999;
    }
    gpu_spline_handler->InitGPU_SplineMonolith(&cpu_total_weights, int(NEvents), nKnots, NSplines_valid, NTF1_valid, _max_knots * nParams);
    gpu_spline_handler->CopyToGPU_SplineMonolith(&flat, cpu_coeff_TF1_many, cpu_paramNo_TF1_arr, int(NEvents), cpu_nParamPerEvent,
                                                 cpu_nParamPerEvent_tf1, nParams, NSplines_valid, _max_knots, nKnots, NTF1_valid);
    // the host copies are not needed once the device holds the re-laid-out store (ref: SplineMonolith.cpp:305-310)
    if (!keep_host_copy) {
      flat = SplineMonoStruct();
      std::vector<float>().swap(cpu_coeff_TF1_many);
      std::vector<short>().swap(cpu_paramNo_TF1_arr);
      std::vector<unsigned>().swap(cpu_nParamPerEvent);
      std::vector<unsigned>().swap(cpu_nParamPerEvent_tf1);
    }
  }

  std::unique_ptr<SMonolithGPU> gpu_spline_handler;
  SplineMonoStruct flat;
  std::vector<short> npts;
  std::vector<float> cpu_coeff_TF1_many;
  std::vector<short> cpu_paramNo_TF1_arr;
  std::vector<unsigned> cpu_nParamPerEvent, cpu_nParamPerEvent_tf1;
  unsigned NEvents = 0, NSplines_valid = 0, NTF1_valid = 0, nKnots = 0;
  short _max_knots = 0;
  bool keep_host_copy = false;
};
