// Core.h -- types, constants and the error convention the per-step path relies on.
// Restates what the hot path needs from the reference's Manager/Core.h:24-92 and
// Manager/MaCh3Exception.h:23-49 (no spdlog, no ROOT).
#pragma once

#include <cmath>
#include <cstdio>
#include <exception>
#include <limits>
#include <string>

namespace M3 {
// SMonolith only exists in the reference's _LOW_MEMORY_STRUCTS_ build
// (Samples/SampleHandlerFD.cpp:1244-1254), so the per-step types are fixed to that build.
using float_t = float;
using int_t = short;
using uint_t = unsigned short;

constexpr double _BAD_DOUBLE_ = -999.99;
constexpr int _BAD_INT_ = -999;
constexpr double _DEFAULT_RETURN_VAL_ = -999999.123456;
constexpr float_t Unity = 1.f;
constexpr float_t Zero = 0.f;
constexpr double KinematicLowBound = std::numeric_limits<double>::lowest();
constexpr double KinematicUpBound = std::numeric_limits<double>::max();
/// returned by the parameter side for out-of-bounds proposals; such steps never reach Reweight()
/// (ref: Fitters/MR2T2.cpp:51-58)
constexpr double _LARGE_LOGL_ = 1234567890.0;
/// lowest MC prediction a bin likelihood is evaluated at
constexpr double _LOW_MC_BOUND_ = .00001;
constexpr int UnderOverFlowBin = -1;
}  // namespace M3

/// Thrown wherever the reference throws MaCh3Exception(__FILE__, __LINE__); carries the
/// engine's message when a C-ABI call failed.
class MaCh3Exception : public std::exception {
 public:
  MaCh3Exception(const std::string& file, int line, const std::string& message = "")
      : msg_((message.empty() ? std::string("Terminating MaCh3") : message) + " [" + file + ":" + std::to_string(line) + "]") {
    std::fprintf(stderr, "[MaCh3 error] %s\n", msg_.c_str());
  }
  const char* what() const noexcept override { return msg_.c_str(); }

 private:
  std::string msg_;
};

/// Samples/SampleStructs.h:105-112 (numeric values are what m3hip_set_test_statistic takes)
enum TestStatistic { kPoisson, kBarlowBeeston, kIceCube, kPearson, kDembinskiAbdelmotteleb, kNTestStatistics };

inline std::string TestStatistic_ToString(TestStatistic t) {
  switch (t) {
    case kPoisson: return "Poisson";
    case kBarlowBeeston: return "BarlowBeeston";
    case kIceCube: return "IceCube";
    case kPearson: return "Pearson";
    case kDembinskiAbdelmotteleb: return "DembinskiAbdelmotteleb";
    default: return "Unknown";
  }
}
