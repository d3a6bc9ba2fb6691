// Sharding.h -- which events a rank owns when one sample is spread over several GPUs.
//
// The reference runs on one device (Manager/gpuUtils.cu:71: "currently I only support one GPU"), so there is nothing to
// mirror: this is the host half of the new seam.  Events are independent up to the per-bin sum
// (Samples/SampleHandlerFD.cpp:390-448), so a sample is cut into contiguous event blocks, one per rank; every rank builds its
// SMonolith and registers its events for its block only, and SampleHandlerFD::InitComm / OpenExchange make Reweight() end in
// ONE exchange of the bin vector before the likelihood.  Same rule, same cuts as mach3_amd/dist.py:shard_ranges (the python
// harness of tests/ and bench.py); tests/test_cpp_facade.py holds the two against each other.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <utility>
#include <vector>

namespace M3 {

/// shard boundaries stay on the engine's classification granularity (m3hip_monolith_append: every chunk but the last holds
/// a multiple of 1024 events)
constexpr uint64_t kShardGroup = 1024;

/// Contiguous [first, last) per rank.  Without weights the events are split evenly (block size rounded up to 1024); with
/// per-event weights -- the responses an event carries, for sparse samples: every rank then streams the same bytes -- the
/// cuts balance the summed weight instead, each rounded to the nearest multiple of 1024 and never before the previous one.
inline std::vector<std::pair<uint64_t, uint64_t>> ShardRanges(uint64_t nEvents, int nRanks, const std::vector<double>& Weights = {}) {
  if (nRanks <= 0) throw std::invalid_argument("ShardRanges: nRanks must be positive");
  if (!Weights.empty() && Weights.size() != nEvents) throw std::invalid_argument("ShardRanges: one weight per event");
  std::vector<uint64_t> cuts;
  cuts.reserve(size_t(nRanks) + 1);
  if (Weights.empty()) {
    uint64_t per = (nEvents + uint64_t(nRanks) - 1) / uint64_t(nRanks);
    per = (per + kShardGroup - 1) / kShardGroup * kShardGroup;
    for (int r = 0; r <= nRanks; ++r) cuts.push_back(std::min(nEvents, uint64_t(r) * per));
  } else {
    std::vector<double> c(size_t(nEvents) + 1, 0.0);  // running sum, in event order
    for (uint64_t e = 0; e < nEvents; ++e) c[e + 1] = c[e] + Weights[e];
    cuts.push_back(0);
    for (int r = 1; r < nRanks; ++r) {
      const double target = c[nEvents] * double(r) / double(nRanks);
      const uint64_t e = uint64_t(std::lower_bound(c.begin(), c.end(), target) - c.begin());  // first event whose running sum reaches it
      const uint64_t rounded = uint64_t(std::nearbyint(double(e) / double(kShardGroup))) * kShardGroup;  // (ties to even, like python's round)
      cuts.push_back(std::min(nEvents, std::max(cuts.back(), rounded)));
    }
    cuts.push_back(nEvents);
  }
  std::vector<std::pair<uint64_t, uint64_t>> out;
  for (int r = 0; r < nRanks; ++r) out.emplace_back(cuts[size_t(r)], cuts[size_t(r) + 1]);
  return out;
}

}  // namespace M3
