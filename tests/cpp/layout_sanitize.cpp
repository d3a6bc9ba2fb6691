// Host re-layout (mach3_amd/csrc/m3hip_layout.cpp) under AddressSanitizer + UBSan: random dense, mixed and
// ragged chunks are planned and filled into exactly-sized stores; any out-of-bounds read of the input arrays or
// write past a store aborts.  GPU sanitizers are not available on the pool, so this is the CPU-side check.
// Built and run by tests/test_layout.py::test_layout_under_sanitizers.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "../../mach3_amd/csrc/m3hip_layout.h"

using namespace m3hip;

struct Chunk {
  std::vector<float> coeff_many, coeff_tf1;
  std::vector<unsigned int> nknots_arr, npe, npe_tf1;
  std::vector<int16_t> param_no, param_no_tf1;
};

static int run_case(uint64_t seed, uint64_t n_events, int n_params, double occupancy, bool mixed, bool uniform_lists, double pad_min_fill = kPadMinFillDefault,
                    int groups = 1) {
  g_pad_min_fill = pad_min_fill;
  std::mt19937_64 rng(seed);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  const int K = 7;
  std::vector<int16_t> npts(static_cast<size_t>(n_params));
  for (int p = 0; p < n_params; ++p) npts[size_t(p)] = (mixed && p % 2) ? int16_t(-999) : int16_t(3 + 2 * (p % 3));  // 3, 5, 7 knots
  Chunk c;
  for (uint64_t e = 0; e < n_events; ++e) {
    const unsigned s0 = unsigned(c.param_no.size()), t0 = unsigned(c.param_no_tf1.size());
    for (int p = 0; p < n_params; ++p) {
      if (!uniform_lists && U(rng) >= occupancy) continue;
      if (npts[size_t(p)] > 0) {
        c.nknots_arr.push_back(unsigned(c.coeff_many.size() / 4));
        c.param_no.push_back(int16_t(p));
        for (int k = 0; k < npts[size_t(p)] * 4; ++k) c.coeff_many.push_back(float(U(rng)));
      } else {
        c.param_no_tf1.push_back(int16_t(p));
        c.coeff_tf1.push_back(float(U(rng)));
        c.coeff_tf1.push_back(1.0f);
      }
    }
    c.npe.push_back(unsigned(c.param_no.size()) - s0);
    c.npe.push_back(s0);
    c.npe_tf1.push_back(unsigned(c.param_no_tf1.size()) - t0);
    c.npe_tf1.push_back(t0);
  }
  // exact-size copies: one element past any of them is poisoned by ASAN
  ChunkInput in;
  in.n_events = n_events;
  in.coeff_many = c.coeff_many.data();
  in.total_nknots = c.coeff_many.size() / 4;
  in.nknots_arr = c.nknots_arr.data();
  in.param_no = c.param_no.data();
  in.n_splines = c.param_no.size();
  in.coeff_tf1 = c.coeff_tf1.data();
  in.param_no_tf1 = c.param_no_tf1.data();
  in.n_tf1 = c.param_no_tf1.size();
  in.npe = c.npe.data();
  in.npe_tf1 = c.npe_tf1.data();
  ChunkPlan plan;
  std::string err;
  if (!plan_chunk(n_params, K, npts.data(), in, &plan, &err, groups)) {
    std::printf("seed %llu: plan refused: %s\n", (unsigned long long)seed, err.c_str());
    return 0;  // a refusal is a valid outcome (e.g. too many stages); crashing is not
  }
  std::vector<float> spl(plan.spl_float4 * 4), tf1(plan.tf1_float2 * 2);
  std::vector<int32_t> meta(plan.meta_int32);
  fill_chunk(n_params, npts.data(), in, plan, spl.data(), tf1.data(), meta.data());
  std::printf("seed %llu: E=%llu P=%d occ=%.2f mixed=%d uniform=%d G=%d -> %llu uniform tiles, %llu bricks, %zu words, %.1f MB\n",
              (unsigned long long)seed, (unsigned long long)n_events, n_params, occupancy, int(mixed), int(uniform_lists), groups,
              (unsigned long long)plan.n_uniform, (unsigned long long)plan.n_brick, plan.brick_words.size(), double(spl.size()) * 4e-6);
  return 0;
}

int main() {
  int rc = 0;
  rc |= run_case(1, 1, 1, 1.0, false, true);          // one event
  rc |= run_case(2, 777, 6, 1.0, true, true);         // uniform, mixed kinds, partial last tile
  rc |= run_case(3, 2048 + 300, 48, 0.2, true, false);  // ragged, three groups
  rc |= run_case(4, 1500, 40, 0.3, false, false);     // ragged, TSpline3 only
  rc |= run_case(5, 1024, 200, 0.9, true, false, 2.0);     // deep brick (many stages; identity padding off)
  rc |= run_case(6, 1024, 1500, 0.95, false, false, 2.0);  // beyond the stage limit: must be refused cleanly
  rc |= run_case(9, 4096 + 2048 + 300, 48, 0.2, true, false, kPadMinFillDefault, 2);  // bricks of 2048 events, a short last one
  rc |= run_case(10, 2 * 4096 + 1500, 64, 0.1, true, false, kPadMinFillDefault, 4);   // bricks of 4096 events
  rc |= run_case(11, 4096, 300, 0.9, false, false, 2.0, 4);  // heavy events: the planner cuts the bricks short of 4096 events
  rc |= run_case(8, 2048 + 300, 60, 0.85, true, false);    // nearly uniform lists: identity-padded uniform tiles
  rc |= run_case(7, 3000, 16, 0.0, true, false);      // no responses at all
  g_pad_runs = 1;  // runs padded to whole lines in the response lists whatever that costs (default: where it is cheap)
  rc |= run_case(12, 1024 + 700, 48, 0.03, true, false);   // short runs: mostly padding entries
  rc |= run_case(13, 2 * 4096 + 99, 64, 0.1, true, false, kPadMinFillDefault, 4);
  g_pad_runs = 0;
  rc |= run_case(14, 2048 + 300, 48, 0.2, true, false);    // never padded
  g_pad_runs = 2;
  std::printf("OK\n");
  return rc;
}
