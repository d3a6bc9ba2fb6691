// m3hip_layout.cpp -- see m3hip_layout.h.
#include "m3hip_layout.h"

#include <algorithm>
#include <cstring>
#include <thread>

namespace m3hip {

namespace {

template <class F>
void parallel_for(uint64_t n, F&& body) {
  unsigned nt = std::thread::hardware_concurrency();
  if (nt == 0) nt = 4;
  if (nt > 16) nt = 16;  // one GPU's share of the host
  if (n < 2 || nt == 1) {
    for (uint64_t i = 0; i < n; ++i) body(i);
    return;
  }
  if (nt > n) nt = unsigned(n);
  std::vector<std::thread> th;
  th.reserve(nt);
  for (unsigned t = 0; t < nt; ++t) {
    th.emplace_back([&, t]() {
      for (uint64_t i = t; i < n; i += nt) body(i);
    });
  }
  for (auto& x : th) x.join();
}

// true when events [e0, e1) all carry the parameter lists of event e0
bool tile_is_uniform(const ChunkInput& in, uint64_t e0, uint64_t e1) {
  const unsigned int c0 = in.npe[2 * e0], s0 = in.npe[2 * e0 + 1];
  const unsigned int t0c = in.npe_tf1[2 * e0], t0s = in.npe_tf1[2 * e0 + 1];
  for (uint64_t e = e0 + 1; e < e1; ++e) {
    if (in.npe[2 * e] != c0 || in.npe_tf1[2 * e] != t0c) return false;
    const unsigned int s = in.npe[2 * e + 1], t = in.npe_tf1[2 * e + 1];
    if (c0 && std::memcmp(in.param_no + s, in.param_no + s0, sizeof(int16_t) * c0) != 0) return false;
    if (t0c && std::memcmp(in.param_no_tf1 + t, in.param_no_tf1 + t0s, sizeof(int16_t) * t0c) != 0) return false;
  }
  return true;
}

}  // namespace

bool plan_chunk(int n_params, int max_knots, const int16_t* npts, const ChunkInput& in, ChunkPlan* plan,
                std::string* err) {
  plan->tiles.clear();
  plan->spl_float4 = plan->tf1_float2 = plan->meta_int32 = 0;
  plan->n_uniform = plan->n_brick = 0;
  if (in.n_events == 0) return true;
  if (!in.npe || !in.npe_tf1) { *err = "n_param_per_event arrays are NULL"; return false; }

  // ---- validation of the reference-layout arrays (the reference trusts them blindly)
  std::vector<int> bad(1, 0);
  std::string msg;
  for (uint64_t e = 0; e < in.n_events; ++e) {
    const uint64_t c = in.npe[2 * e], s = in.npe[2 * e + 1];
    const uint64_t ct = in.npe_tf1[2 * e], st = in.npe_tf1[2 * e + 1];
    if (s + c > in.n_splines || st + ct > in.n_tf1) {
      *err = "event " + std::to_string(e) + ": {count,start} run exceeds n_splines / n_tf1";
      return false;
    }
  }
  for (uint64_t s = 0; s < in.n_splines; ++s) {
    const int p = in.param_no[s];
    if (p < 0 || p >= n_params) { *err = "spline " + std::to_string(s) + ": paramNo out of range"; return false; }
    const int np = npts[p];
    if (np < 2 || np > max_knots) {
      *err = "spline " + std::to_string(s) + ": parameter " + std::to_string(p) + " has no valid knot count (nPts=" +
             std::to_string(np) + ")";
      return false;
    }
    if (uint64_t(in.nknots_arr[s]) + uint64_t(np) > in.total_nknots) {
      *err = "spline " + std::to_string(s) + ": knots exceed total_nknots";
      return false;
    }
  }
  for (uint64_t t = 0; t < in.n_tf1; ++t) {
    const int p = in.param_no_tf1[t];
    if (p < 0 || p >= n_params) { *err = "TF1 " + std::to_string(t) + ": paramNo out of range"; return false; }
  }

  // ---- tiles
  const uint64_t n_tiles = (in.n_events + M3_TE - 1) / M3_TE;
  plan->tiles.resize(n_tiles);
  std::vector<uint8_t> uniform(n_tiles, 0);
  parallel_for(n_tiles, [&](uint64_t t) {
    const uint64_t e0 = t * M3_TE, e1 = std::min<uint64_t>(in.n_events, e0 + M3_TE);
    uniform[t] = tile_is_uniform(in, e0, e1) ? 1 : 0;
  });
  for (uint64_t t = 0; t < n_tiles; ++t) {
    TileHost& th = plan->tiles[t];
    const uint64_t e0 = t * M3_TE, e1 = std::min<uint64_t>(in.n_events, e0 + M3_TE);
    th.n_events = uint32_t(e1 - e0);
    if (!uniform[t]) {
      th.kind = kTileBrick;
      plan->n_brick++;
      *err = "tile " + std::to_string(t) + ": events carry different parameter lists (ragged tiles not supported yet)";
      return false;
    }
    th.kind = kTileUniform;
    plan->n_uniform++;
    th.n_spl_slots = in.npe[2 * e0];
    th.n_tf1_slots = in.npe_tf1[2 * e0];
    uint64_t planes = 0;
    const unsigned int s0 = in.npe[2 * e0 + 1];
    for (uint32_t j = 0; j < th.n_spl_slots; ++j) planes += uint64_t(npts[in.param_no[s0 + j]] - 1);
    th.n_planes = planes;
    th.spl_off = plan->spl_float4;
    th.tf1_off = plan->tf1_float2;
    th.meta_off = plan->meta_int32;
    plan->spl_float4 += planes * M3_TE;
    plan->tf1_float2 += uint64_t(th.n_tf1_slots) * M3_TE;
    plan->meta_int32 += 2ull * th.n_spl_slots + th.n_tf1_slots;
  }
  return true;
}

void fill_chunk(int /*n_params*/, const int16_t* npts, const ChunkInput& in, const ChunkPlan& plan, float* spl,
                float* tf1, int32_t* meta) {
  parallel_for(plan.tiles.size(), [&](uint64_t t) {
    const TileHost& th = plan.tiles[t];
    if (th.kind != kTileUniform) return;
    const uint64_t e0 = t * M3_TE;
    const unsigned int s0 = in.npe[2 * e0 + 1];
    const unsigned int t0 = in.npe_tf1[2 * e0 + 1];
    int32_t* m = meta + th.meta_off;
    float* tile_spl = spl + 4 * th.spl_off;
    float* tile_tf1 = tf1 + 2 * th.tf1_off;
    // slot meta: {param, first plane}
    uint64_t plane0 = 0;
    std::vector<uint32_t> slot_plane(th.n_spl_slots), slot_nseg(th.n_spl_slots);
    for (uint32_t j = 0; j < th.n_spl_slots; ++j) {
      const int p = in.param_no[s0 + j];
      m[2 * j] = p;
      m[2 * j + 1] = int32_t(plane0);
      slot_plane[j] = uint32_t(plane0);
      slot_nseg[j] = uint32_t(npts[p] - 1);
      plane0 += slot_nseg[j];
    }
    for (uint32_t j = 0; j < th.n_tf1_slots; ++j) m[2 * th.n_spl_slots + j] = in.param_no_tf1[t0 + j];
    // coefficients: event-major AoS records -> [plane][lane] float4
    for (uint32_t le = 0; le < uint32_t(M3_TE); ++le) {
      if (le < th.n_events) {
        const uint64_t e = e0 + le;
        const unsigned int se = in.npe[2 * e + 1];
        for (uint32_t j = 0; j < th.n_spl_slots; ++j) {
          const float* src = in.coeff_many + 4ull * in.nknots_arr[se + j];
          float* dst = tile_spl + (uint64_t(slot_plane[j]) * M3_TE + le) * 4;
          for (uint32_t k = 0; k < slot_nseg[j]; ++k) std::memcpy(dst + uint64_t(k) * M3_TE * 4, src + 4 * k, 16);
        }
        const unsigned int te = in.npe_tf1[2 * e + 1];
        for (uint32_t j = 0; j < th.n_tf1_slots; ++j)
          std::memcpy(tile_tf1 + (uint64_t(j) * M3_TE + le) * 2, in.coeff_tf1 + 2ull * (te + j), 8);
      } else {
        // padding lanes of the last tile: neutral response (weight 1)
        for (uint64_t pl = 0; pl < th.n_planes; ++pl) {
          float* dst = tile_spl + (pl * M3_TE + le) * 4;
          dst[0] = 1.f; dst[1] = dst[2] = dst[3] = 0.f;
        }
        for (uint32_t j = 0; j < th.n_tf1_slots; ++j) {
          float* dst = tile_tf1 + (uint64_t(j) * M3_TE + le) * 2;
          dst[0] = 0.f; dst[1] = 1.f;
        }
      }
    }
  });
}

}  // namespace m3hip
