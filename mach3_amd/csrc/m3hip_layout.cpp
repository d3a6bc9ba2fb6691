// m3hip_layout.cpp -- see m3hip_layout.h.
#include "m3hip_layout.h"

#include <algorithm>
#include <cstring>
#include <exception>
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
  std::vector<std::exception_ptr> failed(nt);  // an exception must not leave a thread: it is rethrown in the caller
  th.reserve(nt);
  for (unsigned t = 0; t < nt; ++t) {
    th.emplace_back([&, t]() {
      try {
        for (uint64_t i = t; i < n; i += nt) body(i);
      } catch (...) {
        failed[t] = std::current_exception();
      }
    });
  }
  for (auto& x : th) x.join();
  for (const std::exception_ptr& f : failed)
    if (f) std::rethrow_exception(f);
}

// true when events [e0, e1) all carry the parameter lists of event e0
// Ascending union of the tile's spline / TF1 parameters and the fill fraction of the padded tile.  Returns false
// when an event does not list its parameters in ascending order (then the per-event merge below cannot keep
// the event's own product order).
bool tile_union(int n_params, const int16_t* npts, const ChunkInput& in, uint64_t e0, uint64_t e1, std::vector<int16_t>* us,
                std::vector<int16_t>* ut, double* fill) {
  std::vector<uint8_t> seen_s(size_t(n_params), 0), seen_t(size_t(n_params), 0);
  double real_bytes = 0;
  for (uint64_t e = e0; e < e1; ++e) {
    const unsigned int c = in.npe[2 * e], s0 = in.npe[2 * e + 1];
    for (unsigned int i = 0; i < c; ++i) {
      if (i && in.param_no[s0 + i] <= in.param_no[s0 + i - 1]) return false;
      seen_s[size_t(in.param_no[s0 + i])] = 1;
    }
    const unsigned int ct = in.npe_tf1[2 * e], t0 = in.npe_tf1[2 * e + 1];
    for (unsigned int i = 0; i < ct; ++i) {
      if (i && in.param_no_tf1[t0 + i] <= in.param_no_tf1[t0 + i - 1]) return false;
      seen_t[size_t(in.param_no_tf1[t0 + i])] = 1;
    }
    real_bytes += 16.0 * c + 8.0 * ct;
  }
  us->clear();
  ut->clear();
  for (int p = 0; p < n_params; ++p) {
    if (seen_s[size_t(p)]) us->push_back(int16_t(p));
    if (seen_t[size_t(p)]) ut->push_back(int16_t(p));
  }
  (void)npts;
  const double padded_bytes = double(e1 - e0) * (16.0 * double(us->size()) + 8.0 * double(ut->size()));
  *fill = padded_bytes > 0 ? real_bytes / padded_bytes : 1.0;
  return true;
}

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


// ------------------------------------------------------------------------------------
// Brick metadata stream (uint32 words, offsets relative to the brick's first word).  G = the engine's brick_groups: a brick
// holds up to G x 1024 events, local event le = 1024 h + lane (slab h, lane of the workgroup); wave-block b = 16 h + w.
//   [0] n_stages (TSpline3)   [1] off_tf1   [2] TF1 slots of the brick (codes, skips included)   [3] G   [4] 1: the TF1 lane-slots
//   take the slab's events in order of descending code count (sigma below), 0: in their own order   [5] 1: no event has more than
//   63 TF1 codes and the counts travel in the sigma words   [6..7] 0
//   [8 + 4 s ..]  stage header: n_resp | n_runs << 16 | rows << 26, off_runs, off_pairs, off_resp   (rows = the most
//             responses one event holds in the stage, <= M3_STAGE_EVENT_MAX)
//   off_runs  n_runs x {int32 coef_off - start_r, stride | param << 16}.  coef_off: float4 index of the run's first plane
//             in the brick's spline store; start_r: first response of the run in the stage; so the quad of segment k of
//             response r sits at entry[0] + k * stride + r.  stride = n_run rounded up to M3_PLANE_ALIGN.
//   off_pairs M3_STAGE_PAIR_WORDS words: low half of word g = the run of response group g's first response (n_runs past
//             the stage's last group); high half of word b (b < 16 G) = slot of wave-block b's first row in the stage's
//             compact scatter array.  Then the events' response counts of the stage, a nibble each, stored the way a lane
//             fetches them: G = 1: byte i = count of event i | count of event i + 512 << 4 (512 bytes); G = 2: byte l =
//             event l | event l + 1024 << 4; G = 4: uint16 l = events l, l + 1024, l + 2048, l + 3072, a nibble each.
//   off_resp  one uint16 per response: slot << 2 | last response of its run (bits 2..14 are the byte offset of the
//             response's slot in the scatter array).  Slots: wave-block after wave-block, inside a block event after event
//             (lane order), an event's responses in ascending parameter order.  A lane's words
//             lie next to each other (the kernel fetches them with one 12-byte load): record [l] of lane l = nw uint16, word u =
//             response l + 1024 u, nw = stage_record_words(n_resp) = 1 .. 5 -- records of 2 nw bytes, so with an odd nw every
//             other one starts on a 2-byte boundary (gfx950 loads from there; rounding the records up to whole dwords cost a
//             stage of 4097 .. 5120 entries a sixth of these bytes); one record for every lane that holds a response, the
//             last dword of the list filled up with zeros
//   off_tf1   [0 .. 16 G)       float2 index (from the brick's TF1 store) of the first slot of wave-block b
//             [16 G .. 32 G)    word offset (from the brick's first word) of wave-block b's first code dword
//             [32 G .. 32 G + 256 G)  codes per lane-slot, uint8, four per word (slot k = 1024 h + l: what lane l works on in slab h);
//                               NOT THERE when [5] is set: the count is then bits 10..15 of the slot's sigma word
//             [.. + 512 G)      sigma: the slot's event relative to its slab, uint16 (bits 0..9), two per word.  Inside a slab the events are
//                               handed to the slots in order of descending code count, so that a wave-block's 64 events have
//                               about equal counts (a wave executes as many rows as its longest event has codes)
//             then the code dwords: per wave-block, per four rows 4 q .. 4 q + 3, one dword for every lane with more than 4 q
//             codes (lanes packed, ascending), byte i = code of row 4 q + i.  A code is the step from the event's previous
//             TF1 parameter to this one in the chunk's TF1 dictionary (first: from -1); M3_TF1_SKIP steps 254 without a response.
//   TF1 store: per wave-block, per row j, one float2 {a,b} for every lane with more than j codes (lanes packed, ascending)
// ------------------------------------------------------------------------------------
bool plan_brick(int n_params, const int16_t* npts, const ChunkInput& in, const std::vector<int32_t>& dict_index, uint64_t e0,
                uint64_t e1, uint32_t G, BrickHost* bh, std::vector<uint32_t>* words_all, std::string* err) {
  const uint32_t ne = uint32_t(e1 - e0);
  bh->n_events = ne;
  bh->groups = G;
  bh->first_event = e0;
  bh->words_off = words_all->size();
  bh->bytes_tables = bh->bytes_book = bh->bytes_codes = 0;
  std::vector<uint32_t> w;  // this brick's stream

  // the per-event product must visit parameters in the event's own order; the brick regroups
  // by parameter, which preserves that order only if every event lists them ascending
  for (uint64_t e = e0; e < e1; ++e) {
    const unsigned int c = in.npe[2 * e], s0 = in.npe[2 * e + 1];
    for (unsigned int i = 1; i < c; ++i)
      if (in.param_no[s0 + i] <= in.param_no[s0 + i - 1]) {
        *err = "event " + std::to_string(e) + ": spline parameters are not in ascending order (required for ragged tiles)";
        return false;
      }
    const unsigned int ct = in.npe_tf1[2 * e], t0 = in.npe_tf1[2 * e + 1];
    for (unsigned int i = 1; i < ct; ++i)
      if (in.param_no_tf1[t0 + i] <= in.param_no_tf1[t0 + i - 1]) {
        *err = "event " + std::to_string(e) + ": TF1 parameters are not in ascending order (required for ragged tiles)";
        return false;
      }
  }

  // ---- TSpline3: inverted index, responses per parameter (counting sort keeps events ascending)
  std::vector<uint32_t> cnt_s(n_params + 1, 0);
  for (uint64_t e = e0; e < e1; ++e) {
    const unsigned int c = in.npe[2 * e], s0 = in.npe[2 * e + 1];
    for (unsigned int i = 0; i < c; ++i) cnt_s[in.param_no[s0 + i] + 1]++;
  }
  for (int p = 0; p < n_params; ++p) cnt_s[p + 1] += cnt_s[p];
  const uint32_t S = cnt_s[n_params];
  std::vector<uint32_t> inv_s(S), inv_ev(S);  // spline index / local event, parameter-major
  {
    std::vector<uint32_t> cur(cnt_s.begin(), cnt_s.end() - 1);
    for (uint64_t e = e0; e < e1; ++e) {
      const unsigned int c = in.npe[2 * e], s0 = in.npe[2 * e + 1];
      for (unsigned int i = 0; i < c; ++i) {
        const uint32_t pos = cur[in.param_no[s0 + i]]++;
        inv_s[pos] = s0 + i;
        inv_ev[pos] = uint32_t(e - e0);
      }
    }
  }

  // ---- stages: ascending parameter ranges; <= M3_STAGE_CAP responses, <= M3_STAGE_MAX_RUNS runs, no event more than
  // M3_STAGE_EVENT_MAX times.  Stages of about equal size (a brick whose responses just exceed a multiple of the cap would
  // otherwise end in a stage of a few responses that costs the same barriers as a full one).
  struct Range { int p0, p1; };
  // Extent of a run in a stage's response list.  Padded: its length rounded up to whole 128-byte lines, i.e. the list is laid
  // out like the memory behind it, a group of 64 responses is eight whole lines and no line is asked for by two waves
  // (config 3 -0.7 %, config 5 -1.2 % kernel time; the padding entries load the plane's padding, evaluate it and drop the
  // result into the scatter array's slack).  Only where the runs are long enough for that to cost few extra entries:
  // g_pad_runs = 2 pads the bricks whose lists grow by at most 8 %.
  auto round_line = [](uint32_t n) -> uint32_t { return (n + uint32_t(M3_PLANE_ALIGN) - 1) & ~uint32_t(M3_PLANE_ALIGN - 1); };
  uint32_t S_lines = 0;
  for (int p = 0; p < n_params; ++p) S_lines += round_line(cnt_s[p + 1] - cnt_s[p]);
  const bool pad_runs = g_pad_runs == 1 || (g_pad_runs == 2 && uint64_t(S_lines) * 100 <= uint64_t(S) * 108);
  auto ext = [&](uint32_t n) -> uint32_t { return pad_runs ? round_line(n) : n; };
  const uint32_t S_ext = pad_runs ? S_lines : S;
  // `n_aim` = 0: every stage as long as the limits allow (the fewest stages there can be).  Otherwise: aim at `n_aim` stages
  // of equal size, re-aimed after every stage (one that the row limit closed early must not leave a stub of a few
  // responses behind: it would cost the same barriers as a full stage).
  // `n_4k`: the first n_4k stages hold at most 4096 responses and are filled to that; the others share what is left
  auto split = [&](uint32_t n_aim, uint32_t n_4k) {
    std::vector<Range> st;
    std::vector<uint8_t> mult(ne, 0);
    int p0 = 0;
    uint32_t done = 0;  // responses in the stages closed so far
    while (p0 < n_params) {
      const bool is_4k = st.size() < n_4k;
      const uint32_t hard = is_4k ? 4096u : uint32_t(M3_STAGE_CAP);
      uint32_t soft = hard;
      if (n_aim && !is_4k) {
        const uint32_t left = S_ext - done, n_left = n_aim > st.size() ? n_aim - uint32_t(st.size()) : 1u;
        soft = std::min<uint32_t>(soft, (left + n_left - 1) / n_left);
      }
      std::fill(mult.begin(), mult.end(), uint8_t(0));
      uint32_t tot = 0, runs = 0;
      int p1 = p0;
      while (p1 < n_params) {
        const uint32_t a = cnt_s[p1], b = cnt_s[p1 + 1], n_run = ext(b - a);
        if (n_run == 0) { ++p1; continue; }
        if (tot > 0) {
          bool fits = tot + n_run <= hard && runs < uint32_t(M3_STAGE_MAX_RUNS);
          for (uint32_t r = a; fits && r < b; ++r) fits = mult[inv_ev[r]] < uint8_t(M3_STAGE_EVENT_MAX);
          if (!fits) break;
        }
        for (uint32_t r = a; r < b; ++r) mult[inv_ev[r]]++;
        tot += n_run;
        ++runs;
        ++p1;
        if (tot >= soft) break;
      }
      if (tot > 0) st.push_back({p0, p1});
      done += tot;
      p0 = p1;
    }
    return st;
  };
  std::vector<Range> st_s = split(0, 0);
  if (st_s.size() > 1) {
    std::vector<Range> even = split(uint32_t(st_s.size()), 0);
    if (even.size() <= st_s.size()) st_s.swap(even);
    // A lane's response words of a stage are stored next to each other, two per dword: a stage of up to 4096 responses takes two
    // dwords per lane, one of up to 5120 three, whatever it holds -- so as many stages as the count allows are filled to 4096
    // exactly and the others share the rest (config 3, 17.3 k list entries in four stages: 3 x 4096 + 5012; four stages of
    // 4330 would carry a third dword per lane for 234 responses each)
    const uint32_t n = uint32_t(st_s.size());
    const uint32_t room = n * uint32_t(M3_STAGE_CAP) > S_ext ? n * uint32_t(M3_STAGE_CAP) - S_ext : 0u;
    // (only where equal shares would need the third dword; run-count cuts and light bricks keep their even stages)
    uint32_t n4k = (S_ext + n - 1) / n > 4096u ? std::min<uint32_t>(n, room / uint32_t(M3_STAGE_CAP - 4096)) : 0u;
    // (a stage closes up to one run short of its limit: try with fewer 4096-stages until the count holds)
    for (; n4k > 0; --n4k) {
      std::vector<Range> packed = split(n, n4k);
      if (packed.size() <= n) { st_s.swap(packed); break; }
    }
  }
  const uint32_t n_stages = uint32_t(st_s.size());
  if (n_stages > uint32_t(M3_BRICK_MAX_STAGES)) {
    *err = "brick: " + std::to_string(n_stages) + " stages exceed the supported " + std::to_string(M3_BRICK_MAX_STAGES) +
           " (more than ~1.2M TSpline3 responses in one brick)";
    return false;
  }
  bh->n_stages = n_stages;
  w.assign(size_t(M3_BRICK_HDR0) + size_t(M3_STAGE_HDR) * n_stages, 0u);
  w[0] = n_stages;
  w[3] = G;

  uint64_t spl_f4 = 0;
  bh->spl_src.clear();
  bh->spl_src.reserve(S);
  const uint32_t n_blocks = 16u * G;  // wave-blocks of the brick
  std::vector<uint8_t> cnt(size_t(G) * M3_GROUP), rank_of(size_t(G) * M3_GROUP);
  std::vector<uint16_t> slot(size_t(G) * M3_GROUP * 16);  // slot of the j-th response of local event le: [le * 16 + j]
  for (uint32_t stage = 0; stage < n_stages; ++stage) {
    const Range& rg = st_s[stage];
    const uint32_t r0 = cnt_s[rg.p0], r1 = cnt_s[rg.p1];
    uint32_t n_resp = 0;
    for (int p = rg.p0; p < rg.p1; ++p) n_resp += ext(cnt_s[p + 1] - cnt_s[p]);
    // the events' counts in the stage, then the slots: wave-block after wave-block, event after event inside a block
    std::fill(cnt.begin(), cnt.end(), uint8_t(0));
    for (uint32_t r = r0; r < r1; ++r) cnt[inv_ev[r]]++;
    std::vector<uint16_t> woff(n_blocks, uint16_t(0));
    uint32_t next_slot = 0, rows = 0;
    for (uint32_t blk = 0; blk < n_blocks; ++blk) {
      woff[blk] = uint16_t(next_slot);
      const uint32_t l0 = blk * uint32_t(M3_WAVE);
      for (uint32_t i = 0; i < uint32_t(M3_WAVE); ++i) {
        rows = std::max<uint32_t>(rows, cnt[l0 + i]);
        for (uint32_t j = 0; j < cnt[l0 + i]; ++j) slot[size_t(l0 + i) * 16 + j] = uint16_t(next_slot++);
      }
    }
    std::fill(rank_of.begin(), rank_of.end(), uint8_t(0));
    std::vector<uint32_t> runs;
    std::vector<uint16_t> resp(n_resp), grp(size_t(M3_STAGE_PAIR_WORDS), uint16_t(0));  // groups past the end: filled below
    uint32_t n_runs = 0, i0 = 0;  // i0: the run's first entry in the stage's list
    for (int p = rg.p0; p < rg.p1; ++p) {
      const uint32_t a = cnt_s[p], b = cnt_s[p + 1];
      if (a == b) continue;
      const uint32_t n_run = b - a, nseg = uint32_t(npts[p] - 1);
      // every segment plane of a run starts on a 128-byte line: a step reads ONE plane per run, and the memory side
      // fetches whole lines, so an unaligned plane would drag in the neighbouring planes' lines at both ends
      const uint32_t stride = (n_run + uint32_t(M3_PLANE_ALIGN) - 1) & ~uint32_t(M3_PLANE_ALIGN - 1);
      if (spl_f4 + uint64_t(stride) * nseg > 0x06000000ull) { *err = "brick: spline store exceeds 1.6 GB"; return false; }
      runs.push_back(uint32_t(int32_t(spl_f4) - int32_t(i0)));
      runs.push_back(stride | (uint32_t(p) << 16));
      const uint32_t n_ext = ext(n_run);
      for (uint32_t j = 0; j < n_ext; ++j) {
        const uint32_t i = i0 + j;
        if (i % 64 == 0) grp[i / 64] = uint16_t(n_runs);
        if (j < n_run) {
          const uint32_t r = a + j, ev = inv_ev[r];
          const uint32_t rk = rank_of[ev]++;
          resp[i] = uint16_t(uint32_t(slot[size_t(ev) * 16 + rk]) << 2);
          bh->spl_src.push_back(inv_s[r]);
        } else {  // padding of the run's last line: its quad is loaded with the line, evaluated and dropped into the array's slack
          resp[i] = uint16_t((uint32_t(M3_STAGE_CAP) + (i & 15u)) << 2);
        }
      }
      resp[i0 + n_ext - 1] |= uint16_t(1u);  // last entry of its run
      i0 += n_ext;
      spl_f4 += uint64_t(stride) * nseg;
      ++n_runs;
    }
    // groups past the stage's end point behind the last run (the kernel keeps harmless entries there for lanes without a response)
    for (size_t g = (size_t(n_resp) + 63) / 64; g < grp.size(); ++g) grp[g] = uint16_t(n_runs);
    if (w.size() & 1) w.push_back(0u);  // run entries are read as 8-byte pairs
    const uint32_t off_runs = uint32_t(w.size());
    w.insert(w.end(), runs.begin(), runs.end());
    const uint32_t off_pairs = uint32_t(w.size());
    for (uint32_t g = 0; g < uint32_t(M3_STAGE_PAIR_WORDS); ++g) w.push_back(uint32_t(grp[g]) | (g < n_blocks ? uint32_t(woff[g]) << 16 : 0u));
    if (G == 1) {
      for (uint32_t i = 0; i < 512; i += 4) {
        uint32_t d = 0;
        for (uint32_t k = 0; k < 4; ++k) d |= (uint32_t(cnt[i + k]) | uint32_t(cnt[i + k + 512]) << 4) << (8 * k);
        w.push_back(d);
      }
    } else if (G == 2) {
      for (uint32_t l = 0; l < 1024; l += 4) {
        uint32_t d = 0;
        for (uint32_t k = 0; k < 4; ++k) d |= (uint32_t(cnt[l + k]) | uint32_t(cnt[l + k + 1024]) << 4) << (8 * k);
        w.push_back(d);
      }
    } else {
      for (uint32_t l = 0; l < 1024; l += 2) {
        uint32_t d = 0;
        for (uint32_t k = 0; k < 2; ++k)
          d |= (uint32_t(cnt[l + k]) | uint32_t(cnt[l + k + 1024]) << 4 | uint32_t(cnt[l + k + 2048]) << 8 | uint32_t(cnt[l + k + 3072]) << 12) << (16 * k);
        w.push_back(d);
      }
    }
    // response words, a lane's next to each other (one load fetches them): lane l of the workgroup works on responses l, l + 1024,
    // l + 2048, ... of the stage; record [l] = nw uint16, word u = response l + 1024 u, nw = stage_record_words(n_resp); records
    // for the lanes that hold a response
    while (w.size() % size_t(M3_BRICK_ALIGN_WORDS)) w.push_back(0u);  // a wave's 64 records of 8 or 10 bytes are then whole lines
    const uint32_t off_resp = uint32_t(w.size());
    {
      const uint32_t nw = stage_record_words(n_resp), lanes = std::min<uint32_t>(uint32_t(M3_GROUP), n_resp);
      w.resize(size_t(off_resp) + (size_t(nw) * lanes + 1) / 2, 0u);
      for (uint32_t l = 0; l < lanes; ++l)
        for (uint32_t u = 0; u < nw; ++u) {
          const uint32_t r = u * uint32_t(M3_GROUP) + l, at = nw * l + u;
          if (r < n_resp) w[size_t(off_resp) + at / 2] |= uint32_t(resp[r]) << (16 * (at & 1u));
        }
    }
    bh->bytes_tables += 4ull * (off_resp - off_runs);
    uint32_t* h = &w[size_t(M3_BRICK_HDR0) + size_t(M3_STAGE_HDR) * stage];
    h[0] = n_resp | (n_runs << 16) | (rows << 26);
    h[1] = off_runs;
    h[2] = off_pairs;
    h[3] = off_resp;
  }
  bh->n_spl_f4 = spl_f4;

  // ---- TF1: jagged rows per wave of 64 events, one code byte per response
  bh->tf1_src.clear();
  std::vector<std::vector<uint8_t>> codes(ne);
  std::vector<std::vector<uint32_t>> srcs(ne);
  for (uint32_t le = 0; le < ne; ++le) {
    const uint64_t e = e0 + le;
    const unsigned int ct = in.npe_tf1[2 * e], t0 = in.npe_tf1[2 * e + 1];
    int32_t prev = -1;
    for (unsigned int i = 0; i < ct; ++i) {
      const int32_t k = dict_index[size_t(in.param_no_tf1[t0 + i])];
      int32_t gap = k - prev;  // >= 1: parameters ascend and the dictionary is ascending
      while (gap > 254) {  // real steps are 1 .. 254; a skip advances 254, so that a step of at least 1 remains
        codes[le].push_back(uint8_t(M3_TF1_SKIP));
        srcs[le].push_back(0xFFFFFFFFu);
        gap -= 254;
      }
      codes[le].push_back(uint8_t(gap));
      srcs[le].push_back(t0 + i);
      prev = k;
    }
    if (codes[le].size() > size_t(M3_TF1_MAX_CODES)) {
      *err = "event " + std::to_string(e) + ": more than " + std::to_string(M3_TF1_MAX_CODES) + " TF1 responses in a ragged event";
      return false;
    }
  }
  // Lane-slots of the TF1 phase: slot k = 1024 h + l is what lane l works on in slab h.  Inside a slab the events are handed
  // to the slots in order of DESCENDING code count (stable), so that the 64 events of a wave-block have about equal counts:
  // a wave executes as many rows as its longest event has codes, and with the events in their own order every wave of a
  // config-3 brick ran ~27 rows for 16 codes per event on average.  sigma[k] = the slot's event, relative to its slab.
  std::vector<uint16_t> sigma(size_t(G) * M3_GROUP);
  // ... when that pays: the first batch of the kernel's TF1 phase holds 16 rows, every further 8 rows are one more exposed
  // round trip per wave; with every wave inside the first batch either way (config 5: 8 codes per event) the sorted order only
  // costs (its events' sample words are gathered instead of streamed, the products change lanes through LDS)
  bool sort_slots = g_tf1_sort == 1;
  if (g_tf1_sort == 2) {
    for (uint32_t le = 0; le < ne && !sort_slots; ++le) sort_slots = codes[le].size() > 16;
  }
  auto slot_event = [&](uint32_t k) -> uint32_t { return (k / uint32_t(M3_GROUP)) * uint32_t(M3_GROUP) + sigma[k]; };  // local event (may be >= ne)
  for (uint32_t h = 0; h < G; ++h) {
    std::vector<uint16_t> idx(static_cast<size_t>(M3_GROUP), uint16_t(0));
    for (uint32_t i = 0; i < uint32_t(M3_GROUP); ++i) idx[i] = uint16_t(i);
    auto key = [&](uint16_t i) -> uint32_t {  // existing events first, longest first
      const uint32_t le = h * uint32_t(M3_GROUP) + i;
      return le < ne ? uint32_t(codes[le].size()) + 1u : 0u;
    };
    if (sort_slots) std::stable_sort(idx.begin(), idx.end(), [&](uint16_t x, uint16_t y) { return key(x) > key(y); });
    std::copy(idx.begin(), idx.end(), sigma.begin() + size_t(h) * M3_GROUP);
  }
  if (sort_slots) {  // events that come in that order already (MC registered so, or few distinct counts): nothing to permute
    bool identity = true;
    for (size_t k = 0; k < sigma.size() && identity; ++k) identity = sigma[k] == uint16_t(k % size_t(M3_GROUP));
    if (identity) sort_slots = false;
  }
  auto slot_codes = [&](uint32_t k) -> const std::vector<uint8_t>* { const uint32_t le = slot_event(k); return le < ne ? &codes[le] : nullptr; };
  const uint32_t off_tf1 = uint32_t(w.size());
  w[1] = off_tf1;
  const uint32_t n_waves = n_blocks;  // wave-blocks
  const uint32_t n_slots = G * uint32_t(M3_GROUP);
  // the slots' code counts: in the spare six bits of their sigma words where every count fits (a byte per event less to read),
  // else a byte array in front of the sigma words
  bool cnt_in_sigma = G != 2 || kTf1CntInSigmaG2;
  for (uint32_t le = 0; le < ne && cnt_in_sigma; ++le) cnt_in_sigma = codes[le].size() <= 63;
  const size_t cnt_words = cnt_in_sigma ? 0 : size_t(n_slots) / 4;
  w.resize(w.size() + 2 * n_waves + cnt_words + size_t(n_slots) / 2, 0u);
  bh->bytes_book = 4ull * (w.size() - off_tf1);
  const size_t codes_from = w.size();
  for (uint32_t k = 0; k < n_slots; ++k) {
    const std::vector<uint8_t>* c = slot_codes(k);
    const uint32_t n_codes = uint32_t(c ? c->size() : 0);
    if (!cnt_in_sigma) w[off_tf1 + 2 * n_waves + k / 4] |= n_codes << (8 * (k % 4));
    w[off_tf1 + 2 * n_waves + cnt_words + k / 2] |= (uint32_t(sigma[k]) | (cnt_in_sigma ? n_codes << 10 : 0u)) << (16 * (k % 2));
  }
  uint64_t tf1_f2 = 0;
  for (uint32_t wv = 0; wv < n_waves; ++wv) {
    // a wave-block's rows and its code dwords start on 128-byte lines: whole rows (64 lanes: four lines of {a,b}, two of codes) then
    // ARE whole lines.  Laid end to end, nearly every row shared its first and last line with a neighbour -- two non-temporal
    // requests, often two fetches (config 3 -0.8 %, one GPU's share of config 5 -2 %: profiles/r05_brick_stream_alignment_ab.txt)
    while (tf1_f2 % (128 / 8)) { bh->tf1_src.push_back(0xFFFFFFFFu); ++tf1_f2; }
    while (w.size() % size_t(M3_BRICK_ALIGN_WORDS)) w.push_back(0u);
    w[off_tf1 + wv] = uint32_t(tf1_f2);
    w[off_tf1 + n_waves + wv] = uint32_t(w.size());
    const uint32_t k0 = wv * uint32_t(M3_WAVE), k1 = k0 + uint32_t(M3_WAVE);
    size_t rows = 0;
    for (uint32_t k = k0; k < k1; ++k)
      if (const std::vector<uint8_t>* c = slot_codes(k)) rows = std::max(rows, c->size());
    for (size_t j = 0; j < rows; ++j)
      for (uint32_t k = k0; k < k1; ++k) {
        const std::vector<uint8_t>* c = slot_codes(k);
        if (c && c->size() > j) { bh->tf1_src.push_back(srcs[slot_event(k)][j]); ++tf1_f2; }
      }
    for (size_t q = 0; 4 * q < rows; ++q)
      for (uint32_t k = k0; k < k1; ++k) {
        const std::vector<uint8_t>* c = slot_codes(k);
        if (c && c->size() > 4 * q) {
          uint32_t d = 0;
          for (size_t i = 0; i < 4 && 4 * q + i < c->size(); ++i) d |= uint32_t((*c)[4 * q + i]) << (8 * i);
          w.push_back(d);
        }
      }
  }
  while (tf1_f2 % (128 / 8)) { bh->tf1_src.push_back(0xFFFFFFFFu); ++tf1_f2; }  // (the next brick's store starts on a line as well)
  w[2] = uint32_t(tf1_f2);
  w[4] = sort_slots ? 1u : 0u;
  w[5] = cnt_in_sigma ? 1u : 0u;
  bh->tf1_sorted = sort_slots;
  bh->tf1_cnt_in_sigma = cnt_in_sigma;
  bh->n_tf1_f2 = tf1_f2;
  while (w.size() % size_t(M3_BRICK_ALIGN_WORDS)) w.push_back(0u);  // the next brick's stream starts on a line
  bh->bytes_codes = 4ull * (w.size() - codes_from);
  bh->n_words = w.size();
  words_all->insert(words_all->end(), w.begin(), w.end());
  return true;
}

// copies a brick's coefficients: TSpline3 runs become [segment][event-in-run] planes, TF1 responses jagged rows
void fill_brick(const int16_t* npts, const ChunkInput& in, const BrickHost& bh, const uint32_t* words, float* spl, float* tf1) {
  float* bs = spl + 4 * bh.spl_off;
  float* bt = tf1 + 2 * bh.tf1_off;
  const uint32_t n_stages = words[0];
  size_t src_i = 0;
  for (uint32_t s = 0; s < n_stages; ++s) {
    const uint32_t* h = words + size_t(M3_BRICK_HDR0) + size_t(M3_STAGE_HDR) * s;
    const uint32_t n_resp = h[0] & 0xFFFFu, n_runs = (h[0] >> 16) & 0x3FFu;
    const uint32_t* runs = words + h[1];
    uint32_t start_r = 0;
    for (uint32_t r = 0; r < n_runs; ++r) {
      const uint32_t stride = runs[2 * r + 1] & 0xFFFFu, p = runs[2 * r + 1] >> 16;
      const uint64_t coef_off = uint64_t(int64_t(int32_t(runs[2 * r])) + int64_t(start_r));
      // responses of the run: up to the next run's start (recovered from the entries: coef_off is cumulative)
      const uint32_t nseg = uint32_t(npts[p] - 1);
      uint32_t n_run;
      if (r + 1 < n_runs) {
        // next.coef_off = coef_off + stride * nseg, next.entry = next.coef_off - next.start_r
        const int64_t next_start = int64_t(coef_off) + int64_t(stride) * nseg - int64_t(int32_t(runs[2 * (r + 1)]));
        n_run = uint32_t(next_start) - start_r;
      } else {
        n_run = n_resp - start_r;
      }
      // (with runs padded to whole lines in the list -- g_pad_runs -- the entries behind the run's last response are padding:
      // their words point into the scatter array's slack)
      auto word_at = [&](uint32_t i) -> uint32_t {
        const uint32_t u = i / uint32_t(M3_GROUP), l = i % uint32_t(M3_GROUP);
        const uint32_t at = stage_record_words(n_resp) * l + u, d = words[h[3] + at / 2];
        return (at & 1u) ? d >> 16 : d & 0xFFFFu;
      };
      while (n_run > 0 && ((word_at(start_r + n_run - 1) & 0x7FFCu) >> 2) >= uint32_t(M3_STAGE_CAP)) --n_run;
      const uint32_t n_ext = uint32_t((r + 1 < n_runs ? int64_t(coef_off) + int64_t(stride) * nseg - int64_t(int32_t(runs[2 * (r + 1)])) : int64_t(n_resp)) - int64_t(start_r));
      for (uint32_t k = 0; k < nseg; ++k)  // the padding of every plane is defined (never used in a product)
        std::memset(bs + (coef_off + uint64_t(k) * stride + n_run) * 4, 0, size_t(stride - n_run) * 16);
      for (uint32_t q = 0; q < n_run; ++q) {
        const float* src = in.coeff_many + 4ull * in.nknots_arr[bh.spl_src[src_i++]];
        for (uint32_t k = 0; k < nseg; ++k) std::memcpy(bs + (coef_off + uint64_t(k) * stride + q) * 4, src + 4 * k, 16);
      }
      start_r += n_ext;
    }
  }
  static const float kIdentityTf1[2] = {0.f, 1.f};
  for (size_t i = 0; i < bh.tf1_src.size(); ++i)
    std::memcpy(bt + 2 * i, bh.tf1_src[i] == 0xFFFFFFFFu ? kIdentityTf1 : in.coeff_tf1 + 2ull * bh.tf1_src[i], 8);
}

}  // namespace

double g_pad_min_fill = kPadMinFillDefault;
int g_tf1_sort = 2;
int g_pad_runs = 2;

bool plan_chunk(int n_params, int max_knots, const int16_t* npts, const ChunkInput& in, ChunkPlan* plan,
                std::string* err, int brick_groups, uint64_t short_from, uint64_t short_to) {
  if (brick_groups != 1 && brick_groups != 2 && brick_groups != 4) { *err = "brick_groups must be 1, 2 or 4"; return false; }
  const uint32_t G = uint32_t(brick_groups);
  plan->tiles.clear();
  plan->tiles.reserve((in.n_events + M3_TE - 1) / M3_TE);
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
  // only the responses the chunk's events point at: a caller may hand over the arrays of a whole sample with every
  // chunk of it (the {count,start} pairs index them), and must not pay for all of them each time
  for (uint64_t e = 0; e < in.n_events; ++e) {
    const uint64_t c = in.npe[2 * e], s0 = in.npe[2 * e + 1];
    for (uint64_t s = s0; s < s0 + c; ++s) {
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
    const uint64_t ct = in.npe_tf1[2 * e], t0 = in.npe_tf1[2 * e + 1];
    for (uint64_t t = t0; t < t0 + ct; ++t) {
      const int p = in.param_no_tf1[t];
      if (p < 0 || p >= n_params) { *err = "TF1 " + std::to_string(t) + ": paramNo out of range"; return false; }
    }
  }

  // ---- tiles: classify per group of M3_GROUP events
  const uint64_t n_tiles = (in.n_events + M3_TE - 1) / M3_TE;
  std::vector<uint8_t> uniform(n_tiles, 0);  // 1: every event carries the same list; 2: paddable to the union list
  std::vector<std::vector<int16_t>> uni_s(n_tiles), uni_t(n_tiles);
  parallel_for(n_tiles, [&](uint64_t t) {
    const uint64_t e0 = t * M3_TE, e1 = std::min<uint64_t>(in.n_events, e0 + M3_TE);
    if (tile_is_uniform(in, e0, e1)) {
      uniform[t] = 1;
      const unsigned int c = in.npe[2 * e0], s0 = in.npe[2 * e0 + 1], ct = in.npe_tf1[2 * e0], t0 = in.npe_tf1[2 * e0 + 1];
      uni_s[t].assign(in.param_no + s0, in.param_no + s0 + c);
      uni_t[t].assign(in.param_no_tf1 + t0, in.param_no_tf1 + t0 + ct);
    } else {
      double fill = 0;
      if (tile_union(n_params, npts, in, e0, e1, &uni_s[t], &uni_t[t], &fill) && fill >= g_pad_min_fill) uniform[t] = 2;
    }
  });
  // Tiles that are uniform (exactly or after identity padding) stay tiles; every maximal run of the other
  // tiles, cut at G x M3_GROUP events (sooner when the events are so heavy that the brick's stage headers would not fit
  // the header block), becomes one brick.  (Mode blocks of a few thousand events therefore lose one
  // 256-event tile at each boundary, not a whole 1024-event group.)
  const uint64_t tiles_per_group = M3_GROUP / M3_TE;
  plan->bricks.clear();
  plan->brick_words.clear();
  plan->tile_params.clear();
  plan->n_padded = plan->pad_spl = plan->pad_tf1 = 0;
  // the bricks first, each on its own (the inverted index and the metadata stream of a brick are the expensive part
  // of planning a ragged chunk, and bricks do not depend on one another) ...
  struct BrickJob { uint64_t t0, t1; BrickHost bh; std::vector<uint32_t> words; std::string err; bool ok = true; };
  std::vector<BrickJob> jobs;
  auto tile_resp = [&](uint64_t t) {  // TSpline3 responses of a tile's events
    uint64_t n = 0;
    for (uint64_t e = t * M3_TE, e1 = std::min<uint64_t>(in.n_events, (t + 1) * M3_TE); e < e1; ++e) n += in.npe[2 * e];
    return n;
  };
  for (uint64_t t0 = 0; t0 < n_tiles;) {
    uint64_t t1 = t0 + 1;
    if (!uniform[t0]) {
      // stages hold 3 000 - 5 120 responses (an event may hold 15 of a stage): stay well inside M3_BRICK_MAX_STAGES
      uint64_t resp = tile_resp(t0);
      const uint64_t cap = t0 * M3_TE >= short_from && t0 * M3_TE < short_to ? tiles_per_group * G * 3 / 4 : tiles_per_group * G;
      while (t1 < n_tiles && t1 - t0 < cap && t1 * M3_TE != short_from && t1 * M3_TE != short_to && !uniform[t1] &&
             (t1 - t0 < tiles_per_group || resp + tile_resp(t1) <= 3000ull * uint64_t(M3_BRICK_MAX_STAGES))) {
        resp += tile_resp(t1);
        ++t1;
      }
      jobs.emplace_back();
      jobs.back().t0 = t0;
      jobs.back().t1 = t1;
    }
    t0 = t1;
  }
  // the dictionary the bricks' TF1 codes step through: every parameter with a TF1 response in a brick of this chunk
  plan->tf1_dict.clear();
  plan->dict_off = 0;
  std::vector<int32_t> dict_index(size_t(n_params), -1);
  if (!jobs.empty()) {
    std::vector<uint8_t> seen(size_t(n_params), 0);
    for (const BrickJob& job : jobs) {
      const uint64_t e0 = job.t0 * M3_TE, e1 = std::min<uint64_t>(in.n_events, job.t1 * M3_TE);
      for (uint64_t e = e0; e < e1; ++e) {
        const unsigned int ct = in.npe_tf1[2 * e], t0 = in.npe_tf1[2 * e + 1];
        for (unsigned int i = 0; i < ct; ++i) seen[size_t(in.param_no_tf1[t0 + i])] = 1;
      }
    }
    for (int p = 0; p < n_params; ++p)
      if (seen[size_t(p)]) {
        dict_index[size_t(p)] = int32_t(plan->tf1_dict.size());
        plan->tf1_dict.push_back(uint16_t(p));
      }
  }
  parallel_for(jobs.size(), [&](uint64_t j) {
    BrickJob& job = jobs[j];
    const uint64_t e0 = job.t0 * M3_TE, e1 = std::min<uint64_t>(in.n_events, job.t1 * M3_TE);
    job.ok = plan_brick(n_params, npts, in, dict_index, e0, e1, G, &job.bh, &job.words, &job.err);
  });
  for (const BrickJob& job : jobs)
    if (!job.ok) { *err = job.err; return false; }  // the first failing brick in event order
  // ... then tiles and bricks take their places in the stores in event order
  size_t next_job = 0;
  for (uint64_t t0 = 0; t0 < n_tiles;) {
    uint64_t t1 = t0 + 1;
    if (uniform[t0]) {
      for (uint64_t t = t0; t < t1; ++t) {
        TileHost th;
        const uint64_t e0 = t * M3_TE, e1 = std::min<uint64_t>(in.n_events, e0 + M3_TE);
        th.first_event = e0;
        th.n_events = uint32_t(e1 - e0);
        th.kind = kTileUniform;
        th.padded = uniform[t] == 2 ? 1u : 0u;
        th.n_spl_slots = uint32_t(uni_s[t].size());
        th.n_tf1_slots = uint32_t(uni_t[t].size());
        th.list_off = plan->tile_params.size();
        plan->tile_params.insert(plan->tile_params.end(), uni_s[t].begin(), uni_s[t].end());
        plan->tile_params.insert(plan->tile_params.end(), uni_t[t].begin(), uni_t[t].end());
        uint64_t planes = 0;
        for (uint32_t j = 0; j < th.n_spl_slots; ++j) planes += uint64_t(npts[uni_s[t][j]] - 1);
        if (th.padded) {
          plan->n_padded++;
          for (uint64_t e = e0; e < e1; ++e) {
            plan->pad_spl += th.n_spl_slots - in.npe[2 * e];
            plan->pad_tf1 += th.n_tf1_slots - in.npe_tf1[2 * e];
          }
        }
        th.n_planes = planes;
        th.spl_off = plan->spl_float4;
        th.tf1_off = plan->tf1_float2;
        th.meta_off = plan->meta_int32;
        plan->spl_float4 += planes * M3_TE;
        plan->tf1_float2 += uint64_t(th.n_tf1_slots) * M3_TE;
        plan->meta_int32 += 2ull * th.n_spl_slots + th.n_tf1_slots;
        plan->tiles.push_back(th);
        plan->n_uniform++;
      }
    } else {
      BrickJob& job = jobs[next_job++];
      t1 = job.t1;
      BrickHost& bh = job.bh;
      bh.words_off = plan->brick_words.size();
      plan->brick_words.insert(plan->brick_words.end(), job.words.begin(), job.words.end());
      std::vector<uint32_t>().swap(job.words);
      bh.spl_off = plan->spl_float4;
      bh.tf1_off = plan->tf1_float2;
      plan->spl_float4 += bh.n_spl_f4;
      plan->tf1_float2 += bh.n_tf1_f2;
      plan->bricks.push_back(std::move(bh));
      plan->n_brick++;
    }
    t0 = t1;
  }
  plan->dict_off = plan->brick_words.size();
  for (size_t i = 0; i < plan->tf1_dict.size(); i += 2)
    plan->brick_words.push_back(uint32_t(plan->tf1_dict[i]) | (uint32_t(i + 1 < plan->tf1_dict.size() ? plan->tf1_dict[i + 1] : 0) << 16));
  return true;
}

void fill_chunk(int /*n_params*/, const int16_t* npts, const ChunkInput& in, const ChunkPlan& plan, float* spl,
                float* tf1, int32_t* meta) {
  parallel_for(plan.tiles.size(), [&](uint64_t t) {
    const TileHost& th = plan.tiles[t];
    if (th.kind != kTileUniform) return;
    const uint64_t e0 = th.first_event;
    const int16_t* list_s = plan.tile_params.data() + th.list_off;
    const int16_t* list_t = list_s + th.n_spl_slots;
    int32_t* m = meta + th.meta_off;
    float* tile_spl = spl + 4 * th.spl_off;
    float* tile_tf1 = tf1 + 2 * th.tf1_off;
    // slot meta: {param, first plane}
    uint64_t plane0 = 0;
    std::vector<uint32_t> slot_plane(th.n_spl_slots), slot_nseg(th.n_spl_slots);
    for (uint32_t j = 0; j < th.n_spl_slots; ++j) {
      const int p = list_s[j];
      m[2 * j] = p;
      m[2 * j + 1] = int32_t(plane0);
      slot_plane[j] = uint32_t(plane0);
      slot_nseg[j] = uint32_t(npts[p] - 1);
      plane0 += slot_nseg[j];
    }
    for (uint32_t j = 0; j < th.n_tf1_slots; ++j) m[2 * th.n_spl_slots + j] = list_t[j];
    static const float kIdentitySpl[4] = {1.f, 0.f, 0.f, 0.f};  // y = 1, b = c = d = 0: the cubic is exactly 1.0f
    static const float kIdentityTf1[2] = {0.f, 1.f};            // a = 0, b = 1: fmaf(0, x, 1) is exactly 1.0f
    // coefficients: event-major AoS records -> [plane][lane] float4.  The event's own list is merged into the
    // tile's list (both ascending for a padded tile; identical for an exactly uniform one): a slot the event
    // does not have gets the identity response, which leaves its float product bit for bit unchanged.
    for (uint32_t le = 0; le < uint32_t(M3_TE); ++le) {
      if (le < th.n_events) {
        const uint64_t e = e0 + le;
        const unsigned int se = in.npe[2 * e + 1], ce = in.npe[2 * e];
        unsigned int i = 0;
        for (uint32_t j = 0; j < th.n_spl_slots; ++j) {
          float* dst = tile_spl + (uint64_t(slot_plane[j]) * M3_TE + le) * 4;
          const bool has = i < ce && in.param_no[se + i] == list_s[j];
          const float* src = has ? in.coeff_many + 4ull * in.nknots_arr[se + i] : nullptr;
          for (uint32_t k = 0; k < slot_nseg[j]; ++k) std::memcpy(dst + uint64_t(k) * M3_TE * 4, has ? src + 4 * k : kIdentitySpl, 16);
          i += has;
        }
        const unsigned int te = in.npe_tf1[2 * e + 1], cte = in.npe_tf1[2 * e];
        i = 0;
        for (uint32_t j = 0; j < th.n_tf1_slots; ++j) {
          const bool has = i < cte && in.param_no_tf1[te + i] == list_t[j];
          std::memcpy(tile_tf1 + (uint64_t(j) * M3_TE + le) * 2, has ? in.coeff_tf1 + 2ull * (te + i) : kIdentityTf1, 8);
          i += has;
        }
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
  parallel_for(plan.bricks.size(), [&](uint64_t b) {
    const BrickHost& bh = plan.bricks[b];
    fill_brick(npts, in, bh, plan.brick_words.data() + bh.words_off, spl, tf1);
  });
}

}  // namespace m3hip
