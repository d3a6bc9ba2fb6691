// m3hip_layout.h -- host-side re-layout of a reference-format monolith chunk into the
// engine's tile store.  Pure C++ (no HIP): the engine calls it from m3hip_monolith_append,
// and tests inspect its output directly through the m3hip_layout_* debug exports.
//
// Reference layout in (Splines/SplineMonolith.cpp:53-250): event-major CSR, one
// 16*nKnots-byte AoS record {y,b,c,d}[knot] per spline, of which a step reads ONE quad.
//
// Engine layout out ("segment-major tiles"):
//   events are cut into tiles of M3_TE = 256 consecutive events.
//   UNIFORM tile  (every event of the tile has the same ordered parameter list):
//       slot j of the tile <-> parameter plist[j]; the tile stores, per slot and per
//       spline SEGMENT k (npts-1 of them), one plane of M3_TE float4 {y,b,c,d}:
//           spl[(plane0[j] + k) * M3_TE + lane]
//       so the quads a step needs for (slot j, segment seg[p]) are 4 KB contiguous and a
//       wavefront's 64 loads are one 1 KB line-aligned burst.  TF1: tf1[j * M3_TE + lane]
//       float2 {a,b}.  Lane = event inside the tile; the per-event product runs over j in
//       the reference's order (splines in event order, then TF1s).
//   BRICK (ragged parameter lists): G x 1024 consecutive events in which the lists differ, G = 1, 2 or 4 (the planner's
//   `brick_groups`); lane l of the brick's 1024-lane workgroup owns events l, l + 1024, ... ("slabs" h = 0 .. G-1).  Larger
//   bricks spread the fixed phases of a workgroup's life (first round trips, pipeline fill and drain, flush, dispatch)
//   over more events and lengthen the runs below.
//   TSpline3 responses: a per-brick CSC.  They are regrouped by PARAMETER so that the quads a step needs for
//   (parameter p, segment seg[p]) are again contiguous:
//       run (p) = the n_run events of the brick that carry p, ascending event order;
//       spl[coef_off(p) + k * stride + q]   float4 of segment k for the q-th event of the run
//       (stride = n_run rounded up to 8 quads: every plane starts on a 128-B line, the unit the memory side fetches --
//       tools/sector_fetch.hip: a line costs the same whether 16 or 128 of its bytes are used).
//   Runs are packed into STAGES (ascending parameter ranges) of <= M3_STAGE_CAP responses in which no event holds more
//   than M3_STAGE_EVENT_MAX responses.  A stage's weights are evaluated lane-per-response (coalesced) and scattered into a
//   COMPACT LDS array of one float per response, ordered so that the event's lane can multiply its own responses in
//   ascending parameter order (the reference's order: the float product is bit-identical): wave-block after wave-block
//   (slab h, wave w = events 1024 h + 64 w ... + 63), event after event, an event's responses next to each other.  Per response: ONE uint16 {slot:13 << 2 | last-of-its-run:1} (bits 2..14 are the byte
//   offset of the response's slot); per event and stage: a 4-bit count; per wave-block and stage: the uint16 slot of its
//   first row.  The array holds M3_STAGE_CAP floats whatever the brick's size (round 2's [rank][event] array grew with it).
//   TF1 responses have no segment -- every step reads all of them -- so they need no regrouping: they are stored
//   event-major per wave-block as jagged rows (row j = the j-th TF1 of each event that has one, lanes packed),
//   read lane-per-event straight into the register product.  Per response: ONE byte, the step from the previous
//   parameter of the event in the chunk's dictionary of TF1 parameters (255: step 254 and no response).
//   Per-brick metadata is a uint32 word stream (layout in m3hip_layout.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace m3hip {

constexpr int M3_TE = 256;       // events per tile (= workgroup size of the reweight kernels)
constexpr int M3_GROUP = 1024;   // classification granularity; chunk sizes are multiples of it
constexpr int M3_STAGE_CAP = 5120;       // TSpline3 responses per brick stage (5 per lane of the 1024-lane workgroup)
constexpr int M3_STAGE_MAX_RUNS = 512;   // runs per stage (the resolved run table lives in LDS)
constexpr int M3_STAGE_EVENT_MAX = 15;   // responses one event may hold in a stage (its count is a nibble)
constexpr int M3_BRICK_MAX_GROUPS = 4;   // a brick holds up to 4 x M3_GROUP events
// 16-bit response words a lane holds in a stage of n_resp list entries = the length of its record (m3hip_layout.cpp)
constexpr uint32_t stage_record_words(uint32_t n_resp) { return (n_resp + 1023u) / 1024u; }
constexpr int M3_STAGE_PAIR_WORDS = M3_STAGE_CAP / 64;  // {run of the group's first response, first slot of wave-block} pairs per stage
constexpr int M3_BRICK_MAX_STAGES = 250; // stage headers are staged in LDS (one load per lane of the workgroup)
constexpr int M3_BRICK_HDR0 = 8;         // words in front of the stage headers
constexpr int M3_STAGE_HDR = 4;          // words per stage header
constexpr int M3_RUN_WORDS = 2;          // words per run entry
constexpr int M3_PLANE_ALIGN = 8;        // quads: segment planes of a run start on 128-B lines
constexpr int M3_WAVE = 64;              // events per jagged TF1 block
constexpr int M3_TF1_MAX_CODES = 255;    // TF1 codes (responses + skips) per event
constexpr bool kTf1CntInSigmaG2 = false;  // bricks of 2048 events keep the TF1 code counts as a byte array (m3hip_kernels.hip: book_word)
constexpr int M3_TF1_SKIP = 255;         // code: advance 254 dictionary entries, no response
// idle lanes of the brick kernel read (and drop) what lies behind the list they index: every chunk's stores are allocated
// with this much room behind them
constexpr int M3_STORE_SLACK_BYTES = 32 * 1024;
constexpr int M3_BRICK_ALIGN_WORDS = 32; // every brick's stream starts on a 128-B line

enum TileKind : uint32_t { kTileUniform = 0, kTileBrick = 1 };

// Host-side tile descriptor: offsets into the chunk's stores.
struct TileHost {
  uint64_t spl_off = 0;   // float4 index of the tile's first plane
  uint64_t tf1_off = 0;   // float2 index
  uint64_t meta_off = 0;  // int32 index: [2*n_spl_slots] {param, plane0}, then [n_tf1_slots] {param}
  uint32_t n_spl_slots = 0;
  uint32_t n_tf1_slots = 0;
  uint32_t n_events = 0;  // valid lanes (<= M3_TE)
  uint32_t kind = kTileUniform;
  uint64_t n_planes = 0;  // spline planes in the tile (sum of segments over slots)
  uint64_t first_event = 0;  // chunk-local index of lane 0
  // the tile's ordered slot lists (parameters), in ChunkPlan::tile_params: [n_spl_slots] then [n_tf1_slots].
  // For an exactly uniform tile this is every event's own list; for a PADDED tile it is the ascending union
  // over the tile's events, and an event that lacks a parameter gets the identity response in that slot.
  uint64_t list_off = 0;
  uint32_t padded = 0;
};

// Host-side brick descriptor.
struct BrickHost {
  uint64_t spl_off = 0;    // float4 index of the brick's spline store
  uint64_t tf1_off = 0;    // float2 index of the brick's TF1 store
  uint64_t words_off = 0;  // uint32 index of the brick's metadata stream
  uint64_t n_spl_f4 = 0, n_tf1_f2 = 0, n_words = 0;
  uint32_t n_events = 0;
  uint64_t first_event = 0;  // chunk-local
  // fill-time copy lists: source spline / TF1 index of every response, in store order
  std::vector<uint32_t> spl_src;   // per spline response (run-major): index into the chunk's spline arrays
  std::vector<uint32_t> tf1_src;   // per TF1 slot (wave-major, jagged rows inside); 0xFFFFFFFF: skip slot -> identity {0,1}
  uint64_t bytes_tables = 0, bytes_book = 0, bytes_codes = 0;  // of the metadata stream: the stages' tables (run entries, pair words, counts), the TF1 bookkeeping, the TF1 codes
  uint32_t n_stages = 0;           // TSpline3 stages
  uint32_t groups = 1;             // G: the brick spans up to G x M3_GROUP events
  bool tf1_sorted = false;         // the TF1 lane-slots take the events in order of descending code count
  bool tf1_cnt_in_sigma = false;   // no event has more than 63 TF1 codes: the slots' counts travel in their sigma words
};

struct ChunkInput {
  uint64_t n_events = 0;
  const float* coeff_many = nullptr;
  uint64_t total_nknots = 0;
  const unsigned int* nknots_arr = nullptr;
  const int16_t* param_no = nullptr;
  uint64_t n_splines = 0;
  const float* coeff_tf1 = nullptr;
  const int16_t* param_no_tf1 = nullptr;
  uint64_t n_tf1 = 0;
  const unsigned int* npe = nullptr;
  const unsigned int* npe_tf1 = nullptr;
};

struct ChunkPlan {
  std::vector<TileHost> tiles;
  uint64_t spl_float4 = 0;  // size of the spline store, float4 units
  uint64_t tf1_float2 = 0;  // size of the TF1 store, float2 units
  uint64_t meta_int32 = 0;  // size of the slot-meta store
  uint64_t n_uniform = 0, n_brick = 0;
  std::vector<BrickHost> bricks;
  std::vector<uint32_t> brick_words;  // metadata streams of all bricks, concatenated; then the TF1 dictionary
  // ascending list of the parameters that carry a TF1 response somewhere in the chunk's bricks; the bricks' TF1 codes
  // step through it.  Stored behind the streams in brick_words, two uint16 per word, from word dict_off on.
  std::vector<uint16_t> tf1_dict;
  uint64_t dict_off = 0;
  std::vector<int16_t> tile_params;   // slot lists of all uniform tiles, concatenated (TileHost::list_off)
  uint64_t n_padded = 0, pad_spl = 0, pad_tf1 = 0;  // padded tiles and the identity responses they add
};

// A ragged group is laid out as identity-padded uniform tiles instead of a brick when the responses its events
// really have fill at least this fraction of the padded slots.  Measured break-even on a 1e6-event, 128-parameter
// sample (scripts/kbench.py pad, round 2's brick kernel): at 90 % fill padded tiles and bricks take the same time
// (305.5 / 306.1 us), at 80 % the bricks are 12 % faster (269 vs 305 us), at 70 % 22 % (237 vs 306 us) -- long runs make
// bricks stream at 76 % of the HBM peak, and padded tiles stream their padding.  (Round 1's brick kernel broke even at
// 75 %.)  The padded form also costs 1/fill of the store.  Settable for experiments (M3HIP_PAD_MIN_FILL; > 1 switches the
// padding off).
constexpr double kPadMinFillDefault = 0.88;
extern double g_pad_min_fill;
// TF1 lane-slots by descending code count: 0 never, 1 always, 2 (default) when an event of the brick has more than 16 codes
// (the planner's rule; the other two values are for the CPU tests of the layout)
extern int g_tf1_sort;
extern int g_pad_runs;  // runs padded to whole lines in the stages' response lists: 0 never, 1 always, 2 where it costs <= 8 % more entries (env M3HIP_PAD_RUNS)

// Validates the chunk against (n_params, npts) and sizes the tile store.
// Returns false and sets err on inconsistent input.
// brick_groups: G of the bricks (1, 2 or 4); a brick whose TSpline3 responses would need more stages than the header block
// holds falls back to a smaller G.
// [short_from, short_to): chunk-local events (multiples of M3_TE) whose bricks are cut at three quarters of the size -- four short
// bricks where three full ones would be -- and whose ends no brick spans: the engine's way of handing every compute unit the same
// number of bricks (m3hip_upload.hip: ShortBrickPlan)
bool plan_chunk(int n_params, int max_knots, const int16_t* npts, const ChunkInput& in, ChunkPlan* plan,
                std::string* err, int brick_groups = 1, uint64_t short_from = ~0ull, uint64_t short_to = ~0ull);

// Fills caller-provided stores (sizes from the plan).  Multi-threaded.
void fill_chunk(int n_params, const int16_t* npts, const ChunkInput& in, const ChunkPlan& plan, float* spl /*float4*/,
                float* tf1 /*float2*/, int32_t* meta);

}  // namespace m3hip
