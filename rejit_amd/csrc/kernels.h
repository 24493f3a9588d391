// rejit_amd/csrc/kernels.h -- parameter blocks and launchers of the HIP kernels.
#ifndef REJIT_AMD_KERNELS_H_
#define REJIT_AMD_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dense_streams.h"
#include "device_program.h"
#include "run_scan.h"

namespace rejit_amd {

// device counters (unsigned long long[kCntSize])
enum { kCntCands = 0, kCntFinal = 1, kCntOverflow = 2, kCntAdjacent = 3, kCntHits = 4, kCntMaxRegion = 5, kCntOverrun = 6,
  kCntUnordered = 7,  // check_and_interleave: the candidates are not already the result
  kCntConflict = 8,   // behind mode: a candidate hidden by an earlier match ends after it (the run is repeated dense)
  kCntSharedMax = 9,  // plane scan (plane_scan.hip): fullest SHARED candidate region when one overflowed, else 0
  kCntLongWalks = 10, // walks of the run that went past kLongWalk bytes (device_program.h: rj_lane_longest)
  kCntSlowStarts = 11, // dense_streams.hip: starts that outlived the register steps and took the scalar walk
  kCntSize = 12 };
// offsets_gather_check keeps 2 x kOgcMaxBlocks granules behind the counters of a scan (kernels.hip): a counter block is
// kCntSize + 2 * kOgcMaxBlocks words, zeroed when it is allocated
constexpr uint32_t kOgcMaxBlocks = 256;
constexpr size_t kCounterBlockWords = kCntSize + 2 * kOgcMaxBlocks;
static_assert(kCntLongWalks - kCntOverrun == kLongWalksAfterOverrun, "rj_lane_longest finds the long-walk counter behind the overrun flag");

constexpr uint64_t kNoMatch = ~0ull;  // cand_end of a hit at which nothing matches

constexpr int kFinalizeCap = 2048;  // candidates the single-workgroup finalize sorts in LDS

struct ScanParams {
  const uint8_t* text;   // 16-byte aligned; bytes [0, n) are readable
  uint64_t n;
  uint64_t sb, se;       // candidate starts lie in [sb, se), se <= n + 1
  uint64_t wlo, whi;     // window positions scanned (windows mode): [wlo, whi)
  uint64_t span_chunks;  // wave w owns chunks [first + w*span, first + (w+1)*span)
  uint64_t* hits;        // n_regions regions of region_cap entries, region w = wave w
  uint32_t region_cap;
  uint32_t* hit_counts;  // [n_regions] hits each wave found (may exceed region_cap)
  // scan_windows: when set, the kernel zeroes the counter block (its first use is in the kernels
  // that follow on the stream), which saves the host a memset command per run
  unsigned long long* zero_counters;
};

struct WindowSet {
  uint32_t value0[kDevMaxWindows], mask0[kDevMaxWindows];
  uint32_t value1[kDevMaxWindows], mask1[kDevMaxWindows];
  uint32_t offset;
  uint32_t len;      // 1..8 bytes
  uint32_t masked;   // some mask byte inside [0,len) is a wildcard / len is not 4 or 8
  uint32_t two_level;  // test the first dword of all positions before the second (large alphabets)
  // nibble filter (len > 4, small alphabets): value0/mask0 hold the LOW NIBBLES of the 8 window
  // bytes packed into one dword (byte i < 4 -> bits 8i..8i+3, byte i >= 4 -> bits 8(i-4)+4..+7);
  // a superset test with one compare per window instead of two -- verify removes the aliases
  uint32_t nibble;
};

struct VerifyParams {
  const uint8_t* text;
  uint64_t n;
  uint64_t* hits;
  const uint64_t* offsets;  // [n_regions + 1], offsets[n_regions] = number of hits
  uint32_t n_regions;
  uint32_t region_cap;
  uint64_t sb, se;          // candidate starts must lie in [sb, se)
  uint32_t expand;          // slots per hit (floating windows: one per possible start), >= 1
  uint32_t float_max;       // slot (hit h, delta d): start = hit - float_max + d
  uint64_t* cand_begin;     // one slot per (hit, delta), in hit (= text) order
  uint64_t* cand_end;       // kNoMatch when nothing matches at that start
  unsigned long long* counters;  // kCntOverrun is set when a walk hits kMaxSimSteps
};

struct FinalizeParams {
  const uint64_t* cand_begin;
  const uint64_t* cand_end;
  uint64_t cands_cap;
  uint64_t* out;         // (begin,end) pairs, ordered
  uint64_t out_cap;
  unsigned long long* counters;
  // state carried in from the text before this range (multi-GPU / segmented runs):
  uint64_t carry_cur;       // smallest begin the first match may have
  uint64_t carry_prev_end;  // end of the previous match (zero-length rule)
  int have_prev;
  int detect_adjacent;      // set counters[kCntAdjacent] when a candidate begins where another ends
  int detect_conflict;      // behind mode: set counters[kCntConflict] when a skipped candidate ends after the match hiding it
  uint32_t expand;          // candidate slots per hit (see VerifyParams)
};

// grid of the scan kernels for a run over `chunks` 1-KiB chunks: workgroups (4 waves each),
// regions (= waves) and chunks per wave
// Several patterns in one pass over the text: every pattern has up to two 5..8-byte
// windows in nibble form (see WindowSet::nibble) and its own hit regions; patterns are processed in
// groups of kFuseGroup with their constants in scalar registers.
constexpr int kMaxFused = 18;
constexpr int kFuseGroup = 3;
struct FusedParams {
  const uint8_t* text;
  uint64_t n;
  uint64_t sb, se;        // candidate starts [sb, se)
  uint64_t span_chunks;
  uint32_t n_patterns;    // a multiple of kFuseGroup (padding entries have region_cap 0)
  // shared prefilter: every window is within one nibble of base[b] for some b < n_bases (0 = no such
  // bases: every pattern is tested at every position); value / mask then hold 3-bit nibbles
  uint32_t n_bases;
  uint32_t base[2];
  uint32_t value[kMaxFused][2], mask[kMaxFused][2];
  uint32_t offset[kMaxFused], len[kMaxFused];
  uint64_t* hits[kMaxFused];
  uint32_t region_cap[kMaxFused];
  uint32_t* hit_counts[kMaxFused];
  unsigned long long* zero_counters[kMaxFused];  // may be null
};
// several patterns' scans as ONE launch (scan_windows_train): per pattern what ScanParams + WindowSet hold
struct TrainParams {
  const uint8_t* text;
  uint64_t n;
  uint64_t sb, se;
  uint64_t span_chunks;
  uint32_t n_patterns;
  uint32_t value[kMaxFused][2], mask[kMaxFused][2];   // nibble form, see WindowSet::nibble
  uint32_t offset[kMaxFused], len[kMaxFused];
  uint64_t wlo[kMaxFused], whi[kMaxFused];
  uint64_t* hits[kMaxFused];
  uint32_t region_cap[kMaxFused];
  uint32_t* hit_counts[kMaxFused];
  unsigned long long* zero_counters[kMaxFused];
};
void launch_scan_windows_train(const TrainParams& t, bool masked, int grid, hipEvent_t t0, hipEvent_t t1, hipStream_t st);

// ---- small texts: ONE kernel, one workgroup, one host synchronise (match_small): scan + verify + selection
// for a text of <= kSmallMaxText bytes and a lane-sized automaton.  hdr[0] = matches, hdr[1] = 1 when the
// general pipeline has to take the run (too many candidates, a walk limit, a Q8-sensitive adjacency).
constexpr uint32_t kSmallMaxText = 32768;
constexpr uint32_t kSmallMaxCands = 4096;
constexpr uint32_t kSmallMaxTableWords = 4096;
struct SmallParams {
  const uint8_t* text;        // device-accessible (HBM or pinned host memory)
  uint32_t n;
  uint32_t sb, se;            // starts [sb, se), se <= n + 1
  uint64_t carry_cur, carry_prev_end;
  int have_prev;
  int q8_risk;
  uint64_t* out;              // (begin, end) pairs, device-accessible
  uint32_t out_cap;
  unsigned long long* hdr;    // [2]
};
size_t small_lds_bytes(const DevProgram& P, uint32_t n);
size_t small_lds_limit();   // dynamic LDS a workgroup of match_small may use on this device
void launch_match_small(const SmallParams& a, const DevProgram& P, hipStream_t st);

struct BoundsParams {
  int n_lists;
  const uint64_t* spans[kMaxFused];
  uint64_t count[kMaxFused];
};
void launch_first_last(const BoundsParams& a, uint64_t* pinned_bounds, hipStream_t st);
void launch_bounds_rows(const BoundsParams& a, int64_t offset, int first_round, int64_t* d_rows, hipStream_t st);
void launch_carry_decide(const int64_t* d_all, int world, int rank, int n_patterns, int64_t* out, hipStream_t st);
// out[i] = local[i] + offset for the 2 * count offsets of a shard's result (rj_scan_gather_spans: local -> global)
void launch_globalize_spans(const uint64_t* local, uint64_t count, int64_t offset, uint64_t* out, hipStream_t st);
void launch_scan_windows_fused(const FusedParams& a, int grid, hipEvent_t t0, hipEvent_t t1, hipStream_t st);

// ---- plane scan (plane_scan.hip): the one-pass scan for SEVERAL patterns whose 8-byte windows all lie within
// one byte of <= 2 base windows over an alphabet of <= 4 symbols (regexdna: every window is `agggtaaa` or
// `tttaccct` with at most one position turned into a class).  The text is reduced to 2-bit symbol codes
// ((byte >> code_shift) & 3), held as bit planes, and "at most one symbol differs from base b" is evaluated
// for 32 positions per VALU instruction.  Positions that pass go to ONE candidate list shared by all
// patterns (regions as everywhere: region w = wave w, sorted by construction); classify_shared_multi then
// tests every candidate against every pattern's own windows and automaton.
struct PlaneParams {
  const uint8_t* text;   // 16-byte aligned
  uint64_t n;
  uint64_t sb, se;       // candidate starts [sb, se)
  uint64_t span_pairs;   // wave w owns the 2-KiB pairs [first + w*span, first + (w+1)*span)
  uint32_t offset;       // window offset inside a match (the same for all patterns); windows are 8 bytes
  uint32_t code_shift;
  uint32_t n_bases;      // 1 or 2
  // base b, window byte i: lo[b][i] / hi[b][i] = 0 when the low / high bit of its symbol code is 1, else ~0
  uint32_t lo[2][8], hi[2][8];
  uint64_t* hits;        // shared candidate regions (starts s = w - offset)
  uint32_t region_cap;
  uint32_t* hit_counts;  // [n_regions]
  uint32_t n_zero;
  unsigned long long* zero_counters[kMaxFused];  // counter blocks the kernel clears (one per pattern)
};
void launch_plane_scan(const PlaneParams& a, int grid, hipEvent_t t0, hipEvent_t t1, hipStream_t st);

// ---- plane count (plane_count.hip, exact_count.h): the same scan for callers that want COUNTS (MatchAllCount,
// reference src/rejit.cc:203-208), scan + exact classification + counting in ONE kernel.  The caller zeroes `acc`
// once (kPcAccWords words); a run leaves its results in `host_out` (pinned, kPcHostWords words) and a device copy in `acc`.
enum { kPcConflict = 1,   // matches of one pattern that overlap in a way the kernel does not resolve itself (three candidates in a row, each
                          // closer to its neighbour than a match is long, or a pair across two waves' spans): run void
       kPcVoid = 2 };     // a 2-KiB block held more candidates than a wave's ring: run void
constexpr uint32_t kPcBounds = 0;                      // acc: [32][2] first / last match (kPcNone, or begin | length << kPcLenShift) of the last run (device copy)
constexpr uint32_t kPcTotals = kPcBounds + 64;         // [32]: the last run's counts (device copy)
constexpr uint32_t kPcFinishGroups = 8;                // workgroups of plane_count_finish
constexpr uint32_t kPcTicket = kPcTotals + 32;         // their ticket (zero between runs)
constexpr uint32_t kPcGroupRows = kPcTicket + 16;      // [kPcFinishGroups][3][32]: their sums (slot 31: the flags) | first | last row + 1 with a match (0: none)
constexpr uint32_t kPcAccWords = kPcGroupRows + kPcFinishGroups * 96;
constexpr uint32_t kPcHostCount = 0, kPcHostFlags = 32, kPcHostBounds = 40, kPcHostWords = 40 + 64;
constexpr unsigned long long kPcNone = ~0ull;          // bounds: no match
constexpr uint32_t kPcLenShift = 56;                   // a bound = begin | length << 56
struct PlaneCountParams {
  const uint8_t* text;   // 16-byte aligned
  uint64_t n;
  uint64_t sb, se;       // match begins [sb, se)
  uint64_t first_block, end_block;   // the 2-KiB blocks that hold their window positions
  uint64_t span_blocks;  // wave w owns span_blocks blocks, the first span_extra waves one more, one after the other from first_block
  uint32_t span_extra;
  uint32_t code_shift, n_bases, n_patterns;
  uint32_t batch_at;     // a wave classifies what its ring holds (<= 64 at a time) when that many are waiting, 1..64
  uint32_t mask_bits;    // bit 16 b + 2 i / + 1: the low / high bit of the symbol code of base b's window byte i is 0
  uint32_t base_lo[2], base_hi[2];   // the bases' 8 bytes (ExactCountPlan)
  const uint32_t* table;             // ExactCountPlan::table in device memory (the general shape: the blob of descriptors + automaton tables)
  uint32_t table_words;              // the general shape: words of that blob (a multiple of 4), copied to LDS by every workgroup
  unsigned long long* acc;
  uint32_t* wg_rows;                 // [grid][32]: every workgroup's counts (slot 31: its flags)
  unsigned long long* wg_bounds;     // [grid][32][2]: every workgroup's first / last match per pattern (as kPcBounds)
  unsigned long long* host_out;
};
void launch_plane_count(const PlaneCountParams& a, int grid, hipEvent_t t0, hipEvent_t t1, hipStream_t st);
// The same streaming loop and test with the candidates written to the shared candidate regions of the span pipeline instead of
// classified in place (round 6: plane_scan<NB>'s successor -- 32 contiguous bytes per lane, code planes through the VGPR index
// mode; region w = wave w, in text order; c.first_block / span_blocks / span_extra deal the blocks out, c.text / n / code_shift /
// n_bases / mask_bits as for the count).  What plane_scan leaves behind, for classify_shared_multi.
struct PlaneListParams {
  PlaneCountParams c;
  uint64_t* hits;        // shared candidate regions (starts s = w - offset)
  uint32_t region_cap;
  uint32_t offset;       // window offset inside a match (the same for all patterns)
  uint32_t* hit_counts;  // [n_regions]
  uint32_t n_zero;
  unsigned long long* zero_counters[kMaxFused];  // counter blocks the kernel clears (one per pattern)
};
void launch_plane_list(const PlaneListParams& a, int grid, hipEvent_t t0, hipEvent_t t1, hipStream_t st);
// the rows of that launch (grid workgroups) added up: counts, flags, bounds -> a.host_out and a.acc
void launch_plane_count_finish(const PlaneCountParams& a, int grid, hipStream_t st);
// launch_bounds_rows after a counts run: a pattern without a list (spans[p] == nullptr) takes its count and first /
// last match (8 bytes long) from the kernel's device copy in `acc`
void launch_bounds_rows_counts(const BoundsParams& a, const unsigned long long* acc, int64_t offset, int first_round, int64_t* d_rows, hipStream_t st);

// Dense mode as bit streams (dense_streams.h / dense_streams.hip): patterns whose candidates cannot overlap, one pass, the
// pairs written once at their final place (tiles of 32 KiB, the prefix scan of tile_lookback.h).  counters[kCntFinal] = the number of
// matches (also beyond out_cap: the host grows `out` and runs again); counters[kCntOverrun] = 1: a walk beyond max_walk
// or a time-out of the prefix scan, the run is void; counters[kCntSlowStarts] = starts that took the scalar walk.
struct StreamParams {
  const uint8_t* text;   // 16-byte aligned
  uint64_t n;
  uint64_t sb, se;       // starts [sb, se), se <= n + 1
  uint64_t first_tile, n_tiles;   // stream_tiles()
  unsigned long long* granules;   // set by the launcher
  unsigned long long* ticket;
  uint64_t* out;
  uint64_t out_cap;
  unsigned long long* counters;
  unsigned long long* host_counters;
  uint32_t max_walk;     // DevProgram::max_walk: a walk this long voids the run (the carry scan takes it)
  StreamPlan plan;
};
uint64_t stream_tiles(uint64_t sb, uint64_t se, uint64_t n, uint64_t* first_tile);
size_t stream_scratch_bytes(uint64_t n_tiles);
void launch_dense_streams(StreamParams a, unsigned long long* scratch, hipEvent_t t0, hipEvent_t t1, hipStream_t st);

// Patterns with ONE long-lived thread in one loop position (run_scan.h / run_scan.hip, round 6): tiles of kRunTile bytes,
// run_summary -> run_resolve -> run_emit.  counters[kCntFinal] (and host_counters) = the number of matches after run_resolve.
constexpr uint64_t kRunTile = 8192;
struct RunSummary {
  unsigned long long r1;             // the tile's first break (~0: none)
  unsigned long long a1, b1, b1a;    // before it (or in the whole tile): the first A, the last B, the last B behind that A
  unsigned long long open_s, open_q; // the segment open at the tile's end (tiles with a break)
  unsigned long long cnt;            // matches closed by the tile's other breaks
  unsigned long long pad;
};
struct RunTileIn {
  unsigned long long s, q;           // the segment open at the tile's begin
  unsigned long long off;            // the tile's first output pair
  unsigned long long cnt;            // the pairs it writes (0: run_emit skips the tile)
};
struct RunParams {
  const uint8_t* text;   // 16-byte aligned
  uint64_t n;
  uint64_t sb, se;       // match begins [sb, se)
  uint64_t min_start;    // starts below are not looked at (the own range's begin, or where a carried-in match ends)
  uint32_t blocked_in;   // the segment that holds min_start has had its match (a carried-in match with a B that is no break)
  uint64_t first_tile, n_tiles;
  uint64_t tile_bytes;   // a wave's share of the text (run_tile_bytes)
  uint64_t block_tiles;  // (set by launch_run_resolve: tiles per block of the two-level resolve)
  RunPlan plan;
  RunSummary* summaries;
  RunTileIn* tile_in;
  uint64_t* out;
  uint64_t out_cap;
  unsigned long long* counters;
  unsigned long long* host_counters;
};
uint64_t run_tile_bytes(uint64_t span);   // bytes per tile for a run over `span` bytes (kRunTile or a multiple)
uint64_t run_tiles(uint64_t sb, uint64_t n, uint64_t tile_bytes, uint64_t* first_tile);
uint64_t run_resolve_slots(uint64_t n_tiles);   // elements of `summaries` and of `tile_in`: the tiles + the blocks of the two-level resolve
void launch_run_summary(const RunParams& a, hipEvent_t t0, hipEvent_t t1, hipStream_t st);
void launch_run_resolve(const RunParams& a, hipStream_t st);
void launch_run_emit(const RunParams& a, hipEvent_t t1, hipStream_t st);
// the PAIR shape (`"[^"]*"`: run_scan.h), the same three steps over the same buffers
void launch_pair_summary(const RunParams& a, hipEvent_t t0, hipEvent_t t1, hipStream_t st);
void launch_pair_resolve(const RunParams& a, hipStream_t st);
void launch_pair_emit(const RunParams& a, hipEvent_t t1, hipStream_t st);

struct ScanGeometry {
  int grid;
  uint32_t n_regions;
  uint64_t span_chunks;
};
ScanGeometry scan_geometry(uint64_t chunks, uint64_t chunks_per_block = 128);

// t0 / t1: events stamped with the kernel's own start / end (hipExtLaunchKernelGGL), no extra
// stream commands
void launch_scan_windows(const ScanParams& a, const WindowSet& ws, int n_windows, int grid, hipEvent_t t0, hipEvent_t t1,
                         hipStream_t st);
// dense mode with a lane-sized automaton whose tables fit LDS: candidates are found, walked and
// compacted in one kernel; the regions then hold verified (begin | end) like after
// verify_in_regions.  hit_counts = survivors per region (clamped to region_cap; overflow flagged).
bool dense_walk_fits(const DevProgram& P);
void launch_scan_dense_walk(const ScanParams& a, const DevProgram& P, int grid, uint64_t* region_ends,
                            unsigned long long* counters, hipEvent_t t0, hipEvent_t t1, hipStream_t st);
void launch_scan_dense(const ScanParams& a, const DevProgram& P, int grid, hipEvent_t t0, hipEvent_t t1, hipStream_t st);
// assertion-only patterns (n_pos == 0) in one pass, the pairs written once at their final place through a decoupled
// prefix scan over 32-KiB tiles (emit_scan.hip, tile_lookback.h).  scratch: emit_scratch_bytes(sb, se) bytes (zeroed by the launcher).
// counters[kCntFinal] (and host_counters, pinned, when given) = the number of matches, also when it exceeds out_cap;
// [kCntOverrun] = 1 when the prefix scan timed out (the caller takes the dense kernel instead).
uint64_t emit_tiles(uint64_t sb, uint64_t se);
size_t emit_scratch_bytes(uint64_t sb, uint64_t se);
void launch_emit_assertions(const uint8_t* text, uint64_t n, uint64_t sb, uint64_t se, uint32_t nullable, unsigned long long* scratch,
                            uint64_t* out, uint64_t out_cap, unsigned long long* counters, unsigned long long* host_counters, hipEvent_t t0,
                            hipEvent_t t1, hipStream_t st);
void launch_region_offsets(const uint32_t* counts, uint32_t n_regions, uint32_t cap, uint64_t* offsets,
                           unsigned long long* counters, hipStream_t st);
// what verify_in_regions + offsets_gather_check need for one pattern; rj_multi runs the tails of all
// its patterns in two launches (grid.y = pattern) from a device array of these
struct MultiTail {
  VerifyParams verify;
  DevProgram program;
  const uint32_t* hit_counts;
  uint32_t* valid_counts;
  uint64_t* region_ends;
  uint64_t* out;
  uint64_t out_cap;
  unsigned long long* host_counters;
};
void launch_tails_multi(const MultiTail* d_tails, int n_patterns, uint32_t n_regions, hipStream_t st);
// the same tails fed from ONE shared candidate list (plane scan): classify_shared_multi verifies every candidate
// against every pattern (exact window test first, automaton for the patterns whose window matches) and compacts
// the survivors into the patterns' own regions; then offsets_gather_check_multi as above.  A shared region that
// overflowed is reported in pattern 0's counters[kCntSharedMax].
// Every pattern must be short (DevProgram::short_max != 0, n_words <= 2).  What the kernel needs of the patterns
// travels as ONE blob that every workgroup copies into LDS: ClassifyDesc[n_patterns] (padded to 16 bytes), then
// the patterns' automaton tables one after the other (each padded to 4 words), <= kClassifyMaxTableWords in all.
constexpr uint32_t kClassifyMaxTableWords = 12288;
constexpr int kClassifyMaxWindows = 4;
struct ClassifyDesc {
  uint32_t n_windows;   // <= kClassifyMaxWindows (classify_shared_multi, the regexdna shape, looks at the first two)
  uint32_t v0[kClassifyMaxWindows], m0[kClassifyMaxWindows], v1[kClassifyMaxWindows], m1[kClassifyMaxWindows];
  uint32_t tab;         // the pattern's tables, word offset behind the descriptors
  uint32_t n_words, n_pos, n_rows, short_max, nullable;
  uint32_t rowbits[2];  // non-linear positions whose follow set is not empty
  uint32_t region_cap;
  uint32_t win_offset, win_len;  // classify_shared_general: the pattern's own window offset and length
  // (28 32-bit words: the pointers below start on an 8-byte boundary without implicit padding)
  uint64_t* begins;     // the pattern's own regions
  uint64_t* ends;
  uint32_t* valid_counts;
  unsigned long long* counters;
};
struct SharedHits {
  const uint64_t* hits;
  const uint32_t* counts;
  uint32_t cap;
  uint32_t n_regions;
  uint32_t n_patterns;
  uint32_t win_offset;
  const uint8_t* text;
  uint64_t n, sb, se;
  const uint32_t* blob;   // descriptors + tables (device memory)
  uint32_t desc_words;    // words the descriptors take (a multiple of 4)
  uint32_t blob_words;    // descriptors + tables (a multiple of 4)
  uint32_t reverse;       // classify_shared_multi: take the regions from the last one down
};
// max_words / max_short: the largest n_words / short_max among the patterns (selects the kernel instantiation)
void launch_tails_shared(const MultiTail* d_tails, const SharedHits& sh, int max_words, uint32_t max_short, unsigned long long* counters0,
                         hipStream_t st);
// The general one-pass scan (round 4): any set of patterns with fixed-offset windows of >= 4 bytes and short bounded
// automata.  The windows' first n_cmp bytes are compared, as 2-bit codes (byte >> code_shift) & 3 -- codes may alias:
// a superset test like every fast-forward filter --, with <= kPlaneMaxBases base windows, exactly (tolerance 0) or up to
// one differing code (tolerance 1: windows with a class byte).  Candidates are WINDOW POSITIONS (the patterns' offsets
// differ); classify_shared_general applies every pattern's own offset, exact window test (<= 4 windows) and automaton.
constexpr int kPlaneMaxBases = 12;  // (the kernel loops over the bases without unrolling: one base's masks in scalar registers at a time)
struct PlaneGParams {
  const uint8_t* text;
  uint64_t n;
  uint64_t wlo, whi;     // window positions to look at
  uint64_t span_pairs;
  uint32_t code_shift, n_bases, n_cmp, tolerance;
  uint32_t lo[kPlaneMaxBases][8], hi[kPlaneMaxBases][8];
  uint64_t* hits;
  uint32_t region_cap;
  uint32_t* hit_counts;
  uint32_t n_zero;
  unsigned long long* zero_counters[kMaxFused];
};
void launch_plane_scan_general(const PlaneGParams& a, int grid, hipEvent_t t0, hipEvent_t t1, hipStream_t st);
// MatchAllCount in one kernel for the sets the GENERAL plan takes (round 6; plane_count.hip: GeneralShape): the filter of
// plane_scan_general in plane_count's 32-bytes-per-lane layout, candidates in the wave's LDS ring, every candidate
// classified by the patterns' exact window tests + automata out of LDS (c.table = the blob classify_shared_general
// stages: ClassifyDesc[n_patterns] padded to 16 bytes, then the tables; c.table_words of it), the left-most-longest
// selection applied per pattern along its matches, nothing written but counts and first / last matches.
struct PlaneCountGParams {
  PlaneCountParams c;    // (c.mask_bits, c.base_lo / _hi unused)
  uint32_t n_cmp, tolerance;
  uint32_t lmax;         // the longest match of any pattern (<= 16)
  uint32_t desc_words;   // words the descriptors take inside the blob
  uint32_t idx[kPlaneMaxBases][8];   // base b, compared byte i: its symbol code 0..3, or 4 for a byte beyond n_cmp (always fits)
};
// plane_count<GeneralListShape>: the general test, the candidates (WINDOW positions: every pattern has its own offset, the
// classification subtracts it) written to the shared regions -- plane_scan_general in plane_count's layout (round 6).
struct PlaneListGParams {
  PlaneCountGParams g;   // (g.c.table, g.lmax, g.desc_words unused)
  uint64_t* hits;
  uint32_t region_cap;
  uint32_t offset;       // 0: the slots hold window positions
  uint32_t* hit_counts;
  uint32_t n_zero;
  unsigned long long* zero_counters[kMaxFused];
};
void launch_plane_list_general(const PlaneListGParams& a, int grid, hipEvent_t t0, hipEvent_t t1, hipStream_t st);
constexpr uint32_t kCountMaxBlobWords = 6144;   // 24 KiB of descriptors + tables per workgroup
void launch_plane_count_general(const PlaneCountGParams& g, int max_words, uint32_t max_short, int grid, hipEvent_t t0, hipEvent_t t1, hipStream_t st);
void launch_tails_shared_general(const MultiTail* d_tails, const SharedHits& sh, int max_words, uint32_t max_short, unsigned long long* counters0,
                                 hipStream_t st);
void launch_offsets_gather_check_multi(const MultiTail* d_tails, int n_patterns, uint32_t n_regions, hipStream_t st);
// plane_scan + classify_shared_multi as ONE kernel (plane_scan.hip, round 4): a wave keeps its span's candidates in LDS and
// classifies them at the end of the span; sh.hits / sh.counts / sh.cap are unused.  The kernel does not clear the
// counters it may set (kCntOverflow, kCntMaxRegion of every pattern, kCntSharedMax of pattern 0): the caller keeps them
// zero.  kCntSharedMax != 0 afterwards: a span held more candidates than the LDS slots, the run is void.  Returns
// false (nothing launched) when the pattern set's shape has no instantiation: then launch_plane_scan + launch_tails_shared.
constexpr uint32_t kFusedMaxBlobWords = 4096;  // 16 KiB of descriptors + tables per workgroup
bool launch_plane_scan_classify(const PlaneParams& a, const SharedHits& sh, int max_words, uint32_t max_short, unsigned long long* counters0,
                                int grid, hipEvent_t t0, hipEvent_t t1, hipStream_t st);

// windows mode, lane-sized automaton: verify + compact inside every region (16 lanes each), then
// offsets_gather_check lays the survivors out
void launch_verify_in_regions(const VerifyParams& a, const DevProgram& P, const uint32_t* hit_counts, uint32_t* valid_counts,
                              uint64_t* region_ends, hipStream_t st);
// windows behind an unbounded prefix (behind_walk.h): one candidate per hit -- forward check from the cut,
// reverse automaton R to the left-most start, forward longest; survivors compacted in place like
// verify_in_regions (begins are NOT ordered: the caller sorts before selecting)
void launch_verify_behind_in_regions(const VerifyParams& a, const DevProgram& P, const DevProgram& R, const uint32_t* hit_counts,
                                     uint32_t* valid_counts, uint64_t* region_ends, hipStream_t st);
// floating windows: the candidate starts of every hit (ranges clipped so that each start is verified
// once, in order); survivors go to region_begins / region_ends
void launch_verify_floating_in_regions(const VerifyParams& a, const DevProgram& P, const uint32_t* hit_counts,
                                       uint32_t* valid_counts, uint64_t* region_begins, uint64_t* region_ends, hipStream_t st);
// The same two tails over PADDED tables in LDS (verify_lds.hip, lds_walk.h): automata of <= 128 positions.  A step of
// a walk is two LDS reads instead of a chain of 10-20 flat loads, the text around a hit is staged in LDS in one
// trip; the floating kernel can also make the left-most-longest selection among the candidates of its own region
// (`select`), which usually leaves the candidates "already the result" for offsets_gather_check -- when that check
// fails the caller repeats the launch with select = false (the general selection needs every candidate).
struct WalkDesc {
  const uint64_t* blob;      // forward tables, lds_walk.h layout
  const uint64_t* rev_blob;  // reverse tables (behind mode), same shape; else null
  uint32_t words;            // uint64 words of one blob, rounded up to an even number
  int32_t nq;                // 64-bit words per row: 1 or 2
  int32_t n_ctx, n_pos;
  uint32_t nullable, max_walk;
};
// false: the tables + text windows do not fit the workgroup's LDS (the caller takes the kernels above)
bool launch_verify_floating_lds(const VerifyParams& a, const DevProgram& P, const WalkDesc& d, const uint32_t* hit_counts,
                                uint32_t* valid_counts, uint64_t* region_begins, uint64_t* region_ends, bool select, hipStream_t st);
bool launch_verify_behind_lds(const VerifyParams& a, const DevProgram& P, const WalkDesc& d, const uint32_t* hit_counts,
                              uint32_t* valid_counts, uint64_t* region_ends, hipStream_t st);
// region offsets + gather + check_and_interleave in one launch (see the kernel); host_counters
// (pinned, may be null) receives the counter block directly.  With offsets_scratch / prev_scratch
// ([n_regions] each) the first launch only lays the regions out and a second one copies and checks
// with a wave per region -- for runs whose regions hold many candidates
void launch_offsets_gather_check(const uint32_t* counts, const uint64_t* region_begins, const uint64_t* region_ends,
                                 uint32_t n_regions, uint32_t region_cap, uint64_t carry_cur, uint64_t* out, uint64_t out_cap,
                                 unsigned long long* counters, unsigned long long* host_counters, uint64_t* offsets_scratch,
                                 uint64_t* prev_scratch, hipStream_t st, uint64_t carry_prev_end = 0, int have_prev = 0);
// (begin,end) pairs -> begin[] / end[]; n read from device memory
void launch_split_pairs(const uint64_t* pairs, const unsigned long long* n_ptr, uint64_t n_upper, uint64_t* keys, uint64_t* vals,
                        hipStream_t st);
void launch_verify(const VerifyParams& a, const DevProgram& P, uint64_t expected_hits, hipStream_t st);
// large path: drop the kNoMatch slots, keeping the order
void launch_mark_valid(const uint64_t* cand_end, uint64_t n, uint64_t* flags, hipStream_t st);
void launch_compact_valid(const uint64_t* cand_begin, const uint64_t* cand_end, const uint64_t* flags,
                          const uint64_t* pos, uint64_t n, uint64_t* keys, uint64_t* vals,
                          unsigned long long* counters, hipStream_t st);
void launch_match_full(const uint8_t* text, uint64_t n, const DevProgram& P, int* result, hipStream_t st);
void launch_finalize_small(const FinalizeParams& a, hipStream_t st);
// writes the sorted candidates as (begin,end) pairs and sets *unordered unless they already are a
// valid result (pairwise disjoint, no empty match, first begin >= carry_cur)
void launch_check_and_interleave(const uint64_t* keys, const uint64_t* vals, const unsigned long long* n_ptr,
                                 uint64_t n_upper, uint64_t carry_cur, uint64_t* out, uint64_t cap, unsigned long long* unordered,
                                 hipStream_t st);
// counters[kCntAdjacent] = 1 when some non-empty candidate ends exactly where another begins
void launch_detect_adjacent(const uint64_t* keys, const uint64_t* vals, uint64_t n_upper, unsigned long long* counters,
                            hipStream_t st);
// the reference's no-fast-forward algorithm on one lane (exact incl. its ring-slot artefact)
void launch_exact_sequential(const uint8_t* text, uint64_t n, const DevGraph& G, int64_t* ring, uint64_t* out,
                             uint64_t out_cap, unsigned long long* counters, hipStream_t st);
// Replace: match lengths (input of the prefix sum) and the gather itself; the result length is
// left in counters[kCntFinal]; long_gaps needs 3 * (m + 1) uint64
void launch_match_lengths(const uint64_t* spans, uint64_t m, uint64_t* len, hipStream_t st);
void launch_replace_gather(const uint8_t* text, uint64_t n, const uint64_t* spans, const uint64_t* removed, uint64_t m,
                           const uint8_t* with, uint64_t with_len, uint8_t* out, uint64_t out_cap, uint64_t* long_gaps,
                           unsigned long long* counters, hipStream_t st);
// taken[i] = 1 for the candidates the greedy left-most-longest rule takes (keys / vals sorted by begin,
// pmax = exclusive prefix max of vals); nxt, G: n uint64 of scratch each, blocks_scratch:
// chain_select_scratch_bytes(n)
size_t chain_select_scratch_bytes(uint64_t n);
void launch_chain_select(const uint64_t* keys, const uint64_t* vals, const uint64_t* pmax, uint64_t n, uint64_t carry_cur,
                         uint8_t* taken, uint64_t* nxt, uint64_t* G, uint64_t* blocks_scratch, hipStream_t st);
void launch_taken_index(const uint8_t* taken, uint64_t n, uint64_t* idx, hipStream_t st);
// (conflict: when not null, *conflict = 1 if a candidate that was not taken ends after the match that hides it)
void launch_zero_length_rule(const uint64_t* keys, const uint64_t* vals, const uint8_t* taken,
                             const uint64_t* last_taken, uint64_t n, uint64_t carry_prev_end, int have_prev,
                             uint64_t* keep, unsigned long long* conflict, hipStream_t st);
void launch_compact_kept(const uint64_t* keys, const uint64_t* vals, const uint64_t* keep, const uint64_t* pos,
                         uint64_t n, uint64_t* out, uint64_t out_cap, unsigned long long* counters, hipStream_t st);


// ---- the linear-time carry scan (carry_kernels.hip, carry_scan.h).  R = the REVERSE automaton.
// A run covers the sub-chunks [a0 + i*sub, ..), i < m, up to the end of the text; the first m_own of
// them hold the starts [sb, se).  vals [m][P] uint64 (summaries, resolved in place), mats [m][P*W].
int cs_state_words(const DevProgram& R);                       // 1, 2, 4, 8, 16, 32; 0 = automaton too wide (> 1024 positions)
size_t cs_scratch_bytes(const DevProgram& R, uint64_t lanes);  // global slab for the lanes' private state
void launch_cs_summarize(const DevProgram& R, const uint8_t* text, uint64_t n, uint64_t a0, uint64_t sub, uint64_t m, uint64_t* vals,
                         uint32_t* mats, uint8_t* scratch, hipStream_t st);
size_t cs_resolve_scratch_bytes(const DevProgram& R, uint64_t m);
void launch_cs_resolve(const DevProgram& R, uint64_t m, uint64_t* vals, const uint32_t* mats, uint8_t* scratch, hipStream_t st);
void launch_cs_emit(const DevProgram& R, const uint8_t* text, uint64_t n, uint64_t a0, uint64_t sub, uint64_t m, uint64_t m_own,
                    uint64_t sb, uint64_t se, const uint64_t* vals, uint64_t* E, uint8_t* scratch, unsigned long long* longest,
                    hipStream_t st);   // *longest = max E(s) - s (atomicMax; the caller zeroes it)
// local chains, the hop over the sub-chunks from `cur`, and the taken matches compacted to the front
// of every sub-chunk's slab (begins in G, ends in E, counts[i] of them; *total = their sum)
void launch_cs_chain(uint64_t* E, uint64_t* G, uint64_t a0, uint64_t sub, uint64_t m_own, uint64_t sb, uint64_t se, uint64_t cur,
                     uint64_t* entry, uint32_t* counts, unsigned long long* total, hipStream_t st);

}  // namespace rejit_amd
#endif
