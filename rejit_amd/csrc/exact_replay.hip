// rejit_amd/csrc/exact_replay.hip -- the reference's answer, ring artefact included, on texts of any
// size (exact_replay.h has the argument): synchronisation points from the all-starts position automaton,
// then the reference's own loop per segment, one lane per segment.
//
//   xr_lookup      the two ends of the ownership range: a proven synchronisation point >= sb, >= se (lane per chunk)
//   xr_chunk_sync  per 1-KiB chunk the first proven synchronisation point (lane per chunk, early exit)
//   xr_segments    per chunk that has one: where its segment ends; the last point and the longest segment
//   xr_replay      lane per segment: rj_replay_segment, ring in LDS (lane-interleaved) when it fits, pairs
//                  into the segment's own stretch of a scratch array (a segment has at most one match
//                  per byte)
//   xr_offsets     exclusive scan of the per-chunk counts (one workgroup), running total across batches
//   xr_compact     scratch -> result pairs
//
// The text is taken in batches of 64 MiB so that the scratch is bounded (1 GiB); a batch ends at its last
// synchronisation point, where the next one begins.
//
// Round 4: LONG segments.  One lane replays a segment at 2-6 us per byte (tools/replay_probe.py: `.{0,2}.` over a text
// without line breaks, 4 MiB = 25 s), so until round 4 a stretch of more than 16 MiB without a proven synchronisation point
// was not replayed at all.  Now a segment of more than kLongSegment bytes is taken in parts (exact_replay.h, "speculate and
// verify"):
//   xr_round       lane per part of kPart bytes: round 0 replays it from kWarm bytes before with a free ring and notes the
//                  ring's ORDER PATTERN on entering the part and the ring on leaving it; round k replays the parts from
//                  candidate pattern k
//   xr_walk        one wave carries the TRUE ring over the parts: a part that was replayed from its order pattern hands on its
//                  exit ring (inherited ranks replaced by the true starts); a pattern no part was replayed from stops the
//                  walk and becomes the next round's candidate
//   xr_emit        lane per part: once more from its verified ring, its matches through the sink as if nothing came before
//   xr_join        one lane over the PARTS: a part pops from the lists before it while begin >= its smallest raw begin, and
//                  drops its first entry when the sink's filter says so; the parts' lists then go to xr_compact as they are
// and xr_offsets / xr_compact carry on as for the other segments.  What remains: a segment must fit a batch (which grows
// from 64 MiB to 1 GiB when it has to), the cuts must not meet more than kReplayMaxRounds different order patterns, and rings of more than kWalkSlots slots keep the one-lane replay (<= 16 MiB): otherwise run_exact
// reports "not done" and the caller keeps the result of the parallel pipeline (documented semantics).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "engine_internal.h"
#include "exact_replay.h"

namespace rejit_amd {
namespace {

constexpr uint64_t kChunk = 1024;
constexpr uint64_t kBatchChunks = 65536;        // 64 MiB of text per batch ...
constexpr uint64_t kBigBatchChunks = 1u << 20;  // ... and up to 1 GiB (16 GiB of scratch) when a batch holds no second synchronisation point
constexpr uint64_t kMaxSegment = 16ull << 20;   // longest stretch ONE lane is asked to replay (rings too large for the walk)
constexpr uint64_t kLongSegment = 64ull << 10;  // a longer segment is taken in parts
constexpr uint64_t kPart = 2048, kWarm = 512;   // bytes per part; bytes before a part its replay starts at
constexpr int kWalkSlots = 448;                 // times x states up to this: xr_walk holds the true ring, its pattern and the candidates in LDS (16 x 448 x 8 B)
constexpr int kReplayLanes = 64;                // lanes per workgroup of xr_replay
constexpr int kLdsRingSlots = 96;               // times x states up to this: ring in LDS (64 lanes x 96 x 8 B = 48 KiB)

// state words shared by the kernels of one run
enum { kXrY0 = 0, kXrY1, kXrLastSync, kXrMaxGap, kXrTotal, kXrRounds, kXrWalkNext, kXrWalkStuck, kXrStateSize };

// An end of the ownership range: the first PROVEN synchronisation point at or after x among the chunks [first_chunk,
// first_chunk + n_chunks) of the text's own 1-KiB grid -- a lane per chunk walks it from its beginning with every position
// alive (chunk 0: from nothing alive, exactly) and reports the first position >= x at which nothing is.  Any proven point
// will do as a cut as long as the two ranges that share it compute it the same way: this is a function of the text and x
// alone.  (Until round 4 ONE lane walked from x -- back and forth until its superset had died out -- to the next point: on a
// text without synchronisation points that was the whole text at half a microsecond per byte, twice per call.)
template <int NQ>
__global__ void xr_lookup(DevProgram P, const uint8_t* t, uint64_t n, uint64_t x, uint64_t first_chunk, uint64_t n_chunks, unsigned long long* y) {
  const uint64_t k = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k >= n_chunks) return;
  const uint64_t c0 = (first_chunk + k) * kChunk;
  if (c0 > n) return;
  const uint64_t p = rj_chunk_first_sync<NQ>(P, t, n, c0, std::min(c0 + kChunk, n + 1), c0 == 0, x);
  if (p != kNoSync) atomicMin(y, static_cast<unsigned long long>(p));
}

template <int NQ>
__global__ void xr_chunk_sync(DevProgram P, const uint8_t* t, uint64_t n, uint64_t ys, uint64_t y1, uint64_t n_chunks,
                              uint64_t* sync) {
  const uint64_t c = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (c >= n_chunks) return;
  const uint64_t c0 = ys + c * kChunk;
  sync[c] = rj_chunk_first_sync<NQ>(P, t, n, c0, std::min(c0 + kChunk, y1), c == 0);
}

// seg_end[c]: end of the segment that begins at sync[c] -- the next synchronisation point of the batch, y1
// in the last batch (`final`), kNoSync for the batch's last point otherwise (the next batch starts there)
__global__ void xr_segments(const uint64_t* sync, uint64_t n_chunks, uint64_t y1, int final, uint64_t* seg_end,
                            unsigned long long* state) {
  const uint64_t c = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (c >= n_chunks) return;
  const uint64_t a = sync[c];
  uint64_t b = kNoSync;
  if (a != kNoSync) {
    for (uint64_t k = c + 1; k < n_chunks && b == kNoSync; k++) b = sync[k];
    if (b == kNoSync) {
      atomicMax(&state[kXrLastSync], static_cast<unsigned long long>(a));
      if (final) b = y1;
    }
    if (b != kNoSync) atomicMax(&state[kXrMaxGap], static_cast<unsigned long long>(b - a));
  }
  seg_end[c] = b;
}

__global__ void xr_begin_batch(unsigned long long* state, uint64_t ys) {
  state[kXrLastSync] = ys;
  state[kXrMaxGap] = 0;
}

struct LdsRing {
  int64_t* base;  // + lane; slot stride = kReplayLanes
  __device__ int64_t& operator()(int slot) const { return base[slot * kReplayLanes]; }
};
struct GlobalRing {
  int64_t* base;  // + lane; slot stride = total lanes
  uint64_t stride;
  __device__ int64_t& operator()(int slot) const { return base[static_cast<uint64_t>(slot) * stride]; }
};

template <bool LDS>
__global__ void __launch_bounds__(kReplayLanes)
xr_replay(DevGraph G, const uint8_t* t, uint64_t n, uint64_t ys, const uint64_t* sync, const uint64_t* seg_end, uint64_t n_chunks,
          int64_t* ring_mem, uint64_t* scratch, uint32_t* counts, uint64_t max_len) {
  extern __shared__ int64_t lds_ring[];
  const uint64_t lanes = static_cast<uint64_t>(gridDim.x) * kReplayLanes;
  const uint64_t lane = static_cast<uint64_t>(blockIdx.x) * kReplayLanes + threadIdx.x;
  for (uint64_t c = lane; c < n_chunks; c += lanes) {
    const uint64_t a = sync[c], b = seg_end[c];
    uint32_t m = 0;
    if (a != kNoSync && b != kNoSync && b - a > max_len) continue;  // (a long segment: xr_join sets the counts of its parts)
    if (a != kNoSync && b != kNoSync) {
      uint64_t* out = scratch + 2 * (a - ys);
      if (LDS) m = static_cast<uint32_t>(rj_replay_segment(G, t, n, a, b, LdsRing{lds_ring + threadIdx.x}, out));
      else m = static_cast<uint32_t>(rj_replay_segment(G, t, n, a, b, GlobalRing{ring_mem + lane, lanes}, out));
    }
    counts[c] = m;
  }
}

// ---- long segments (exact_replay.h: speculate and verify)
struct CandFrom {
  uint64_t v[kReplayMaxRounds + 1];  // the first part that has been replayed from candidate k
};

__device__ __forceinline__ uint64_t part_warm_start(uint64_t a, uint64_t i) {
  const uint64_t c0 = a + i * kPart;
  return i == 0 ? a : std::max(a, c0 - std::min(c0, kWarm));
}

// exits: [n_parts][slots] of this round (ring values at the part's end); entry0 (round 0 only): [n_parts][slots] order
// patterns; cand: the round's candidate pattern (null: round 0); pat_scratch: 2 x slots words per lane
template <bool LDS>
__global__ void __launch_bounds__(kReplayLanes)
xr_round(DevGraph G, const uint8_t* t, uint64_t n, uint64_t a, uint64_t b, uint64_t n_parts, uint64_t from, const int64_t* cand,
         int64_t* ring_mem, int64_t* entry0, int64_t* exits, int64_t* pat_scratch) {
  extern __shared__ int64_t lds_ring[];
  const uint64_t lanes = static_cast<uint64_t>(gridDim.x) * kReplayLanes;
  const uint64_t lane = static_cast<uint64_t>(blockIdx.x) * kReplayLanes + threadIdx.x;
  const uint64_t slots = static_cast<uint64_t>(G.n_states) * G.times;
  for (uint64_t i = from + lane; i < n_parts; i += lanes) {
    const uint64_t c0 = a + i * kPart, c1 = i + 1 == n_parts ? b : c0 + kPart;
    const uint64_t start = cand ? c0 : part_warm_start(a, i);
    int64_t* entry = cand ? nullptr : entry0 + i * slots;
    int64_t* scratch = pat_scratch + lane * 2 * slots;
    if (LDS) rj_replay_raw(G, t, n, start, cand, c0, c1, LdsRing{lds_ring + threadIdx.x}, entry, exits + i * slots, static_cast<uint64_t*>(nullptr),
                           static_cast<const int64_t*>(nullptr), scratch);
    else rj_replay_raw(G, t, n, start, cand, c0, c1, GlobalRing{ring_mem + lane, lanes}, entry, exits + i * slots, static_cast<uint64_t*>(nullptr),
                       static_cast<const int64_t*>(nullptr), scratch);
  }
}

// One wave.  cands: [kReplayMaxRounds + 1][slots] order patterns (row 0 unused: round 0's entry patterns are the parts' own);
// exits: [kReplayMaxRounds + 1][n_parts][slots]; walk_ring: the TRUE ring (start offsets, time order) at part
// state[kXrWalkNext] -- in: where to carry on; out: where the walk stands; walk_pat (out): the pattern that stopped the walk;
// true_starts: [n_parts][slots], row i = the real start offsets of the threads at part i's entry, oldest first (for xr_emit).
// state[kXrWalkStuck] = 1 when it stopped before the last part.
__global__ void __launch_bounds__(64) xr_walk(uint64_t slots, uint64_t a, uint64_t n_parts, int n_cand, CandFrom cand_from, const int64_t* cands,
                                              const int64_t* entry0, const int64_t* exits, int64_t* walk_ring, int64_t* walk_pat,
                                              int64_t* true_starts, int32_t* chosen, unsigned long long* state) {
  extern __shared__ int64_t lds[];  // the true ring | its pattern | its distinct starts, ascending | first-slot flags | the candidates 1 .. n_cand - 1
  int64_t* T = lds;
  int64_t* pat = lds + slots;
  int64_t* sorted = lds + 2 * slots;
  int64_t* first_slot = lds + 3 * slots;  // 1: the slot is the first one that holds its start
  int64_t* cand_lds = lds + 4 * slots;    // candidate c at (c - 1) * slots
  const uint32_t lane = threadIdx.x;
  for (uint64_t k = lane; k < slots; k += 64) T[k] = walk_ring[k];
  for (int c = 1; c < n_cand; c++)
    for (uint64_t k = lane; k < slots; k += 64) cand_lds[static_cast<uint64_t>(c - 1) * slots + k] = cands[static_cast<uint64_t>(c) * slots + k];
  __syncthreads();
  uint64_t i = state[kXrWalkNext];
  for (; i < n_parts; i++) {
    // the true ring's order pattern: a slot's rank = the distinct starts below its own (each counted at its first slot)
    for (uint64_t k = lane; k < slots; k += 64) {
      const int64_t v = T[k];
      bool first = v >= 0;
      for (uint64_t q = 0; q < k && first; q++) first = T[q] != v;
      first_slot[k] = first ? 1 : 0;
    }
    __syncthreads();
    for (uint64_t k = lane; k < slots; k += 64) {
      const int64_t v = T[k];
      int64_t rank = -1;
      if (v >= 0) {
        rank = 0;
        for (uint64_t j = 0; j < slots; j++) rank += (first_slot[j] != 0 && T[j] < v) ? 1 : 0;
        if (first_slot[k] != 0) sorted[rank] = v;
      }
      pat[k] = rank;
    }
    __syncthreads();
    int found = -1;
    {
      bool same = true;
      for (uint64_t k = lane; k < slots; k += 64) same = same && pat[k] == entry0[i * slots + k];
      if (__ballot(!same) == 0) found = 0;
    }
    for (int c = 1; c < n_cand && found < 0; c++) {
      if (i < cand_from.v[c]) continue;
      bool same = true;
      for (uint64_t k = lane; k < slots; k += 64) same = same && pat[k] == cand_lds[static_cast<uint64_t>(c - 1) * slots + k];
      if (__ballot(!same) == 0) found = c;
    }
    if (found < 0) break;
    // the number of distinct starts, the part's true starts for xr_emit, and the ring the part leaves
    int64_t cnt = 0;
    for (uint64_t k = lane; k < slots; k += 64) cnt = pat[k] + 1 > cnt ? pat[k] + 1 : cnt;
    for (int o = 32; o > 0; o >>= 1) {
      const int64_t other = __shfl_xor(cnt, o);
      cnt = other > cnt ? other : cnt;
    }
    for (uint64_t k = lane; k < static_cast<uint64_t>(cnt); k += 64) true_starts[i * slots + k] = sorted[k];
    if (lane == 0) chosen[i] = found;
    const int64_t c0 = static_cast<int64_t>(a + i * kPart);
    const int64_t* x = exits + (static_cast<uint64_t>(found) * n_parts + i) * slots;
    int64_t next[(kWalkSlots + 63) / 64];
    int u = 0;
    for (uint64_t k = lane; k < slots; k += 64, u++) {
      const int64_t v = x[k];
      next[u] = v < 0 ? -1 : v < c0 ? sorted[v - (c0 - cnt)] : v;
    }
    __syncthreads();
    u = 0;
    for (uint64_t k = lane; k < slots; k += 64, u++) T[k] = next[u];
    __syncthreads();
  }
  for (uint64_t k = lane; k < slots; k += 64) {
    walk_ring[k] = T[k];
    walk_pat[k] = pat[k];
  }
  if (lane == 0) {
    state[kXrWalkNext] = i;
    state[kXrWalkStuck] = i < n_parts ? 1 : 0;
  }
}

template <bool LDS>
__global__ void __launch_bounds__(kReplayLanes)
xr_emit(DevGraph G, const uint8_t* t, uint64_t n, uint64_t a, uint64_t b, uint64_t ys, uint64_t n_parts, const int32_t* chosen, const int64_t* cands,
        const int64_t* true_starts, int64_t* ring_mem, int64_t* pat_scratch, uint64_t* scratch, uint32_t* raw_n, uint64_t* min_begin) {
  extern __shared__ int64_t lds_ring[];
  const uint64_t lanes = static_cast<uint64_t>(gridDim.x) * kReplayLanes;
  const uint64_t lane = static_cast<uint64_t>(blockIdx.x) * kReplayLanes + threadIdx.x;
  const uint64_t slots = static_cast<uint64_t>(G.n_states) * G.times;
  for (uint64_t i = lane; i < n_parts; i += lanes) {
    const uint64_t c0 = a + i * kPart, c1 = i + 1 == n_parts ? b : c0 + kPart;
    const int k = chosen[i];
    const int64_t* init = k == 0 ? nullptr : cands + static_cast<uint64_t>(k) * slots;
    const uint64_t start = k == 0 ? part_warm_start(a, i) : c0;
    uint64_t* out = scratch + 2 * (c0 - ys);
    const int64_t* ts = true_starts + i * slots;
    int64_t* ps = pat_scratch + lane * 2 * slots;
    uint64_t m;
    if (LDS) m = rj_replay_raw(G, t, n, start, init, c0, c1, LdsRing{lds_ring + threadIdx.x}, static_cast<int64_t*>(nullptr), static_cast<int64_t*>(nullptr), out, ts, ps);
    else m = rj_replay_raw(G, t, n, start, init, c0, c1, GlobalRing{ring_mem + lane, lanes}, static_cast<int64_t*>(nullptr), static_cast<int64_t*>(nullptr), out, ts, ps);
    uint64_t lowest = ~0ull, kept = 0;
    for (uint64_t j = 0; j < m; j++) {
      const uint64_t pb = out[2 * j], pe = out[2 * j + 1];
      lowest = pb < lowest ? pb : lowest;
      kept = rj_sink_append(out, kept, static_cast<int64_t>(pb), static_cast<int64_t>(pe));
    }
    min_begin[i] = lowest;
    raw_n[i] = static_cast<uint32_t>(kept);
  }
}

// One lane: the parts' own lists (xr_emit) joined in order: a part pops from the list before it while begin >= the smallest
// begin among its raw matches; and the filter of the sink for the one entry whose decision the part could not take on its
// own -- its list's first, when that is an empty match right at the end of the entry before it (which belongs to an earlier
// part): dropped (first[i] = 1).  Every later decision of the part came out the same with that entry standing in for the
// one before it: both end where it begins.  What is left of part i is handed to xr_offsets / xr_compact as if the part were
// a segment of its own: sync[] / counts[] of the chunk its first byte lies in (parts are two chunks long; a long segment has
// no synchronisation point of its own inside).  keep / first / prev: n_parts words each.
__global__ void xr_join(uint64_t a, uint64_t ys, uint64_t n_parts, const uint32_t* raw_n, const uint64_t* min_begin, const uint64_t* scratch,
                        uint32_t* keep, uint32_t* first, int32_t* prev, uint64_t* sync, uint32_t* counts) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  int64_t top_part = -1;
  for (uint64_t i = 0; i < n_parts; i++) {
    const uint32_t l = raw_n[i];
    const uint64_t mb = min_begin[i];
    while (top_part >= 0 && mb != ~0ull) {
      const uint64_t* list = scratch + 2 * (a + static_cast<uint64_t>(top_part) * kPart - ys);
      uint32_t k = keep[top_part];
      const uint32_t f = first[top_part];
      while (k > f && list[2 * (k - 1)] >= mb) k--;
      keep[top_part] = k;
      if (k != f) break;
      top_part = prev[top_part];
    }
    keep[i] = l;
    uint32_t f = 0;
    if (l != 0 && top_part >= 0) {
      const uint64_t* mine = scratch + 2 * (a + i * kPart - ys);
      const uint64_t* list = scratch + 2 * (a + static_cast<uint64_t>(top_part) * kPart - ys);
      if (mine[0] == mine[1] && mine[0] == mb && list[2 * (keep[top_part] - 1) + 1] == mine[0]) f = 1;
    }
    first[i] = f;
    if (l != f) {
      prev[i] = static_cast<int32_t>(top_part);
      top_part = static_cast<int64_t>(i);
    }
  }
  for (uint64_t i = 0; i < n_parts; i++) {
    const uint64_t c0 = a + i * kPart;
    const uint64_t c = (c0 - ys) / kChunk;
    // (xr_compact copies counts[c] pairs from scratch + 2 (sync[c] - ys): one pair further when the first entry is dropped)
    if (i != 0 || first[i] != 0) sync[c] = c0 + first[i];
    counts[c] = keep[i] - first[i];
  }
}

// one workgroup: offs[c] = total so far + exclusive prefix of counts; total += sum
__global__ void __launch_bounds__(1024) xr_offsets(const uint32_t* counts, uint64_t n_chunks, uint64_t* offs, unsigned long long* state) {
  __shared__ uint64_t part[1024];
  const uint64_t per = (n_chunks + 1023) / 1024;
  const uint64_t lo = std::min(n_chunks, threadIdx.x * per), hi = std::min(n_chunks, lo + per);
  uint64_t sum = 0;
  for (uint64_t c = lo; c < hi; c++) sum += counts[c];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const uint64_t v = threadIdx.x >= static_cast<unsigned>(d) ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  const uint64_t base = state[kXrTotal];
  uint64_t run = base + part[threadIdx.x] - sum;
  for (uint64_t c = lo; c < hi; c++) {
    offs[c] = run;
    run += counts[c];
  }
  __syncthreads();
  if (threadIdx.x == 1023) state[kXrTotal] = base + part[1023];
}

// a wave per chunk: the segment's pairs to their place in the result
__global__ void xr_compact(const uint64_t* sync, const uint32_t* counts, const uint64_t* offs, uint64_t n_chunks, uint64_t ys,
                           const uint64_t* scratch, uint64_t* out, uint64_t cap) {
  const uint64_t c = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const uint32_t l = threadIdx.x & 63;
  if (c >= n_chunks) return;
  const uint32_t m = counts[c];
  if (m == 0) return;
  const uint64_t* from = scratch + 2 * (sync[c] - ys);
  const uint64_t to = offs[c];
  for (uint32_t j = l; j < 2 * m; j += 64)
    if (to + (j >> 1) < cap) out[2 * to + j] = from[j];
}

template <int NQ>
int run_exact_nq(rj_scan* s, const uint8_t* d_text, uint64_t n, uint64_t sb, uint64_t se, hipStream_t st) {
  const rj_program* rp = s->prog;
  const DevProgram& P = rp->dev;
  const DevGraph& G = rp->graph;
  RJ_HIP(s->xr_state.reserve(kXrStateSize * sizeof(unsigned long long)));
  unsigned long long* state = s->xr_state.as<unsigned long long>();
  unsigned long long h[kXrStateSize];
  uint64_t y0 = 0, y1 = n + 1;
  if (!(sb == 0 && se == n + 1)) {
    // the two ends, each in windows of 4096 chunks until a point shows up (on ordinary text: in the first)
    auto first_point = [&](uint64_t x, uint64_t* out) -> int {
      if (x == 0 || x > n) {
        *out = x == 0 ? 0 : n + 1;
        return RJ_OK;
      }
      constexpr uint64_t kWindow = 4096;
      for (uint64_t c = x / kChunk; c * kChunk <= n; c += kWindow) {
        h[0] = n + 1;
        RJ_HIP(hipMemcpyAsync(state + kXrY0, h, sizeof(unsigned long long), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(xr_lookup<NQ>, dim3(static_cast<unsigned>(kWindow / 256)), dim3(256), 0, st, P, d_text, n, x, c, kWindow, state + kXrY0);
        RJ_HIP(hipMemcpyAsync(h, state + kXrY0, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        RJ_HIP(hipStreamSynchronize(st));
        if (h[0] != n + 1) break;
      }
      *out = h[0];
      return RJ_OK;
    };
    int rc = first_point(sb, &y0);
    if (rc != RJ_OK) return rc;
    rc = first_point(se, &y1);
    if (rc != RJ_OK) return rc;
  }
  const int slots = G.n_states * G.times;
  const bool lds = slots <= kLdsRingSlots;
  // (the pairs go to a buffer of their own: when a later batch turns out not to be replayable the caller
  // keeps the result it has)
  uint64_t batch_chunks = kBatchChunks;
  for (int attempt = 0; attempt < 8; attempt++) {
    if (y0 >= y1) {  // nothing begins in this range
      s->result_count = 0;
      s->result = s->out.as<uint64_t>();
      return 1;
    }
    RJ_HIP(hipMemsetAsync(state + kXrTotal, 0, sizeof(unsigned long long), st));
    s->xr_parts = 0;
    s->xr_rounds = 0;
    if (s->xr_out_cap < (y1 - y0) / 8 + 1024) {  // (a first guess; the count decides below)
      s->xr_out_cap = (y1 - y0) / 8 + 1024;
      RJ_HIP(s->xr_out.reserve(s->xr_out_cap * 2 * sizeof(uint64_t)));
    }
    const uint64_t all_chunks = (y1 - y0 + kChunk - 1) / kChunk;
    const uint64_t cap_chunks = std::min(all_chunks, batch_chunks);
    RJ_HIP(s->xr_sync.reserve(cap_chunks * sizeof(uint64_t)));
    RJ_HIP(s->xr_seg_end.reserve(cap_chunks * sizeof(uint64_t)));
    RJ_HIP(s->xr_offs.reserve(cap_chunks * sizeof(uint64_t)));
    RJ_HIP(s->xr_counts.reserve(cap_chunks * sizeof(uint32_t)));
    if (s->xr_scratch.reserve((cap_chunks * kChunk + 1) * 2 * sizeof(uint64_t)) != hipSuccess) {
      (void)hipGetLastError();
      if (batch_chunks > kBatchChunks) return 0;  // (a grown batch: 16 bytes of scratch per text byte did not fit -- not replayed)
      return rj_fail(RJ_DEVICE_ERROR, "exact replay: out of device memory");
    }
    const int replay_blocks = static_cast<int>(std::min<uint64_t>((cap_chunks + kReplayLanes - 1) / kReplayLanes, 2048));
    if (!lds) RJ_HIP(s->ring.reserve(static_cast<size_t>(replay_blocks) * kReplayLanes * slots * sizeof(int64_t)));
    bool grow_batch = false;
    for (uint64_t ys = y0; ys < y1;) {
      const uint64_t nb = std::min((y1 - ys + kChunk - 1) / kChunk, batch_chunks);
      const int final = ys + nb * kChunk >= y1 ? 1 : 0;
      hipLaunchKernelGGL(xr_begin_batch, dim3(1), dim3(1), 0, st, state, ys);
      const int grid = static_cast<int>((nb + 255) / 256);
      hipLaunchKernelGGL(xr_chunk_sync<NQ>, dim3(grid), dim3(256), 0, st, P, d_text, n, ys, y1, nb, s->xr_sync.as<uint64_t>());
      hipLaunchKernelGGL(xr_segments, dim3(grid), dim3(256), 0, st, s->xr_sync.as<uint64_t>(), nb, y1, final,
                         s->xr_seg_end.as<uint64_t>(), state);
      RJ_HIP(hipMemcpyAsync(h, state, sizeof(h), hipMemcpyDeviceToHost, st));
      RJ_HIP(hipStreamSynchronize(st));
      const uint64_t last = h[kXrLastSync];
      // the batch's last point opens a segment that the next batch replays; when it is the batch's first
      // point as well, more than a batch of text has no synchronisation point
      const bool can_walk = slots <= kWalkSlots;
      if (!final && last == ys) {
        // (larger batches, from the beginning: the parts make a segment of a gigabyte a matter of seconds)
        if (!can_walk || batch_chunks >= kBigBatchChunks) return 0;
        batch_chunks *= 4;
        grow_batch = true;
        break;
      }
      if (!can_walk && h[kXrMaxGap] > kMaxSegment) return 0;  // (one lane would take minutes)
      const bool has_long = can_walk && h[kXrMaxGap] > kLongSegment;
      const uint64_t max_len = has_long ? kLongSegment : ~0ull;
      const int rb = static_cast<int>(std::min<uint64_t>((nb + kReplayLanes - 1) / kReplayLanes, static_cast<uint64_t>(replay_blocks)));
      if (lds)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(xr_replay<true>), dim3(rb), dim3(kReplayLanes),
                           static_cast<size_t>(slots) * kReplayLanes * sizeof(int64_t), st, G, d_text, n, ys, s->xr_sync.as<uint64_t>(),
                           s->xr_seg_end.as<uint64_t>(), nb, nullptr, s->xr_scratch.as<uint64_t>(), s->xr_counts.as<uint32_t>(), max_len);
      else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(xr_replay<false>), dim3(rb), dim3(kReplayLanes), 0, st, G, d_text, n, ys,
                           s->xr_sync.as<uint64_t>(), s->xr_seg_end.as<uint64_t>(), nb, s->ring.as<int64_t>(),
                           s->xr_scratch.as<uint64_t>(), s->xr_counts.as<uint32_t>(), max_len);
      if (has_long) {
        // the long segments, one after the other: rounds of parts in parallel, the walk, the raw matches, the sink
        std::vector<uint64_t> h_sync(nb), h_end(nb);
        RJ_HIP(hipMemcpyAsync(h_sync.data(), s->xr_sync.p, nb * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        RJ_HIP(hipMemcpyAsync(h_end.data(), s->xr_seg_end.p, nb * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
        RJ_HIP(hipStreamSynchronize(st));
        const size_t ring_bytes = static_cast<size_t>(slots) * sizeof(int64_t);
        const size_t lds_bytes = lds ? ring_bytes * kReplayLanes : 0;
        int64_t* ring_mem = lds ? nullptr : s->ring.as<int64_t>();
        for (uint64_t c = 0; c < nb; c++) {
          const uint64_t a = h_sync[c], b = h_end[c];
          if (a == kNoSync || b == kNoSync || b - a <= kLongSegment) continue;
          // (the last part takes the remainder, up to 2 kPart - 1 bytes: every part then begins at least kPart bytes before b, so
          // the chunk its first byte lies in holds no synchronisation point of its own -- xr_join hands the part's list to
          // xr_compact through that chunk's sync[] / counts[] words)
          const uint64_t n_parts = (b - a) / kPart;
          // snaps: the candidate patterns [kReplayMaxRounds + 1][slots] | the walk's ring and the pattern that stopped it [2][slots] |
          // round 0's entry patterns [n_parts][slots] | the parts' true starts [n_parts][slots] | the exit rings
          // [kReplayMaxRounds + 1][n_parts][slots] | pattern scratch [lanes][2 slots]
          const int blocks = static_cast<int>(std::min<uint64_t>((n_parts + kReplayLanes - 1) / kReplayLanes, static_cast<uint64_t>(replay_blocks)));
          const size_t lanes = static_cast<size_t>(blocks) * kReplayLanes;
          // (a wide ring over a gigabyte asks for tens of GB here: no room means "not replayed", not an error)
          if (s->xr_snaps.reserve((static_cast<size_t>(kReplayMaxRounds) + 3 + (static_cast<size_t>(kReplayMaxRounds) + 3) * n_parts + 2 * lanes) * ring_bytes) != hipSuccess ||
              s->xr_raw_n.reserve(n_parts * 2 * sizeof(uint32_t)) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
          }
          int64_t* cands = s->xr_snaps.as<int64_t>();
          int64_t* walk_ring = cands + static_cast<size_t>(kReplayMaxRounds + 1) * slots;
          int64_t* walk_pat = walk_ring + slots;
          int64_t* entry0 = walk_pat + slots;
          int64_t* true_starts = entry0 + n_parts * static_cast<size_t>(slots);
          int64_t* exits = true_starts + n_parts * static_cast<size_t>(slots);
          int64_t* pat_scratch = exits + static_cast<size_t>(kReplayMaxRounds + 1) * n_parts * slots;
          uint32_t* raw_n = s->xr_raw_n.as<uint32_t>();
          int32_t* chosen = reinterpret_cast<int32_t*>(raw_n + n_parts);
          auto round = [&](uint64_t from, const int64_t* cand, int r) {
            if (lds)
              hipLaunchKernelGGL(HIP_KERNEL_NAME(xr_round<true>), dim3(blocks), dim3(kReplayLanes), lds_bytes, st, G, d_text, n, a, b, n_parts, from,
                                 cand, ring_mem, entry0, exits + static_cast<size_t>(r) * n_parts * slots, pat_scratch);
            else
              hipLaunchKernelGGL(HIP_KERNEL_NAME(xr_round<false>), dim3(blocks), dim3(kReplayLanes), 0, st, G, d_text, n, a, b, n_parts, from,
                                 cand, ring_mem, entry0, exits + static_cast<size_t>(r) * n_parts * slots, pat_scratch);
          };
          round(0, nullptr, 0);
          RJ_HIP(hipMemsetAsync(walk_ring, 0xFF, ring_bytes, st));  // (the segment begins with a free ring: every slot -1)
          RJ_HIP(hipMemsetAsync(state + kXrWalkNext, 0, 2 * sizeof(unsigned long long), st));
          CandFrom cand_from{};
          int n_cand = 1;
          for (;;) {
            hipLaunchKernelGGL(xr_walk, dim3(1), dim3(64), ring_bytes * static_cast<size_t>(n_cand + 3), st, static_cast<uint64_t>(slots), a, n_parts, n_cand,
                               cand_from, cands, entry0, exits, walk_ring, walk_pat, true_starts, chosen, state);
            RJ_HIP(hipMemcpyAsync(h, state, sizeof(h), hipMemcpyDeviceToHost, st));
            RJ_HIP(hipStreamSynchronize(st));
            if (h[kXrWalkStuck] == 0) break;
            if (n_cand > kReplayMaxRounds) return 0;  // (more order patterns at the cuts than rounds: given up)
            // the pattern that stopped the walk becomes candidate n_cand: every part from there on is replayed from it
            RJ_HIP(hipMemcpyAsync(cands + static_cast<size_t>(n_cand) * slots, walk_pat, ring_bytes, hipMemcpyDeviceToDevice, st));
            cand_from.v[n_cand] = h[kXrWalkNext];
            round(h[kXrWalkNext], cands + static_cast<size_t>(n_cand) * slots, n_cand);
            n_cand++;
            s->xr_rounds++;
          }
          uint64_t* min_begin = reinterpret_cast<uint64_t*>(exits);  // (the exit rings are done with: at least 3 n_parts words)
          if (lds)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(xr_emit<true>), dim3(blocks), dim3(kReplayLanes), lds_bytes, st, G, d_text, n, a, b, ys, n_parts, chosen,
                               cands, true_starts, ring_mem, pat_scratch, s->xr_scratch.as<uint64_t>(), raw_n, min_begin);
          else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(xr_emit<false>), dim3(blocks), dim3(kReplayLanes), 0, st, G, d_text, n, a, b, ys, n_parts, chosen,
                               cands, true_starts, ring_mem, pat_scratch, s->xr_scratch.as<uint64_t>(), raw_n, min_begin);
          uint32_t* keep = reinterpret_cast<uint32_t*>(min_begin + n_parts);
          uint32_t* first = keep + n_parts;
          int32_t* prev = reinterpret_cast<int32_t*>(first + n_parts);
          hipLaunchKernelGGL(xr_join, dim3(1), dim3(64), 0, st, a, ys, n_parts, raw_n, min_begin, s->xr_scratch.as<uint64_t>(), keep, first, prev,
                             s->xr_sync.as<uint64_t>(), s->xr_counts.as<uint32_t>());
          s->xr_parts += n_parts;
        }
      }
      hipLaunchKernelGGL(xr_offsets, dim3(1), dim3(1024), 0, st, s->xr_counts.as<uint32_t>(), nb, s->xr_offs.as<uint64_t>(), state);
      hipLaunchKernelGGL(xr_compact, dim3(static_cast<int>((nb * 64 + 255) / 256)), dim3(256), 0, st, s->xr_sync.as<uint64_t>(),
                         s->xr_counts.as<uint32_t>(), s->xr_offs.as<uint64_t>(), nb, ys, s->xr_scratch.as<uint64_t>(),
                         s->xr_out.as<uint64_t>(), s->xr_out_cap);
      ys = final ? y1 : last;
    }
    if (grow_batch) continue;
    RJ_HIP(hipMemcpyAsync(h, state, sizeof(h), hipMemcpyDeviceToHost, st));
    RJ_HIP(hipStreamSynchronize(st));
    RJ_HIP(hipGetLastError());
    const uint64_t total = h[kXrTotal];
    if (total <= s->xr_out_cap) {
      s->result_count = total;
      s->result = s->xr_out.as<uint64_t>();
      return 1;
    }
    // the result did not fit: make room and replay once more
    s->xr_out_cap = total + 1;
    RJ_HIP(s->xr_out.reserve(s->xr_out_cap * 2 * sizeof(uint64_t)));
  }
  return rj_fail(RJ_DEVICE_ERROR, "exact replay: result kept overflowing");
}

}  // namespace

// (the position set of the synchronisation-point walk lives in registers: up to 1024 positions, 16 x 64 bits)
bool exact_replay_fits(const rj_program* rp) { return rp->host->q8_risk && rp->dev.n_words <= 32 && rp->graph.n_states > 0; }

int run_exact(rj_scan* s, const uint8_t* d_text, uint64_t n, uint64_t sb, uint64_t se, hipStream_t st) {
  const rj_program* rp = s->prog;
  if (!exact_replay_fits(rp)) return 0;
  if (se > n + 1) se = n + 1;
  if (rp->dev.n_words <= 2) return run_exact_nq<1>(s, d_text, n, sb, se, st);
  if (rp->dev.n_words <= 4) return run_exact_nq<2>(s, d_text, n, sb, se, st);
  if (rp->dev.n_words <= 8) return run_exact_nq<4>(s, d_text, n, sb, se, st);
  if (rp->dev.n_words <= 16) return run_exact_nq<8>(s, d_text, n, sb, se, st);
  return run_exact_nq<16>(s, d_text, n, sb, se, st);
}

}  // namespace rejit_amd
