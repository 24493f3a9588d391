// rejit_amd/csrc/dense_streams.hip -- dense mode for patterns whose candidates cannot overlap (dense_streams.h): the
// no-fast-forward seeding + NFA loop of the reference (src/x64/codegen-x64.cc:535-677; class scan :1431-1565) as bit
// streams, ONE pass, every (begin, end) pair written ONCE at its final place.
//
// scan_dense_walk (kernels.hip) spends 23.5 VALU lane-operations per text byte on `[a-f]+[0-9]` (0.21 of the HBM rate) and
// writes its pairs twice (wave region, then gather_regions_by_wave).  Here:
//   * a wave iteration takes 2 KiB: a lane loads its 32 bytes (two dwordx4, the next iteration's already in flight),
//     turns them into one 32-bit stream per byte range and automaton position (dense_streams.h) and runs the first
//     automaton steps of its 32 starts position-major: 4 operations per position and step for 32 starts;
//   * the starts of a lane lie 16 bytes BEFORE its bytes, so every shifted stream comes from the lane's own word and the
//     word of the lane below (DPP; lane 0: the last lane of the iteration before, a scalar) -- no halo loads;
//   * a start still alive after the plan's depth (16 bytes for patterns with a loop) is walked by its lane with the
//     scalar walker (rj_lane_longest): rare on random text, counted (kCntSlowStarts) -- the host goes back to
//     scan_dense_walk for a scan object whose text makes it common;
//   * output as in emit_scan.hip: a wave owns a TILE of 32 KiB (16 iterations), stages its pairs in LDS (4 bytes each:
//     begin relative to the tile | length), publishes the tile's count as a {status, value} granule, finds the count of
//     everything before the tile by a decoupled look-back (Merrill & Garland) over tiles handed out in ARRIVAL order (a
//     ticket per workgroup and round, so every tile a wave waits for belongs to a wave that has started), and then
//     writes the staged pairs with coalesced 16-byte stores.  A tile with more pairs than the stage holds is computed a
//     second time with its base known, writing directly (`[a-p]` on a text of a..p).
// Every spin is bounded; a time-out or a walk beyond max_walk flags the run (kCntOverrun) and the engine repeats it on
// scan_dense_walk (and, from there, the carry scan).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "dense_streams.h"
#include "device_program.h"
#include "kernels.h"

namespace rejit_amd {

namespace {

constexpr int kWave = 64;
constexpr uint64_t kIter = 2048;                    // bytes per wave iteration: 64 lanes x 32
constexpr int kTileIters = 16;
constexpr uint64_t kTile = kIter * kTileIters;      // 32 KiB: a wave's unit of look-back
constexpr uint32_t kStage = 1024;                   // staged pairs per wave
constexpr uint32_t kLenBits = 17;                   // staged entry: begin - tile start (15 bits) << 17 | length
constexpr int kTilesPerTicket = 4;
constexpr unsigned long long kStatusAggregate = 1ull << 62, kStatusInclusive = 2ull << 62, kValueMask = (1ull << 62) - 1;
constexpr uint32_t kSpinLimit = 1u << 22;

__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x) & (kWave - 1); }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t x) {
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), CTRL, ROW_MASK, 0xF, true));
}
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x) {
  x += dpp_or_zero<0x111, 0xF>(x);
  x += dpp_or_zero<0x112, 0xF>(x);
  x += dpp_or_zero<0x114, 0xF>(x);
  x += dpp_or_zero<0x118, 0xF>(x);
  x += dpp_or_zero<0x142, 0xA>(x);
  x += dpp_or_zero<0x143, 0xC>(x);
  return x;
}
__device__ __forceinline__ uint32_t wave_from_lane_below(uint32_t x) { return dpp_or_zero<0x138, 0xF>(x); }  // wave_shr:1, lane 0 gets 0
__device__ __forceinline__ uint32_t wave_last_lane(uint32_t x) { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(x), kWave - 1)); }

struct AnyLane {
  __device__ __forceinline__ bool operator()(uint32_t x) const { return __ballot(x != 0) != 0; }
};

// the lane's 32 bytes at `at`, zeros beyond the end of the text; *valid: bit j = byte j is inside the text
__device__ __forceinline__ void load32_guarded(const uint8_t* text, uint64_t n, uint64_t at, uint32_t (&x)[8], uint32_t* valid) {
  if (at + 32 <= n) {
    const uint4 v0 = *reinterpret_cast<const uint4*>(text + at), v1 = *reinterpret_cast<const uint4*>(text + at + 16);
    x[0] = v0.x; x[1] = v0.y; x[2] = v0.z; x[3] = v0.w;
    x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w;
    *valid = ~0u;
    return;
  }
#pragma unroll
  for (int q = 0; q < 8; q++) x[q] = 0;
  uint32_t v = 0;
#pragma unroll 1
  for (int j = 0; j < 32; j++)
    if (at + j < n) {
      x[j >> 2] |= static_cast<uint32_t>(text[at + j]) << (8 * (j & 3));
      v |= 1u << j;
    }
  *valid = v;
}

// The scalar walk of a start that outlived the register steps: its own function, so that the program descriptor (122
// dwords, loaded from device memory here) does not compete for the scalar registers of the kernel's chunk loop.
// Returns the longest match's length (0: none) in the low kLenBits bits, bit 31 when the walk hit max_walk.
constexpr uint32_t kSlowOverrun = 1u << 31;
__device__ __attribute__((noinline)) uint32_t slow_longest(const DevProgram* Pd, const uint8_t* text, uint64_t n, uint64_t s,
                                                           unsigned long long* counters) {
  const DevProgram P = *Pd;
  uint64_t e = 0;
  bool ov = false;
  const bool found = rj_lane_longest<1>(P, text, n, s, &e, &ov, counters + kCntOverrun);
  uint32_t r = ov ? kSlowOverrun : 0u;
  if (!found) return r;
  uint64_t l = e - s;
  if (l >= (1u << kLenBits)) {  // (beyond what a staged entry holds: the run goes to the carry scan like an overrun)
    r |= kSlowOverrun;
    l = (1u << kLenBits) - 1u;
  }
  return r | static_cast<uint32_t>(l);
}

struct TileOut {
  uint32_t* stage;        // the wave's kStage staged entries (LDS)
  uint64_t* out;
  uint64_t out_cap;
  uint64_t direct_base;   // DIRECT: where the tile's first pair goes
};

// One tile: its matches counted and (DIRECT) written at direct_base onwards / (not DIRECT) staged in LDS.  Returns the
// tile's count (wave-uniform); *slow = starts that took the scalar walk; *overrun = a walk hit max_walk.
template <int NP, int NR, bool DIRECT>
__device__ __forceinline__ uint32_t stream_tile(const StreamParams& a, const DevProgram* Pd, const StreamMasks<NP>& mk, uint64_t base,
                                                const TileOut& o, uint32_t* slow, bool* overrun) {
  const int lane = lane_id();
  const StreamPlan& pl = a.plan;
  const uint64_t lim = a.se < a.n ? a.se : a.n;  // starts s in [sb, lim)
  // the streams of the 32 bytes before the tile: the "lane below" of lane 0 in the first iteration (every lane computes the
  // same words; a tile begins at a multiple of 32 KiB, so those bytes exist unless the tile is the text's first)
  uint32_t carry[NP];
  {
    uint32_t x[8], valid = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) x[q] = 0;
    if (base >= 32) load32_guarded(a.text, a.n, base - 32, x, &valid);
    rj_stream_classes<NP, NR>(pl, x, valid, carry);
#pragma unroll
    for (int k = 0; k < NP; k++) carry[k] = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(carry[k])));
  }
  // a tile whose bytes all lie inside the text and whose starts all lie inside the own range needs no guards (uniform)
  const bool inner = base + kTile <= a.n && base >= a.sb + kStreamShift && base + kTile <= lim + kStreamShift;
  const uint8_t* lane_text = a.text + base + static_cast<uint64_t>(lane) * 32;
  uint4 n0 = make_uint4(0, 0, 0, 0), n1 = make_uint4(0, 0, 0, 0);
  if (inner) {
    n0 = *reinterpret_cast<const uint4*>(lane_text);
    n1 = *reinterpret_cast<const uint4*>(lane_text + 16);
  }
  uint32_t count = 0;
#pragma unroll 1
  for (int it = 0; it < kTileIters; it++) {
    const uint64_t at = base + static_cast<uint64_t>(it) * kIter + static_cast<uint64_t>(lane) * 32;
    uint32_t x[8], valid = ~0u, start_mask = ~0u;
    if (inner) {
      x[0] = n0.x; x[1] = n0.y; x[2] = n0.z; x[3] = n0.w;
      x[4] = n1.x; x[5] = n1.y; x[6] = n1.z; x[7] = n1.w;
      if (it + 1 < kTileIters) {  // the next iteration's bytes, in flight while this one is evaluated
        n0 = *reinterpret_cast<const uint4*>(lane_text + static_cast<uint64_t>(it + 1) * kIter);
        n1 = *reinterpret_cast<const uint4*>(lane_text + static_cast<uint64_t>(it + 1) * kIter + 16);
      }
    } else {
      if (base + static_cast<uint64_t>(it) * kIter >= lim + kStreamShift) break;  // (uniform: no start of the range reaches this far)
      load32_guarded(a.text, a.n, at, x, &valid);
      // starts p = at - 16 + j inside [sb, lim)
      const uint64_t lo_p = a.sb + kStreamShift, hi_p = lim + kStreamShift;  // bit j counts iff lo_p <= at + j < hi_p
      const uint32_t lo = lo_p > at ? (lo_p - at < 32 ? static_cast<uint32_t>(lo_p - at) : 32u) : 0u;
      const uint32_t hi = hi_p > at ? (hi_p - at < 32 ? static_cast<uint32_t>(hi_p - at) : 32u) : 0u;
      const uint32_t below_hi = hi >= 32u ? ~0u : (1u << hi) - 1u, below_lo = lo >= 32u ? ~0u : (1u << lo) - 1u;
      start_mask = below_hi & ~below_lo;
    }
    uint32_t S[NP], Sb[NP];
    rj_stream_classes<NP, NR>(pl, x, valid, S);
#pragma unroll
    for (int k = 0; k < NP; k++) {
      const uint32_t below = wave_from_lane_below(S[k]);
      Sb[k] = lane == 0 ? carry[k] : below;
      carry[k] = wave_last_lane(S[k]);
    }
    uint32_t matched, alive, len[4], cand;
    rj_stream_steps<NP>(pl, mk, S, Sb, start_mask, AnyLane(), &matched, &alive, len, &cand);
    uint32_t fin = matched & ~alive;
    uint32_t walked = 0;  // alive starts whose scalar walk found a match
    if (__ballot(alive != 0) != 0) {
      // starts that outlived the register steps: the scalar walk, once now for "does it match" (the ranks below need the
      // count) and once more when the pair is written (no per-start storage)
      for (uint32_t m = alive; m; m &= m - 1) {
        const int j = __builtin_ctz(m);
        const uint32_t r = slow_longest(Pd, a.text, a.n, at + static_cast<uint64_t>(j) - kStreamShift, a.counters);
        if ((r & ~kSlowOverrun) != 0) walked |= 1u << j;
        if (r & kSlowOverrun) *overrun = true;
        (*slow)++;
      }
    }
    const uint32_t take = fin | walked;
    if (__ballot(take != 0) == 0) continue;
    const uint32_t mine = __popc(take);
    const uint32_t inc = wave_inclusive_sum(mine);
    uint32_t idx = count + inc - mine;
    const uint32_t rel0 = static_cast<uint32_t>(it) * static_cast<uint32_t>(kIter) + static_cast<uint32_t>(lane) * 32u;
    for (uint32_t m = take; m; m &= m - 1, idx++) {
      const int j = __builtin_ctz(m);
      uint32_t l;
      if ((walked >> j) & 1u) {
        l = slow_longest(Pd, a.text, a.n, at + static_cast<uint64_t>(j) - kStreamShift, a.counters) & ~kSlowOverrun;
      } else {
        l = rj_stream_len(len, j);
      }
      if (DIRECT) {
        const uint64_t s = at + static_cast<uint64_t>(j) - kStreamShift;
        const uint64_t pos = o.direct_base + idx;
        if (pos < o.out_cap) *reinterpret_cast<ulonglong2*>(o.out + 2 * pos) = make_ulonglong2(s, s + l);
      } else if (idx < kStage) {
        o.stage[idx] = ((rel0 + static_cast<uint32_t>(j)) << kLenBits) | l;
      }
    }
    count += wave_last_lane(inc);
  }
  return count;
}

// the tile's count published, the count of everything before it from the look-back (emit_scan.hip: emit_tile)
__device__ __forceinline__ bool look_back(unsigned long long* granules, uint64_t t, unsigned long long k, unsigned long long* before) {
  const int lane = lane_id();
  unsigned long long before_tile = 0;
  if (t == 0) {
    if (lane == 0) __hip_atomic_store(&granules[0], kStatusInclusive | k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *before = 0;
    return true;
  }
  if (lane == 0) __hip_atomic_store(&granules[t], kStatusAggregate | k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint64_t window_end = t;  // tiles [window_end - 64, window_end) are looked at, lane l reads tile window_end - 1 - l
  for (;;) {
    const bool valid = window_end >= static_cast<uint64_t>(lane) + 1;
    const uint64_t tile = valid ? window_end - 1 - static_cast<uint64_t>(lane) : 0;
    unsigned long long g = 0;
    uint32_t spins = 0;
    for (;;) {
      g = valid ? __hip_atomic_load(&granules[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kStatusInclusive;
      const uint64_t inclusive = __ballot((g >> 62) == 2);
      const uint64_t missing = __ballot((g >> 62) == 0);
      const uint64_t upto = inclusive ? (inclusive & (0 - inclusive)) : 0;
      const uint64_t needed = upto ? (upto | (upto - 1)) : ~0ull;
      if ((missing & needed) == 0) break;
      if (++spins > kSpinLimit) return false;
      __builtin_amdgcn_s_sleep(2);
    }
    const uint64_t inclusive = __ballot((g >> 62) == 2);
    const int stop = inclusive ? __builtin_ctzll(inclusive) : kWave;
    unsigned long long part = lane <= stop ? (g & kValueMask) : 0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    before_tile += part;
    if (inclusive) break;
    window_end -= kWave;
  }
  if (lane == 0) __hip_atomic_store(&granules[t], kStatusInclusive | (before_tile + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  *before = before_tile;
  return true;
}

}  // namespace

template <int NP, int NR>
__global__ __launch_bounds__(256) void dense_streams(StreamParams a, const DevProgram* Pd) {
  __shared__ unsigned long long s_ticket;
  __shared__ uint32_t s_stage[4][kStage];
  const int wv = static_cast<int>(threadIdx.x) >> 6;
  const int lane = lane_id();
  const StreamMasks<NP> mk = rj_stream_masks<NP>(a.plan);
  for (;;) {
    if (threadIdx.x == 0) s_ticket = atomicAdd(a.ticket, 1ull);
    __syncthreads();
    const uint64_t tk = s_ticket;
    __syncthreads();
    if (tk * kTilesPerTicket >= a.n_tiles) return;
    const uint64_t t = tk * kTilesPerTicket + static_cast<uint64_t>(wv);
    if (t >= a.n_tiles) continue;
    const uint64_t base = (a.first_tile + t) * kTile;
    // the run is void already (a time-out or an overrun elsewhere): publish a count so that nobody waits for this tile
    const bool void_run = __hip_atomic_load(a.counters + kCntOverrun, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    TileOut o{s_stage[wv], a.out, a.out_cap, 0};
    uint32_t slow = 0;
    bool overrun = false;
    unsigned long long k = 0;
    if (!void_run) k = stream_tile<NP, NR, false>(a, Pd, mk, base, o, &slow, &overrun);
    overrun = __ballot(overrun) != 0;
    unsigned long long before = 0;
    bool ok = !void_run && !overrun;
    if (ok) ok = look_back(a.granules, t, k, &before);
    if (!ok) {
      if (lane == 0) {
        __hip_atomic_store(&a.granules[t], kStatusInclusive | 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.counters[kCntOverrun] = 1;
        if (a.host_counters) a.host_counters[kCntOverrun] = 1;
      }
      continue;
    }
    if (t == a.n_tiles - 1 && lane == 0) {
      a.counters[kCntFinal] = before + k;
      a.counters[kCntCands] = before + k;
      a.counters[kCntHits] = before + k;
      if (a.host_counters) {
        a.host_counters[kCntFinal] = before + k;
        a.host_counters[kCntCands] = before + k;
        a.host_counters[kCntHits] = before + k;
      }
    }
    if (slow != 0) {
      uint32_t total = slow;
#pragma unroll
      for (int w = 32; w > 0; w >>= 1) total += __shfl_xor(total, w);
      if (lane == 0) atomicAdd(a.counters + kCntSlowStarts, static_cast<unsigned long long>(total));
    }
    if (k <= kStage) {
      // the staged pairs to their final place: lane i takes pair i, 16 bytes each -- one KiB per wave store
      const uint64_t tile_start = base - kStreamShift;
      for (uint32_t i = static_cast<uint32_t>(lane); i < static_cast<uint32_t>(k); i += kWave) {
        const uint32_t e = o.stage[i];
        const uint64_t b = tile_start + (e >> kLenBits);
        const uint64_t pos = before + i;
        if (pos < a.out_cap) *reinterpret_cast<ulonglong2*>(a.out + 2 * pos) = make_ulonglong2(b, b + (e & ((1u << kLenBits) - 1u)));
      }
    } else {
      // more pairs than the stage holds: the tile once more, its base known now
      o.direct_base = before;
      uint32_t slow2 = 0;
      bool ov2 = false;
      (void)stream_tile<NP, NR, true>(a, Pd, mk, base, o, &slow2, &ov2);
    }
  }
}

uint64_t stream_tiles(uint64_t sb, uint64_t se, uint64_t n, uint64_t* first_tile) {
  const uint64_t lim = se < n ? se : n;
  *first_tile = (sb + kStreamShift) / kTile;
  if (lim <= sb) return 0;
  return (lim - 1 + kStreamShift) / kTile - *first_tile + 1;
}

size_t stream_scratch_bytes(uint64_t n_tiles) { return (n_tiles + 1) * sizeof(unsigned long long); }

// scratch: [0] the ticket counter, [1 ..] one granule per tile; cleared here
namespace {
template <int NP>
void launch_np(const StreamParams& a, const DevProgram* Pd, dim3 g, hipEvent_t t0, hipEvent_t t1, hipStream_t st) {
  const dim3 b(256);
  const uint32_t nr = a.plan.n_ranges;
  if (nr <= 1) hipExtLaunchKernelGGL((dense_streams<NP, 1>), g, b, 0, st, t0, t1, 0, a, Pd);
  else if (nr <= 2) hipExtLaunchKernelGGL((dense_streams<NP, 2>), g, b, 0, st, t0, t1, 0, a, Pd);
  else if (nr <= 4) hipExtLaunchKernelGGL((dense_streams<NP, 4>), g, b, 0, st, t0, t1, 0, a, Pd);
  else hipExtLaunchKernelGGL((dense_streams<NP, 8>), g, b, 0, st, t0, t1, 0, a, Pd);
}
}  // namespace

// d_program: the pattern's DevProgram in device memory (the scalar walk of the rare long-lived start reads it there)
void launch_dense_streams(StreamParams a, const DevProgram* d_program, unsigned long long* scratch, hipEvent_t t0, hipEvent_t t1, hipStream_t st) {
  (void)hipMemsetAsync(scratch, 0, stream_scratch_bytes(a.n_tiles), st);
  a.ticket = scratch;
  a.granules = scratch + 1;
  uint64_t blocks = (a.n_tiles + kTilesPerTicket - 1) / kTilesPerTicket;
  blocks = blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks;  // persistent: workgroups take tickets until none is left
  const dim3 g(static_cast<unsigned>(blocks));
  const uint32_t np = a.plan.n_pos;
  if (np <= 1) launch_np<1>(a, d_program, g, t0, t1, st);
  else if (np <= 2) launch_np<2>(a, d_program, g, t0, t1, st);
  else if (np <= 3) launch_np<3>(a, d_program, g, t0, t1, st);
  else if (np <= 4) launch_np<4>(a, d_program, g, t0, t1, st);
  else if (np <= 6) launch_np<6>(a, d_program, g, t0, t1, st);
  else launch_np<8>(a, d_program, g, t0, t1, st);
}

}  // namespace rejit_amd
