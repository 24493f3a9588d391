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
//     plan's own scalar walk (rj_stream_walk): rare on random text, counted (kCntSlowStarts) -- the host goes back to
//     scan_dense_walk for a scan object whose text makes it common;
//   * output as in emit_scan.hip: a wave owns a TILE of 32 KiB (16 iterations) and stages its pairs in LDS (4 bytes each:
//     begin relative to the tile | length); the workgroup publishes the count of its four tiles (tiles are handed out in
//     ARRIVAL order: a ticket per workgroup and round, so everything a wave waits for belongs to a wave that has started),
//     computes its NEXT four tiles into the other stage, and only then looks up the count of everything before the staged
//     ones (tile_lookback.h: published long ago by then) and writes them with coalesced 16-byte stores.  A tile with more
//     pairs than a stage holds is computed a second time with its base known, writing directly (`[a-p]` on a text of a..p).
// Every spin is bounded; a time-out or a walk beyond max_walk flags the run (kCntOverrun) and the engine repeats it on
// scan_dense_walk (and, from there, the carry scan).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdlib>

#include "dense_streams.h"
#include "device_program.h"
#include "kernels.h"
#include "tile_lookback.h"

namespace rejit_amd {

namespace {

constexpr int kWave = 64;
constexpr uint64_t kIter = 2048;                    // bytes per wave iteration: 64 lanes x 32
constexpr int kTileIters = 16;
constexpr uint64_t kTile = kIter * kTileIters;      // 32 KiB: a wave's tile (four of them are a unit of the prefix scan)
constexpr uint32_t kStage = 640;                    // staged pairs per wave and stage (two stages: 20 KiB per workgroup)
constexpr uint32_t kLenBits = 17;                   // staged entry: begin - tile start (15 bits) << 17 | length (engine.hip caps max_walk below 2^17)
constexpr int kTilesPerTicket = 4;
#ifndef RJ_DS_AHEAD
#define RJ_DS_AHEAD 1
#endif
constexpr int kAhead = RJ_DS_AHEAD;                 // iterations of text in flight per wave (2: two buffers, each reloaded behind its class streams)

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

struct TileOut {
  uint32_t* stage;        // the wave's kStage staged entries (LDS)
  uint64_t* out;
  uint64_t out_cap;
  uint64_t direct_base;   // DIRECT: where the tile's first pair goes
};

// One tile: its matches counted and (DIRECT) written at direct_base onwards / (not DIRECT) staged in LDS.  Returns the
// tile's count (wave-uniform); *slow = starts that took the scalar walk; *overrun = a walk hit max_walk.
//
// SELECT (StreamPlan::select: candidates may overlap, matches of at most 16 bytes): the left-most-longest selection is made
// here.  Inside a lane it is a short chain over the lane's matches (rj_stream_select); from lane to lane the one thing that
// travels is d = how many of a lane's first starts lie inside a match selected below it.  Every lane first assumes d = 0,
// then takes the d its neighbour below computed and repeats while anything changes (lane k is exact after k rounds; on
// text whose matches are not packed, after one or two).  From iteration to iteration d is a scalar.  A TILE's entry state
// comes from the 2 KiB before it (one iteration more, `it` = -1, nothing written): a lane without any match there resets
// the chain whatever came before, so everything behind the LAST such lane is exact.  No such lane in 2 KiB (64 lanes each
// with a match: a text packed with matches): *unsure -- the run is void and the engine repeats it on scan_dense_walk.
// RUN (StreamPlan::run_shape: `X+`, `X+ Y`; round 6): the steps are ONE 64-bit addition per lane (dense_streams.h: rj_stream_runs)
// -- `[a-f]+[0-9]` spent 75 of its 267 instructions per lane and 32 bytes in the generic steps, at the VALU issue rate.
template <int NP, int NR, bool HIGH, bool DIRECT, bool SELECT, bool RUN>
__device__ __forceinline__ uint32_t stream_tile(const StreamParams& a, const StreamMasks<NP>& mk, const StreamRangeMasks<NP, NR>& rm, uint64_t base,
                                                const TileOut& o, uint32_t* slow, bool* overrun) {
  const int lane = lane_id();
  const StreamPlan& pl = a.plan;
  const uint64_t lim = a.se < a.n ? a.se : a.n;  // starts s in [sb, lim)
  const int it0 = (SELECT && base >= kIter) ? -1 : 0;   // (tiles begin at multiples of 32 KiB: the first one of a text has nothing before it)
  const uint64_t first_at = base - (it0 < 0 ? kIter : 0u);
  uint32_t d_carry = 0;   // SELECT: starts at the beginning of the iteration that lie inside a selected match (uniform)
  // the streams of the 32 bytes before the tile: the "lane below" of lane 0 in the first iteration (every lane computes the
  // same words; a tile begins at a multiple of 32 KiB, so those bytes exist unless the tile is the text's first)
  uint32_t carry[NP];
  {
    uint32_t x[8], valid = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) x[q] = 0;
    if (first_at >= 32) load32_guarded(a.text, a.n, first_at - 32, x, &valid);
    rj_stream_classes<NP, NR, HIGH>(pl, rm, x, valid, carry);
#pragma unroll
    for (int k = 0; k < NP; k++) carry[k] = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(carry[k])));
  }
  // a tile whose bytes all lie inside the text and whose starts all lie inside the own range needs no guards (uniform)
  const bool inner = base + kTile <= a.n && first_at >= a.sb + kStreamShift && base + kTile <= lim + kStreamShift;
  const uint8_t* lane_text = a.text + base + static_cast<uint64_t>(lane) * 32;
  // The lane's 32 bytes are loaded one iteration ahead.  Where the load sits decides what the back edge of the loop costs:
  // one buffer, reloaded AHEAD of its last readers (the class streams), lands in fresh registers and is copied twice -- 20
  // moves per iteration, 7 % of the kernel's instructions.  Measured on one box, alternating builds (`[a-f]+[0-9]` /
  // `[0-9][0-9][0-9]` over 5 GB): that form 1.25 / 1.20 ms; the reload BEHIND the class streams into the same registers
  // (69 registers, 7 waves per SIMD) 1.20 / 1.29; two buffers used alternately, the loop unrolled by two (81 registers, 5
  // waves) 1.25 / 1.15.  So: the plain kernels reload late into the one buffer, the SELECT kernels alternate two.
  uint4 bufa0 = make_uint4(0, 0, 0, 0), bufa1 = make_uint4(0, 0, 0, 0), bufb0 = make_uint4(0, 0, 0, 0), bufb1 = make_uint4(0, 0, 0, 0);
  if (inner) {
    bufa0 = *reinterpret_cast<const uint4*>(lane_text + static_cast<int64_t>(it0) * static_cast<int64_t>(kIter));
    bufa1 = *reinterpret_cast<const uint4*>(lane_text + static_cast<int64_t>(it0) * static_cast<int64_t>(kIter) + 16);
    if (kAhead == 2) {
      bufb0 = *reinterpret_cast<const uint4*>(lane_text + static_cast<int64_t>(it0 + 1) * static_cast<int64_t>(kIter));
      bufb1 = *reinterpret_cast<const uint4*>(lane_text + static_cast<int64_t>(it0 + 1) * static_cast<int64_t>(kIter) + 16);
    }
  }
  uint32_t count = 0;
  // one iteration (2 KiB of the tile); false: the tile ends here
  auto iteration = [&](const int it, uint4& n0, uint4& n1, uint4& next0, uint4& next1) __attribute__((always_inline)) -> bool {
    const uint64_t at = base + static_cast<uint64_t>(static_cast<int64_t>(it) * static_cast<int64_t>(kIter)) + static_cast<uint64_t>(lane) * 32;
    uint32_t x[8], valid = ~0u, start_mask = ~0u;
    if (inner) {
      x[0] = n0.x; x[1] = n0.y; x[2] = n0.z; x[3] = n0.w;
      x[4] = n1.x; x[5] = n1.y; x[6] = n1.z; x[7] = n1.w;
      if (SELECT && kAhead == 1 && it + 1 < kTileIters) {  // the next iteration's bytes, in flight while this one is evaluated
        next0 = *reinterpret_cast<const uint4*>(lane_text + static_cast<int64_t>(it + 1) * static_cast<int64_t>(kIter));
        next1 = *reinterpret_cast<const uint4*>(lane_text + static_cast<int64_t>(it + 1) * static_cast<int64_t>(kIter) + 16);
      }
    } else {
      if (it >= 0 && base + static_cast<uint64_t>(it) * kIter >= lim + kStreamShift) return false;  // (uniform: no start of the range reaches this far)
      load32_guarded(a.text, a.n, at, x, &valid);
      // starts p = at - 16 + j inside [sb, lim)
      const uint64_t lo_p = a.sb + kStreamShift, hi_p = lim + kStreamShift;  // bit j counts iff lo_p <= at + j < hi_p
      const uint32_t lo = lo_p > at ? (lo_p - at < 32 ? static_cast<uint32_t>(lo_p - at) : 32u) : 0u;
      const uint32_t hi = hi_p > at ? (hi_p - at < 32 ? static_cast<uint32_t>(hi_p - at) : 32u) : 0u;
      const uint32_t below_hi = hi >= 32u ? ~0u : (1u << hi) - 1u, below_lo = lo >= 32u ? ~0u : (1u << lo) - 1u;
      start_mask = below_hi & ~below_lo;
    }
    uint32_t S[NP], Sb[NP];
    rj_stream_classes<NP, NR, HIGH>(pl, rm, x, valid, S);
    if ((!SELECT || kAhead == 2) && inner && it + kAhead < kTileIters) {  // (next0 / next1 ARE n0 / n1 here: the streams are pinned, then the reload)
#pragma unroll
      for (int k = 0; k < NP; k++) asm volatile("" : "+v"(S[k]));
      next0 = *reinterpret_cast<const uint4*>(lane_text + static_cast<int64_t>(it + kAhead) * static_cast<int64_t>(kIter));
      next1 = *reinterpret_cast<const uint4*>(lane_text + static_cast<int64_t>(it + kAhead) * static_cast<int64_t>(kIter) + 16);
    }
#pragma unroll
    for (int k = 0; k < NP; k++) {
      const uint32_t below = wave_from_lane_below(S[k]);
      Sb[k] = lane == 0 ? carry[k] : below;
      carry[k] = wave_last_lane(S[k]);
    }
    const uint32_t rel0 = static_cast<uint32_t>(it) * static_cast<uint32_t>(kIter) + static_cast<uint32_t>(lane) * 32u;
    auto put = [&](int j, uint32_t l, uint32_t idx) {
      if (DIRECT) {
        const uint64_t s = at + static_cast<uint64_t>(j) - kStreamShift;
        const uint64_t pos = o.direct_base + idx;
        if (pos < o.out_cap) *reinterpret_cast<ulonglong2*>(o.out + 2 * pos) = make_ulonglong2(s, s + l);
      } else if (idx < kStage) {
        o.stage[idx] = ((rel0 + static_cast<uint32_t>(j)) << kLenBits) | l;
      }
    };
    if (RUN) {
      uint64_t starts, ends;
      uint32_t alive_r;
      rj_stream_runs(pl.run_shape, S[0], Sb[0], NP > 1 ? S[NP > 1 ? 1 : 0] : 0u, NP > 1 ? Sb[NP > 1 ? 1 : 0] : 0u, start_mask, &starts, &ends, &alive_r);
      uint32_t walked_l = 0;   // the run that reaches the window's top (at most one per lane): its scalar walk
      if (__ballot(alive_r != 0) != 0 && alive_r != 0) {
        bool ov = false;
        walked_l = rj_stream_walk(pl, a.text, a.n, at + static_cast<uint64_t>(__builtin_ctz(alive_r)) - kStreamShift, a.max_walk, &ov);
        if (ov) *overrun = true;
        (*slow)++;
      }
      const uint32_t mine = static_cast<uint32_t>(__popcll(ends)) + (walked_l != 0 ? 1u : 0u);
      if (__ballot(mine != 0) == 0) return true;
      const uint32_t inc = wave_inclusive_sum(mine);
      uint32_t idx = count + inc - mine;
      while (ends != 0) {
        int j;
        const uint32_t l = rj_stream_run_next(pl.run_shape, starts, &ends, &j);
        put(j, l, idx++);
      }
      if (walked_l != 0) put(__builtin_ctz(alive_r), walked_l, idx);   // (the top run is the lane's last)
      count += wave_last_lane(inc);
      return true;
    }
    uint32_t matched, alive, len[4], cand;
    rj_stream_steps<NP>(pl, mk, S, Sb, start_mask, AnyLane(), &matched, &alive, len, &cand);
    uint32_t fin = matched & ~alive;
    uint32_t walked = 0;  // alive starts whose scalar walk found a match
    if (__ballot(alive != 0) != 0) {
      // starts that outlived the register steps: the scalar walk (dense_streams.h: state in a register, classes from the
      // plan's ranges, a byte load per step), once now for "does it match" (the ranks below need the count) and once more
      // when the pair is written (no per-start storage)
      for (uint32_t m = alive; m; m &= m - 1) {
        const int j = __builtin_ctz(m);
        bool ov = false;
        if (rj_stream_walk(pl, a.text, a.n, at + static_cast<uint64_t>(j) - kStreamShift, a.max_walk, &ov) != 0) walked |= 1u << j;
        if (ov) *overrun = true;
        (*slow)++;
      }
    }
    uint32_t take = fin | walked;
    if (SELECT) {
      if (it < 0) {
        const uint64_t quiet = __ballot(take == 0);
        if (quiet == 0) *overrun = true;   // (no lane resets the chain: the tile's entry state is not known)
        const int last_quiet = quiet != 0 ? 63 - static_cast<int>(__builtin_clzll(quiet)) : 63;
        if (lane <= last_quiet) take = 0;  // (what lies before the last reset does not matter)
        d_carry = 0;
      }
      if (__ballot(take != 0) == 0) {
        d_carry = 0;   // (2 KiB without a match)
        return true;
      }
      uint32_t d_in = lane == 0 ? d_carry : 0u, d_out;
      uint32_t sel = rj_stream_select(take, len, d_in, &d_out);
      for (;;) {
        uint32_t want = wave_from_lane_below(d_out);
        if (lane == 0) want = d_carry;
        const bool redo = want != d_in;
        if (__ballot(redo) == 0) break;
        if (redo) {
          d_in = want;
          sel = rj_stream_select(take, len, d_in, &d_out);
        }
      }
      d_carry = wave_last_lane(d_out);
      if (it < 0) return true;
      take = sel;
    }
    if (__ballot(take != 0) == 0) return true;
    const uint32_t mine = __popc(take);
    const uint32_t inc = wave_inclusive_sum(mine);
    uint32_t idx = count + inc - mine;
    for (uint32_t m = take; m; m &= m - 1, idx++) {
      const int j = __builtin_ctz(m);
      uint32_t l;
      if ((walked >> j) & 1u) {
        bool ov = false;
        l = rj_stream_walk(pl, a.text, a.n, at + static_cast<uint64_t>(j) - kStreamShift, a.max_walk, &ov);
      } else {
        l = rj_stream_len(len, j);
      }
      put(j, l, idx);
    }
    count += wave_last_lane(inc);
    return true;
  };
  if (kAhead == 2) {
#pragma unroll 1
    for (int it = it0; it < kTileIters; it += 2) {
      if (!iteration(it, bufa0, bufa1, bufa0, bufa1)) break;
      if (it + 1 >= kTileIters || !iteration(it + 1, bufb0, bufb1, bufb0, bufb1)) break;
    }
  } else if (SELECT) {
#pragma unroll 1
    for (int it = it0; it < kTileIters; it += 2) {
      if (!iteration(it, bufa0, bufa1, bufb0, bufb1)) break;
      if (it + 1 >= kTileIters || !iteration(it + 1, bufb0, bufb1, bufa0, bufa1)) break;
    }
  } else {
#pragma unroll 1
    for (int it = it0; it < kTileIters; it++)
      if (!iteration(it, bufa0, bufa1, bufa0, bufa1)) break;
  }
  return count;
}

}  // namespace

// One unit of the prefix scan per WORKGROUP and round (its four tiles, one per wave), resolved ONE ROUND LATE
// (tile_lookback.h): a round computes its tiles, stages their pairs in LDS and publishes their count; then the count of
// everything before the PREVIOUS round's tiles is looked up -- published long ago by then, no waiting -- and that round's
// staged pairs go out.  Two stages per wave, used alternately.  History: a granule per tile, resolved at once: the
// look-back set the pace (`[@#]`, one step, took as long as `[a-f]+[0-9]`); a granule per workgroup: 1.62 ms per 5 GB;
// two-level look-back: 1.53; with no look-back at all (wrong output) 1.17 -- the rest was waiting for the slowest of the
// ~1800 units in flight, which this form no longer does.
#ifdef RJ_DS_WAVES
#define RJ_DS_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(RJ_DS_WAVES, RJ_DS_WAVES)))
#else
#define RJ_DS_WAVES_ATTR
#endif
template <int NP, int NR, bool HIGH, bool SELECT, bool RUN = false>
__global__ __launch_bounds__(256) RJ_DS_WAVES_ATTR void dense_streams(StreamParams a) {
  __shared__ unsigned long long s_ticket, s_before;
  __shared__ uint32_t s_count[2][kTilesPerTicket], s_bad;
  __shared__ uint32_t s_stage[2][kTilesPerTicket][kStage];
  const int wv = static_cast<int>(threadIdx.x) >> 6;
  const int lane = lane_id();
  const StreamMasks<NP> mk = rj_stream_masks<NP>(a.plan);
  const StreamRangeMasks<NP, NR> rm = rj_stream_range_masks<NP, NR>(a.plan);
  const uint64_t n_tickets = (a.n_tiles + kTilesPerTicket - 1) / kTilesPerTicket;
  if (threadIdx.x == 0) s_bad = 0;
  uint64_t prev_tk = ~0ull;  // the round whose pairs wait in s_stage[cur ^ 1]
  uint32_t prev_k = 0;
  int cur = 0;
  for (;;) {
    if (threadIdx.x == 0) s_ticket = atomicAdd(a.ticket, 1ull);
    __syncthreads();
    const uint64_t tk = s_ticket;
    const bool have = tk < n_tickets;
    const uint64_t t = tk * kTilesPerTicket + static_cast<uint64_t>(wv);
    const uint64_t base = (a.first_tile + t) * kTile;
    uint32_t k = 0;
    if (have) {
      // (the run is void already -- a time-out or an overrun elsewhere: only publish a count so that nobody waits)
      const bool void_run = __hip_atomic_load(a.counters + kCntOverrun, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
      const TileOut o{s_stage[cur][wv], a.out, a.out_cap, 0};
      uint32_t slow = 0;
      bool overrun = false;
      if (!void_run && t < a.n_tiles) k = stream_tile<NP, NR, HIGH, false, SELECT, RUN>(a, mk, rm, base, o, &slow, &overrun);
      if (__ballot(overrun) != 0 && lane == 0) {
        a.counters[kCntOverrun] = 1;
        if (a.host_counters) a.host_counters[kCntOverrun] = 1;
      }
      if (slow != 0) {
        uint32_t total = slow;
#pragma unroll
        for (int w = 32; w > 0; w >>= 1) total += __shfl_xor(total, w);
        if (lane == 0) atomicAdd(a.counters + kCntSlowStarts, static_cast<unsigned long long>(total));
      }
      if (lane == 0) s_count[cur][wv] = k;
    }
    // wave 0, its own tile done, looks up the count of everything before the PREVIOUS round (the other waves are still at
    // their tiles); behind the barrier it publishes this round's count
    unsigned long long prev_total = 0;
    if (wv == 0 && prev_tk != ~0ull) {
      unsigned long long before = 0;
      const bool ok = lookback::resolve(a.granules, n_tickets, prev_tk, &before);
      if (lane == 0) {
#pragma unroll
        for (int w = 0; w < kTilesPerTicket; w++) prev_total += s_count[cur ^ 1][w];
        if (!ok) {
          a.counters[kCntOverrun] = 1;
          if (a.host_counters) a.host_counters[kCntOverrun] = 1;
          s_bad = 1;
        } else if (prev_tk == n_tickets - 1) {
          a.counters[kCntFinal] = before + prev_total;
          a.counters[kCntCands] = before + prev_total;
          a.counters[kCntHits] = before + prev_total;
          if (a.host_counters) {
            a.host_counters[kCntFinal] = before + prev_total;
            a.host_counters[kCntCands] = before + prev_total;
            a.host_counters[kCntHits] = before + prev_total;
          }
        }
        s_before = before;
      }
    }
    __syncthreads();
    if (have && threadIdx.x == 0) {
      unsigned long long total = 0;
#pragma unroll
      for (int w = 0; w < kTilesPerTicket; w++) total += s_count[cur][w];
      lookback::publish(a.granules, n_tickets, tk, total);
    }
    if (prev_tk != ~0ull && s_bad == 0) {
      const uint64_t pt = prev_tk * kTilesPerTicket + static_cast<uint64_t>(wv);
      const uint64_t pbase = (a.first_tile + pt) * kTile;
      unsigned long long before = s_before;
      for (int w = 0; w < wv; w++) before += s_count[cur ^ 1][w];
      if (pt >= a.n_tiles) {
        // (a wave beyond the last tile)
      } else if (prev_k <= kStage) {
        // the staged pairs to their final place: lane i takes pair i, 16 bytes each -- one KiB per wave store
        const uint32_t* stage = s_stage[cur ^ 1][wv];
        const uint64_t tile_start = pbase - kStreamShift;
        for (uint32_t i = static_cast<uint32_t>(lane); i < prev_k; i += kWave) {
          const uint32_t e = stage[i];
          const uint64_t b = tile_start + (e >> kLenBits);
          const uint64_t pos = before + i;
          if (pos < a.out_cap) *reinterpret_cast<ulonglong2*>(a.out + 2 * pos) = make_ulonglong2(b, b + (e & ((1u << kLenBits) - 1u)));
        }
      } else {
        // more pairs than the stage holds: the tile once more, its base known now
        const TileOut o{s_stage[cur ^ 1][wv], a.out, a.out_cap, before};
        uint32_t slow2 = 0;
        bool ov2 = false;
        (void)stream_tile<NP, NR, HIGH, true, SELECT, RUN>(a, mk, rm, pbase, o, &slow2, &ov2);
      }
    }
    if (!have) return;
    // (two barriers per round: s_ticket is rewritten when every wave has read it -- before the second one --, s_count[cur ^ 1]
    // and s_before behind the next round's first)
    prev_tk = tk;
    prev_k = k;
    cur ^= 1;
  }
}

#ifndef RJ_DENSE_SELECT_TU
uint64_t stream_tiles(uint64_t sb, uint64_t se, uint64_t n, uint64_t* first_tile) {
  const uint64_t lim = se < n ? se : n;
  *first_tile = (sb + kStreamShift) / kTile;
  if (lim <= sb) return 0;
  return (lim - 1 + kStreamShift) / kTile - *first_tile + 1;
}

size_t stream_scratch_bytes(uint64_t n_tiles) {
  return (lookback::granule_words((n_tiles + kTilesPerTicket - 1) / kTilesPerTicket) + 1) * sizeof(unsigned long long);
}
#endif

// The kernels are instantiated in TWO translation units that compile side by side: this file (the plain kernels) and
// dense_streams_select.hip, which includes this file with RJ_DENSE_SELECT_TU defined (the SELECT kernels) -- 96 kernels in one unit
// took 3.4 of a clean build's 4 minutes.
void launch_dense_streams_shape(const StreamParams& a, dim3 g, hipEvent_t t0, hipEvent_t t1, hipStream_t st);         // this unit
void launch_dense_streams_select_shape(const StreamParams& a, dim3 g, hipEvent_t t0, hipEvent_t t1, hipStream_t st);  // the other

namespace {
#ifdef RJ_DENSE_SELECT_TU
constexpr bool kSelectUnit = true;
#else
constexpr bool kSelectUnit = false;
#endif
template <int NP, int NR>
void launch_nr(const StreamParams& a, dim3 g, hipEvent_t t0, hipEvent_t t1, hipStream_t st) {
  const dim3 b(256);
  if (a.plan.high_half == 0) hipExtLaunchKernelGGL((dense_streams<NP, NR, false, kSelectUnit>), g, b, 0, st, t0, t1, 0, a);
  else hipExtLaunchKernelGGL((dense_streams<NP, NR, true, kSelectUnit>), g, b, 0, st, t0, t1, 0, a);
}
template <int NP>
void launch_np(const StreamParams& a, dim3 g, hipEvent_t t0, hipEvent_t t1, hipStream_t st) {
  const uint32_t nr = a.plan.n_ranges;
  if (nr <= 1) launch_nr<NP, 1>(a, g, t0, t1, st);
  else if (nr <= 2) launch_nr<NP, 2>(a, g, t0, t1, st);
  else if (nr <= 4) launch_nr<NP, 4>(a, g, t0, t1, st);
  else launch_nr<NP, 8>(a, g, t0, t1, st);
}
#ifndef RJ_DENSE_SELECT_TU
// the run form (`X+`, `X+ Y`): NP 1 / 2, ranges rounded up to 2 / 8 -- eight kernels; the six-position instantiations of both
// units went in exchange (plans of five and six positions take the eight-position kernels): 96 -> 88 kernels in all
template <int NP, int NR>
void launch_run_nr(const StreamParams& a, dim3 g, hipEvent_t t0, hipEvent_t t1, hipStream_t st) {
  const dim3 b(256);
  if (a.plan.high_half == 0) hipExtLaunchKernelGGL((dense_streams<NP, NR, false, false, true>), g, b, 0, st, t0, t1, 0, a);
  else hipExtLaunchKernelGGL((dense_streams<NP, NR, true, false, true>), g, b, 0, st, t0, t1, 0, a);
}
void launch_run_shape(const StreamParams& a, dim3 g, hipEvent_t t0, hipEvent_t t1, hipStream_t st) {
  const bool few = a.plan.n_ranges <= 2;
  if (a.plan.n_pos <= 1) {
    if (few) launch_run_nr<1, 2>(a, g, t0, t1, st);
    else launch_run_nr<1, 8>(a, g, t0, t1, st);
  } else {
    if (few) launch_run_nr<2, 2>(a, g, t0, t1, st);
    else launch_run_nr<2, 8>(a, g, t0, t1, st);
  }
}
#endif
void launch_shape(const StreamParams& a, dim3 g, hipEvent_t t0, hipEvent_t t1, hipStream_t st) {
  const uint32_t np = a.plan.n_pos;
#ifndef RJ_DENSE_SELECT_TU
  if (a.plan.run_shape != 0 && !a.plan.select && np <= 2) return launch_run_shape(a, g, t0, t1, st);
#endif
  if (np <= 1) launch_np<1>(a, g, t0, t1, st);
  else if (np <= 2) launch_np<2>(a, g, t0, t1, st);
  else if (np <= 3) launch_np<3>(a, g, t0, t1, st);
  else if (np <= 4) launch_np<4>(a, g, t0, t1, st);
  else launch_np<8>(a, g, t0, t1, st);
}
}  // namespace

#ifdef RJ_DENSE_SELECT_TU
void launch_dense_streams_select_shape(const StreamParams& a, dim3 g, hipEvent_t t0, hipEvent_t t1, hipStream_t st) { launch_shape(a, g, t0, t1, st); }
#else
void launch_dense_streams_shape(const StreamParams& a, dim3 g, hipEvent_t t0, hipEvent_t t1, hipStream_t st) { launch_shape(a, g, t0, t1, st); }

// scratch: [0] the ticket counter, [1 ..] one granule per ticket (four tiles), then one per group of tickets; cleared here
void launch_dense_streams(StreamParams a, unsigned long long* scratch, hipEvent_t t0, hipEvent_t t1, hipStream_t st) {
  (void)hipMemsetAsync(scratch, 0, stream_scratch_bytes(a.n_tiles), st);
  a.ticket = scratch;
  a.granules = scratch + 1;
  uint64_t blocks = (a.n_tiles + kTilesPerTicket - 1) / kTilesPerTicket;
  blocks = blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks;  // persistent: workgroups take tickets until none is left
  const dim3 g(static_cast<unsigned>(blocks));
  if (a.plan.select) launch_dense_streams_select_shape(a, g, t0, t1, st);
  else launch_dense_streams_shape(a, g, t0, t1, st);
}
#endif

}  // namespace rejit_amd
