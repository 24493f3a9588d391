// rejit_amd/csrc/verify_lds.hip -- the verify tails of the fast-forward scans for automata of <= 128 positions, built
// for LATENCY: a call's hits are few (a thousand over 5 GB), so what the caller waits for is one hit's chain of
// dependent memory trips, not throughput.
//
// Replaces, for lane-sized automata, verify_floating_in_regions / verify_behind_in_regions (kernels.hip), i.e. the
// reference's NFA loop run from a fast-forward hit (GenerateMatchDirection, src/x64/codegen-x64.cc:535-677; the
// backward pass from the hit, :643-650, :166-189).  Measured on the old kernels (tools/verify_trace.py, one hit):
//   tables staged 3.4 us -> count read 6.2 -> hit read 6.7 -> text read 13.7 -> three walks of 8 + 4 + 12 steps: 29.9 us,
// 0.6-0.9 us per automaton step with the tables already in LDS (flat loads one 32-bit word at a time, each waiting
// for the one before), 35 us for a 42-byte match of the complex benchmark regex.  Here:
//   * the chain is  {count, first hit}  ->  {tables, text window, previous region's last hit}  ->  walks in LDS  ->  stores:
//     the region's first hit is loaded together with its count (speculatively: an empty region's slot holds rubbish
//     that nobody uses), and the text around the hit (1 KiB per wave / 128 B per lane) comes in one trip, with the
//     table copy in flight at the same time;
//   * a workgroup whose four regions are all empty leaves after the count read, before staging anything;
//   * tables are the padded blob of lds_walk.h (NQ 64-bit words per row, rows by position): a step is one ds_read per
//     live non-linear position (two per round) plus one for the class row;
//   * floating windows, `select`: the wave applies the left-most-longest rule to its own candidates (they come out
//     in begin order), so that offsets_gather_check usually finds them "already the result" and the four selection
//     launches are not needed.  Proof that this is the global selection whenever that check passes: the local rule
//     always takes the region's first candidate a0; the check says a0 begins at or after every earlier end, i.e. at
//     or after the global rule's `cur` on entering the region, so the global rule takes a0 too (it is the first
//     candidate it sees there) and both continue identically.  When the check fails the engine repeats this launch
//     with select = false and runs the general selection on all candidates.
#include <hip/hip_runtime.h>

#include "trace_stamp.h"
RJ_TRACE_EXPORT(rj_debug_trace_lds)

#include "device_program.h"
#include "kernels.h"
#include "lds_walk.h"

namespace rejit_amd {

namespace {

constexpr int kWave = 64;
constexpr uint32_t kWinBytes = 1024;      // floating: text window of a wave (64 lanes x 16 B)
constexpr uint32_t kLaneWin = 128;        // behind: text window of a lane ...
constexpr uint32_t kLaneWinStride = 144;  // ... at this stride (16-byte aligned slots, 8 banks apart)

__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x) & (kWave - 1); }

// 16 bytes of the text at `at` (16-byte aligned); bytes at or beyond n read as 0.  The partial block at the end of
// the text is out of line: it is needed once per text at most, and inlined at every use it was most of the code.
__device__ __noinline__ uint4 load16_tail(const uint8_t* text, uint64_t n, uint64_t at) {
  uint32_t w[4] = {0u, 0u, 0u, 0u};
  for (uint32_t k = 0; k < 16 && at + k < n; k++) w[k >> 2] |= static_cast<uint32_t>(text[at + k]) << (8 * (k & 3));
  return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ uint4 load16(const uint8_t* text, uint64_t n, uint64_t at) {
  if (at + 16 <= n) return *reinterpret_cast<const uint4*>(text + at);
  return load16_tail(text, n, at);
}

// The text as the walkers read it: a window in LDS, and beyond it 16 bytes in registers -- a walk that leaves the
// window (a match of several hundred bytes) still makes one trip per 16 steps.
struct WinText {
  const uint8_t* win;
  uint64_t base;
  uint32_t len;
  const uint8_t* text;
  uint64_t n;
  mutable uint64_t far_base, far_lo, far_hi;
  __device__ WinText(const uint8_t* w, uint64_t b, uint32_t l, const uint8_t* t, uint64_t tn)
      : win(w), base(b), len(l), text(t), n(tn), far_base(~0ull), far_lo(0), far_hi(0) {}
  __device__ __forceinline__ uint8_t operator[](uint64_t p) const {
    const uint64_t d = p - base;
    if (d < len) return win[d];
    const uint64_t b = p & ~15ull;
    if (b != far_base) {
      far_base = b;
      const uint4 v = load16(text, n, b);
      far_lo = (static_cast<uint64_t>(v.y) << 32) | v.x;
      far_hi = (static_cast<uint64_t>(v.w) << 32) | v.z;
    }
    // (an arithmetic select, see RjCachedText)
    const uint64_t m = 0ull - ((p >> 3) & 1ull);
    return static_cast<uint8_t>(((far_lo & ~m) | (far_hi & m)) >> (8 * (p & 7)));
  }
};

__device__ __forceinline__ void stage16(uint8_t* dst, const uint8_t* text, uint64_t n, uint64_t at) {
  *reinterpret_cast<uint4*>(dst) = load16(text, n, at);
}

// a lane's window of kLaneWin bytes: the loads of the common case are unconditional, so that they leave together
__device__ __forceinline__ void stage_lane_window(uint8_t* dst, const uint8_t* text, uint64_t n, uint64_t at) {
  if (at + kLaneWin <= n) {
    uint4 v[kLaneWin / 16];
#pragma unroll
    for (uint32_t k = 0; k < kLaneWin / 16; k++) v[k] = *reinterpret_cast<const uint4*>(text + at + 16 * k);
#pragma unroll
    for (uint32_t k = 0; k < kLaneWin / 16; k++) *reinterpret_cast<uint4*>(dst + 16 * k) = v[k];
  } else {
#pragma unroll 1
    for (uint32_t k = 0; k < kLaneWin / 16; k++) stage16(dst + 16 * k, text, n, at + 16 * k);
  }
}

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void copy_blob(uint64_t* dst, const uint64_t* src, uint32_t words) {
  const uint4* s4 = reinterpret_cast<const uint4*>(src);
  uint4* d4 = reinterpret_cast<uint4*>(dst);
  for (uint32_t i = threadIdx.x; i < words / 2; i += blockDim.x) d4[i] = s4[i];
}

__device__ __forceinline__ uint64_t window_base(uint64_t lo) { return lo >= 16 ? (lo - 16) & ~15ull : 0ull; }

// Floating windows: a hit at w makes every s in [w - float_max, w - float_min] a candidate start.  A wave per
// region, a lane per start; the start ranges of consecutive hits are clipped against each other (also against the
// last hit of the region before) so that every start is verified once and the survivors come out sorted by begin.
template <int NQ, bool CTX, bool SELECT>
__global__ __launch_bounds__(256) void verify_floating_lds(VerifyParams a, WalkDesc d, const uint32_t* hit_counts, uint32_t* valid_counts,
                                                           uint64_t* region_begins, uint64_t* region_ends, uint32_t float_min) {
  extern __shared__ uint64_t lds[];
  const int wave = static_cast<int>(threadIdx.x) >> 6, sub = lane_id();
  uint8_t* win = reinterpret_cast<uint8_t*>(lds + d.words) + static_cast<uint32_t>(wave) * kWinBytes;
  const uint64_t r = static_cast<uint64_t>(blockIdx.x) * 4 + static_cast<uint64_t>(wave);
  const bool mine = r < a.n_regions;
  if (blockIdx.x == 0 && threadIdx.x == 0) RJ_STAMP(0);
  // trip 1: the count, the neighbour's count, and (speculatively) the region's first hit
  const uint32_t raw = mine ? hit_counts[r] : 0u;
  const uint32_t prev_raw = mine && r > 0 ? hit_counts[r - 1] : 0u;
  const uint64_t* region = a.hits + r * a.region_cap;
  uint64_t w = mine ? region[0] : 0ull;
  const uint32_t cnt = raw < a.region_cap ? raw : a.region_cap;
  if (!__syncthreads_or(cnt != 0)) {  // nothing to verify in these four regions
    if (mine && sub == 0) valid_counts[r] = 0;
    return;
  }
  if (cnt != 0 && sub == 0) RJ_STAMP(2);
  // trip 2: tables, the previous region's last hit, the text around the first hit
  uint64_t prev_w = 0;
  if (prev_raw != 0) prev_w = a.hits[(r - 1) * a.region_cap + (prev_raw < a.region_cap ? prev_raw : a.region_cap) - 1];
  uint64_t wbase = 0;
  if (cnt != 0) {
    wbase = window_base(w >= a.float_max ? w - a.float_max : 0);
    stage16(win + 16 * sub, a.text, a.n, wbase + 16 * static_cast<uint64_t>(sub));
  }
  copy_blob(lds, d.blob, d.words);
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) RJ_STAMP(1);
  if (cnt == 0) {
    if (mine && sub == 0) valid_counts[r] = 0;
    return;
  }
  if (raw > a.region_cap && sub == 0) {  // the host grows the regions and runs again
    a.counters[kCntOverflow] = 1;
    atomicMax(&a.counters[kCntMaxRegion], static_cast<unsigned long long>(raw));
  }
  const WalkTab<NQ> T = lw_point<NQ>(lds, d.n_ctx, d.n_pos, d.nullable, d.max_walk);
  // first start not yet covered by an earlier hit.  Start ranges are at most 256 wide (lowering.cc: plan_floating)
  // and a region spans >= 1 KiB, so of all earlier hits only the last one of the region right before can reach
  // into this region's ranges.
  uint64_t next_lo = a.sb;
  if (prev_raw != 0 && prev_w >= float_min && prev_w - float_min + 1 > next_lo) next_lo = prev_w - float_min + 1;
  uint64_t* begins = region_begins + r * a.region_cap;
  uint64_t* ends = region_ends + r * a.region_cap;
  uint32_t kept = 0;
  uint64_t cur = 0;  // SELECT: end of the last match taken in this region
  for (uint32_t i = 0; i < cnt; i++) {
    if (i != 0) {
      w = region[i];
      wave_lds_fence();  // every lane is done with the previous window
      wbase = window_base(w >= a.float_max ? w - a.float_max : 0);
      stage16(win + 16 * sub, a.text, a.n, wbase + 16 * static_cast<uint64_t>(sub));
    }
    wave_lds_fence();
    if (sub == 0) RJ_STAMP(3);
    if (w < float_min) continue;
    const uint64_t hi = w - float_min;                     // last start of this hit
    uint64_t lo = w >= a.float_max ? w - a.float_max : 0;  // first
    if (lo < next_lo) lo = next_lo;
    const uint64_t avail = a.n - wbase;
    const WinText t(win, wbase, avail < kWinBytes ? static_cast<uint32_t>(avail) : kWinBytes, a.text, a.n);
    for (uint64_t base = lo; base <= hi; base += kWave) {
      const uint64_t s = base + static_cast<uint64_t>(sub);
      uint64_t e = 0;
      bool overrun = false;
      const bool found = s <= hi && s >= a.sb && s < a.se && lw_longest<NQ, CTX>(T, t, a.n, s, &e, &overrun, a.counters + kCntOverrun);
      if (overrun) a.counters[kCntOverrun] = 1;
      if (found) RJ_STAMP(7);
      uint64_t took = __ballot(found);
      if (SELECT) {
        // the left-most-longest rule over this round's candidates, in begin order (matches here are never empty:
        // they contain the window's literal)
        uint64_t m = took;
        took = 0;
        while (m) {
          const int l = __builtin_ctzll(m);
          m &= m - 1;
          const uint64_t sl = base + static_cast<uint64_t>(l);
          if (sl < cur) continue;
          took |= 1ull << l;
          cur = (static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(e >> 32), l))) << 32) |
                static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(e), l));
        }
      }
      const uint32_t pos = kept + __popcll(took & ((1ull << sub) - 1ull));
      if (((took >> sub) & 1ull) && pos < a.region_cap) {
        begins[pos] = s;
        ends[pos] = e;
        RJ_STAMP(9);
      }
      kept += __popcll(took);
    }
    if (hi + 1 > next_lo) next_lo = hi + 1;
  }
  if (kept > a.region_cap && sub == 0) {  // more candidates than the region holds: grow and run again
    a.counters[kCntOverflow] = 1;
    atomicMax(&a.counters[kCntMaxRegion], static_cast<unsigned long long>(kept));
  }
  if (sub == 0) {
    valid_counts[r] = kept < a.region_cap ? kept : a.region_cap;
    RJ_STAMP(10);
  }
}

// Windows behind an unbounded prefix: a wave per region, a lane per hit, the per-hit procedure of behind_walk.h on
// the padded tables (forward from the cut, backwards to the left-most start, forward to the longest end);
// survivors compacted in place like verify_in_regions.
template <int NQ, bool CTX>
__global__ __launch_bounds__(256) void verify_behind_lds(VerifyParams a, DevProgram P, WalkDesc d, const uint32_t* hit_counts,
                                                         uint32_t* valid_counts, uint64_t* region_ends) {
  extern __shared__ uint64_t lds[];
  const int wave = static_cast<int>(threadIdx.x) >> 6, sub = lane_id();
  uint8_t* win = reinterpret_cast<uint8_t*>(lds + 2 * d.words) + (static_cast<uint32_t>(wave) * kWave + static_cast<uint32_t>(sub)) * kLaneWinStride;
  const uint64_t r = static_cast<uint64_t>(blockIdx.x) * 4 + static_cast<uint64_t>(wave);
  const bool mine = r < a.n_regions;
  if (blockIdx.x == 0 && threadIdx.x == 0) RJ_STAMP(0);
  const uint32_t raw = mine ? hit_counts[r] : 0u;
  uint64_t* region = a.hits + r * a.region_cap;
  // (speculative: slot `sub` of an empty or shorter region holds rubbish that nobody uses)
  uint64_t w = mine && static_cast<uint32_t>(sub) < a.region_cap ? region[sub] : 0ull;
  const uint32_t cnt = raw < a.region_cap ? raw : a.region_cap;
  if (!__syncthreads_or(cnt != 0)) {
    if (mine && sub == 0) valid_counts[r] = 0;
    return;
  }
  if (static_cast<uint32_t>(sub) < cnt) RJ_STAMP(2);
  uint64_t wbase = 0;
  if (static_cast<uint32_t>(sub) < cnt) {
    wbase = w >= 64 ? (w - 64) & ~15ull : 0ull;
    stage_lane_window(win, a.text, a.n, wbase);
  }
  copy_blob(lds, d.blob, d.words);
  copy_blob(lds + d.words, d.rev_blob, d.words);
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) RJ_STAMP(1);
  if (cnt == 0) {
    if (mine && sub == 0) valid_counts[r] = 0;
    return;
  }
  if (raw > a.region_cap && sub == 0) {
    a.counters[kCntOverflow] = 1;
    atomicMax(&a.counters[kCntMaxRegion], static_cast<unsigned long long>(raw));
  }
  const WalkTab<NQ> F = lw_point<NQ>(lds, d.n_ctx, d.n_pos, d.nullable, d.max_walk);
  const WalkTab<NQ> R = lw_point<NQ>(lds + d.words, d.n_ctx, d.n_pos, d.nullable, d.max_walk);
  uint64_t* ends = region_ends + r * a.region_cap;
  uint32_t kept = 0;
  for (uint32_t base = 0; base < cnt; base += kWave) {
    const uint32_t k = base + static_cast<uint32_t>(sub);
    if (base != 0 && k < cnt) {
      w = region[k];
      wbase = w >= 64 ? (w - 64) & ~15ull : 0ull;
      stage_lane_window(win, a.text, a.n, wbase);
    }
    if (k < cnt) RJ_STAMP(3);
    const uint64_t avail = a.n - wbase;
    const WinText t(win, wbase, avail < kLaneWin ? static_cast<uint32_t>(avail) : kLaneWin, a.text, a.n);
    uint64_t b = 0, e = 0;
    bool overrun = false;
    const bool found = k < cnt && *static_cast<const volatile unsigned long long*>(a.counters + kCntOverrun) == 0 &&
                       lw_behind_candidate<NQ, CTX>(P, F, R, t, a.n, w, &b, &e, &overrun, a.counters + kCntOverrun) && b >= a.sb && b < a.se;
    if (overrun) a.counters[kCntOverrun] = 1;
    const uint64_t took = __ballot(found);
    const uint32_t pos = kept + __popcll(took & ((1ull << sub) - 1ull));
    if (found) {  // pos <= k, and every lane of the wave has read its hit already
      region[pos] = b;
      ends[pos] = e;
      RJ_STAMP(9);
    }
    kept += __popcll(took);
  }
  if (sub == 0) {
    valid_counts[r] = kept;
    RJ_STAMP(10);
  }
}

constexpr size_t kLdsLimit = 64 * 1024;

}  // namespace

template <int NQ, bool CTX>
static void launch_floating(bool select, dim3 g, size_t lds, hipStream_t st, const VerifyParams& a, const WalkDesc& d, const uint32_t* hit_counts,
                            uint32_t* valid_counts, uint64_t* region_begins, uint64_t* region_ends, uint32_t float_min) {
  if (select) hipLaunchKernelGGL((verify_floating_lds<NQ, CTX, true>), g, dim3(256), lds, st, a, d, hit_counts, valid_counts, region_begins, region_ends, float_min);
  else hipLaunchKernelGGL((verify_floating_lds<NQ, CTX, false>), g, dim3(256), lds, st, a, d, hit_counts, valid_counts, region_begins, region_ends, float_min);
}

bool launch_verify_floating_lds(const VerifyParams& a, const DevProgram& P, const WalkDesc& d, const uint32_t* hit_counts,
                                uint32_t* valid_counts, uint64_t* region_begins, uint64_t* region_ends, bool select, hipStream_t st) {
  const size_t lds = static_cast<size_t>(d.words) * 8 + 4 * kWinBytes;
  if (d.blob == nullptr || lds > kLdsLimit || (d.nq != 1 && d.nq != 2)) return false;
  const unsigned blocks = (a.n_regions + 3) / 4 > 0 ? (a.n_regions + 3) / 4 : 1;  // a wave per region
  const uint32_t float_min = P.float_max + 1 - P.float_range;
  const dim3 g(blocks);
  const bool ctx = d.n_ctx > 1;
  if (d.nq == 1 && !ctx) launch_floating<1, false>(select, g, lds, st, a, d, hit_counts, valid_counts, region_begins, region_ends, float_min);
  else if (d.nq == 1) launch_floating<1, true>(select, g, lds, st, a, d, hit_counts, valid_counts, region_begins, region_ends, float_min);
  else if (!ctx) launch_floating<2, false>(select, g, lds, st, a, d, hit_counts, valid_counts, region_begins, region_ends, float_min);
  else launch_floating<2, true>(select, g, lds, st, a, d, hit_counts, valid_counts, region_begins, region_ends, float_min);
  return true;
}

bool launch_verify_behind_lds(const VerifyParams& a, const DevProgram& P, const WalkDesc& d, const uint32_t* hit_counts,
                              uint32_t* valid_counts, uint64_t* region_ends, hipStream_t st) {
  const size_t lds = static_cast<size_t>(d.words) * 16 + 4 * kWave * kLaneWinStride;
  if (d.blob == nullptr || d.rev_blob == nullptr || lds > kLdsLimit || (d.nq != 1 && d.nq != 2)) return false;
  const unsigned blocks = (a.n_regions + 3) / 4 > 0 ? (a.n_regions + 3) / 4 : 1;
  const dim3 g(blocks), b(256);
  const bool ctx = d.n_ctx > 1;
  if (d.nq == 1 && !ctx) hipLaunchKernelGGL((verify_behind_lds<1, false>), g, b, lds, st, a, P, d, hit_counts, valid_counts, region_ends);
  else if (d.nq == 1) hipLaunchKernelGGL((verify_behind_lds<1, true>), g, b, lds, st, a, P, d, hit_counts, valid_counts, region_ends);
  else if (!ctx) hipLaunchKernelGGL((verify_behind_lds<2, false>), g, b, lds, st, a, P, d, hit_counts, valid_counts, region_ends);
  else hipLaunchKernelGGL((verify_behind_lds<2, true>), g, b, lds, st, a, P, d, hit_counts, valid_counts, region_ends);
  return true;
}

}  // namespace rejit_amd
