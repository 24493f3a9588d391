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
//   * a workgroup looks at 64 regions (one count and one first hit per lane of its first wave) and leaves at once when
//     all are empty -- before staging anything; its four waves share the regions that have hits.  (A wave per region
//     and 4 regions per workgroup meant 5000 workgroups for a 5 GB text: the workgroup with the hit was dispatched
//     in a second round, 4-8 us after the first);
//   * the walks run the plain steps in a loop with one exit (lds_walk.h): a lone wave pays every instruction in full,
//     and the general iteration spent most of its ~150 instructions on the masks of its five exits;
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
constexpr uint32_t kRegionsPerBlock = 32;  // (64: a fifth of the workgroups of a 1000-hit run had more regions with hits than waves; 16: no faster)
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

// how the window classes of lds_walk.h fetch 16 bytes of the text into LDS
struct DeviceLoader {
  static __device__ __forceinline__ void block16(uint8_t* dst, const uint8_t* text, uint64_t n, uint64_t at) { stage16(dst, text, n, at); }
};
using WaveWinText = WaveWindowText<DeviceLoader>;
using LaneWinText = LaneWindowText<DeviceLoader, kLaneWin>;

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The table copy in two halves: the loads leave with the workgroup's first trip (before it is known whether any of
// its regions has a hit -- the tables are a few KiB that every workgroup finds in L2), the LDS stores follow once
// one has.  kBlobRegs x 256 lanes x 16 bytes are held in registers, a longer blob's rest is copied the plain way.
constexpr uint32_t kBlobRegs = 3;
struct BlobRegs {
  uint4 v[kBlobRegs];
};
__device__ __forceinline__ BlobRegs blob_fetch(const uint64_t* src, uint32_t words) {
  BlobRegs b;
  const uint4* s4 = reinterpret_cast<const uint4*>(src);
#pragma unroll
  for (uint32_t k = 0; k < kBlobRegs; k++) {
    const uint32_t i = threadIdx.x + k * 256u;
    b.v[k] = i < words / 2 ? s4[i] : make_uint4(0u, 0u, 0u, 0u);
  }
  return b;
}
__device__ __forceinline__ void blob_store(uint64_t* dst, const uint64_t* src, uint32_t words, const BlobRegs& b) {
  const uint4* s4 = reinterpret_cast<const uint4*>(src);
  uint4* d4 = reinterpret_cast<uint4*>(dst);
#pragma unroll
  for (uint32_t k = 0; k < kBlobRegs; k++) {
    const uint32_t i = threadIdx.x + k * 256u;
    if (i < words / 2) d4[i] = b.v[k];
  }
  for (uint32_t i = threadIdx.x + kBlobRegs * 256u; i < words / 2; i += 256u) d4[i] = s4[i];
}

__device__ __forceinline__ uint64_t window_base(uint64_t lo) { return lo >= 16 ? (lo - 16) & ~15ull : 0ull; }

// What a workgroup knows about its 64 regions after the first trip (one region per lane of wave 0).
struct BlockRegions {
  uint32_t raw[kWave];     // hit counts as the scan left them (a region per lane of the first wave)
  uint64_t first[kWave];   // the region's first hit (rubbish when the region is empty)
  uint64_t before[kWave];  // floating: 1 + the last hit of the region right before, 0 when that one is empty
  uint64_t todo;           // bit k: region k has hits
};

// Floating windows: a hit at w makes every s in [w - float_max, w - float_min] a candidate start.  A wave per
// region with hits, a lane per start; the start ranges of consecutive hits are clipped against each other (also
// against the last hit of the region before) so that every start is verified once and the survivors come out
// sorted by begin.
template <int NQ, bool CTX, bool SELECT>
__global__ __launch_bounds__(256) void verify_floating_lds(VerifyParams a, WalkDesc d, const uint32_t* hit_counts, uint32_t* valid_counts,
                                                           uint64_t* region_begins, uint64_t* region_ends, uint32_t float_min) {
  extern __shared__ uint64_t lds[];
  __shared__ BlockRegions br;
  const int wave = static_cast<int>(threadIdx.x) >> 6, sub = lane_id();
  uint8_t* win = reinterpret_cast<uint8_t*>(lds + d.words) + static_cast<uint32_t>(wave) * kWinBytes;
  uint8_t* slot = reinterpret_cast<uint8_t*>(lds + d.words) + 4 * kWinBytes + (static_cast<uint32_t>(wave) * kWave + static_cast<uint32_t>(sub)) * 16;
  const uint64_t r0 = static_cast<uint64_t>(blockIdx.x) * kRegionsPerBlock;
  if (threadIdx.x == 0) RJ_STAMP(0);
  const BlobRegs tables = blob_fetch(d.blob, d.words);
  // trip 1 (wave 0, a region per lane): the count, the count of the region before, and -- speculatively -- the first hit
  bool any = false;
  uint64_t pw = 0;     // wave 0: the last hit of the region before this lane's region ...
  bool has_pw = false; // ... when both have hits (trip 2, in flight across the first barrier)
  if (wave == 0) {
    const uint64_t r = r0 + static_cast<uint64_t>(sub);
    const bool mine = r < a.n_regions && static_cast<uint32_t>(sub) < kRegionsPerBlock;
    const uint32_t raw = mine ? hit_counts[r] : 0u;
    const uint32_t prev_raw = mine && r > 0 ? hit_counts[r - 1] : 0u;
    const uint64_t w0 = mine ? a.hits[r * a.region_cap] : 0ull;
    br.raw[sub] = raw;
    br.first[sub] = w0;
    has_pw = raw != 0 && prev_raw != 0;
    if (has_pw) pw = a.hits[(r - 1) * a.region_cap + (prev_raw < a.region_cap ? prev_raw : a.region_cap) - 1];
    if (mine && raw == 0) valid_counts[r] = 0;
    any = raw != 0;
    const uint64_t mask = __ballot(any);
    if (sub == 0) br.todo = mask;
  }
  if (!__syncthreads_or(any)) return;  // nothing to verify in these regions
  blob_store(lds, d.blob, d.words, tables);
  const uint64_t todo = br.todo;  // the regions with hits, shared out among the four waves
  // this wave's first region: its text window can leave together with the table copy
  uint32_t mine_k = 0;
  {
    uint64_t m = todo;
    for (int skip = 0; skip < wave && m != 0; skip++) m &= m - 1;
    mine_k = m != 0 ? static_cast<uint32_t>(__builtin_ctzll(m)) : kWave;
  }
  uint64_t w = 0, wbase = 0;
  if (mine_k < kWave) {
    w = br.first[mine_k];
    wbase = window_base(w >= a.float_max ? w - a.float_max : 0);
    stage16(win + 16 * sub, a.text, a.n, wbase + 16 * static_cast<uint64_t>(sub));
  }
  if (wave == 0) br.before[sub] = has_pw ? pw + 1 : 0;  // (+ 1: 0 means none)
  __syncthreads();
  if (threadIdx.x == 0) RJ_STAMP(1);
  const WalkTab<NQ> T = lw_point<NQ>(lds, d.n_ctx, d.n_pos, d.nullable, d.max_walk);
  uint32_t nth = static_cast<uint32_t>(wave);  // this wave takes the nth, (nth + 4)th, ... region with hits
  for (uint64_t m = todo; m != 0; m &= m - 1, nth--) {
    if (nth != 0) continue;
    nth = 4;
    const uint32_t k = static_cast<uint32_t>(__builtin_ctzll(m));
    const uint64_t r = r0 + k;
    const uint32_t raw = br.raw[k];
    const uint32_t cnt = raw < a.region_cap ? raw : a.region_cap;
    if (sub == 0) RJ_STAMP(2);
    if (raw > a.region_cap && sub == 0) {  // the host grows the regions and runs again
      a.counters[kCntOverflow] = 1;
      atomicMax(&a.counters[kCntMaxRegion], static_cast<unsigned long long>(raw));
    }
    // first start not yet covered by an earlier hit.  Start ranges are at most 256 wide (lowering.cc:
    // plan_floating) and a region spans >= 1 KiB, so of all earlier hits only the last one of the region right
    // before can reach into this region's ranges.
    uint64_t next_lo = a.sb;
    {
      const uint64_t pw1 = br.before[k];
      if (pw1 != 0 && pw1 - 1 >= float_min && pw1 - float_min > next_lo) next_lo = pw1 - float_min;
    }
    const uint64_t* region = a.hits + r * a.region_cap;
    uint64_t* begins = region_begins + r * a.region_cap;
    uint64_t* ends = region_ends + r * a.region_cap;
    uint32_t kept = 0;
    uint64_t cur = 0;  // SELECT: end of the last match taken in this region
    for (uint32_t i = 0; i < cnt; i++) {
      if (i != 0 || k != mine_k) {
        w = i == 0 ? br.first[k] : region[i];
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
      const WaveWinText t(win, slot, wbase, avail < kWinBytes ? static_cast<uint32_t>(avail) : kWinBytes, a.text, a.n);
      for (uint64_t base = lo; base <= hi; base += kWave) {
        const uint64_t s = base + static_cast<uint64_t>(sub);
        uint64_t e = 0;
        bool overrun = false;
        const bool found = s <= hi && s >= a.sb && s < a.se && lw_longest<NQ, CTX>(T, t, a.n, s, &e, &overrun, a.counters + kCntOverrun);
        if (overrun) a.counters[kCntOverrun] = 1;
        if (found) RJ_STAMP(7);
        uint64_t took = __ballot(found);
        if (SELECT) {
          // the left-most-longest rule over this round's candidates, in begin order (matches here are never
          // empty: they contain the window's literal)
          uint64_t mm = took;
          took = 0;
          while (mm) {
            const int l = __builtin_ctzll(mm);
            mm &= mm - 1;
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
}

// Windows behind an unbounded prefix: a wave per region with hits, a lane per hit, the per-hit procedure of
// behind_walk.h on the padded tables (forward from the cut, backwards to the left-most start, forward to the
// longest end); survivors compacted in place like verify_in_regions.
template <int NQ, bool CTX>
__global__ __launch_bounds__(256) void verify_behind_lds(VerifyParams a, DevProgram P, WalkDesc d, const uint32_t* hit_counts,
                                                         uint32_t* valid_counts, uint64_t* region_ends) {
  extern __shared__ uint64_t lds[];
  __shared__ BlockRegions br;
  const int wave = static_cast<int>(threadIdx.x) >> 6, sub = lane_id();
  uint8_t* win = reinterpret_cast<uint8_t*>(lds + 2 * d.words) + (static_cast<uint32_t>(wave) * kWave + static_cast<uint32_t>(sub)) * kLaneWinStride;
  const uint64_t r0 = static_cast<uint64_t>(blockIdx.x) * kRegionsPerBlock;
  if (threadIdx.x == 0) RJ_STAMP(0);
  const BlobRegs fwd_tables = blob_fetch(d.blob, d.words), rev_tables = blob_fetch(d.rev_blob, d.words);
  bool any = false;
  if (wave == 0) {
    const uint64_t r = r0 + static_cast<uint64_t>(sub);
    const bool mine = r < a.n_regions && static_cast<uint32_t>(sub) < kRegionsPerBlock;
    const uint32_t raw = mine ? hit_counts[r] : 0u;
    const uint64_t w0 = mine ? a.hits[r * a.region_cap] : 0ull;
    br.raw[sub] = raw;
    br.first[sub] = w0;
    if (mine && raw == 0) valid_counts[r] = 0;
    any = raw != 0;
    const uint64_t mask = __ballot(any);
    if (sub == 0) br.todo = mask;
  }
  if (!__syncthreads_or(any)) return;
  blob_store(lds, d.blob, d.words, fwd_tables);
  blob_store(lds + d.words, d.rev_blob, d.words, rev_tables);
  const uint64_t todo = br.todo;
  __syncthreads();
  if (threadIdx.x == 0) RJ_STAMP(1);
  const WalkTab<NQ> F = lw_point<NQ>(lds, d.n_ctx, d.n_pos, d.nullable, d.max_walk);
  const WalkTab<NQ> R = lw_point<NQ>(lds + d.words, d.n_ctx, d.n_pos, d.nullable, d.max_walk);
  uint32_t nth = static_cast<uint32_t>(wave);
  for (uint64_t m = todo; m != 0; m &= m - 1, nth--) {
    if (nth != 0) continue;
    nth = 4;
    const uint32_t kr = static_cast<uint32_t>(__builtin_ctzll(m));
    const uint64_t r = r0 + kr;
    const uint32_t raw = br.raw[kr];
    const uint32_t cnt = raw < a.region_cap ? raw : a.region_cap;
    if (raw > a.region_cap && sub == 0) {
      a.counters[kCntOverflow] = 1;
      atomicMax(&a.counters[kCntMaxRegion], static_cast<unsigned long long>(raw));
    }
    uint64_t* region = a.hits + r * a.region_cap;
    uint64_t* ends = region_ends + r * a.region_cap;
    uint32_t kept = 0;
    for (uint32_t base = 0; base < cnt; base += kWave) {
      const uint32_t k = base + static_cast<uint32_t>(sub);
      uint64_t w = 0, wbase = 0;
      if (k < cnt) {
        RJ_STAMP(2);
        w = k == 0 ? br.first[kr] : region[k];
        wbase = w >= 64 ? (w - 64) & ~15ull : 0ull;
        stage_lane_window(win, a.text, a.n, wbase);
        RJ_STAMP(3);
      }
      const uint64_t avail = a.n - wbase;
      const LaneWinText t(win, wbase, avail < kLaneWin ? static_cast<uint32_t>(avail) : kLaneWin, a.text, a.n);
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
  const size_t lds = static_cast<size_t>(d.words) * 8 + 4 * kWinBytes + 4 * kWave * 16;
  if (d.blob == nullptr || lds + sizeof(BlockRegions) > kLdsLimit || (d.nq != 1 && d.nq != 2)) return false;
  const unsigned blocks = (a.n_regions + kRegionsPerBlock - 1) / kRegionsPerBlock > 0 ? (a.n_regions + kRegionsPerBlock - 1) / kRegionsPerBlock : 1;
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
  if (d.blob == nullptr || d.rev_blob == nullptr || lds + sizeof(BlockRegions) > kLdsLimit || (d.nq != 1 && d.nq != 2)) return false;
  const unsigned blocks = (a.n_regions + kRegionsPerBlock - 1) / kRegionsPerBlock > 0 ? (a.n_regions + kRegionsPerBlock - 1) / kRegionsPerBlock : 1;
  const dim3 g(blocks), b(256);
  const bool ctx = d.n_ctx > 1;
  if (d.nq == 1 && !ctx) hipLaunchKernelGGL((verify_behind_lds<1, false>), g, b, lds, st, a, P, d, hit_counts, valid_counts, region_ends);
  else if (d.nq == 1) hipLaunchKernelGGL((verify_behind_lds<1, true>), g, b, lds, st, a, P, d, hit_counts, valid_counts, region_ends);
  else if (!ctx) hipLaunchKernelGGL((verify_behind_lds<2, false>), g, b, lds, st, a, P, d, hit_counts, valid_counts, region_ends);
  else hipLaunchKernelGGL((verify_behind_lds<2, true>), g, b, lds, st, a, P, d, hit_counts, valid_counts, region_ends);
  return true;
}

}  // namespace rejit_amd
