// rejit_amd/csrc/kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the hot path.
//
// What the reference JITs per pattern (src/x64/codegen-x64.cc) is here a fixed family of
// kernels that interpret the lowered program (device_program.h).  The common pipeline:
//
//   scan_windows<K,...>    the "fast-forward" scan (FastForwardGen::VisitSingleMultipleChar /
//                          ::Generate multi-literal branch, codegen-x64.cc:1102-1403): every wave
//                          streams a contiguous span of the text (one coalesced 16-byte load per
//                          lane, three chunks in flight), builds the lane's 16 unaligned windows
//                          with v_alignbyte_b32 and compares them with <= 8 window constants held
//                          in SGPRs, VALU only (xor/and/min, one v_cmp + ballot per chunk); exact,
//                          two-level or nibble-packed form.  Hit offsets go to the wave's own
//                          region of the hit list: no atomics, sorted by construction.
//   scan_windows_fused     the same for several patterns in one pass over the text (rj_multi mode 0: a shared
//                          nibble-distance prefilter, per-pattern classification on hit lanes only);
//   scan_windows_train     every pattern's own scan in ONE launch, a wave going through its span pattern
//                          after pattern (rj_multi mode 1, the bench headline).
//   verify_in_regions<NQ>  the NFA inner loop (GenerateMatchDirection + GenerateTransitions,
//                          codegen-x64.cc:535-677): longest match from every hit, automaton state
//                          in NQ 64-bit registers per lane, survivors compacted inside their region
//                          (bounded patterns of <= 16 bytes: rj_lane_longest_short, the candidate's text in
//                          two loads); verify_floating_in_regions for floating windows;
//                          verify_behind_in_regions for windows behind an unbounded prefix (behind_walk.h:
//                          forward check, REVERSE automaton back to the left-most start, forward longest).
//   offsets_gather_check   region offsets + gather + "the candidates already are the result" check
//                          (MatchAllAppendFilter + the non-overlap rule, src/codegen.cc:36-86,
//                          codegen-x64.cc:448-460) in one multi-workgroup launch.
//   scan_dense_walk<NW,CTX,PD> dense mode (no fast-forward window; GenerateMatchDirection seeding
//                          every position, codegen-x64.cc:544-554) in ONE kernel: candidate masks,
//                          pre-steps in registers (PD > 0: lane-packed, four starts per register, class rows
//                          by byte-parallel range tests -- dense_swar.h), chunks whose candidates are all
//                          decided emit directly, the others go through persistent walker lanes and an
//                          in-region compaction.
//   match_small<NQ>        texts of a few KiB: the whole MatchAll in one workgroup, one launch.
//
// and around it: finalize_small / chain_* & co (the selection when candidates overlap: blocked chains, a wave
// per block, staged in LDS), verify_wave + region_offsets + mark/compact (automata of more than 128
// positions), match_full (kMatchFull, codegen-x64.cc:162-164), exact_sequential (the reference's whole loop
// on one lane: automata too wide for exact_replay.hip), replace_gather (rejit::Replace, src/rejit.cc:97-112).
// The linear-time carry scan lives in carry_kernels.hip, the reference-exact replay in exact_replay.hip.
//
// No MFMA: there is no contraction anywhere on this path (integer compares on a byte stream);
// the roofline is HBM read bandwidth.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "trace_stamp.h"
RJ_TRACE_EXPORT(rj_debug_trace)


#include "behind_walk.h"
#include "dense_swar.h"
#include "device_program.h"
#include "kernel_util.h"
#include "kernels.h"

namespace rejit_amd {

// (the shared wave-level helpers: kernel_util.h)

// ---------------------------------------------------------------------------------------
// Region bookkeeping: offsets[r] = sum of min(count, cap) of the regions before r (so hit i of
// the run lives in region upper_bound(offsets, i) - 1), the total, and the fullest region.
// One workgroup; n_regions <= 64 Ki.
template <int PER>
__global__ __launch_bounds__(1024) void region_offsets(const uint32_t* counts, uint32_t n_regions, uint32_t cap,
                                                       uint64_t* offsets, unsigned long long* counters) {
  __shared__ uint64_t wave_sum[16];
  __shared__ uint32_t wave_max[16];
  const int lane = lane_id(), wv = threadIdx.x >> 6;
  const uint32_t lo = threadIdx.x * PER;
  // the thread's slice lives in registers: PER independent loads in flight instead of a
  // dependent chain (the slice loop was the whole 20-50 us of this kernel)
  uint32_t pre[PER];
#pragma unroll
  for (int k = 0; k < PER; k++) pre[k] = (lo + k < n_regions) ? counts[lo + k] : 0u;
  uint64_t sum = 0;
  uint32_t mx = 0;
#pragma unroll
  for (int k = 0; k < PER; k++) {
    mx = pre[k] > mx ? pre[k] : mx;
    sum += pre[k] < cap ? pre[k] : cap;
  }
  // inclusive scan inside the wave, then across the 16 waves
  uint64_t inc = sum;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const uint64_t v = __shfl_up(inc, o);
    if (lane >= o) inc += v;
    const uint32_t m2 = __shfl_xor(mx, o);
    mx = m2 > mx ? m2 : mx;
  }
  if (lane == kWave - 1) {
    wave_sum[wv] = inc;
    wave_max[wv] = mx;
  }
  __syncthreads();
  uint64_t before = 0, total = 0;
  uint32_t maxc = 0;
  for (int w = 0; w < 16; w++) {
    if (w < wv) before += wave_sum[w];
    total += wave_sum[w];
    maxc = wave_max[w] > maxc ? wave_max[w] : maxc;
  }
  uint64_t run = before + inc - sum;  // exclusive prefix of this thread's slice
#pragma unroll
  for (int k = 0; k < PER; k++) {
    if (lo + k < n_regions) offsets[lo + k] = run;
    run += pre[k] < cap ? pre[k] : cap;
  }
  if (threadIdx.x == 0) {
    offsets[n_regions] = total;
    counters[kCntHits] = total;
    counters[kCntMaxRegion] = maxc;
    if (maxc > cap) counters[kCntOverflow] = 1;
  }
}

namespace {
// hit i of the run: region = upper_bound(offsets, i) - 1
__device__ __forceinline__ uint64_t hit_at(const VerifyParams& a, uint64_t i) {
  uint32_t lo = 0, hi = a.n_regions;  // invariant: offsets[lo] <= i < offsets[hi]
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a.offsets[mid] <= i) lo = mid; else hi = mid;
  }
  return a.hits[static_cast<uint64_t>(lo) * a.region_cap + (i - a.offsets[lo])];
}
}  // namespace

// ---------------------------------------------------------------------------------------
// Verify: longest match from every hit; survivors become (begin,end) candidates.
// The result goes to slot i of the candidate arrays (end = kNoMatch when nothing matches at
// that start), so the candidates stay sorted by begin and need no atomic either.
template <int NQ>
__global__ __launch_bounds__(256) void verify_lane(VerifyParams a, DevProgram P) {
  const uint64_t n_slots = a.offsets[a.n_regions] * a.expand;
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n_slots; i += stride) {
    // slot i = (hit i / expand, delta i % expand); fixed-offset windows have expand == 1
    const uint64_t w = hit_at(a, i / a.expand);
    const uint64_t back = static_cast<uint64_t>(a.float_max) - (i % a.expand);
    uint64_t e = 0;
    bool overrun = false, found = false;
    const uint64_t s = w - back;
    if (w >= back && s >= a.sb && s < a.se) found = rj_lane_longest<NQ>(P, a.text, a.n, s, &e, &overrun, a.counters + kCntOverrun);
    if (overrun) a.counters[kCntOverrun] = 1;
    a.cand_begin[i] = s;
    a.cand_end[i] = found ? e : kNoMatch;
  }
}

namespace {

// Automaton state spread over the wave: lane l holds 32-bit words l, l+64, ... (NR of them).
template <int NR>
__device__ bool wave_longest(const DevProgram& P, const uint8_t* t, uint64_t n, uint64_t s, uint64_t* end,
                             bool anchored_full, bool* overrun) {
  const int lane = lane_id();
  const int W = P.n_words;
  const bool ctxed = P.n_ctx > 1;
  int ctx = ctxed ? rj_context(t, n, s) : 0;
  bool found = false;
  if (!anchored_full || s == n) {
    if ((P.nullable >> ctx) & 1u) {
      found = true;
      *end = s;
    }
  }
  if (s >= n || P.n_pos == 0) return found;
  uint32_t S[NR], lin[NR];
  {
    const uint32_t* fr = P.first + ctx * W;
    const uint32_t* cr = P.cls + static_cast<uint32_t>(t[s]) * W;
#pragma unroll
    for (int r = 0; r < NR; r++) {
      const int wi = r * kWave + lane;
      S[r] = wi < W ? (fr[wi] & cr[wi]) : 0u;
      lin[r] = wi < W ? P.linear[wi] : 0u;
    }
  }
  uint64_t p = s + 1;
  for (;;) {
    bool alive = false;
#pragma unroll
    for (int r = 0; r < NR; r++) alive |= S[r] != 0;
    if (__ballot(alive) == 0) break;
    ctx = ctxed ? rj_context(t, n, p) : 0;
    if (!anchored_full || p == n) {
      const uint32_t* lr = P.last + ctx * W;
      bool acc = false;
#pragma unroll
      for (int r = 0; r < NR; r++) {
        const int wi = r * kWave + lane;
        if (wi < W) acc |= (S[r] & lr[wi]) != 0;
      }
      if (__ballot(acc) != 0) {
        found = true;
        *end = p;
      }
    }
    if (p == n) break;
    if (!anchored_full && p - s >= P.max_walk) {
      *overrun = true;
      break;
    }
    uint32_t T[NR];
    uint32_t carry_in = 0;  // bit shifted out of the previous register's lane 63
#pragma unroll
    for (int r = 0; r < NR; r++) {
      const uint32_t x = S[r] & lin[r];
      const uint32_t hi = x >> 31;
      uint32_t up = __shfl_up(hi, 1);
      if (lane == 0) up = carry_in;
      T[r] = (x << 1) | up;
      carry_in = __shfl(hi, kWave - 1);
    }
#pragma unroll
    for (int r = 0; r < NR; r++) {
      uint32_t sp = S[r] & ~lin[r];
      for (;;) {
        const unsigned long long m = __ballot(sp != 0);
        if (m == 0) break;
        const int src = __ffsll(static_cast<long long>(m)) - 1;
        const uint32_t spv = __shfl(sp, src);
        const int b = __ffs(static_cast<int>(spv)) - 1;
        const int pos = (r * kWave + src) * 32 + b;
        const uint32_t* row = P.rows + (static_cast<size_t>(ctx) * P.n_rows + P.row_of[pos]) * W;
#pragma unroll
        for (int r2 = 0; r2 < NR; r2++) {
          const int wi = r2 * kWave + lane;
          if (wi < W) T[r2] |= row[wi];
        }
        if (lane == src) sp &= sp - 1;
      }
    }
    const uint32_t* cr = P.cls + static_cast<uint32_t>(t[p]) * W;
#pragma unroll
    for (int r = 0; r < NR; r++) {
      const int wi = r * kWave + lane;
      S[r] = wi < W ? (T[r] & cr[wi]) : 0u;
    }
    p++;
  }
  return found;
}

}  // namespace

// Dense mode: regions are long (thousands of hits each), so a wave walks whole regions and
// needs no per-hit search for the region (135M hits/GB made that search the whole run time).
template <int NQ>
__global__ __launch_bounds__(256) void verify_lane_regions(VerifyParams a, DevProgram P) {
  const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_waves = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 6;
  const int lane = lane_id();
  for (uint64_t r = wave; r < a.n_regions; r += n_waves) {
    const uint64_t lo = a.offsets[r], hi = a.offsets[r + 1];
    const uint64_t* region = a.hits + r * a.region_cap;
    for (uint64_t k = lane; k < hi - lo; k += kWave) {
      const uint64_t s = region[k];
      uint64_t e = 0;
      bool overrun = false;
      const bool found = rj_lane_longest<NQ>(P, a.text, a.n, s, &e, &overrun, a.counters + kCntOverrun);
      if (overrun) a.counters[kCntOverrun] = 1;
      a.cand_begin[lo + k] = s;
      a.cand_end[lo + k] = found ? e : kNoMatch;
    }
  }
}

// Fast-forward windows with a lane-sized automaton: 16 lanes take one hit region, verify its
// hits and compact the survivors IN PLACE (begins stay in the region, ends go to the same slot
// of region_ends).  The count per region then drives offsets_gather_check, so the candidates need
// no global compaction pass at all.
// Copy the automaton tables into LDS (dynamic shared memory, `lds_words` = 0 when they do not fit)
// and return a descriptor whose table pointers point there: the per-lane walk then reads LDS
// instead of chasing global loads (every step is a chain of dependent table reads).
__device__ __forceinline__ DevProgram stage_tables(const DevProgram& P, uint32_t* tab, uint32_t lds_words) {
  DevProgram Q = P;
  if (lds_words >= P.table_words) {
    for (uint32_t i = threadIdx.x; i < P.table_words; i += blockDim.x) tab[i] = P.first[i];
    const int W = P.n_words, C = P.n_ctx, NP = P.n_pos > 0 ? P.n_pos : 1;
    Q.first = tab;
    Q.last = tab + C * W;
    Q.linear = tab + 2 * C * W;
    Q.row_of = reinterpret_cast<const int32_t*>(tab + 2 * C * W + W);
    Q.rows = tab + 2 * C * W + W + NP;
    Q.cls = tab + 2 * C * W + W + NP + C * P.n_rows * W;
  }
  __syncthreads();
  return Q;
}

// G = lanes per region (a power of two <= 16)
template <int NQ, int G = 16>
__device__ __forceinline__ void verify_in_regions_body(const VerifyParams& a, const DevProgram& P, const uint32_t* hit_counts,
                                                       uint32_t* valid_counts, uint64_t* region_ends) {
  constexpr int kShift = G == 16 ? 4 : G == 8 ? 3 : 2;
  static_assert(G == 16 || G == 8 || G == 4, "lanes per region");
  const uint64_t tid = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const uint64_t n_groups = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> kShift;
  const int lane = lane_id(), sub = lane & (G - 1), shift = lane & (64 - G);
  for (uint64_t r = tid >> kShift; r < a.n_regions; r += n_groups) {
    const uint32_t raw = hit_counts[r];
    const uint32_t cnt = raw < a.region_cap ? raw : a.region_cap;
    if (raw > a.region_cap && sub == 0) {  // the host grows the regions and runs again
      a.counters[kCntOverflow] = 1;
      atomicMax(&a.counters[kCntMaxRegion], static_cast<unsigned long long>(raw));
    }
    uint64_t* region = a.hits + r * a.region_cap;
    uint64_t* ends = region_ends + r * a.region_cap;
    uint32_t kept = 0;
    for (uint32_t base = 0; base < cnt; base += G) {
      const uint32_t k = base + sub;
      const uint64_t w = k < cnt ? region[k] : 0;
      const uint64_t s = w - a.float_max;
      uint64_t e = 0;
      bool overrun = false;
      const bool in_range = k < cnt && w >= a.float_max && s >= a.sb && s < a.se;
      // (short bounded patterns: the candidate's text in two loads, DevProgram::short_max)
      const bool found = in_range && (NQ == 1 && P.short_max != 0 ? rj_lane_longest_short(P, a.text, a.n, s, &e)
                                                                   : rj_lane_longest<NQ>(P, a.text, a.n, s, &e, &overrun, a.counters + kCntOverrun));
      if (overrun) a.counters[kCntOverrun] = 1;
      const uint32_t mine = static_cast<uint32_t>(__ballot(found) >> shift) & ((1u << G) - 1u);
      const uint32_t pos = kept + __popc(mine & ((1u << sub) - 1u));
      if (found) {  // pos <= k, and every lane of the group has read its hit already
        region[pos] = s;
        ends[pos] = e;
      }
      kept += __popc(mine);
    }
    if (sub == 0) valid_counts[r] = kept;
  }
}

template <int NQ>
__global__ __launch_bounds__(256) void verify_in_regions(VerifyParams a, DevProgram P, const uint32_t* hit_counts,
                                                         uint32_t* valid_counts, uint64_t* region_ends) {
  verify_in_regions_body<NQ>(a, P, hit_counts, valid_counts, region_ends);
}

// Floating windows (a hit at w makes every s in [w - float_max, w - float_min] a candidate start):
// the same in-region scheme.  The start ranges of consecutive hits are clipped against each other
// -- also across regions, against the last hit of the nearest non-empty region before -- so every
// start is verified once and the survivors come out sorted by begin; they go to separate arrays
// (one hit can yield several candidates, so compacting in place could overtake unread hits).
template <int NQ>
__global__ __launch_bounds__(256) void verify_floating_in_regions(VerifyParams a, DevProgram P, const uint32_t* hit_counts,
                                                                  uint32_t* valid_counts, uint64_t* region_begins,
                                                                  uint64_t* region_ends, uint32_t float_min,
                                                                  uint32_t lds_words) {
  extern __shared__ uint32_t tab[];
  if (blockIdx.x == 0 && threadIdx.x == 0) RJ_STAMP(0);
  const DevProgram Q = stage_tables(P, tab, lds_words);
  if (blockIdx.x == 0 && threadIdx.x == 0) RJ_STAMP(1);
  const uint64_t tid = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  // a whole wave per region: one hit has up to 256 starts and every walk is a chain of dependent
  // steps, so the width of a round is what bounds the kernel's latency (16 lanes: 126 us at 1000 hits)
  const uint64_t n_groups = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 6;
  const int sub = lane_id();
  for (uint64_t r = tid >> 6; r < a.n_regions; r += n_groups) {
    const uint32_t raw = hit_counts[r];
    const uint32_t cnt = raw < a.region_cap ? raw : a.region_cap;
    if (raw > a.region_cap && sub == 0) {
      a.counters[kCntOverflow] = 1;
      atomicMax(&a.counters[kCntMaxRegion], static_cast<unsigned long long>(raw));
    }
    if (cnt == 0) {
      if (sub == 0) valid_counts[r] = 0;
      continue;
    }
    // first start not yet covered by an earlier hit.  Start ranges are at most 256 wide
    // (lowering.cc: plan_floating) and a region spans >= 1 KiB, so of all earlier hits only the
    // last one of the region right before can reach into this region's ranges.
    uint64_t next_lo = a.sb;
    if (r > 0) {
      const uint32_t c = hit_counts[r - 1];
      if (c != 0) {
        const uint64_t w = a.hits[(r - 1) * a.region_cap + (c < a.region_cap ? c : a.region_cap) - 1];
        if (w >= float_min && w - float_min + 1 > next_lo) next_lo = w - float_min + 1;
      }
    }
    const uint64_t* region = a.hits + r * a.region_cap;
    uint64_t* begins = region_begins + r * a.region_cap;
    uint64_t* ends = region_ends + r * a.region_cap;
    uint32_t kept = 0;
    if (sub == 0) RJ_STAMP(2);
    for (uint32_t i = 0; i < cnt; i++) {
      const uint64_t w = region[i];
      if (sub == 0 && w != 0) RJ_STAMP(3);
      if (w < float_min) continue;
      const uint64_t hi = w - float_min;                       // last start of this hit
      uint64_t lo = w >= a.float_max ? w - a.float_max : 0;    // first
      if (lo < next_lo) lo = next_lo;
      for (uint64_t base = lo; base <= hi; base += kWave) {
        const uint64_t s = base + sub;
        uint64_t e = 0;
        bool overrun = false;
        const RjCachedText ct(a.text, a.n);  // (the walk reads the text 16 bytes at a time, device_program.h)
        const bool found = s <= hi && s >= a.sb && s < a.se && rj_lane_longest<NQ>(Q, ct, a.n, s, &e, &overrun, a.counters + kCntOverrun);
        if (overrun) a.counters[kCntOverrun] = 1;
        if (found) RJ_STAMP(7);
        const uint64_t mine = __ballot(found);
        if (sub == 0) RJ_STAMP(8);
        const uint32_t pos = kept + __popcll(mine & ((1ull << sub) - 1ull));
        if (found && pos < a.region_cap) {
          begins[pos] = s;
          ends[pos] = e;
          RJ_STAMP(9);
        }
        kept += __popcll(mine);
      }
      if (hi + 1 > next_lo) next_lo = hi + 1;
    }
    if (kept > a.region_cap && sub == 0) {  // more candidates than the region holds: grow and run again
      a.counters[kCntOverflow] = 1;
      atomicMax(&a.counters[kCntMaxRegion], static_cast<unsigned long long>(kept));
    }
    if (sub == 0) valid_counts[r] = kept < a.region_cap ? kept : a.region_cap;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) RJ_STAMP(10);
}

// Windows behind an unbounded prefix: 16 lanes per region, one hit per lane and round, the per-hit
// procedure of behind_walk.h (three walks: forward from the cut, backwards to the left-most start,
// forward to the longest end), survivors compacted in place.  Both automata's tables in LDS.
template <int NW, int NQ>
__global__ __launch_bounds__(256) void verify_behind_in_regions(VerifyParams a, DevProgram P, DevProgram R, const uint32_t* hit_counts,
                                                                uint32_t* valid_counts, uint64_t* region_ends, uint32_t lds_words) {
  extern __shared__ uint32_t tab[];
  if (blockIdx.x == 0 && threadIdx.x == 0) RJ_STAMP(0);
  const DevProgram Pq = stage_tables(P, tab, lds_words >= P.table_words + R.table_words ? P.table_words : 0);
  const DevProgram Rq = stage_tables(R, tab + P.table_words, lds_words >= P.table_words + R.table_words ? R.table_words : 0);
  if (blockIdx.x == 0 && threadIdx.x == 0) RJ_STAMP(1);
  constexpr int G = 16;
  const uint64_t tid = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const uint64_t n_groups = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 4;
  const int lane = lane_id(), sub = lane & (G - 1), shift = lane & (64 - G);
  for (uint64_t r = tid >> 4; r < a.n_regions; r += n_groups) {
    const uint32_t raw = hit_counts[r];
    const uint32_t cnt = raw < a.region_cap ? raw : a.region_cap;
    if (raw > a.region_cap && sub == 0) {
      a.counters[kCntOverflow] = 1;
      atomicMax(&a.counters[kCntMaxRegion], static_cast<unsigned long long>(raw));
    }
    uint64_t* region = a.hits + r * a.region_cap;
    uint64_t* ends = region_ends + r * a.region_cap;
    uint32_t kept = 0;
    for (uint32_t base = 0; base < cnt; base += G) {
      const uint32_t k = base + sub;
      if (k < cnt) RJ_STAMP(2);
      const uint64_t w = k < cnt ? region[k] : 0;
      if (k < cnt && w != 0) RJ_STAMP(3);
      uint64_t b = 0, e = 0;
      bool overrun = false;
      const bool found = k < cnt && *static_cast<const volatile unsigned long long*>(a.counters + kCntOverrun) == 0 &&
                         rj_behind_candidate<NW, NQ>(Pq, Rq, a.text, a.n, w, &b, &e, &overrun, a.counters + kCntOverrun) && b >= a.sb &&
                         b < a.se;
      if (overrun) a.counters[kCntOverrun] = 1;
      const uint32_t mine = static_cast<uint32_t>(__ballot(found) >> shift) & ((1u << G) - 1u);
      const uint32_t pos = kept + __popc(mine & ((1u << sub) - 1u));
      if (found) {  // pos <= k, and every lane of the group has read its hit already
        region[pos] = b;
        ends[pos] = e;
        RJ_STAMP(9);
      }
      kept += __popc(mine);
    }
    if (sub == 0) valid_counts[r] = kept;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) RJ_STAMP(10);
}

// The tails of several patterns in one launch (rj_multi): blockIdx.y selects the pattern, whose
// parameters are read from a device array.
__global__ __launch_bounds__(256) void verify_in_regions_multi(const MultiTail* tails) {
  // by value: the parameters are read once (uniform address: scalar loads) instead of through a
  // reference that has to be re-read after the stores of the body
  const VerifyParams a = tails[blockIdx.y].verify;
  const DevProgram P = tails[blockIdx.y].program;
  const uint32_t* hit_counts = tails[blockIdx.y].hit_counts;
  uint32_t* valid_counts = tails[blockIdx.y].valid_counts;
  uint64_t* region_ends = tails[blockIdx.y].region_ends;
  // 4 lanes per region here: with nine patterns 16 lanes per region made several rounds of
  // workgroups, each a latency chain, and regions hold a handful of hits -- a quarter of the
  // threads fits the GPU in one round
  if (P.n_words <= 2) verify_in_regions_body<1, 4>(a, P, hit_counts, valid_counts, region_ends);
  else verify_in_regions_body<2, 4>(a, P, hit_counts, valid_counts, region_ends);
}

// region offsets + gather + disjointness check in one multi-workgroup launch: workgroup b owns
// regions [256 b, 256 b + 256), one per thread.  It sums the counts of ALL regions before its
// own (<= 64 Ki counts, read as uint4 by 256 threads: cheaper than another launch, and no
// workgroup waits for another), scans its own counts, and copies its regions' survivors to their
// final place in `out` (pairs).  The candidates already are the result iff every one is non-empty
// and begins at or after the end of all earlier ones (begins are sorted, so "all earlier" =
// running maximum); the maximum before a workgroup's first candidate is the last end of the
// nearest non-empty region before it.  Otherwise counters[kCntUnordered] is set and the host
// splits the pairs and runs the cluster-parallel selection.
constexpr int kOgcThreads = 256;
// Is candidate (b, e) in place as it stands?  Ordered and disjoint: it begins at or after every earlier
// end (`before_it`: their running maximum, or the carried-in cur).  An EMPTY match additionally must begin
// strictly behind that end -- one that begins where the previous match ended is dropped by the zero-length
// rule (src/codegen.cc:65-73), which the selection kernels apply -- except the very first candidate of a
// run into which no match ending there was carried (`^` at offset 0).  With this the line table of a
// grep-like caller (16 M empty matches per GB) and class hits take the no-selection path.
__device__ __forceinline__ bool candidate_in_place(uint64_t b, uint64_t e, uint64_t before_it, bool first_overall, uint64_t carry_pe) {
  return b >= before_it && (e > b || b > before_it || (first_overall && carry_pe != b));
}
__device__ __forceinline__ void offsets_gather_check_body(const uint32_t* counts, const uint64_t* region_begins,
                                                          const uint64_t* region_ends, uint32_t n_regions,
                                                          uint32_t region_cap, uint64_t carry_cur, uint64_t* out,
                                                          uint64_t out_cap, unsigned long long* counters,
                                                          unsigned long long* host_counters, uint64_t carry_pe, uint64_t epoch,
                                                          uint64_t* offsets_out = nullptr, uint64_t* prev_out = nullptr) {
  // Adjacency (a candidate begins exactly where an earlier one ends) is wanted for the Q8 check.
  // When the list is ordered and disjoint -- the only case in which this kernel's verdict is
  // used -- only neighbours can be adjacent, so comparing with the running maximum is exact.
  bool adjacent = false;
  constexpr int kWaves = kOgcThreads / kWave;
  __shared__ uint64_t wave_sum[kWaves], wave_before[kWaves], wave_end[kWaves];
  __shared__ int64_t wave_last[kWaves];
  const int lane = lane_id(), wv = threadIdx.x >> 6;
  if (threadIdx.x == 0) RJ_STAMP_AT(blockIdx.x, 0);
  const uint32_t first = blockIdx.x * kOgcThreads;  // a multiple of 4
  // 1. everything before this workgroup: candidate count, nearest non-empty region (its index and count packed in
  //    one word so that the maximum carries both).  Every workgroup used to add up the counts of ALL regions before its
  //    own: the last workgroup of a 65 536-region run (a 5 GB text) made 63 dependent trips per lane, 22 of the kernel's
  //    25 us (tools/ogc_trace.py: per-workgroup stamps).  Now a workgroup publishes the total and the last non-empty
  //    region of ITS 256 regions in two 8-byte granules {epoch, value} (relaxed agent-scope stores into the tail of the
  //    scan's counter block; the epoch is this launch's number, so nothing needs clearing) and reads the granules of
  //    the workgroups before it, one per lane: one trip.  It waits for lower-numbered workgroups only -- dispatched no
  //    later than itself -- and a lane whose granule does not show up in time adds that workgroup's counts up itself.
  uint64_t before = 0;
  int64_t last = -1;
  const uint4* counts4 = reinterpret_cast<const uint4*>(counts);
  const bool exchange = gridDim.x <= kOgcMaxBlocks && gridDim.x > 1 && (epoch >> 63) == 0;
  unsigned long long* granules = counters + kCntSize;  // [kOgcMaxBlocks] totals, [kOgcMaxBlocks] last non-empty regions
  auto sum_counts = [&](uint32_t q_begin, uint32_t q_end, uint32_t stride) {  // uint4 indices
    constexpr uint32_t kBatch = 8;
    for (uint32_t q0 = q_begin; q0 < q_end; q0 += stride * kBatch) {
      uint4 c[kBatch];
#pragma unroll
      for (uint32_t j = 0; j < kBatch; j++) {
        const uint32_t q = q0 + j * stride;
        c[j] = q < q_end ? counts4[q] : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (uint32_t j = 0; j < kBatch; j++) {
        before += static_cast<uint64_t>(c[j].x) + c[j].y + c[j].z + c[j].w;
        const uint32_t i = 4 * (q0 + j * stride);
        if (c[j].x) last = (static_cast<int64_t>(i) << 32) | c[j].x;  // q grows along the loop
        if (c[j].y) last = (static_cast<int64_t>(i + 1) << 32) | c[j].y;
        if (c[j].z) last = (static_cast<int64_t>(i + 2) << 32) | c[j].z;
        if (c[j].w) last = (static_cast<int64_t>(i + 3) << 32) | c[j].w;
      }
    }
  };
  if (!exchange) sum_counts(threadIdx.x, first / 4, kOgcThreads);
  // 2. own region: the first kHeld entries are loaded at once and kept in registers (regions hold
  //    a few candidates; one round trip instead of one per entry), the largest end
  constexpr int kHeld = 4;
  const uint32_t r = first + threadIdx.x;
  const uint32_t cnt = r < n_regions ? counts[r] : 0u;
  const uint64_t src = static_cast<uint64_t>(r) * region_cap;
  uint64_t hb[kHeld], he[kHeld];
  // (only the regions with survivors: loading the slots of ALL regions ahead of the count -- one trip less -- was
  // measured slower, 160 000 scattered loads against a thousand)
#pragma unroll
  for (int k = 0; k < kHeld; k++) {
    hb[k] = static_cast<uint32_t>(k) < cnt ? region_begins[src + k] : 0;
    he[k] = static_cast<uint32_t>(k) < cnt ? region_ends[src + k] : 0;
  }
  // the region's largest end: the held ones and the LAST one.  (Exact whenever the region is
  // ordered inside, and when it is not the copy below flags the list anyway.)
  uint64_t my_end = 0;
#pragma unroll
  for (int k = 0; k < kHeld; k++) my_end = he[k] > my_end ? he[k] : my_end;
  if (cnt > kHeld) {
    const uint64_t e = region_ends[src + cnt - 1];
    my_end = e > my_end ? e : my_end;
  }
  uint64_t inc = cnt, inc_end = my_end;
  int64_t own_last = cnt ? ((static_cast<int64_t>(r) << 32) | cnt) : -1;  // this workgroup's last non-empty region
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const uint64_t v = __shfl_up(inc, o);
    const uint64_t ve = __shfl_up(inc_end, o);
    if (lane >= o) {
      inc += v;
      inc_end = ve > inc_end ? ve : inc_end;
    }
    const int64_t ol = __shfl_xor(own_last, o);
    own_last = ol > own_last ? ol : own_last;
  }
  __shared__ int64_t wave_own_last[kWaves];
  if (lane == kWave - 1) {
    wave_sum[wv] = inc;
    wave_end[wv] = inc_end;
  }
  if (lane == 0) wave_own_last[wv] = own_last;
  __syncthreads();
  if (threadIdx.x == 0) RJ_STAMP_AT(blockIdx.x, 1);
  uint64_t own_before = 0, own_total = 0, end_before = 0;
#pragma unroll
  for (int w = 0; w < kWaves; w++) {
    if (w < wv) {
      own_before += wave_sum[w];
      end_before = wave_end[w] > end_before ? wave_end[w] : end_before;
    }
    own_total += wave_sum[w];
  }
  // (BOTH granules carry the launch number's LOW bits: 32 beside the total -- a workgroup's own 256 regions hold < 2^28
  // candidates -- and 26 beside the packed last region.  Rounds 3-4 put the number's bits 32..57 beside the last region:
  // zero for the first four billion launches, so that granule had no tag at all -- a reader that saw the new total and the
  // OLD last region (two relaxed stores, two relaxed loads: no order between them) took "no match before this workgroup"
  // or an earlier launch's region for the truth, and an empty match that begins where the previous match ends was
  // reported in place: `\xffa?|.{1,3}` over 70 001 bytes gave an extra (n, n) in one run of five under memory churn;
  // found by tests/test_gpu_mid.py's random patterns in round 5.)
  const unsigned long long tag_sum = (epoch & 0xFFFFFFFFull) << 32, tag_last = (epoch & 0x3FFFFFFull) << 38;
  if (exchange) {
    if (threadIdx.x == 0) {
      int64_t bl = -1;
#pragma unroll
      for (int w = 0; w < kWaves; w++) bl = wave_own_last[w] > bl ? wave_own_last[w] : bl;
      // (region index + 1 in 17 bits, count in 21: a region holds at most 2^20 candidates; 0 = none)
      const unsigned long long packed = bl < 0 ? 0ull : ((static_cast<unsigned long long>(bl >> 32) + 1) << 21) | static_cast<unsigned long long>(bl & 0x1FFFFF);
      __hip_atomic_store(&granules[blockIdx.x], tag_sum | (own_total & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&granules[kOgcMaxBlocks + blockIdx.x], tag_last | packed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x < blockIdx.x) {  // lane t: the workgroup t before this one
      unsigned long long gs = 0, gl = 0;
      bool have = false;
      for (int spin = 0; spin < 4096 && !have; spin++) {
        gs = __hip_atomic_load(&granules[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        gl = __hip_atomic_load(&granules[kOgcMaxBlocks + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        have = (gs >> 32) == (tag_sum >> 32) && (gl >> 38) == (tag_last >> 38);
        if (!have) __builtin_amdgcn_s_sleep(1);
      }
      if (have) {
        before = gs & 0xFFFFFFFFull;
        const unsigned long long pk = gl & 0x3FFFFFFFFFull;
        last = pk == 0 ? -1 : ((static_cast<int64_t>((pk >> 21) - 1) << 32) | static_cast<int64_t>(pk & 0x1FFFFF));
      } else {
        sum_counts(threadIdx.x * (kOgcThreads / 4), (threadIdx.x + 1) * (kOgcThreads / 4), 1);  // that workgroup's 256 counts
      }
    }
  }
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    before += __shfl_xor(before, o);
    const int64_t l2 = __shfl_xor(last, o);
    last = l2 > last ? l2 : last;
  }
  if (lane == 0) {
    wave_before[wv] = before;
    wave_last[wv] = last;
  }
  __syncthreads();
  if (threadIdx.x == 0) RJ_STAMP_AT(blockIdx.x, 2);
  uint64_t base = 0;
  int64_t nearest = -1;
#pragma unroll
  for (int w = 0; w < kWaves; w++) {
    base += wave_before[w];
    nearest = wave_last[w] > nearest ? wave_last[w] : nearest;
  }
  // running maximum of the ends before this thread's first candidate
  uint64_t prev = carry_cur;
  if (nearest >= 0) {
    const uint64_t e = region_ends[static_cast<uint64_t>(nearest >> 32) * region_cap + (nearest & 0xFFFFFFFF) - 1];
    prev = e > prev ? e : prev;
  }
  prev = end_before > prev ? end_before : prev;
  const uint64_t up = __shfl_up(inc_end, 1);  // inclusive maximum of the lanes below
  if (lane > 0) prev = up > prev ? up : prev;
  if (threadIdx.x == 0 && prev != ~0ull) RJ_STAMP_AT(blockIdx.x, 3);
  // 3. copy + check
  const uint64_t off = base + own_before + inc - cnt;
  bool ok = true;
  if (offsets_out != nullptr) {
    // offsets only: a second launch (gather_regions_by_wave) copies and checks, one wave per region
    if (r < n_regions) {
      offsets_out[r] = off;
      prev_out[r] = prev;
    }
  } else if (own_total > 4 * kOgcThreads) {
    // many candidates per region (dense patterns, `^` over a file, one-letter replaces): the
    // workgroup copies them together, entry e of the workgroup by thread e mod 256 -- coalesced
    // 16-byte stores instead of one thread walking hundreds of entries.  The region of an entry
    // comes from a binary search in the workgroup's prefix (LDS); an entry is checked against the
    // entry before it (the running maximum before the region for its first one).
    __shared__ uint32_t s_pre[kOgcThreads];
    __shared__ uint64_t s_prev[kOgcThreads];
    s_pre[threadIdx.x] = static_cast<uint32_t>(own_before + inc - cnt);
    s_prev[threadIdx.x] = prev;
    __syncthreads();
    for (uint64_t e = threadIdx.x; e < own_total; e += kOgcThreads) {
      uint32_t lo = 0, hi = kOgcThreads;  // last t with s_pre[t] <= e  (an empty region shares its prefix with the next one)
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_pre[mid] <= e) lo = mid; else hi = mid;
      }
      const uint64_t at = static_cast<uint64_t>(first + lo) * region_cap + (e - s_pre[lo]);
      const uint64_t b = region_begins[at], en = region_ends[at];
      const uint64_t before_it = e > s_pre[lo] ? region_ends[at - 1] : s_prev[lo];
      ok = ok && candidate_in_place(b, en, before_it, base + e == 0, carry_pe);
      adjacent = adjacent || (b == before_it && before_it != 0);
      if (base + e < out_cap) *reinterpret_cast<ulonglong2*>(out + 2 * (base + e)) = make_ulonglong2(b, en);
    }
  } else
  for (uint32_t k = 0; k < cnt; k++) {
    uint64_t b, e;
    if (k < kHeld) {
      b = k == 0 ? hb[0] : k == 1 ? hb[1] : k == 2 ? hb[2] : hb[3];
      e = k == 0 ? he[0] : k == 1 ? he[1] : k == 2 ? he[2] : he[3];
    } else {
      b = region_begins[src + k];
      e = region_ends[src + k];
    }
    ok = ok && candidate_in_place(b, e, prev, off + k == 0, carry_pe);
    adjacent = adjacent || (b == prev && prev != 0);
    prev = e > prev ? e : prev;
    if (off + k < out_cap) *reinterpret_cast<ulonglong2*>(out + 2 * (off + k)) = make_ulonglong2(b, e);
  }
  if (threadIdx.x == 0) RJ_STAMP_AT(blockIdx.x, 4);
  if (!ok) {
    counters[kCntUnordered] = 1;
    if (host_counters) host_counters[kCntUnordered] = 1;
  }
  if (adjacent) {
    counters[kCntAdjacent] = 1;
    if (host_counters) host_counters[kCntAdjacent] = 1;
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    counters[kCntHits] = base + own_total;
    counters[kCntCands] = base + own_total;
    if (host_counters) {
      // straight into pinned host memory (visible when the kernel has completed): the host needs
      // no copy command, only the synchronise.  The flags below were set by EARLIER kernels; the
      // host cleared its kCntUnordered word before the launch.
      host_counters[kCntHits] = base + own_total;
      host_counters[kCntCands] = base + own_total;
      host_counters[kCntOverflow] = counters[kCntOverflow];
      host_counters[kCntMaxRegion] = counters[kCntMaxRegion];
      host_counters[kCntOverrun] = counters[kCntOverrun];
      host_counters[kCntSharedMax] = counters[kCntSharedMax];
      host_counters[kCntFinal] = 0;
    }
    RJ_STAMP_AT(blockIdx.x, 5);
  }
}

__global__ __launch_bounds__(kOgcThreads) void offsets_gather_check(const uint32_t* counts, const uint64_t* region_begins,
                                                                    const uint64_t* region_ends, uint32_t n_regions,
                                                                    uint32_t region_cap, uint64_t carry_cur, uint64_t* out,
                                                                    uint64_t out_cap, unsigned long long* counters,
                                                                    unsigned long long* host_counters, uint64_t carry_pe,
                                                                    uint64_t* offsets_out, uint64_t* prev_out, uint64_t epoch) {
  offsets_gather_check_body(counts, region_begins, region_ends, n_regions, region_cap, carry_cur, out, out_cap, counters,
                            host_counters, carry_pe, epoch, offsets_out, prev_out);
}

// Second half of the two-launch form used when regions hold many candidates (tens and more each:
// dense patterns, `^` over a text, one-letter replaces): one wave per region copies its survivors
// with coalesced 16-byte stores and checks each against the one before it.
__global__ __launch_bounds__(256) void gather_regions_by_wave(const uint32_t* counts, const uint64_t* region_begins,
                                                              const uint64_t* region_ends, const uint64_t* offsets,
                                                              const uint64_t* prev_end, uint32_t n_regions, uint32_t region_cap,
                                                              uint64_t* out, uint64_t out_cap, unsigned long long* counters,
                                                              unsigned long long* host_counters, uint64_t carry_pe) {
  const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_waves = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 6;
  const int lane = lane_id();
  bool ok = true, adjacent = false;
  for (uint64_t r = wave; r < n_regions; r += n_waves) {
    const uint32_t cnt = counts[r];
    if (cnt == 0) continue;
    const uint64_t src = r * region_cap, off = offsets[r], before_region = prev_end[r];
    for (uint32_t k = lane; k < cnt; k += kWave) {
      const uint64_t b = region_begins[src + k], e = region_ends[src + k];
      const uint64_t before_it = k ? region_ends[src + k - 1] : before_region;
      ok = ok && candidate_in_place(b, e, before_it, off + k == 0, carry_pe);
      adjacent = adjacent || (b == before_it && before_it != 0);
      if (off + k < out_cap) *reinterpret_cast<ulonglong2*>(out + 2 * (off + k)) = make_ulonglong2(b, e);
    }
  }
  if (!ok) {
    counters[kCntUnordered] = 1;
    if (host_counters) host_counters[kCntUnordered] = 1;
  }
  if (adjacent) {
    counters[kCntAdjacent] = 1;
    if (host_counters) host_counters[kCntAdjacent] = 1;
  }
}

__global__ __launch_bounds__(kOgcThreads) void offsets_gather_check_multi(const MultiTail* tails, uint64_t epoch) {
  const MultiTail t = tails[blockIdx.y];  // by value, see verify_in_regions_multi
  offsets_gather_check_body(t.valid_counts, t.verify.hits, t.region_ends, t.verify.n_regions, t.verify.region_cap, 0, t.out,
                            t.out_cap, t.verify.counters, t.host_counters, ~0ull, epoch);
}

// First and last match of up to kMaxFused result lists (rj_multi_bounds: what a shard exchanges with
// its neighbours to carry the selection over a cut): bounds[p] = {first begin, first end, last begin,
// last end}, all ~0 when the list is empty.  One thread per list; written straight to pinned memory.
__global__ void first_last_kernel(BoundsParams a, uint64_t* bounds) {
  const int p = threadIdx.x;
  if (p >= a.n_lists) return;
  const uint64_t n = a.count[p];
  const uint64_t* r = a.spans[p];
  bounds[4 * p + 0] = n ? r[0] : ~0ull;
  bounds[4 * p + 1] = n ? r[1] : ~0ull;
  bounds[4 * p + 2] = n ? r[2 * (n - 1)] : ~0ull;
  bounds[4 * p + 3] = n ? r[2 * (n - 1) + 1] : ~0ull;
}

void launch_first_last(const BoundsParams& a, uint64_t* pinned_bounds, hipStream_t st) {
  hipLaunchKernelGGL(first_last_kernel, dim3(1), dim3(64), 0, st, a, pinned_bounds);
}

// The same as rows of the carry exchange between shards (rejit_amd/sharding.py: multi_pattern_counts_device), left in
// device memory, plus an offset that turns the shard's local offsets into global ones.  rows[p] = the 8 integers a
// rank contributes per pattern: count | first begin, first end (written in the FIRST round only: the result under the
// empty carry) | last begin, last end (-1 without a match) | the carry the result was selected under (cur, prev_end,
// have: zeroed in the first round, kept up to date by the caller when it re-runs a pattern).  Nothing waits for it:
// the collective that follows is queued on the same stream.
__global__ void bounds_rows_kernel(BoundsParams a, int64_t offset, int first_round, int64_t* rows) {
  const int p = threadIdx.x;
  if (p >= a.n_lists) return;
  const uint64_t n = a.count[p];
  const uint64_t* r = a.spans[p];
  rows[8 * p + 0] = static_cast<int64_t>(n);
  if (first_round) {  // (selected under the empty carry: the first match every later round is judged by)
    rows[8 * p + 1] = n ? static_cast<int64_t>(r[0]) + offset : -1;
    rows[8 * p + 2] = n ? static_cast<int64_t>(r[1]) + offset : -1;
    rows[8 * p + 5] = rows[8 * p + 6] = rows[8 * p + 7] = 0;
  }
  rows[8 * p + 3] = n ? static_cast<int64_t>(r[2 * (n - 1)]) + offset : -1;
  rows[8 * p + 4] = n ? static_cast<int64_t>(r[2 * (n - 1) + 1]) + offset : -1;
}

void launch_bounds_rows(const BoundsParams& a, int64_t offset, int first_round, int64_t* d_rows, hipStream_t st) {
  hipLaunchKernelGGL(bounds_rows_kernel, dim3(1), dim3(64), 0, st, a, offset, first_round, d_rows);
}

// The decision step of the carry exchange, on the device: all[r][p] = count, first begin / end under the EMPTY carry,
// current last begin / end, the carry the current result was selected under (cur, prev_end, have) -- 8 integers per
// rank and pattern, as gathered.  out: [0, P) the job-wide counts; [P, 2P) 1 when `rank` has to select pattern p again
// (must_rerun, sharding.py); [2P, 4P) the carry (cur, prev_end) to select it under; [4P] 1 when ANY rank re-runs
// anything (another round is needed).
__global__ void carry_decide_kernel(const int64_t* all, int world, int rank, int n_patterns, int64_t* out) {
  const int p = threadIdx.x;
  __shared__ int any_again;
  if (threadIdx.x == 0) any_again = 0;
  __syncthreads();
  if (p < n_patterns) {
    int64_t total = 0;
    int64_t carry_cur = 0, carry_pe = 0;
    bool have = false, mine_again = false, again = false;
    int64_t my_cur = 0, my_pe = 0;
    for (int r = 0; r < world; r++) {
      const int64_t* row = all + (static_cast<int64_t>(r) * n_patterns + p) * 8;
      total += row[0];
      if (r > 0) {
        // carry into rank r = the last match of the nearest rank before it that has one (carried along below)
        const int64_t used_cur = row[5], used_pe = row[6];
        const bool used_have = row[7] != 0;
        const bool same = (have == used_have) && (!have || (carry_cur == used_cur && carry_pe == used_pe));
        bool rerun = false;
        if (!same) {
          if (used_have) {
            rerun = true;  // it re-ran already: its result depends on the carry it used
          } else if (row[1] >= 0 && have) {
            const int64_t fb = row[1], fe = row[2];
            rerun = fb < carry_cur || (fb == fe && carry_pe == fb);
          }
        }
        if (rerun) {
          again = true;
          if (r == rank) {
            mine_again = true;
            my_cur = have ? carry_cur : 0;
            my_pe = have ? carry_pe : 0;
          }
        }
      }
      if (row[3] >= 0) {
        const int64_t lb = row[3], le = row[4];
        carry_cur = le > lb ? le : lb + 1;
        carry_pe = le;
        have = true;
      }
    }
    out[p] = total;
    out[n_patterns + p] = mine_again ? 1 : 0;
    out[2 * n_patterns + 2 * p] = my_cur;
    out[2 * n_patterns + 2 * p + 1] = my_pe;
    if (again) atomicOr(&any_again, 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) out[4 * n_patterns] = any_again;
}

void launch_carry_decide(const int64_t* d_all, int world, int rank, int n_patterns, int64_t* out, hipStream_t st) {
  hipLaunchKernelGGL(carry_decide_kernel, dim3(1), dim3(64), 0, st, d_all, world, rank, n_patterns, out);
}

__global__ void globalize_spans(const uint64_t* local, uint64_t n2, uint64_t offset, uint64_t* out) {
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n2; i += stride) out[i] = local[i] + offset;
}
void launch_globalize_spans(const uint64_t* local, uint64_t count, int64_t offset, uint64_t* out, hipStream_t st) {
  const uint64_t n2 = 2 * count;
  const unsigned blocks = static_cast<unsigned>(std::min<uint64_t>((n2 + 255) / 256, 4096));
  hipLaunchKernelGGL(globalize_spans, dim3(blocks ? blocks : 1), dim3(256), 0, st, local, n2, static_cast<uint64_t>(offset), out);
}

// pairs -> begin[] / end[] for the selection kernels (only when the pairs are not the result yet)
__global__ void split_pairs(const uint64_t* pairs, const unsigned long long* n_ptr, uint64_t* keys, uint64_t* vals) {
  const uint64_t n = *n_ptr;
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    keys[i] = pairs[2 * i];
    vals[i] = pairs[2 * i + 1];
  }
}

// Dense mode, lane-sized automaton: the NFA inner loop proper.  Two things made the plain
// one-walk-per-lane kernel (verify_lane_regions) slow -- 1700 lane-cycles per walk on
// `[A-Z][a-z]+ ...` although the average walk is under two steps:
//   * a wave ran as long as its LONGEST walk (walk lengths are geometric), and
//   * every step was a chain of dependent global loads (text byte -> class row -> follow rows).
// Here the lanes are persistent walkers: each loop iteration advances every live walk by one
// byte, and a lane whose walk has ended takes the next hit of the region at once (wave-shared
// cursor, ballot + mbcnt), so the lanes stay busy regardless of the length distribution.  The
// automaton tables (a few KiB) are staged in LDS once per workgroup, and the next text byte is
// loaded one step ahead.
template <int NQ>
__global__ __launch_bounds__(256) void verify_walkers(VerifyParams a, DevProgram P) {
  extern __shared__ uint32_t tab[];
  const int W = P.n_words, C = P.n_ctx, NP = P.n_pos > 0 ? P.n_pos : 1;
  const int o_last = C * W, o_lin = 2 * C * W, o_rowof = o_lin + W, o_rows = o_rowof + NP, o_cls = o_rows + C * P.n_rows * W;
  for (uint32_t i = threadIdx.x; i < P.table_words; i += blockDim.x) tab[i] = P.first[i];
  __syncthreads();
  const bool ctxed = C > 1;
  const int lane = lane_id();
  const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_waves = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 6;
  uint64_t lin[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    lin[q] = 2 * q < W ? tab[o_lin + 2 * q] : 0u;
    if (2 * q + 1 < W) lin[q] |= static_cast<uint64_t>(tab[o_lin + 2 * q + 1]) << 32;
  }
  auto word2 = [&](int base, int q) -> uint64_t {  // 64-bit word q of a W-word LDS row
    uint64_t v = 2 * q < W ? tab[base + 2 * q] : 0u;
    if (2 * q + 1 < W) v |= static_cast<uint64_t>(tab[base + 2 * q + 1]) << 32;
    return v;
  };
  for (uint64_t r = wave; r < a.n_regions; r += n_waves) {
    const uint64_t lo = a.offsets[r];
    const uint32_t cnt = static_cast<uint32_t>(a.offsets[r + 1] - lo);
    const uint64_t* region = a.hits + r * a.region_cap;
    uint32_t cursor = 0;  // wave-uniform: next hit of the region to hand out
    // lane state
    bool active = false, found = false;
    uint32_t my_k = 0, prevb = 0, curb = 0;  // bytes at p-1 and p
    uint64_t s = 0, p = 0, e = 0, S[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) S[q] = 0;
    for (;;) {
      // ---- refill: idle lanes take the next hits
      const uint64_t idle = __ballot(!active);
      if (idle != 0 && cursor < cnt) {
        const uint32_t k = cursor + __popcll(idle & ((1ull << lane) - 1ull));
        if (!active && k < cnt) {
          my_k = k;
          s = region[k];
          active = true;
          found = false;
          e = 0;
          p = s;
          prevb = s > 0 && s <= a.n ? a.text[s - 1] : '\n';
          curb = s < a.n ? a.text[s] : '\n';
          int ctx = 0;
          if (ctxed) {
            if (s == 0 || rj_line_break(prevb)) ctx |= 1;
            if (s == a.n || rj_line_break(curb)) ctx |= 2;
          }
          if ((P.nullable >> ctx) & 1u) {
            found = true;
            e = s;
          }
          const bool can_start = s < a.n && P.n_pos != 0;
#pragma unroll
          for (int q = 0; q < NQ; q++)
            S[q] = can_start ? (word2(ctx * W, q) & word2(o_cls + static_cast<int>(curb) * W, q)) : 0;
          // p -> s + 1, its byte one step ahead
          p = s + 1;
          prevb = curb;
          curb = p < a.n ? a.text[p] : '\n';
        }
        cursor += __popcll(idle);
        if (cursor > cnt) cursor = cnt;
      }
      if (__ballot(active) == 0) break;  // nothing live and (cursor >= cnt): region done
      // ---- one step of every live walk
      if (active) {
        uint64_t alive = 0;
#pragma unroll
        for (int q = 0; q < NQ; q++) alive |= S[q];
        bool done = alive == 0;
        if (!done) {
          int ctx = 0;
          if (ctxed) {
            if (rj_line_break(prevb)) ctx |= 1;  // p >= 1 here
            if (p == a.n || rj_line_break(curb)) ctx |= 2;
          }
          uint64_t acc = 0;
#pragma unroll
          for (int q = 0; q < NQ; q++) acc |= S[q] & word2(o_last + ctx * W, q);
          if (acc) {
            found = true;
            e = p;
          }
          if (p == a.n) {
            done = true;
          } else if (p - s >= P.max_walk) {
            a.counters[kCntOverrun] = 1;
            done = true;
          } else {
            const uint32_t nextb = p + 1 < a.n ? a.text[p + 1] : '\n';  // issued before the table work
            uint64_t T[NQ];
            uint64_t carry = 0;
#pragma unroll
            for (int q = 0; q < NQ; q++) {
              const uint64_t x = S[q] & lin[q];
              T[q] = (x << 1) | carry;
              carry = x >> 63;
            }
#pragma unroll
            for (int q = 0; q < NQ; q++) {
              uint64_t sp = S[q] & ~lin[q];
              while (sp) {
                const int b = __builtin_ctzll(sp);
                sp &= sp - 1;
                const int row = o_rows + (ctx * P.n_rows + static_cast<int>(tab[o_rowof + q * 64 + b])) * W;
#pragma unroll
                for (int j = 0; j < NQ; j++) T[j] |= word2(row, j);
              }
            }
#pragma unroll
            for (int q = 0; q < NQ; q++) S[q] = T[q] & word2(o_cls + static_cast<int>(curb) * W, q);
            p++;
            prevb = curb;
            curb = nextb;
          }
        }
        if (done) {
          a.cand_begin[lo + my_k] = s;
          a.cand_end[lo + my_k] = found ? e : kNoMatch;
          active = false;
        }
      }
    }
  }
}

template <int NR>
__global__ __launch_bounds__(256) void verify_wave(VerifyParams a, DevProgram P) {
  const uint64_t n_slots = a.offsets[a.n_regions] * a.expand;
  const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_waves = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 6;
  for (uint64_t i = wave; i < n_slots; i += n_waves) {
    const uint64_t w = hit_at(a, i / a.expand);
    const uint64_t back = static_cast<uint64_t>(a.float_max) - (i % a.expand);
    const uint64_t s = w - back;
    uint64_t e = 0;
    bool overrun = false;
    const bool found = w >= back && s >= a.sb && s < a.se && wave_longest<NR>(P, a.text, a.n, s, &e, false, &overrun);
    if (overrun && lane_id() == 0) a.counters[kCntOverrun] = 1;
    if (lane_id() == 0) {
      a.cand_begin[i] = s;
      a.cand_end[i] = found ? e : kNoMatch;
    }
  }
}

// kMatchFull: result[0] = 1 iff the automaton started at 0 accepts exactly at n.
template <int NR>
__global__ __launch_bounds__(64) void match_full(const uint8_t* text, uint64_t n, DevProgram P, int* result) {
  uint64_t e = 0;
  bool overrun = false;
  const bool found = wave_longest<NR>(P, text, n, 0, &e, true, &overrun);
  if (lane_id() == 0) result[0] = (found && e == n) ? 1 : 0;
}

// ---------------------------------------------------------------------------------------
// Small texts in ONE launch.  The general pipeline is three kernels and one synchronise -- ~35 us per call
// however small the text, ~60 us with the copies of a host text -- which a caller that matches file by
// file (the reference's sample/jrep.cc:261-313 does one MatchAll per file, and a second one for the line
// table) pays per file.  Here one workgroup of 1024 lanes does the whole MatchAll of a text of up to 32 KiB:
//   text (read straight from pinned host memory when the caller's buffer is on the host) and tables -> LDS;
//   every lane takes a contiguous slice of start positions: candidate test + longest match from LDS
//   (the same rj_lane_longest), length + 1 into a 16-bit array;
//   a workgroup scan lays the candidates out in position order;
//   "already the result" check in parallel, else the sequential rule on one lane (few candidates);
//   pairs and count go straight to the output (pinned host memory for host callers).
template <int NQ>
__global__ __launch_bounds__(1024) void match_small(SmallParams a, DevProgram P) {
  extern __shared__ uint32_t tab[];
  __shared__ uint32_t wave_sums[16];
  __shared__ int s_flags[4];  // 0: fallback, 1: not in place, 2: adjacent
  __shared__ unsigned long long s_count;
  __shared__ uint32_t fb[8];  // first-byte bitmap (indexed by data: LDS -- indexing the by-value struct put it in scratch)
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 8; k++) fb[k] = P.first_bytes[k];
  }
  // every exit goes through `finish`: results first, then (system-scope release) the header the host polls
  auto finish = [&](unsigned long long count, unsigned long long status) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
      a.hdr[0] = count;
      __threadfence_system();
      a.hdr[1] = status;
    }
  };
  const DevProgram Q = stage_tables(P, tab, P.table_words);
  const uint32_t n = a.n;
  uint8_t* txt = reinterpret_cast<uint8_t*>(tab + ((P.table_words + 3u) & ~3u));
  const uint32_t txt_bytes = (n + 31u) & ~15u;
  uint16_t* len1 = reinterpret_cast<uint16_t*>(txt + txt_bytes);
  uint32_t* cands = reinterpret_cast<uint32_t*>(txt + txt_bytes + ((2u * (n + 2u) + 15u) & ~15u));
  const uint32_t tid = threadIdx.x;
  if (tid < 4) s_flags[tid] = 0;
  for (uint32_t i = tid * 16; i < txt_bytes; i += 1024 * 16) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (i + 16 <= n) {
      v = *reinterpret_cast<const uint4*>(a.text + i);  // (a.text is 16-byte aligned)
    } else {
      uint32_t w[4] = {0, 0, 0, 0};
      for (uint32_t k = 0; k < 16 && i + k < n; k++) w[k >> 2] |= static_cast<uint32_t>(a.text[i + k]) << (8 * (k & 3));
      v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    *reinterpret_cast<uint4*>(txt + i) = v;
  }
  __syncthreads();
  // ---- longest match from every start of this lane's slice
  const uint32_t per = (n + 1 + 1023) / 1024;
  const uint32_t s0 = tid * per, s1 = s0 + per < n + 1 ? s0 + per : n + 1;
  uint32_t mine = 0;
  for (uint32_t s = s0; s < s1; s++) {
    uint32_t l = 0;
    bool cand = s >= a.sb && s < a.se;
    if (cand) {
      bool ok = false;
      if (Q.nullable) ok = (Q.nullable >> (Q.n_ctx > 1 ? rj_context(txt, n, s) : 0)) & 1u;
      if (!ok && s < n) ok = (fb[txt[s] >> 5] >> (txt[s] & 31)) & 1u;
      cand = ok && !(Q.loop_first && s > 0 && ((fb[txt[s - 1] >> 5] >> (txt[s - 1] & 31)) & 1u));
    }
    if (cand) {
      uint64_t e = 0;
      bool overrun = false;
      if (rj_lane_longest<NQ>(Q, txt, n, s, &e, &overrun)) l = static_cast<uint32_t>(e - s) + 1;
      if (overrun) s_flags[0] = 1;
    }
    len1[s] = static_cast<uint16_t>(l);
    mine += l != 0;
  }
  // ---- exclusive scan of the lanes' counts
  const int lane = lane_id(), wv = tid >> 6;
  uint32_t inc = mine;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const uint32_t v = __shfl_up(inc, o);
    if (lane >= o) inc += v;
  }
  if (lane == kWave - 1) wave_sums[wv] = inc;
  __syncthreads();
  uint32_t before = 0, total = 0;
  for (int w = 0; w < 16; w++) {
    if (w < wv) before += wave_sums[w];
    total += wave_sums[w];
  }
  if (total > kSmallMaxCands || total > a.out_cap) {
    if (tid == 0) s_flags[0] = 1;
    total = 0;
  }
  __syncthreads();
  if (s_flags[0]) {
    finish(0, 1);
    return;
  }
  uint32_t at = before + inc - mine;
  for (uint32_t s = s0; s < s1; s++)
    if (len1[s]) cands[at++] = (s << 16) | len1[s];
  __syncthreads();
  // ---- are the candidates the result as they stand?
  for (uint32_t i = tid; i < total; i += 1024) {
    const uint64_t b = cands[i] >> 16, e = b + (cands[i] & 0xFFFFu) - 1;
    const uint64_t before_it = i ? (cands[i - 1] >> 16) + (cands[i - 1] & 0xFFFFu) - 1 : a.carry_cur;
    if (!candidate_in_place(b, e, before_it, i == 0, a.have_prev ? a.carry_prev_end : ~0ull)) s_flags[1] = 1;
    if (i && b == before_it && before_it != 0) s_flags[2] = 1;
  }
  __syncthreads();
  // (Q8, DESIGN.md: where a candidate begins exactly at the end of another the reference's ring artefact can
  // change the answer; the general pipeline knows what to do)
  if (a.q8_risk && (s_flags[1] || s_flags[2])) {
    finish(0, 1);
    return;
  }
  if (!s_flags[1]) {
    for (uint32_t i = tid; i < total; i += 1024) {
      const uint64_t b = cands[i] >> 16, e = b + (cands[i] & 0xFFFFu) - 1;
      *reinterpret_cast<ulonglong2*>(a.out + 2 * i) = make_ulonglong2(b, e);
    }
    finish(total, 0);
    return;
  }
  if (tid == 0) {  // overlapping or empty candidates: the sequential rule
    RjSelectState st;
    st.cur = a.carry_cur;
    st.prev_end = a.carry_prev_end;
    st.have_prev = a.have_prev != 0;
    unsigned long long k = 0;
    for (uint32_t i = 0; i < total; i++) {
      const uint64_t b = cands[i] >> 16, e = b + (cands[i] & 0xFFFFu) - 1;
      bool taken;
      if (rj_select_step(&st, b, e, &taken)) {
        a.out[2 * k] = b;
        a.out[2 * k + 1] = e;
        k++;
      }
    }
    s_count = k;
  }
  __syncthreads();
  finish(s_count, 0);
}

size_t small_lds_bytes(const DevProgram& P, uint32_t n) {
  const size_t tables = ((static_cast<size_t>(P.table_words) + 3) & ~size_t{3}) * 4;
  const size_t txt = (n + 31u) & ~15u;
  const size_t lens = (2u * (n + 2u) + 15u) & ~15u;
  return tables + txt + lens + static_cast<size_t>(kSmallMaxCands) * 4;
}

// dynamic LDS one workgroup of match_small may use on this device (the runtime's per-block limit, raised to
// the hardware's where the runtime allows it); queried once
size_t small_lds_limit() {
  static const size_t limit = [] {
    int dev = 0, per_block = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&per_block, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || per_block <= 0) per_block = 64 * 1024;
    size_t lim = static_cast<size_t>(per_block);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(match_small<1>), hipFuncAttributeMaxDynamicSharedMemorySize, per_block) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(match_small<2>), hipFuncAttributeMaxDynamicSharedMemorySize, per_block) != hipSuccess)
      lim = std::min<size_t>(lim, 64 * 1024);
    (void)hipGetLastError();  // (a refused attribute must not surface as a later call's error)
    return lim - 256;         // room for the kernel's static LDS
  }();
  return limit;
}

void launch_match_small(const SmallParams& a, const DevProgram& P, hipStream_t st) {
  const size_t lds = small_lds_bytes(P, a.n);
  if (P.n_words <= 2) hipLaunchKernelGGL((match_small<1>), dim3(1), dim3(1024), lds, st, a, P);
  else hipLaunchKernelGGL((match_small<2>), dim3(1), dim3(1024), lds, st, a, P);
}

// The reference's no-fast-forward kMatchAll loop, restated for ONE lane: a ring of
// times x states start offsets (GenerateMatchDirection, codegen-x64.cc:535-640; SetState
// :951-987 "left-most start wins"; CheckMatch + ClearStates :401-466,1075-1097; the sink
// MatchAllAppendFilter, src/codegen.cc:36-86).  It exists for bit-exactness only: it reproduces
// the ring-slot artefact "Q8"  that the parallel pipeline -- which
// implements the documented semantics -- does not, and runs only when a pattern can hit that
// artefact AND two candidates are adjacent.  Sequential by nature: ~1 us per text byte.
__global__ void exact_sequential(const uint8_t* t, uint64_t n, DevGraph G, int64_t* ring, uint64_t* out,
                                 uint64_t out_cap, unsigned long long* counters) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int S = G.n_states, T = G.times;
  const int slots = S * T;
  for (int i = 0; i < slots; i++) ring[i] = -1;
  int base = 0;
  unsigned long long out_n = 0;
  bool pending = false;
  int64_t pb = 0, pe = 0;
  auto at = [&](int time, int st) -> int64_t& {
    int tt = base + time;
    if (tt >= T) tt -= T;
    return ring[tt * S + st];
  };
  auto emit = [&](int64_t b, int64_t e) {
    while (out_n > 0 && static_cast<int64_t>(out[2 * (out_n - 1)]) >= b) out_n--;
    if (b == e && out_n > 0 && static_cast<int64_t>(out[2 * (out_n - 1) + 1]) == b) return;
    if (out_n < out_cap) {
      out[2 * out_n] = static_cast<uint64_t>(b);
      out[2 * out_n + 1] = static_cast<uint64_t>(e);
    }
    out_n++;
  };
  for (uint64_t p = 0;; p++) {
    if (pending) {
      emit(pb, pe);
      pending = false;
      if (static_cast<uint64_t>(pe) == n) break;
    }
    at(0, G.entry) = static_cast<int64_t>(p);
    bool changed = true;
    while (changed) {
      changed = false;
      for (int i = 0; i < G.n_control_edges; i++) {
        const int64_t v = at(0, G.ce_src[i]);
        if (v < 0) continue;
        const int kind = G.ce_kind[i];
        bool ok = true;
        if (kind == 1) ok = p == 0 || rj_line_break(t[p - 1]);
        else if (kind == 2) ok = p == n || rj_line_break(t[p]);
        if (!ok) continue;
        int64_t& tgt = at(0, G.ce_dst[i]);
        if (tgt < 0 || v < tgt) {
          tgt = v;
          changed = true;
        }
      }
    }
    const int64_t xs = at(0, G.exit);
    if (xs >= 0) {
      pending = true;
      pb = xs;
      pe = static_cast<int64_t>(p);
      for (int i = 0; i < slots; i++)
        if (ring[i] > xs && ring[i] < pe) ring[i] = -1;
    }
    if (p == n) {
      if (pending) emit(pb, pe);
      break;
    }
    for (int i = 0; i < G.n_byte_edges; i++) {
      const int64_t v = at(0, G.be_src[i]);
      if (v < 0) continue;
      const int len = G.be_len[i];
      int land = 0;
      if (len > 0) {
        if (p + static_cast<uint64_t>(len) <= n) {
          const uint8_t* lit = G.lit + G.be_off[i];
          bool eq = true;
          for (int k = 0; k < len && eq; k++) eq = t[p + k] == lit[k];
          if (eq) land = len;
        }
      } else {
        const uint32_t c = t[p];
        if ((G.cls[G.be_off[i] * 8 + (c >> 5)] >> (c & 31)) & 1u) land = 1;
      }
      if (land) {
        int64_t& tgt = at(land, G.be_dst[i]);
        if (tgt < 0 || v < tgt) tgt = v;
      }
    }
    for (int s = 0; s < S; s++) at(0, s) = -1;
    base++;
    if (base >= T) base -= T;
  }
  counters[kCntFinal] = out_n;
}

__global__ void mark_valid(const uint64_t* cand_end, uint64_t n, uint64_t* flags) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) flags[i] = cand_end[i] != kNoMatch ? 1 : 0;
}

__global__ void compact_valid(const uint64_t* cand_begin, const uint64_t* cand_end, const uint64_t* flags,
                              const uint64_t* pos, uint64_t n, uint64_t* keys, uint64_t* vals,
                              unsigned long long* counters) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (flags[i]) {
    keys[pos[i]] = cand_begin[i];
    vals[pos[i]] = cand_end[i];
  }
  if (i == n - 1) counters[kCntCands] = pos[i] + flags[i];
}

// Large path, common case in one kernel: emit the sorted candidates as pairs and find out
// whether they already are the result (pairwise disjoint, no empty match, nothing hidden by
// the carry); *unordered is set otherwise and the cluster-parallel selection runs.
__global__ void check_and_interleave(const uint64_t* keys, const uint64_t* vals, const unsigned long long* n_ptr,
                                     uint64_t carry_cur, uint64_t* out, uint64_t cap, unsigned long long* unordered) {
  const uint64_t n = *n_ptr;  // number of candidates, produced earlier on this stream
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint64_t b = keys[i], e = vals[i];
    // (a slot that did not verify holds kNoMatch: it fails here or as the next slot's predecessor)
    const bool ok = e != kNoMatch && e > b && (i == 0 ? b >= carry_cur : b >= vals[i - 1]);
    if (!ok) *unordered = 1;
    if (i < cap) {
      out[2 * i] = b;
      out[2 * i + 1] = e;
    }
  }
}

// ---------------------------------------------------------------------------------------
// replace_gather: rejit::Replace (reference src/rejit.cc:97-112: append the text between the
// matches and the replacement, sequentially) as a parallel gather.  With M ordered,
// non-overlapping matches the text has M+1 gaps; gap i = [end[i-1], begin[i]) lands at
//     dst_i = gap_begin_i - (bytes removed before it) + i * with_len
// followed (for i < M) by the replacement.  `removed` = exclusive prefix sum of the match
// lengths, computed by the caller.
__global__ __launch_bounds__(256) void match_lengths(const uint64_t* spans, uint64_t m, uint64_t* len) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < m) len[i] = spans[2 * i + 1] - spans[2 * i];
}

constexpr uint64_t kLongGap = 4096;  // gaps longer than this are copied by the whole grid

__global__ __launch_bounds__(256) void replace_gather(const uint8_t* text, uint64_t n, const uint64_t* spans,
                                                      const uint64_t* removed, uint64_t m, const uint8_t* with,
                                                      uint64_t with_len, uint8_t* out, uint64_t out_cap,
                                                      uint64_t* long_gaps, unsigned long long* counters) {
  const int lane = lane_id();
  const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_waves = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 6;
  for (uint64_t i = wave; i <= m; i += n_waves) {  // one wave per gap (typically a line: ~60 bytes)
    const uint64_t gb = i == 0 ? 0 : spans[2 * (i - 1) + 1];
    const uint64_t ge = i == m ? n : spans[2 * i];
    const uint64_t rem = i == m ? (m ? removed[m - 1] + (spans[2 * (m - 1) + 1] - spans[2 * (m - 1)]) : 0) : removed[i];
    const uint64_t dst = gb - rem + i * with_len;
    const uint64_t len = ge - gb;
    if (len > kLongGap) {
      if (lane == 0) {  // rare: hand the gap to the grid-wide copy
        const unsigned long long k = atomicAdd(counters + kCntHits, 1ull);
        long_gaps[3 * k] = gb;
        long_gaps[3 * k + 1] = len;
        long_gaps[3 * k + 2] = dst;
      }
    } else {
      for (uint64_t k = lane; k < len; k += kWave)
        if (dst + k < out_cap) out[dst + k] = text[gb + k];
    }
    if (i < m)
      for (uint64_t k = lane; k < with_len; k += kWave)
        if (dst + len + k < out_cap) out[dst + len + k] = with[k];
    if (i == m && lane == 0) counters[kCntFinal] = dst + len;  // length of the result
  }
}

__global__ __launch_bounds__(256) void copy_long_gaps(const uint8_t* text, const uint64_t* long_gaps,
                                                      const unsigned long long* counters, uint8_t* out, uint64_t out_cap) {
  const uint64_t n_gaps = counters[kCntHits];
  const uint64_t tid = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t g = 0; g < n_gaps; g++) {
    const uint64_t src = long_gaps[3 * g], len = long_gaps[3 * g + 1], dst = long_gaps[3 * g + 2];
    for (uint64_t k = tid; k < len; k += stride)
      if (dst + k < out_cap) out[dst + k] = text[src + k];
  }
}









void launch_region_offsets(const uint32_t* counts, uint32_t n_regions, uint32_t cap, uint64_t* offsets,
                           unsigned long long* counters, hipStream_t st) {
  // n_regions <= 64 Ki (scan_geometry caps the grid at 16 Ki workgroups of 4 waves)
  if (n_regions <= 16 * 1024) hipLaunchKernelGGL((region_offsets<16>), dim3(1), dim3(1024), 0, st, counts, n_regions, cap, offsets, counters);
  else if (n_regions <= 32 * 1024) hipLaunchKernelGGL((region_offsets<32>), dim3(1), dim3(1024), 0, st, counts, n_regions, cap, offsets, counters);
  else hipLaunchKernelGGL((region_offsets<64>), dim3(1), dim3(1024), 0, st, counts, n_regions, cap, offsets, counters);
}

void launch_verify(const VerifyParams& a, const DevProgram& P, uint64_t expected_hits, hipStream_t st) {
  const int W = P.n_words;
  if (W <= 4 && P.mode == 0) {  // dense: long regions, one wave per region
    uint64_t blocks = (static_cast<uint64_t>(a.n_regions) + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    static const bool no_walkers = getenv("RJ_NO_WALKERS") != nullptr;  // measurement override
    if (P.table_words <= 12288 && !no_walkers) {  // tables fit 48 KiB of LDS: persistent walkers
      const size_t lds = static_cast<size_t>(P.table_words) * sizeof(uint32_t);
      if (W <= 2) hipLaunchKernelGGL((verify_walkers<1>), dim3(static_cast<unsigned>(blocks)), dim3(256), lds, st, a, P);
      else hipLaunchKernelGGL((verify_walkers<2>), dim3(static_cast<unsigned>(blocks)), dim3(256), lds, st, a, P);
      return;
    }
    if (W <= 2) hipLaunchKernelGGL((verify_lane_regions<1>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, a, P);
    else hipLaunchKernelGGL((verify_lane_regions<2>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, a, P);
    return;
  }
  if (W <= 4) {
    uint64_t blocks = (expected_hits + 255) / 256;
    if (blocks < 4) blocks = 4;
    if (blocks > 2048) blocks = 2048;
    if (W <= 2) hipLaunchKernelGGL((verify_lane<1>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, a, P);
    else hipLaunchKernelGGL((verify_lane<2>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, a, P);
    return;
  }
  uint64_t blocks = (expected_hits + 3) / 4;
  if (blocks < 4) blocks = 4;
  if (blocks > 2048) blocks = 2048;
  const unsigned g = static_cast<unsigned>(blocks);
  if (W <= 64) hipLaunchKernelGGL((verify_wave<1>), dim3(g), dim3(256), 0, st, a, P);
  else if (W <= 128) hipLaunchKernelGGL((verify_wave<2>), dim3(g), dim3(256), 0, st, a, P);
  else hipLaunchKernelGGL((verify_wave<4>), dim3(g), dim3(256), 0, st, a, P);
}

void launch_verify_in_regions(const VerifyParams& a, const DevProgram& P, const uint32_t* hit_counts, uint32_t* valid_counts,
                              uint64_t* region_ends, hipStream_t st) {
  uint64_t blocks = (static_cast<uint64_t>(a.n_regions) + 15) / 16;  // 16 lanes per region
  blocks = blocks < 1 ? 1 : blocks > 4096 ? 4096 : blocks;
  if (P.n_words <= 2) hipLaunchKernelGGL((verify_in_regions<1>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, a, P,
                                         hit_counts, valid_counts, region_ends);
  else hipLaunchKernelGGL((verify_in_regions<2>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, a, P, hit_counts,
                          valid_counts, region_ends);
}

// a number per launch of the gather kernels: its low 32 / low 26 bits tag the two granules of a workgroup, so a granule
// left by an earlier launch never matches (the same low 26 bits come round after 67 M launches); those bits are never
// 0, which is what a fresh counter block holds.  Bit 63 (RJ_OGC_NO_EXCHANGE, debugging): every workgroup adds the counts
// before it up itself.
static uint64_t next_ogc_epoch() {
  static std::atomic<uint64_t> epoch{0};
  static const uint64_t no_exchange = getenv("RJ_OGC_NO_EXCHANGE") ? 1ull << 63 : 0;
  uint64_t e = epoch.fetch_add(1, std::memory_order_relaxed) + 1;
  return (e & 0x3FFFFFFull) == 0 ? next_ogc_epoch() : (e | no_exchange);
}

void launch_offsets_gather_check(const uint32_t* counts, const uint64_t* region_begins, const uint64_t* region_ends,
                                 uint32_t n_regions, uint32_t region_cap, uint64_t carry_cur, uint64_t* out, uint64_t out_cap,
                                 unsigned long long* counters, unsigned long long* host_counters, uint64_t* offsets_scratch,
                                 uint64_t* prev_scratch, hipStream_t st, uint64_t carry_prev_end, int have_prev) {
  const uint64_t carry_pe = have_prev ? carry_prev_end : ~0ull;
  const unsigned blocks = n_regions ? (n_regions + kOgcThreads - 1) / kOgcThreads : 1u;
  hipLaunchKernelGGL(offsets_gather_check, dim3(blocks), dim3(kOgcThreads), 0, st, counts, region_begins, region_ends, n_regions,
                     region_cap, carry_cur, out, out_cap, counters, host_counters, carry_pe, offsets_scratch, prev_scratch, next_ogc_epoch());
  if (offsets_scratch != nullptr) {
    uint64_t wblocks = (static_cast<uint64_t>(n_regions) + 3) / 4;
    wblocks = wblocks < 1 ? 1 : wblocks > 16384 ? 16384 : wblocks;
    hipLaunchKernelGGL(gather_regions_by_wave, dim3(static_cast<unsigned>(wblocks)), dim3(256), 0, st, counts, region_begins,
                       region_ends, offsets_scratch, prev_scratch, n_regions, region_cap, out, out_cap, counters, host_counters, carry_pe);
  }
}

void launch_verify_behind_in_regions(const VerifyParams& a, const DevProgram& P, const DevProgram& R, const uint32_t* hit_counts,
                                     uint32_t* valid_counts, uint64_t* region_ends, hipStream_t st) {
  uint64_t blocks = (static_cast<uint64_t>(a.n_regions) + 15) / 16;  // 16 lanes per region
  blocks = blocks < 1 ? 1 : blocks > 4096 ? 4096 : blocks;
  const uint32_t both = P.table_words + R.table_words;
  const uint32_t lds_words = both <= 12288 ? both : 0;
  const size_t lds = static_cast<size_t>(lds_words) * sizeof(uint32_t);
  const dim3 g(static_cast<unsigned>(blocks)), b(256);
  if (P.n_words <= 1) hipLaunchKernelGGL((verify_behind_in_regions<1, 1>), g, b, lds, st, a, P, R, hit_counts, valid_counts, region_ends, lds_words);
  else if (P.n_words == 2) hipLaunchKernelGGL((verify_behind_in_regions<2, 1>), g, b, lds, st, a, P, R, hit_counts, valid_counts, region_ends, lds_words);
  else hipLaunchKernelGGL((verify_behind_in_regions<4, 2>), g, b, lds, st, a, P, R, hit_counts, valid_counts, region_ends, lds_words);
}

void launch_verify_floating_in_regions(const VerifyParams& a, const DevProgram& P, const uint32_t* hit_counts,
                                       uint32_t* valid_counts, uint64_t* region_begins, uint64_t* region_ends, hipStream_t st) {
  uint64_t blocks = (static_cast<uint64_t>(a.n_regions) + 3) / 4;  // a wave per region
  blocks = blocks < 1 ? 1 : blocks > 4096 ? 4096 : blocks;
  const uint32_t float_min = P.float_max + 1 - P.float_range;
  const uint32_t lds_words = P.table_words <= 12288 ? P.table_words : 0;
  const size_t lds = static_cast<size_t>(lds_words) * sizeof(uint32_t);
  if (P.n_words <= 2)
    hipLaunchKernelGGL((verify_floating_in_regions<1>), dim3(static_cast<unsigned>(blocks)), dim3(256), lds, st, a, P,
                       hit_counts, valid_counts, region_begins, region_ends, float_min, lds_words);
  else
    hipLaunchKernelGGL((verify_floating_in_regions<2>), dim3(static_cast<unsigned>(blocks)), dim3(256), lds, st, a, P,
                       hit_counts, valid_counts, region_begins, region_ends, float_min, lds_words);
}

void launch_tails_multi(const MultiTail* d_tails, int n_patterns, uint32_t n_regions, hipStream_t st) {
  uint64_t vblocks = (static_cast<uint64_t>(n_regions) + 63) / 64;  // 4 lanes per region
  vblocks = vblocks < 1 ? 1 : vblocks > 4096 ? 4096 : vblocks;
  hipLaunchKernelGGL(verify_in_regions_multi, dim3(static_cast<unsigned>(vblocks), n_patterns), dim3(256), 0, st, d_tails);
  launch_offsets_gather_check_multi(d_tails, n_patterns, n_regions, st);
}

void launch_offsets_gather_check_multi(const MultiTail* d_tails, int n_patterns, uint32_t n_regions, hipStream_t st) {
  const unsigned gblocks = n_regions ? (n_regions + kOgcThreads - 1) / kOgcThreads : 1u;
  hipLaunchKernelGGL(offsets_gather_check_multi, dim3(gblocks, n_patterns), dim3(kOgcThreads), 0, st, d_tails, next_ogc_epoch());
}

void launch_split_pairs(const uint64_t* pairs, const unsigned long long* n_ptr, uint64_t n_upper, uint64_t* keys, uint64_t* vals,
                        hipStream_t st) {
  uint64_t blocks = (n_upper + 255) / 256;
  blocks = blocks < 1 ? 1 : blocks > 4096 ? 4096 : blocks;
  hipLaunchKernelGGL(split_pairs, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, pairs, n_ptr, keys, vals);
}

void launch_mark_valid(const uint64_t* cand_end, uint64_t n, uint64_t* flags, hipStream_t st) {
  if (n == 0) return;
  hipLaunchKernelGGL(mark_valid, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, cand_end, n, flags);
}

void launch_compact_valid(const uint64_t* cand_begin, const uint64_t* cand_end, const uint64_t* flags,
                          const uint64_t* pos, uint64_t n, uint64_t* keys, uint64_t* vals,
                          unsigned long long* counters, hipStream_t st) {
  if (n == 0) return;
  hipLaunchKernelGGL(compact_valid, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, cand_begin, cand_end,
                     flags, pos, n, keys, vals, counters);
}

void launch_match_full(const uint8_t* text, uint64_t n, const DevProgram& P, int* result, hipStream_t st) {
  const int W = P.n_words;
  if (W <= 64) hipLaunchKernelGGL((match_full<1>), dim3(1), dim3(64), 0, st, text, n, P, result);
  else if (W <= 128) hipLaunchKernelGGL((match_full<2>), dim3(1), dim3(64), 0, st, text, n, P, result);
  else hipLaunchKernelGGL((match_full<4>), dim3(1), dim3(64), 0, st, text, n, P, result);
}


void launch_check_and_interleave(const uint64_t* keys, const uint64_t* vals, const unsigned long long* n_ptr,
                                 uint64_t n_upper, uint64_t carry_cur, uint64_t* out, uint64_t cap, unsigned long long* unordered,
                                 hipStream_t st) {
  uint64_t blocks = (n_upper + 255) / 256;  // grid-stride: an estimate of n is enough
  blocks = blocks < 1 ? 1 : blocks > 4096 ? 4096 : blocks;
  hipLaunchKernelGGL(check_and_interleave, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, keys, vals,
                     n_ptr, carry_cur, out, cap, unordered);
}


void launch_exact_sequential(const uint8_t* text, uint64_t n, const DevGraph& G, int64_t* ring, uint64_t* out,
                             uint64_t out_cap, unsigned long long* counters, hipStream_t st) {
  hipLaunchKernelGGL(exact_sequential, dim3(1), dim3(64), 0, st, text, n, G, ring, out, out_cap, counters);
}

void launch_match_lengths(const uint64_t* spans, uint64_t m, uint64_t* len, hipStream_t st) {
  if (m == 0) return;
  hipLaunchKernelGGL(match_lengths, dim3(static_cast<unsigned>((m + 255) / 256)), dim3(256), 0, st, spans, m, len);
}

void launch_replace_gather(const uint8_t* text, uint64_t n, const uint64_t* spans, const uint64_t* removed, uint64_t m,
                           const uint8_t* with, uint64_t with_len, uint8_t* out, uint64_t out_cap, uint64_t* long_gaps,
                           unsigned long long* counters, hipStream_t st) {
  uint64_t blocks = (m + 1 + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(replace_gather, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, text, n, spans, removed, m, with,
                     with_len, out, out_cap, long_gaps, counters);
  hipLaunchKernelGGL(copy_long_gaps, dim3(2048), dim3(256), 0, st, text, long_gaps, counters, out, out_cap);
}







}  // namespace rejit_amd
