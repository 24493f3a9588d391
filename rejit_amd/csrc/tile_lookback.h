// rejit_amd/csrc/tile_lookback.h -- the single-pass prefix scan (after Merrill & Garland's decoupled look-back) shared by
// the kernels that write their matches once, at their final place (emit_scan.hip, dense_streams.hip).  Device code only.
//
// Unit t is the t-th unit of work in ARRIVAL order (a ticket), so everything a wave waits for is owned by a workgroup that
// has started.  All words are 8 bytes, written and read with relaxed agent-scope atomics (the data is the flag; each word
// is valid on its own, no ordering between words is assumed).  Every spin is bounded: false = timed out, the run is void.
//
// What round 4 measured on the one-level look-back (a {status, value} granule per unit, windows of 64).  These kernels are
// persistent: ~1800 workgroups are resident, take their tickets one after the other and finish a round nearly in
// lock-step.  (1) When a unit's count is ready, hardly any unit before it has its inclusive prefix yet, so the look-back
// walked over all ~1800 units in flight: 28 dependent trips to L2 per round.  Reading 512 granules per trip made it worse
// (`[a-f]+[0-9]` over 5 GB: 1.62 -> 2.45 ms; every spin of every waiting wave re-reads its window).  Two levels (below)
// make it two trips: 1.53 ms.  (2) With the look-back switched off the kernel takes 1.17 ms: most of the cost is not the
// trips but WAITING -- a unit needs the counts of all units before it, the ~1800 of them started at most one round
// earlier, and the slowest decides: a device-wide barrier per round in all but name.
//
// Hence the protocol is split in two calls, and the kernels call the second one a full round late:
//   publish(t, k)        when unit t's count is known: units[t] = {1, k}, and k is added to the unit's GROUP word (64
//                        consecutive units form a group: {arrivals: 8 bits, sum: 56 bits}, complete at 64 arrivals);
//   resolve(t, &before)  one round LATER (the workgroup has computed -- and published -- its next unit meanwhile, the pairs
//                        of unit t wait in LDS): before = the units of t's own group before t (one trip) + the groups before
//                        it: prefix[g], worked out by the group's first unit from the complete group words back to the
//                        nearest group whose prefix is known (the ~60 groups in flight fit one window: one trip).
// By then everything asked for has been published long ago: two trips, no waiting, whatever the number of units in flight.
#ifndef REJIT_AMD_TILE_LOOKBACK_H_
#define REJIT_AMD_TILE_LOOKBACK_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rejit_amd {
namespace lookback {

constexpr uint32_t kSpinLimit = 1u << 22;
constexpr int kWave = 64;
constexpr uint64_t kGroup = 64;                           // units per group
constexpr unsigned long long kHave = 1ull << 62;          // units[], prefix[]: the word has been written
constexpr unsigned long long kValueMask = (1ull << 62) - 1;
constexpr int kArrivalShift = 56;                         // groups[]: arrivals << 56 | sum of the counts
constexpr unsigned long long kSumMask = (1ull << kArrivalShift) - 1;

__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x) & (kWave - 1); }

// words a run of n_units units needs: units[n_units], groups[n_groups], prefix[n_groups] -- all zero at the start
__host__ __device__ inline uint64_t n_groups(uint64_t n_units) { return (n_units + kGroup - 1) / kGroup; }
__host__ __device__ inline uint64_t granule_words(uint64_t n_units) { return n_units + 2 * n_groups(n_units); }

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}

// ONE LANE: unit t holds k matches (a void unit publishes 0: nobody must wait for it)
__device__ __forceinline__ void publish(unsigned long long* units, uint64_t n_units, uint64_t t, unsigned long long k) {
  __hip_atomic_store(&units[t], kHave | k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_add(&units[n_units + t / kGroup], (1ull << kArrivalShift) | k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ONE WAVE: the count of all groups before g: complete group words, nearest first (lane l: group window_end - 1 - l), back
// to a group whose exclusive prefix is known; leaves prefix[g] behind.
__device__ __forceinline__ bool groups_before(unsigned long long* groups, unsigned long long* prefix, uint64_t g, unsigned long long* out) {
  const int lane = lane_id();
  unsigned long long total = 0;
  uint64_t window_end = g;
  while (window_end != 0) {
    const bool valid = window_end >= static_cast<uint64_t>(lane) + 1;
    const uint64_t h = valid ? window_end - 1 - static_cast<uint64_t>(lane) : 0;
    unsigned long long sum = 0, pre = 0;  // group h: its complete sum; its exclusive prefix (kHave | value)
    bool complete = !valid;
    uint32_t spins = 0;
    uint64_t known = 0;
    for (;;) {
      if (valid && !complete) {
        const unsigned long long w = __hip_atomic_load(&groups[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        complete = (w >> kArrivalShift) == kGroup;
        sum = w & kSumMask;
      }
      if (valid && pre == 0) pre = h == 0 ? kHave : __hip_atomic_load(&prefix[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      known = __ballot(valid && pre != 0);
      const uint64_t upto = known ? (known & (0 - known)) : 0;       // the nearest group with a known prefix
      const uint64_t needed = upto ? (upto | (upto - 1)) : ~0ull;    // it and everything nearer must be complete
      if ((__ballot(!complete) & needed) == 0) break;
      if (++spins > kSpinLimit) return false;
      __builtin_amdgcn_s_sleep(1);
    }
    const int stop = known ? __builtin_ctzll(known) : kWave;
    unsigned long long part = valid && lane <= stop ? sum : 0ull;
    if (lane == stop) part += pre & kValueMask;
    total += wave_sum(part);
    if (known) break;
    window_end = window_end > static_cast<uint64_t>(kWave) ? window_end - kWave : 0;
  }
  if (lane == 0 && g != 0) __hip_atomic_store(&prefix[g], kHave | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  *out = total;
  return true;
}

// ONE WAVE: the count of all units before t (every one of them has been, or will be, published).
// Only a group's FIRST unit looks back over the groups (and leaves prefix[g] for the other 63, who read it in the same trip
// as the words of their own group): when every unit did, ~1800 waves read the same 64 group words at the same moment --
// eight cache lines behind ONE memory channel -- and the resolve took longer than the wait it was meant to avoid
// (`[@#]` over 5 GB: 1.43 -> 1.93 ms).  A unit that has its own group's words but still no prefix[g] after kPatience
// polls works it out itself (the first unit's workgroup may be a round behind: at the end of the run, or when it was slow).
constexpr uint32_t kPatience = 16;
__device__ __forceinline__ bool resolve(unsigned long long* units, uint64_t n_units, uint64_t t, unsigned long long* before) {
  const int lane = lane_id();
  unsigned long long* groups = units + n_units;
  unsigned long long* prefix = groups + n_groups(n_units);
  const uint64_t g = t / kGroup;
  const uint32_t i = static_cast<uint32_t>(t % kGroup);
  if (i == 0) return groups_before(groups, prefix, g, before);
  // lane l < i: unit 64 g + l of the own group; lane 63: the group's exclusive prefix.  A lane whose word has arrived keeps it.
  const bool unit_lane = static_cast<uint32_t>(lane) < i, prefix_lane = lane == kWave - 1;
  unsigned long long v = prefix_lane && g == 0 ? kHave : 0ull;
  const unsigned long long* at = unit_lane ? &units[g * kGroup + static_cast<uint64_t>(lane)] : &prefix[g];
  uint32_t spins = 0;
  bool have_prefix = false;
  for (;;) {
    if ((unit_lane || prefix_lane) && v == 0) v = __hip_atomic_load(at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool units_here = __ballot(unit_lane && v == 0) == 0;
    have_prefix = __ballot(prefix_lane && v == 0) == 0;
    if (units_here && (have_prefix || spins >= kPatience)) break;
    if (++spins > kSpinLimit) return false;
    __builtin_amdgcn_s_sleep(1);
  }
  unsigned long long sum = wave_sum((unit_lane || (prefix_lane && have_prefix)) ? (v & kValueMask) : 0ull);
  if (!have_prefix) {
    unsigned long long gb = 0;
    if (!groups_before(groups, prefix, g, &gb)) return false;
    sum += gb;
  }
  *before = sum;
  return true;
}

}  // namespace lookback
}  // namespace rejit_amd
#endif
