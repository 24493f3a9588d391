// rejit_amd/csrc/tile_lookback.h -- the decoupled look-back (Merrill & Garland's single-pass prefix scan) shared by the
// kernels that write their matches once, at their final place (emit_scan.hip, dense_streams.hip).  Device code only.
//
// A granule is 8 bytes {status: 2 bits, value: 62 bits}, written and read with relaxed agent-scope atomics (the data is the
// flag).  Granule g belongs to the g-th unit of work in ARRIVAL order (a ticket), so every granule a wave waits for is
// owned by a workgroup that has started.  Every spin is bounded: false = timed out, the run is void.
#ifndef REJIT_AMD_TILE_LOOKBACK_H_
#define REJIT_AMD_TILE_LOOKBACK_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rejit_amd {
namespace lookback {

constexpr unsigned long long kStatusAggregate = 1ull << 62, kStatusInclusive = 2ull << 62, kValueMask = (1ull << 62) - 1;
constexpr uint32_t kSpinLimit = 1u << 22;
constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x) & (kWave - 1); }

// ONE WAVE: publish unit t's count k, return in *before the count of all units before t
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


// nobody must wait for a unit whose run is void
__device__ __forceinline__ void publish_void(unsigned long long* granules, uint64_t t) {
  __hip_atomic_store(&granules[t], kStatusInclusive | 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace lookback
}  // namespace rejit_amd
#endif
