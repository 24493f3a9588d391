// rejit_amd/csrc/kernel_util.h -- what the kernel translation units that grew out of kernels.hip share (round 6: kernels.hip was
// 3264 lines and a minute of compile time on its own): wave-level helpers, the hit regions a scanning wave appends to, the span
// of 1-KiB chunks a wave owns, guarded chunk loads.  Everything lives in an anonymous namespace: every unit gets its own copy.
#ifndef REJIT_AMD_KERNEL_UTIL_H_
#define REJIT_AMD_KERNEL_UTIL_H_

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdint>

#include "device_program.h"
#include "kernels.h"

namespace rejit_amd {

namespace {

constexpr int kWave = 64;
constexpr int kChunk = 1024;  // bytes per wave iteration: 64 lanes x 16 B

__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x) & (kWave - 1); }

// Cross-lane moves of the dense kernel through DPP (data-parallel primitives: the operand of a VALU
// instruction comes from another lane of the wave, no LDS crossbar round trip as with ds_bpermute, which
// is what __shfl_up / __shfl_down compile to).  gfx9 family: row_shr within rows of 16 lanes, row_bcast:15 /
// row_bcast:31 to carry a row's total into the next rows, wave_shl / wave_shr by one lane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t x) {
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), CTRL, ROW_MASK, 0xF, true));
}
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x) {
  x += dpp_or_zero<0x111, 0xF>(x);  // row_shr:1
  x += dpp_or_zero<0x112, 0xF>(x);  // row_shr:2
  x += dpp_or_zero<0x114, 0xF>(x);  // row_shr:4
  x += dpp_or_zero<0x118, 0xF>(x);  // row_shr:8
  x += dpp_or_zero<0x142, 0xA>(x);  // row_bcast:15 into rows 1 and 3
  x += dpp_or_zero<0x143, 0xC>(x);  // row_bcast:31 into rows 2 and 3
  return x;
}
__device__ __forceinline__ uint32_t wave_from_lane_below(uint32_t x) { return dpp_or_zero<0x138, 0xF>(x); }  // wave_shr:1, lane 0 gets 0
__device__ __forceinline__ uint32_t wave_from_lane_above(uint32_t x) { return dpp_or_zero<0x130, 0xF>(x); }  // wave_shl:1, lane 63 gets 0
__device__ __forceinline__ uint32_t wave_last_lane(uint32_t x) { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(x), kWave - 1)); }

// Hit offsets of one wave go to the wave's own REGION of the hit list: region w = wave w,
// `cap` entries, filled in position order, no atomics.  Every wave owns a contiguous span of
// the text, so the regions concatenated in wave order are globally sorted by offset -- which
// is what lets the rest of the pipeline run without a sort.
//
// History (all measured, tools/ab_probe.py): appending chunk by chunk to one global counter ran
// into the ~90 atomics/us limit of a single address (11 ns per chunk-with-hits); staging 256
// hits per wave in LDS fixed that but cost one returning atomic per wave, whose latency under
// a saturated memory pipeline (~20 us) made the kernel slower the more waves it had; 16 sharded
// counters helped little.  Regions need no atomic at all.
struct RegionHits {
  uint64_t* slots;   // this wave's region
  uint32_t cap;
  uint32_t count;    // wave-uniform; keeps counting past cap so the host can size a retry

  // bit j of mask16 <-> offset at + j - bias; appended in position order
  __device__ __forceinline__ void push_bits(uint32_t mask16, uint64_t at, uint64_t bias) {
    const int lane = lane_id();
    const uint32_t cnt = __popc(mask16);
    uint32_t inc, total;
    uint64_t hitters = __ballot(cnt != 0);
    if (__popcll(hitters) <= 4) {
      // the usual case in window scans: a handful of lanes hold hits -- walk them (scalar loop,
      // v_readlane) instead of a 6-step wave scan
      uint32_t before = 0;
      total = 0;
      while (hitters) {
        const int l = __builtin_ctzll(hitters);
        hitters &= hitters - 1;
        const uint32_t c = __builtin_amdgcn_readlane(cnt, l);
        before += lane > l ? c : 0u;
        total += c;
      }
      inc = before + cnt;
    } else {
      inc = cnt;
#pragma unroll
      for (int o = 1; o < kWave; o <<= 1) {
        const uint32_t v = __shfl_up(inc, o);
        if (lane >= o) inc += v;
      }
      total = __shfl(inc, kWave - 1);
    }
    uint32_t idx = count + inc - cnt;
    while (mask16) {
      const int j = __ffs(static_cast<int>(mask16)) - 1;
      mask16 &= mask16 - 1;
      if (idx < cap) slots[idx] = at + j - bias;
      idx++;
    }
    count += total;
  }
};

// geometry shared by the scan kernels: wave w owns chunks [c0, c1)
struct WaveSpan {
  uint64_t c0, c1;
};

__device__ __forceinline__ uint64_t scalar_wave_index() {
  // as a scalar, so that chunk addresses and loop branches are wave-uniform
  return __builtin_amdgcn_readfirstlane(
      static_cast<uint32_t>((static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6));
}

__device__ __forceinline__ WaveSpan wave_span(const ScanParams& a, uint64_t wave, uint64_t first_chunk,
                                              uint64_t end_chunk) {
  WaveSpan w;
  w.c0 = first_chunk + wave * a.span_chunks;
  w.c1 = w.c0 + a.span_chunks;
  if (w.c0 > end_chunk) w.c0 = end_chunk;
  if (w.c1 > end_chunk) w.c1 = end_chunk;
  return w;
}

// 16 B of the lane + the 8 B that follow, guarded against the end of the text (tail chunk).
__device__ __forceinline__ void load_guarded(const uint8_t* text, uint64_t n, uint64_t at, uint32_t d[6]) {
#pragma unroll
  for (int q = 0; q < 6; q++) {
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint64_t p = at + 4 * q + k;
      if (p < n) v |= static_cast<uint32_t>(text[p]) << (8 * k);
    }
    d[q] = v;
  }
}

}  // namespace

}  // namespace rejit_amd
#endif
