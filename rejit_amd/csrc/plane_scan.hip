// rejit_amd/csrc/plane_scan.hip -- the one-pass fast-forward scan for SEVERAL patterns (regexdna's nine
// counts, reference sample/regexdna.cc:51-67, which runs FastForwardGen's multi-literal scan,
// src/x64/codegen-x64.cc:1102-1252, once per pattern) as a BIT-PLANE kernel, and the classification of its
// candidates.
//
// Shape it serves (host check: multi_pattern.hip): every pattern has <= 2 windows of 8 bytes at the same
// offset, and every window lies within ONE byte of one of <= 2 base windows without wildcards -- all 18
// regexdna windows are `agggtaaa` or `tttaccct` with at most one position turned into a class.  A text
// position can then only begin a window of ANY pattern if it differs from a base in at most one byte: one
// test for all patterns, and the reference's idea of scanning for what the alternatives have in common
// (FF_finder::ff_alternation_reduce, src/codegen.cc:395-531) carried one step further.
//
// How the test is evaluated.  The kernel of round 2 (scan_windows_fused, kernels.hip) built the window of
// every position in a register and counted differing nibbles: 7 VALU instructions per position, 8.6 per text
// byte with the packing -- VALU-bound at 0.23 of HBM.  Here the text is turned "vertical":
//   * every byte becomes a 2-bit symbol code (byte >> code_shift) & 3 -- a, c, g, t -> 0, 1, 3, 2 -- and four
//     codes pack into a byte with ONE v_dot4_u32_u8 ((d & 0x06060606) . (1, 4, 16, 64));  any byte aliases to
//     some symbol, which makes the test a superset test like every fast-forward filter (the automaton removes
//     the aliases);
//   * the low and the high code bits of two 1-KiB chunks A and B are interleaved into two plane registers per
//     lane: bit 2k = position k of the lane's 16 bytes of A, bit 2k + 1 = position k of its 16 bytes of B;
//   * window byte i of a base is the symbol (l_i, h_i): positions where the text shifted by i matches it are
//     E_i = ~(L_i ^ l_i) & ~(H_i ^ h_i) with L_i = v_alignbit(L_halo, L, 2 i) -- both chunks in one shift;
//   * "no mismatch so far" Z and "at most one" O are carried over the eight window bytes: Z' = Z & E,
//     O' = Z | (O & E) -- one v_bitop3 each.
// 32 positions per instruction: 118 VALU instructions per 2 KiB and wave = 3.7 lane-operations per text byte
// for ALL patterns (the single-pattern nibble scan needs 5.4), so the kernel streams at the HBM rate.
// Candidates (one per ~1.3 KiB on DNA) go to ONE list shared by all patterns, in the wave's own region like
// every hit list of this engine (no atomics, sorted by construction).
//
// classify_shared_multi then takes the place of verify_in_regions_multi: a wave per region; every candidate's
// window bytes are loaded once and tested against the windows of pattern after pattern (exact, with wildcards);
// the automaton (device_program.h) runs only for the patterns whose window matches -- one of nine for a
// candidate one byte off a base -- and the survivors are compacted into that pattern's own region, from where
// offsets_gather_check_multi lays them out as before.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "trace_stamp.h"
RJ_TRACE_EXPORT(rj_debug_trace_plane)

#include <cstdlib>
#include <type_traits>

#include "device_program.h"
#include "kernels.h"
#include "short_walk.h"

namespace rejit_amd {

namespace {

constexpr int kWave = 64;
constexpr uint64_t kChunk = 1024;   // bytes of one chunk: 64 lanes x 16 B
constexpr uint64_t kPair = 2048;    // two chunks per wave iteration (the planes interleave them)

__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x) & (kWave - 1); }

__device__ __forceinline__ uint32_t lanes_below(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}

// 16 B of the lane + the 8 B that follow (the neighbour lane's first bytes: same cache lines, L1 hits)
__device__ __forceinline__ void load_chunk24(const uint8_t* text, uint64_t at, uint32_t (&d)[6]) {
  const uint4 v = *reinterpret_cast<const uint4*>(text + at);
  const uint2 h = *reinterpret_cast<const uint2*>(text + at + 16);
  d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  d[4] = h.x; d[5] = h.y;
}

__device__ __forceinline__ void load_guarded24(const uint8_t* text, uint64_t n, uint64_t at, uint32_t (&d)[6]) {
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

struct PlaneConsts {
  uint32_t cmask;   // 0x03030303 << code_shift
  uint32_t shift;   // code_shift
};

// the 2-bit codes of a dword's four bytes as one byte (times 2^shift)
__device__ __forceinline__ uint32_t codes4(uint32_t d, const PlaneConsts& k) {
  return __builtin_amdgcn_udot4(d & k.cmask, 0x40100401u, 0u, false);
}

// Candidate positions of one pair of chunks: bit 2k = position k of the lane's 16 bytes of chunk A, bit
// 2k + 1 = position k of its 16 bytes of chunk B.
template <int NB>
__device__ __forceinline__ uint32_t plane_candidates(const uint32_t (&dA)[6], const uint32_t (&dB)[6], const PlaneConsts& k,
                                                     const PlaneParams& a) {
  const uint32_t s = k.shift;
  const uint32_t ta = (codes4(dA[0], k) >> s) | (codes4(dA[1], k) << (8 - s)) | (codes4(dA[2], k) << (16 - s)) | (codes4(dA[3], k) << (24 - s));
  const uint32_t tb = (codes4(dB[0], k) >> s) | (codes4(dB[1], k) << (8 - s)) | (codes4(dB[2], k) << (16 - s)) | (codes4(dB[3], k) << (24 - s));
  const uint32_t ha = (codes4(dA[4], k) >> s) | (codes4(dA[5], k) << (8 - s));   // the 8 positions that follow
  const uint32_t hb = (codes4(dB[4], k) >> s) | (codes4(dB[5], k) << (8 - s));
  constexpr uint32_t kEven = 0x55555555u;
  // planes: low / high code bit, A in the even bits, B in the odd ones (v_bfi_b32)
  const uint32_t L = (ta & kEven) | ((tb << 1) & ~kEven);
  const uint32_t H = ((ta >> 1) & kEven) | (tb & ~kEven);
  const uint32_t Ln = (ha & kEven) | ((hb << 1) & ~kEven);
  const uint32_t Hn = ((ha >> 1) & kEven) | (hb & ~kEven);
  uint32_t Z[NB], O[NB];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint32_t Li = i ? __builtin_amdgcn_alignbit(Ln, L, 2 * i) : L;
    const uint32_t Hi = i ? __builtin_amdgcn_alignbit(Hn, H, 2 * i) : H;
#pragma unroll
    for (int b = 0; b < NB; b++) {
      const uint32_t E = (Li ^ a.lo[b][i]) & (Hi ^ a.hi[b][i]);
      if (i == 0) {
        Z[b] = E;
      } else if (i == 1) {
        O[b] = Z[b] | E;
        Z[b] &= E;
      } else {
        O[b] = Z[b] | (O[b] & E);
        if (i < 7) Z[b] &= E;
      }
    }
  }
  return NB > 1 ? (O[0] | O[NB - 1]) : O[0];
}

struct SharedRegion {
  uint64_t* slots;
  uint32_t cap;
  uint32_t count;  // wave-uniform; keeps counting past cap so that the host can size a retry
};

// inclusive prefix sum over the wave (DPP row shifts + row broadcasts, see kernels.hip)
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

// Append the candidates of a pair in position order: all of chunk A (even bits), then all of chunk B.
// `at` = byte offset of the lane's 16 bytes of chunk A; the slot holds the candidate START (w - bias).
__device__ __forceinline__ void push_pair(SharedRegion& r, uint32_t hm, uint64_t at, uint64_t bias) {
  const uint32_t hA = hm & 0x55555555u, hB = hm & 0xAAAAAAAAu;
  const uint64_t mA = __ballot(hA != 0), mB = __ballot(hB != 0);
  const uint64_t several = __ballot(((hA & (hA - 1)) | (hB & (hB - 1))) != 0);
  if (several == 0) {
    // the usual case: no lane holds two candidates of one chunk -- ranks straight from the lane masks
    const uint32_t nA = __popcll(mA), nB = __popcll(mB);
    if (hA != 0) {
      const uint32_t idx = r.count + lanes_below(mA);
      if (idx < r.cap) r.slots[idx] = at + (static_cast<uint32_t>(__builtin_ctz(hA)) >> 1) - bias;
    }
    if (hB != 0) {
      const uint32_t idx = r.count + nA + lanes_below(mB);
      if (idx < r.cap) r.slots[idx] = at + kChunk + (static_cast<uint32_t>(__builtin_ctz(hB)) >> 1) - bias;
    }
    r.count += nA + nB;
    return;
  }
  const uint32_t cA = __popc(hA), cB = __popc(hB);
  const uint32_t incA = wave_inclusive_sum(cA), incB = wave_inclusive_sum(cB);
  const uint32_t totA = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incA), kWave - 1));
  const uint32_t totB = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incB), kWave - 1));
  uint32_t idx = r.count + incA - cA;
  for (uint32_t m = hA; m; m &= m - 1, idx++)
    if (idx < r.cap) r.slots[idx] = at + (static_cast<uint32_t>(__builtin_ctz(m)) >> 1) - bias;
  idx = r.count + totA + incB - cB;
  for (uint32_t m = hB; m; m &= m - 1, idx++)
    if (idx < r.cap) r.slots[idx] = at + kChunk + (static_cast<uint32_t>(__builtin_ctz(m)) >> 1) - bias;
  r.count += totA + totB;
}

// (No clipping here: window positions of a boundary pair that lie outside [wlo, whi) -- before the range, or
// with window bytes beyond the end of the text, where the guarded loads read zeros -- may be reported;
// classify_shared_multi drops them.  The test for it took more registers than the scan itself.)
template <int NB>
__device__ __forceinline__ void plane_pair(const uint32_t (&dA)[6], const uint32_t (&dB)[6], uint64_t at, const PlaneConsts& k,
                                           const PlaneParams& a, SharedRegion& region) {
  const uint32_t hm = plane_candidates<NB>(dA, dB, k, a);
  if (__ballot(hm != 0) == 0) return;  // wave-uniform
  push_pair(region, hm, at, a.offset);
}

}  // namespace

template <int NB>
__global__ __launch_bounds__(256) void plane_scan(PlaneParams a) {
  const int lane = lane_id();
  const uint64_t wave = __builtin_amdgcn_readfirstlane(
      static_cast<uint32_t>((static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6));
  if (wave == 0 && lane < kCntSize)
    for (uint32_t p = 0; p < a.n_zero; p++) a.zero_counters[p][lane] = 0;
  PlaneConsts k;
  k.shift = a.code_shift;
  k.cmask = 0x03030303u << a.code_shift;
  SharedRegion region{a.hits + wave * a.region_cap, a.region_cap, 0u};
  // window positions: w = s + offset, sb <= s < se, and the 8 window bytes must lie inside the text
  const uint64_t wlo = a.sb + a.offset;
  const uint64_t last_w = a.n >= 8 ? a.n - 8 + 1 : 0;
  uint64_t whi = a.se + a.offset;
  if (whi > last_w) whi = last_w;
  const uint64_t first_pair = wlo / kPair;
  const uint64_t end_pair = whi > wlo ? (whi + kPair - 1) / kPair : first_pair;
  uint64_t c0 = first_pair + wave * a.span_pairs, c1 = c0 + a.span_pairs;
  if (c0 > end_pair) c0 = end_pair;
  if (c1 > end_pair) c1 = end_pair;
  // pairs below fast_end can be loaded without guards (2 KiB + 8 B of halo stay < n)
  uint64_t fast_end = a.n >= kPair + 8 ? (a.n - 8) / kPair : 0;
  if (fast_end > c1) fast_end = c1;
  if (fast_end < c0) fast_end = c0;
  const uint64_t lane_off = static_cast<uint64_t>(lane) * 16;
  {
    // two pairs deep: while a pair is evaluated the loads of the next two are in flight (unconditional
    // prologue, so that the compiler can count the loads -- see scan_windows_body in kernels.hip)
    uint32_t a0[6], b0[6], a1[6], b1[6];
    uint64_t c = c0;
    if (c + 3 < fast_end) {
      load_chunk24(a.text, c * kPair + lane_off, a0);
      load_chunk24(a.text, c * kPair + kChunk + lane_off, b0);
      load_chunk24(a.text, (c + 1) * kPair + lane_off, a1);
      load_chunk24(a.text, (c + 1) * kPair + kChunk + lane_off, b1);
      while (c + 3 < fast_end) {
        plane_pair<NB>(a0, b0, c * kPair + lane_off, k, a, region);
        load_chunk24(a.text, (c + 2) * kPair + lane_off, a0);
        load_chunk24(a.text, (c + 2) * kPair + kChunk + lane_off, b0);
        __builtin_amdgcn_sched_barrier(0);
        plane_pair<NB>(a1, b1, (c + 1) * kPair + lane_off, k, a, region);
        load_chunk24(a.text, (c + 3) * kPair + lane_off, a1);
        load_chunk24(a.text, (c + 3) * kPair + kChunk + lane_off, b1);
        __builtin_amdgcn_sched_barrier(0);
        c += 2;
      }
      plane_pair<NB>(a0, b0, c * kPair + lane_off, k, a, region);
      plane_pair<NB>(a1, b1, (c + 1) * kPair + lane_off, k, a, region);
      c += 2;
    }
    for (; c < fast_end; c++) {
      load_chunk24(a.text, c * kPair + lane_off, a0);
      load_chunk24(a.text, c * kPair + kChunk + lane_off, b0);
      plane_pair<NB>(a0, b0, c * kPair + lane_off, k, a, region);
    }
  }
  // tail: the pair(s) that touch the end of the text use guarded byte loads
  for (uint64_t t = fast_end; t < c1; t++) {
    uint32_t dA[6], dB[6];
    load_guarded24(a.text, a.n, t * kPair + lane_off, dA);
    load_guarded24(a.text, a.n, t * kPair + kChunk + lane_off, dB);
    plane_pair<NB>(dA, dB, t * kPair + lane_off, k, a, region);
  }
  if (lane == 0) a.hit_counts[wave] = region.count;
}

void launch_plane_scan(const PlaneParams& a, int grid, hipEvent_t t0, hipEvent_t t1, hipStream_t st) {
  // RJ_PLANE_LDS (measurement): unused dynamic LDS per workgroup, to cap the kernel's residency (160 KiB per CU) and leave
  // wave slots to the tails of the previous step that run beside it
  static const size_t lds = getenv("RJ_PLANE_LDS") ? static_cast<size_t>(atoi(getenv("RJ_PLANE_LDS"))) : 0;
  if (a.n_bases <= 1) hipExtLaunchKernelGGL((plane_scan<1>), dim3(grid), dim3(256), lds, st, t0, t1, 0, a);
  else hipExtLaunchKernelGGL((plane_scan<2>), dim3(grid), dim3(256), lds, st, t0, t1, 0, a);
}

// ---------------------------------------------------------------------------------------
// Classification of the shared candidates: HALF a wave per region (a region holds ~25 candidates on DNA), 32
// candidates per round and region.  The patterns are taken one after the other (wave-uniform loop); the lanes
// whose candidate passes the pattern's exact window test run its automaton, and the survivors go, in position
// order, to the pattern's own region (begins / ends) -- what verify_in_regions_multi leaves behind for
// offsets_gather_check_multi.
//
// Everything a candidate needs except its own text lives in LDS: every workgroup copies, once, ONE contiguous
// blob (kernels.h: ClassifyDesc per pattern -- window constants, output pointers -- and the automaton tables of
// all patterns: short bounded patterns only, DevProgram::short_max, 1.4 KB each for regexdna) and then takes
// regions grid-stride.  A pattern's test is register arithmetic and LDS lookups; a region costs three dependent
// trips to device memory (count, candidates, their text) whatever the number of patterns.  History: reading
// descriptors and tables from device memory pattern after pattern: 82 us; tables staged table by table through
// their pointers (nine dependent copies per workgroup, a workgroup per four regions): 87 us -- the scan itself
// takes 97.
// W / MAXK: state words and step bound of the widest / longest pattern of the set (one instantiation per shape,
// so that the nine 8-byte patterns of regexdna do not carry the registers of a 64-position automaton)
template <int W, int MAXK>
__global__ __launch_bounds__(256) void classify_shared_multi(SharedHits sh, unsigned long long* counters0) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int lane = lane_id();
  const int half = lane >> 5, sub = lane & 31;
  if (threadIdx.x == 0) RJ_STAMP_AT(blockIdx.x, 0);
  const uint64_t wave = __builtin_amdgcn_readfirstlane(
      static_cast<uint32_t>((static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6));
  const uint64_t n_waves = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 6;
  const uint8_t* text = sh.text;
  const uint64_t n = sh.n, sb = sh.sb, se = sh.se;
  // the first region's count and candidates are on their way while the blob is copied (a region has at least
  // 64 slots: the first 32 are read whatever the count says)
  // (regions are taken from the LAST one down: the scan read the text front to back, so the end of the text is what the
  // caches still hold when this kernel starts)
  uint64_t r = wave * 2 + half;
  if (sh.reverse && r < sh.n_regions) r = sh.n_regions - 1 - r;
  uint32_t raw = r < sh.n_regions ? sh.counts[r] : 0u;
  uint64_t s_first = r < sh.n_regions ? sh.hits[r * sh.cap + sub] : 0;
  {
    const uint4* src = reinterpret_cast<const uint4*>(sh.blob);
    uint4* dst = reinterpret_cast<uint4*>(lds);
    for (uint32_t i = threadIdx.x; i < sh.blob_words / 4; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) RJ_STAMP_AT(blockIdx.x, 1);
  const ClassifyDesc* desc = reinterpret_cast<const ClassifyDesc*>(lds);
  const uint32_t* tab = lds + sh.desc_words;
  for (uint64_t r0 = wave * 2; r0 < sh.n_regions; r0 += n_waves * 2) {
    r = r0 + half;
    const bool live = r < sh.n_regions;
    if (sh.reverse && live) r = sh.n_regions - 1 - r;
    if (r0 != wave * 2) {
      raw = live ? sh.counts[r] : 0u;
      s_first = live ? sh.hits[r * sh.cap + sub] : 0;
    }
    const uint32_t cnt = raw < sh.cap ? raw : sh.cap;
    if (raw > sh.cap && sub == 0) atomicMax(&counters0[kCntSharedMax], static_cast<unsigned long long>(raw));
    uint32_t kept = 0;  // lane 32 * half + p: survivors of pattern p in the half's region
    const uint64_t* region = sh.hits + r * sh.cap;
    for (uint32_t base = 0; __ballot(base < cnt) != 0; base += 32) {
      const uint32_t k = base + sub;
      const bool have = k < cnt;
      const uint64_t s = !have ? 0 : base == 0 ? s_first : region[k];
      const uint64_t w = s + sh.win_offset;
      // the candidate's 8 window bytes; the scan does not clip (starts outside [sb, se), windows that reach
      // past the end of the text): dropped here
      const bool in_range = have && s >= sb && s < se && w + 8 <= n;
      uint32_t lo = 0, hi = 0;
      uint64_t t_lo = 0, t_hi = 0;  // the text from s on, loaded once for all patterns
      if (in_range) {
        __builtin_memcpy(&lo, text + w, 4);
        __builtin_memcpy(&hi, text + w + 4, 4);
        rj_load16(text, n, s, &t_lo, &t_hi);
      }
      const uint32_t avail = n - s < 16 ? static_cast<uint32_t>(n - s) : 16u;
      if (threadIdx.x == 0 && (t_lo | lo) != 0x123456789ull) RJ_STAMP_AT(blockIdx.x, r0 == wave * 2 ? 2 : 4);
      // 1. every pattern's exact window test: a bit per pattern that this candidate may match
      uint32_t todo = 0;
      for (uint32_t p = 0; p < sh.n_patterns; p++) {
        const ClassifyDesc& d = desc[p];
        bool win = (((lo ^ d.v0[0]) & d.m0[0]) | ((hi ^ d.v1[0]) & d.m1[0])) == 0;
        if (d.n_windows > 1) win = win || ((((lo ^ d.v0[1]) & d.m0[1]) | ((hi ^ d.v1[1]) & d.m1[1])) == 0);
        todo |= (in_range && win) ? 1u << p : 0u;
      }
      // 2. the automata: in pass r every lane runs the r-th pattern whose window test ITS candidate passed (tables and
      //    descriptor addressed per lane).  A candidate passes one or two tests, so this is two passes where the loop
      //    over the patterns -- the automaton whenever ANY of the 64 candidates passed the pattern's test -- made nine.
      uint32_t matched = 0;  // bit p: pattern p matches at this candidate ...
      uint64_t lens[2] = {0, 0};  // ... with this length (5 bits per pattern, <= 16: twelve patterns per word)
      static_assert(kMaxFused <= 24, "two words of twelve lengths");
      while (__ballot(todo != 0) != 0) {
        const bool act = todo != 0;
        const uint32_t p = act ? static_cast<uint32_t>(__builtin_ctz(todo)) : 0u;
        todo &= todo - 1;
        const ClassifyDesc& d = desc[p];
        const uint32_t* t0 = tab + d.tab;
        uint32_t len = 0;
        bool found;
        if (W == 1 || d.n_words <= 1) found = short_longest_lds<1, MAXK>(t0, d, t_lo, t_hi, avail, &len);
        else found = short_longest_lds<W, MAXK>(t0, d, t_lo, t_hi, avail, &len);
        if (found && act) {
          matched |= 1u << p;
          if (p < 12) lens[0] |= static_cast<uint64_t>(len) << (5 * p);
          else lens[1] |= static_cast<uint64_t>(len) << (5 * (p - 12));
        }
      }
      // 3. the survivors of every pattern, in position order, to the pattern's own region
      for (uint32_t p = 0; p < sh.n_patterns; p++) {
        const bool found = ((matched >> p) & 1u) != 0;
        const uint64_t mine = __ballot(found);
        if (mine == 0) continue;
        const ClassifyDesc& d = desc[p];
        const uint64_t e = s + (((p < 12 ? lens[0] >> (5 * p) : lens[1] >> (5 * (p - 12)))) & 31u);
        const uint32_t mine_half = static_cast<uint32_t>(mine >> (32 * half));
        const uint32_t b0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(kept), static_cast<int>(p)));
        const uint32_t b1 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(kept), static_cast<int>(32 + p)));
        const uint32_t pos = (half ? b1 : b0) + __popc(mine_half & ((1u << sub) - 1u));
        const uint32_t cap_p = d.region_cap;
        if (found && pos < cap_p) {
          d.begins[r * cap_p + pos] = s;
          d.ends[r * cap_p + pos] = e;
        }
        if (sub == static_cast<int>(p)) kept += __popc(mine_half);
      }
    }
    if (live && sub < static_cast<int>(sh.n_patterns)) {
      const ClassifyDesc& d = desc[sub];
      const uint32_t c = kept, cap_p = d.region_cap;
      if (c > cap_p) {  // the host grows this pattern's regions and runs again
        d.counters[kCntOverflow] = 1;
        atomicMax(&d.counters[kCntMaxRegion], static_cast<unsigned long long>(c));
      }
      d.valid_counts[r] = c < cap_p ? c : cap_p;
    }
    if (threadIdx.x == 0) RJ_STAMP_AT(blockIdx.x, r0 == wave * 2 ? 3 : 5);
  }
  if (threadIdx.x == 0) RJ_STAMP_AT(blockIdx.x, 6);
}

// ---------------------------------------------------------------------------------------
// Round 4: scan and classification in ONE kernel.  The step of round 3 was plane_scan (97 us) -> gap ->
// classify_shared_multi (24 us) -> gap -> offsets_gather_check_multi (20 us): a third of it latency-bound tails on
// the same stream.  A wave's span holds two dozen candidates, so the wave that found them can classify them
// itself at the end of its span: candidates stay in LDS (32-bit window positions relative to the span, never
// written to HBM), their text is still in L2, the blob of descriptors and tables is staged once per workgroup
// while the first text loads are in flight, and the prefetch registers of the scan loop are dead by then (no
// more registers than the scan alone).  One launch and one gap less per step; the shared candidate regions in
// device memory and their counts are gone from this path.  A span with more than kFusedCap candidates (13 x the
// density of DNA) flags the run (kCntSharedMax) and the host repeats it with the two kernels above, whose shared
// regions grow.
constexpr uint32_t kFusedCap = 256;  // candidate slots per wave (LDS)

namespace {

struct LdsRegion {
  uint32_t* slots;
  uint32_t count;  // wave-uniform; keeps counting past kFusedCap
};

// push_pair with the slots in LDS: `rel` = offset of the lane's 16 bytes of chunk A from the span's first byte
__device__ __forceinline__ void push_pair_lds(LdsRegion& r, uint32_t hm, uint32_t rel) {
  const uint32_t hA = hm & 0x55555555u, hB = hm & 0xAAAAAAAAu;
  const uint64_t mA = __ballot(hA != 0), mB = __ballot(hB != 0);
  const uint64_t several = __ballot(((hA & (hA - 1)) | (hB & (hB - 1))) != 0);
  if (several == 0) {
    const uint32_t nA = __popcll(mA), nB = __popcll(mB);
    if (hA != 0) {
      const uint32_t idx = r.count + lanes_below(mA);
      if (idx < kFusedCap) r.slots[idx] = rel + (static_cast<uint32_t>(__builtin_ctz(hA)) >> 1);
    }
    if (hB != 0) {
      const uint32_t idx = r.count + nA + lanes_below(mB);
      if (idx < kFusedCap) r.slots[idx] = rel + static_cast<uint32_t>(kChunk) + (static_cast<uint32_t>(__builtin_ctz(hB)) >> 1);
    }
    r.count += nA + nB;
    return;
  }
  const uint32_t cA = __popc(hA), cB = __popc(hB);
  const uint32_t incA = wave_inclusive_sum(cA), incB = wave_inclusive_sum(cB);
  const uint32_t totA = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incA), kWave - 1));
  const uint32_t totB = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incB), kWave - 1));
  uint32_t idx = r.count + incA - cA;
  for (uint32_t m = hA; m; m &= m - 1, idx++)
    if (idx < kFusedCap) r.slots[idx] = rel + (static_cast<uint32_t>(__builtin_ctz(m)) >> 1);
  idx = r.count + totA + incB - cB;
  for (uint32_t m = hB; m; m &= m - 1, idx++)
    if (idx < kFusedCap) r.slots[idx] = rel + static_cast<uint32_t>(kChunk) + (static_cast<uint32_t>(__builtin_ctz(m)) >> 1);
  r.count += totA + totB;
}

template <int NB>
__device__ __forceinline__ void plane_pair_lds(const uint32_t (&dA)[6], const uint32_t (&dB)[6], uint32_t rel, const PlaneConsts& k,
                                               const PlaneParams& a, LdsRegion& region) {
  const uint32_t hm = plane_candidates<NB>(dA, dB, k, a);
  if (__ballot(hm != 0) == 0) return;  // wave-uniform
  push_pair_lds(region, hm, rel);
}

// The wave's candidates (window positions span_base + slots[k], in position order) against every pattern: exact
// window test, short automaton from the LDS tables, survivors in position order into the pattern's own region
// `r` -- what classify_shared_multi does for half a wave per region, here for the whole wave and ONE region.
template <int W, int MAXK>
__device__ __forceinline__ void classify_own_region(const uint32_t* lds, const SharedHits& sh, const uint32_t* slots, uint32_t cnt,
                                                    uint64_t span_base, uint64_t r) {
  const int lane = lane_id();
  const ClassifyDesc* desc = reinterpret_cast<const ClassifyDesc*>(lds);
  const uint32_t* tab = lds + sh.desc_words;
  const uint8_t* text = sh.text;
  const uint64_t n = sh.n, sb = sh.sb, se = sh.se;
  uint32_t kept = 0;  // lane p: survivors of pattern p so far
  for (uint32_t base = 0; base < cnt; base += kWave) {
    const uint32_t k = base + static_cast<uint32_t>(lane);
    const bool have = k < cnt;
    const uint64_t w = span_base + (have ? slots[k] : 0u);
    const uint64_t s = w - sh.win_offset;  // (wraps for a window before the offset: dropped by `s < se`)
    const bool in_range = have && s >= sb && s < se && w + 8 <= n;
    uint32_t lo = 0, hi = 0;
    uint64_t t_lo = 0, t_hi = 0;
    if (in_range) {
      __builtin_memcpy(&lo, text + w, 4);
      __builtin_memcpy(&hi, text + w + 4, 4);
      rj_load16(text, n, s, &t_lo, &t_hi);
    }
    const uint32_t avail = n - s < 16 ? static_cast<uint32_t>(n - s) : 16u;
    uint32_t todo = 0;
    for (uint32_t p = 0; p < sh.n_patterns; p++) {
      const ClassifyDesc& d = desc[p];
      bool win = (((lo ^ d.v0[0]) & d.m0[0]) | ((hi ^ d.v1[0]) & d.m1[0])) == 0;
      if (d.n_windows > 1) win = win || ((((lo ^ d.v0[1]) & d.m0[1]) | ((hi ^ d.v1[1]) & d.m1[1])) == 0);
      todo |= (in_range && win) ? 1u << p : 0u;
    }
    uint32_t matched = 0;
    uint64_t lens[2] = {0, 0};
    while (__ballot(todo != 0) != 0) {
      const bool act = todo != 0;
      const uint32_t p = act ? static_cast<uint32_t>(__builtin_ctz(todo)) : 0u;
      todo &= todo - 1;
      const ClassifyDesc& d = desc[p];
      const uint32_t* t0 = tab + d.tab;
      uint32_t len = 0;
      bool found;
      if (W == 1 || d.n_words <= 1) found = short_longest_lds<1, MAXK>(t0, d, t_lo, t_hi, avail, &len);
      else found = short_longest_lds<W, MAXK>(t0, d, t_lo, t_hi, avail, &len);
      if (found && act) {
        matched |= 1u << p;
        if (p < 12) lens[0] |= static_cast<uint64_t>(len) << (5 * p);
        else lens[1] |= static_cast<uint64_t>(len) << (5 * (p - 12));
      }
    }
    for (uint32_t p = 0; p < sh.n_patterns; p++) {
      const bool found = ((matched >> p) & 1u) != 0;
      const uint64_t mine = __ballot(found);
      if (mine == 0) continue;
      const ClassifyDesc& d = desc[p];
      const uint64_t e = s + (((p < 12 ? lens[0] >> (5 * p) : lens[1] >> (5 * (p - 12)))) & 31u);
      const uint32_t b0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(kept), static_cast<int>(p)));
      const uint32_t pos = b0 + lanes_below(mine);
      const uint32_t cap_p = d.region_cap;
      if (found && pos < cap_p) {
        d.begins[r * cap_p + pos] = s;
        d.ends[r * cap_p + pos] = e;
      }
      if (lane == static_cast<int>(p)) kept += __popcll(mine);
    }
  }
  if (lane < static_cast<int>(sh.n_patterns)) {
    const ClassifyDesc& d = desc[lane];
    const uint32_t c = kept, cap_p = d.region_cap;
    if (c > cap_p) {  // the host grows this pattern's regions and runs again
      d.counters[kCntOverflow] = 1;
      atomicMax(&d.counters[kCntMaxRegion], static_cast<unsigned long long>(c));
    }
    d.valid_counts[r] = c < cap_p ? c : cap_p;
  }
}

}  // namespace

// sh.hits / sh.counts / sh.cap are unused here (no shared regions in device memory); sh.n_regions = the grid's waves.
// The counters this kernel may SET (kCntOverflow, kCntMaxRegion of a pattern, kCntSharedMax of pattern 0) are not
// among those it clears -- a wave may finish before wave 0 has run: the host keeps them clean (multi_pattern.hip).
template <int NB, int W, int MAXK>
__global__ __launch_bounds__(256) void plane_scan_classify(PlaneParams a, SharedHits sh, unsigned long long* counters0) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int lane = lane_id();
  const uint64_t wave = __builtin_amdgcn_readfirstlane(
      static_cast<uint32_t>((static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6));
  if (wave == 0 && lane < kCntSize && lane != kCntOverflow && lane != kCntMaxRegion && lane != kCntSharedMax)
    for (uint32_t p = 0; p < a.n_zero; p++) a.zero_counters[p][lane] = 0;
  PlaneConsts k;
  k.shift = a.code_shift;
  k.cmask = 0x03030303u << a.code_shift;
  LdsRegion region{lds + sh.blob_words + (threadIdx.x >> 6) * kFusedCap, 0u};
  const uint64_t wlo = a.sb + a.offset;
  const uint64_t last_w = a.n >= 8 ? a.n - 8 + 1 : 0;
  uint64_t whi = a.se + a.offset;
  if (whi > last_w) whi = last_w;
  const uint64_t first_pair = wlo / kPair;
  const uint64_t end_pair = whi > wlo ? (whi + kPair - 1) / kPair : first_pair;
  uint64_t c0 = first_pair + wave * a.span_pairs, c1 = c0 + a.span_pairs;
  if (c0 > end_pair) c0 = end_pair;
  if (c1 > end_pair) c1 = end_pair;
  uint64_t fast_end = a.n >= kPair + 8 ? (a.n - 8) / kPair : 0;
  if (fast_end > c1) fast_end = c1;
  if (fast_end < c0) fast_end = c0;
  const uint64_t lane_off = static_cast<uint64_t>(lane) * 16;
  const uint64_t span_base = c0 * kPair;
  const uint32_t lane_rel = static_cast<uint32_t>(lane) * 16u;
  uint32_t a0[6], b0[6], a1[6], b1[6];
  uint64_t c = c0;
  const bool piped = c + 3 < fast_end;  // (wave-uniform)
  if (piped) {
    load_chunk24(a.text, c * kPair + lane_off, a0);
    load_chunk24(a.text, c * kPair + kChunk + lane_off, b0);
    load_chunk24(a.text, (c + 1) * kPair + lane_off, a1);
    load_chunk24(a.text, (c + 1) * kPair + kChunk + lane_off, b1);
  }
  {
    // the blob of descriptors + tables, while the first pairs are on their way
    const uint4* src = reinterpret_cast<const uint4*>(sh.blob);
    uint4* dst = reinterpret_cast<uint4*>(lds);
    for (uint32_t i = threadIdx.x; i < sh.blob_words / 4; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  if (piped) {
    while (c + 3 < fast_end) {
      plane_pair_lds<NB>(a0, b0, static_cast<uint32_t>((c - c0) * kPair) + lane_rel, k, a, region);
      load_chunk24(a.text, (c + 2) * kPair + lane_off, a0);
      load_chunk24(a.text, (c + 2) * kPair + kChunk + lane_off, b0);
      __builtin_amdgcn_sched_barrier(0);
      plane_pair_lds<NB>(a1, b1, static_cast<uint32_t>((c + 1 - c0) * kPair) + lane_rel, k, a, region);
      load_chunk24(a.text, (c + 3) * kPair + lane_off, a1);
      load_chunk24(a.text, (c + 3) * kPair + kChunk + lane_off, b1);
      __builtin_amdgcn_sched_barrier(0);
      c += 2;
    }
    plane_pair_lds<NB>(a0, b0, static_cast<uint32_t>((c - c0) * kPair) + lane_rel, k, a, region);
    plane_pair_lds<NB>(a1, b1, static_cast<uint32_t>((c + 1 - c0) * kPair) + lane_rel, k, a, region);
    c += 2;
  }
  for (; c < fast_end; c++) {
    load_chunk24(a.text, c * kPair + lane_off, a0);
    load_chunk24(a.text, c * kPair + kChunk + lane_off, b0);
    plane_pair_lds<NB>(a0, b0, static_cast<uint32_t>((c - c0) * kPair) + lane_rel, k, a, region);
  }
  for (uint64_t t = fast_end; t < c1; t++) {
    load_guarded24(a.text, a.n, t * kPair + lane_off, a0);
    load_guarded24(a.text, a.n, t * kPair + kChunk + lane_off, b0);
    plane_pair_lds<NB>(a0, b0, static_cast<uint32_t>((t - c0) * kPair) + lane_rel, k, a, region);
  }
  // the wave's own LDS stores, then its own LDS loads: in order on the LDS queue, no barrier
  uint32_t cnt = region.count;
  if (cnt > kFusedCap) {  // the run is void: the host repeats it with regions in device memory
    if (lane == 0) atomicMax(&counters0[kCntSharedMax], static_cast<unsigned long long>(cnt));
    cnt = kFusedCap;
  }
  classify_own_region<W, MAXK>(lds, sh, region.slots, cnt, span_base, wave);
}

size_t plane_fused_lds_bytes(uint32_t blob_words) { return (static_cast<size_t>(blob_words) + 4u * kFusedCap) * sizeof(uint32_t); }

// false: no instantiation for this shape / blob too large for a resident workgroup mix (the caller takes the two kernels)
bool launch_plane_scan_classify(const PlaneParams& a, const SharedHits& sh, int max_words, uint32_t max_short, unsigned long long* counters0,
                                int grid, hipEvent_t t0, hipEvent_t t1, hipStream_t st) {
  if (max_words > 1 || max_short > 8 || sh.blob_words > kFusedMaxBlobWords) return false;
  const size_t lds = plane_fused_lds_bytes(sh.blob_words);
  if (a.n_bases <= 1) hipExtLaunchKernelGGL((plane_scan_classify<1, 1, 8>), dim3(grid), dim3(256), lds, st, t0, t1, 0, a, sh, counters0);
  else hipExtLaunchKernelGGL((plane_scan_classify<2, 1, 8>), dim3(grid), dim3(256), lds, st, t0, t1, 0, a, sh, counters0);
  return true;
}


// ---------------------------------------------------------------------------------------
// The GENERAL one-pass scan (round 4; kernels.h: PlaneGParams): up to 12 base windows, their first n_cmp (4..8) bytes
// compared exactly or with one differing code, any alphabet (codes alias), candidates = window positions.
namespace {

template <int NB, bool TOL>
__device__ __forceinline__ uint32_t plane_candidates_general(const uint32_t (&dA)[6], const uint32_t (&dB)[6], const PlaneConsts& k,
                                                             const PlaneGParams& a) {
  const uint32_t s = k.shift;
  const uint32_t ta = (codes4(dA[0], k) >> s) | (codes4(dA[1], k) << (8 - s)) | (codes4(dA[2], k) << (16 - s)) | (codes4(dA[3], k) << (24 - s));
  const uint32_t tb = (codes4(dB[0], k) >> s) | (codes4(dB[1], k) << (8 - s)) | (codes4(dB[2], k) << (16 - s)) | (codes4(dB[3], k) << (24 - s));
  const uint32_t ha = (codes4(dA[4], k) >> s) | (codes4(dA[5], k) << (8 - s));
  const uint32_t hb = (codes4(dB[4], k) >> s) | (codes4(dB[5], k) << (8 - s));
  constexpr uint32_t kEven = 0x55555555u;
  const uint32_t L = (ta & kEven) | ((tb << 1) & ~kEven);
  const uint32_t H = ((ta >> 1) & kEven) | (tb & ~kEven);
  const uint32_t Ln = (ha & kEven) | ((hb << 1) & ~kEven);
  const uint32_t Hn = ((ha >> 1) & kEven) | (hb & ~kEven);
  // the shifted planes of all compared offsets first (16 registers), then base after base in a loop that is NOT unrolled:
  // one base's 16 masks live in scalar registers at a time (read from the kernel arguments per pair: scalar-cache hits).
  // All bases unrolled kept 4 x 8 x 2 masks live and the compiler spilled ~100 of them to VGPR lanes, a v_readlane per use.
  uint32_t Ls[8], Hs[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    Ls[i] = i ? __builtin_amdgcn_alignbit(Ln, L, 2 * i) : L;
    Hs[i] = i ? __builtin_amdgcn_alignbit(Hn, H, 2 * i) : H;
  }
  uint32_t c = 0;
#pragma clang loop unroll(disable)
  for (uint32_t b = 0; b < a.n_bases; b++) {
    uint32_t Z = ~0u, O = ~0u;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (static_cast<uint32_t>(i) >= a.n_cmp) break;  // (wave-uniform: windows shorter than 8 bytes)
      const uint32_t E = (Ls[i] ^ a.lo[b][i]) & (Hs[i] ^ a.hi[b][i]);
      if (TOL) O = i == 0 ? ~0u : (Z | (O & E));   // at most one code differs so far
      Z &= E;                                        // none differs so far
    }
    c |= TOL ? O : Z;
  }
  return c;
}

}  // namespace

template <int NB, bool TOL>
__global__ __launch_bounds__(256) void plane_scan_general(PlaneGParams a) {
  const int lane = lane_id();
  const uint64_t wave = __builtin_amdgcn_readfirstlane(
      static_cast<uint32_t>((static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6));
  if (wave == 0 && lane < kCntSize)
    for (uint32_t p = 0; p < a.n_zero; p++) a.zero_counters[p][lane] = 0;
  PlaneConsts k;
  k.shift = a.code_shift;
  k.cmask = 0x03030303u << a.code_shift;
  SharedRegion region{a.hits + wave * a.region_cap, a.region_cap, 0u};
  const uint64_t first_pair = a.wlo / kPair;
  const uint64_t end_pair = a.whi > a.wlo ? (a.whi + kPair - 1) / kPair : first_pair;
  uint64_t c0 = first_pair + wave * a.span_pairs, c1 = c0 + a.span_pairs;
  if (c0 > end_pair) c0 = end_pair;
  if (c1 > end_pair) c1 = end_pair;
  uint64_t fast_end = a.n >= kPair + 8 ? (a.n - 8) / kPair : 0;
  if (fast_end > c1) fast_end = c1;
  if (fast_end < c0) fast_end = c0;
  const uint64_t lane_off = static_cast<uint64_t>(lane) * 16;
  auto pair = [&](const uint32_t (&dA)[6], const uint32_t (&dB)[6], uint64_t at) {
    const uint32_t hm = plane_candidates_general<NB, TOL>(dA, dB, k, a);
    if (__ballot(hm != 0) == 0) return;
    push_pair(region, hm, at, 0);   // (the slot holds the WINDOW position; classify_shared_general clips)
  };
  {
    uint32_t a0[6], b0[6], a1[6], b1[6];
    uint64_t c = c0;
    if (c + 3 < fast_end) {
      load_chunk24(a.text, c * kPair + lane_off, a0);
      load_chunk24(a.text, c * kPair + kChunk + lane_off, b0);
      load_chunk24(a.text, (c + 1) * kPair + lane_off, a1);
      load_chunk24(a.text, (c + 1) * kPair + kChunk + lane_off, b1);
      while (c + 3 < fast_end) {
        pair(a0, b0, c * kPair + lane_off);
        load_chunk24(a.text, (c + 2) * kPair + lane_off, a0);
        load_chunk24(a.text, (c + 2) * kPair + kChunk + lane_off, b0);
        __builtin_amdgcn_sched_barrier(0);
        pair(a1, b1, (c + 1) * kPair + lane_off);
        load_chunk24(a.text, (c + 3) * kPair + lane_off, a1);
        load_chunk24(a.text, (c + 3) * kPair + kChunk + lane_off, b1);
        __builtin_amdgcn_sched_barrier(0);
        c += 2;
      }
      pair(a0, b0, c * kPair + lane_off);
      pair(a1, b1, (c + 1) * kPair + lane_off);
      c += 2;
    }
    for (; c < fast_end; c++) {
      load_chunk24(a.text, c * kPair + lane_off, a0);
      load_chunk24(a.text, c * kPair + kChunk + lane_off, b0);
      pair(a0, b0, c * kPair + lane_off);
    }
  }
  for (uint64_t t = fast_end; t < c1; t++) {
    uint32_t dA[6], dB[6];
    load_guarded24(a.text, a.n, t * kPair + lane_off, dA);
    load_guarded24(a.text, a.n, t * kPair + kChunk + lane_off, dB);
    pair(dA, dB, t * kPair + lane_off);
  }
  if (lane == 0) a.hit_counts[wave] = region.count;
}

void launch_plane_scan_general(const PlaneGParams& a, int grid, hipEvent_t t0, hipEvent_t t1, hipStream_t st) {
  const dim3 g(grid), b(256);
  // (the kernel no longer depends on the number of bases at compile time: one instantiation per tolerance)
  if (a.tolerance) hipExtLaunchKernelGGL((plane_scan_general<1, true>), g, b, 0, st, t0, t1, 0, a);
  else hipExtLaunchKernelGGL((plane_scan_general<1, false>), g, b, 0, st, t0, t1, 0, a);
}

// classify_shared_multi for candidates that are WINDOW positions of patterns with their own offsets, window lengths and
// up to four windows each: per pattern the start s = w - offset is clipped to the own range, the exact window test runs on
// the 8 bytes at w (masks cover the pattern's own window length), the automaton reads the 16 bytes at ITS start.
template <int W, int MAXK>
__global__ __launch_bounds__(256) void classify_shared_general(SharedHits sh, unsigned long long* counters0) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int lane = lane_id();
  const int half = lane >> 5, sub = lane & 31;
  const uint64_t wave = __builtin_amdgcn_readfirstlane(
      static_cast<uint32_t>((static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6));
  const uint64_t n_waves = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 6;
  const uint8_t* text = sh.text;
  const uint64_t n = sh.n, sb = sh.sb, se = sh.se;
  {
    const uint4* src = reinterpret_cast<const uint4*>(sh.blob);
    uint4* dst = reinterpret_cast<uint4*>(lds);
    for (uint32_t i = threadIdx.x; i < sh.blob_words / 4; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  const ClassifyDesc* desc = reinterpret_cast<const ClassifyDesc*>(lds);
  const uint32_t* tab = lds + sh.desc_words;
  for (uint64_t r0 = wave * 2; r0 < sh.n_regions; r0 += n_waves * 2) {
    const uint64_t r = r0 + half;
    const bool live = r < sh.n_regions;
    const uint32_t raw = live ? sh.counts[r] : 0u;
    const uint32_t cnt = raw < sh.cap ? raw : sh.cap;
    if (raw > sh.cap && sub == 0) atomicMax(&counters0[kCntSharedMax], static_cast<unsigned long long>(raw));
    uint32_t kept = 0;  // lane 32 * half + p: survivors of pattern p in the half's region
    const uint64_t* region = sh.hits + r * sh.cap;
    for (uint32_t base = 0; __ballot(base < cnt) != 0; base += 32) {
      const uint32_t k = base + sub;
      const bool have = k < cnt;
      const uint64_t w = have ? region[k] : 0;
      uint32_t lo = 0, hi = 0;
      if (have) {
        if (w + 8 <= n) {
          __builtin_memcpy(&lo, text + w, 4);
          __builtin_memcpy(&hi, text + w + 4, 4);
        } else {
          for (uint32_t q = 0; q < 8 && w + q < n; q++) {
            const uint32_t c = text[w + q];
            if (q < 4) lo |= c << (8 * q);
            else hi |= c << (8 * (q - 4));
          }
        }
      }
      // 1. per pattern: the start inside the own range, the window inside the text, one of its windows matches
      uint32_t todo = 0;
      for (uint32_t p = 0; p < sh.n_patterns; p++) {
        const ClassifyDesc& d = desc[p];
        const uint64_t s = w - d.win_offset;
        const bool ok = have && w >= d.win_offset && s >= sb && s < se && w + d.win_len <= n;
        bool win = false;
        for (uint32_t q = 0; q < d.n_windows; q++) win = win || ((((lo ^ d.v0[q]) & d.m0[q]) | ((hi ^ d.v1[q]) & d.m1[q])) == 0);
        todo |= (ok && win) ? 1u << p : 0u;
      }
      // 2. the automata: in pass r every lane runs the r-th pattern whose window test its candidate passed
      uint32_t matched = 0;
      uint64_t lens[2] = {0, 0};
      while (__ballot(todo != 0) != 0) {
        const bool act = todo != 0;
        const uint32_t p = act ? static_cast<uint32_t>(__builtin_ctz(todo)) : 0u;
        todo &= todo - 1;
        const ClassifyDesc& d = desc[p];
        const uint32_t* t0 = tab + d.tab;
        const uint64_t s = w - d.win_offset;
        uint64_t t_lo = 0, t_hi = 0;
        if (act) rj_load16(text, n, s, &t_lo, &t_hi);
        const uint32_t avail = act ? (n - s < 16 ? static_cast<uint32_t>(n - s) : 16u) : 0u;
        uint32_t len = 0;
        bool found;
        if (W == 1 || d.n_words <= 1) found = short_longest_lds<1, MAXK>(t0, d, t_lo, t_hi, avail, &len);
        else found = short_longest_lds<W, MAXK>(t0, d, t_lo, t_hi, avail, &len);
        if (found && act) {
          matched |= 1u << p;
          if (p < 12) lens[0] |= static_cast<uint64_t>(len) << (5 * p);
          else lens[1] |= static_cast<uint64_t>(len) << (5 * (p - 12));
        }
      }
      // 3. the survivors of every pattern, in position order, to the pattern's own region
      for (uint32_t p = 0; p < sh.n_patterns; p++) {
        const bool found = ((matched >> p) & 1u) != 0;
        const uint64_t mine = __ballot(found);
        if (mine == 0) continue;
        const ClassifyDesc& d = desc[p];
        const uint64_t s = w - d.win_offset;
        const uint64_t e = s + (((p < 12 ? lens[0] >> (5 * p) : lens[1] >> (5 * (p - 12)))) & 31u);
        const uint32_t mine_half = static_cast<uint32_t>(mine >> (32 * half));
        const uint32_t b0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(kept), static_cast<int>(p)));
        const uint32_t b1 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(kept), static_cast<int>(32 + p)));
        const uint32_t pos = (half ? b1 : b0) + __popc(mine_half & ((1u << sub) - 1u));
        const uint32_t cap_p = d.region_cap;
        if (found && pos < cap_p) {
          d.begins[r * cap_p + pos] = s;
          d.ends[r * cap_p + pos] = e;
        }
        if (sub == static_cast<int>(p)) kept += __popc(mine_half);
      }
    }
    if (live && sub < static_cast<int>(sh.n_patterns)) {
      const ClassifyDesc& d = desc[sub];
      const uint32_t c = kept, cap_p = d.region_cap;
      if (c > cap_p) {
        d.counters[kCntOverflow] = 1;
        atomicMax(&d.counters[kCntMaxRegion], static_cast<unsigned long long>(c));
      }
      d.valid_counts[r] = c < cap_p ? c : cap_p;
    }
  }
}

void launch_tails_shared_general(const MultiTail* d_tails, const SharedHits& sh, int max_words, uint32_t max_short, unsigned long long* counters0,
                                 hipStream_t st) {
  uint64_t blocks = (static_cast<uint64_t>(sh.n_regions) + 7) / 8;
  blocks = blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks;
  const dim3 g(static_cast<unsigned>(blocks)), b(256);
  const size_t lds = static_cast<size_t>(sh.blob_words) * sizeof(uint32_t);
  if (max_words <= 1 && max_short <= 8) hipLaunchKernelGGL((classify_shared_general<1, 8>), g, b, lds, st, sh, counters0);
  else if (max_words <= 1) hipLaunchKernelGGL((classify_shared_general<1, 16>), g, b, lds, st, sh, counters0);
  else hipLaunchKernelGGL((classify_shared_general<2, 16>), g, b, lds, st, sh, counters0);
  launch_offsets_gather_check_multi(d_tails, static_cast<int>(sh.n_patterns), sh.n_regions, st);
}

void launch_tails_shared(const MultiTail* d_tails, const SharedHits& sh, int max_words, uint32_t max_short, unsigned long long* counters0,
                         hipStream_t st) {
  // half a wave per region; every workgroup copies the blob into LDS first
  uint64_t blocks = (static_cast<uint64_t>(sh.n_regions) + 7) / 8;
  // (at most 2048 workgroups, each taking its regions grid-stride; 1024 / 512 / 256 measured: +1 / +4 / +7 % on the regexdna step)
  blocks = blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks;
  const dim3 g(static_cast<unsigned>(blocks)), b(256);
  const size_t lds = static_cast<size_t>(sh.blob_words) * sizeof(uint32_t);
  if (max_words <= 1 && max_short <= 8) hipLaunchKernelGGL((classify_shared_multi<1, 8>), g, b, lds, st, sh, counters0);
  else if (max_words <= 1) hipLaunchKernelGGL((classify_shared_multi<1, 16>), g, b, lds, st, sh, counters0);
  else hipLaunchKernelGGL((classify_shared_multi<2, 16>), g, b, lds, st, sh, counters0);
  launch_offsets_gather_check_multi(d_tails, static_cast<int>(sh.n_patterns), sh.n_regions, st);
}

}  // namespace rejit_amd
