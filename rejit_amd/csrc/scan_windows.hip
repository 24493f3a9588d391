// rejit_amd/csrc/scan_windows.hip -- the "fast-forward" window scans (split out of kernels.hip in round 6; the pipeline's overview
// is at the top of kernels.hip): scan_windows<K,...> (FastForwardGen::VisitSingleMultipleChar / the multi-literal branch,
// reference src/x64/codegen-x64.cc:1102-1403), scan_windows_train, scan_windows_fused, scan_dense, and the launch geometry.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "behind_walk.h"
#include "dense_swar.h"
#include "device_program.h"
#include "kernel_util.h"
#include "stream_load.h"
#include "kernels.h"

namespace rejit_amd {

// ---------------------------------------------------------------------------------------
// Fast-forward window scan.
//
// Window position w is a hit iff for some k < K
//     (load32(text + w) & mask0[k]) == value0[k]  and, when TWO,
//     (load32(text + w + 4) & mask1[k]) == value1[k];
// the candidate start is s = w - offset.  Scanned w range: [wlo, whi).
// One chunk: d[0..3] = the lane's 16 bytes, d[4..5] = the 8 bytes that follow.
template <int K, bool TWO, bool MASKED, bool TWOLEVEL, bool NIB>
__device__ __forceinline__ void windows_chunk(const uint32_t (&d)[6], uint64_t at, const ScanParams& a,
                                              const WindowSet& ws, RegionHits& hits) {
  if (NIB) {
    // Nibble filter: keep only the low nibble of every byte, so that the 8 bytes of a window pack
    // into ONE dword -- byte i of pk[j] = nibble of text byte j+i | nibble of text byte j+4+i << 4 --
    // and a window costs one v_bitop3 ((pk ^ value) & mask) instead of three VALU ops.  The packing
    // is done on the ALIGNED dwords first (z[q] = nib(d[q]) | nib(d[q+1]) << 4: 6 v_and + 5
    // v_lshl_or) and the unaligned positions are v_alignbyte of neighbouring z: 26 VALU per chunk
    // for all 16 pk[j] (packing after the alignment took 37).  With two masked 8-byte windows
    // (regexdna) the exact form needs ~8.5 VALU per text byte, which bounds the kernel at ~4.6 TB/s
    // on 256 CUs; this form needs ~4.6.
    uint32_t nib[6], z[5], pk[16];
#pragma unroll
    for (int q = 0; q < 6; q++) nib[q] = d[q] & 0x0F0F0F0Fu;
#pragma unroll
    for (int q = 0; q < 5; q++) z[q] = nib[q] | (nib[q + 1] << 4);  // v_lshl_or_b32
#pragma unroll
    for (int q = 0; q < 4; q++) {
      pk[4 * q] = z[q];
      pk[4 * q + 1] = __builtin_amdgcn_alignbyte(z[q + 1], z[q], 1);
      pk[4 * q + 2] = __builtin_amdgcn_alignbyte(z[q + 1], z[q], 2);
      pk[4 * q + 3] = __builtin_amdgcn_alignbyte(z[q + 1], z[q], 3);
    }
    uint32_t accs[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
#pragma unroll
    for (int j = 0; j < 16; j++) {
#pragma unroll
      for (int k = 0; k + 1 < K; k += 2) {  // two windows per v_min3_u32
        uint32_t t0 = pk[j] ^ ws.value0[k], t1 = pk[j] ^ ws.value0[k + 1];
        if (MASKED) {
          t0 &= ws.mask0[k];
          t1 &= ws.mask0[k + 1];
        }
        const uint32_t ab = accs[j & 3] < t0 ? accs[j & 3] : t0;
        accs[j & 3] = ab < t1 ? ab : t1;
      }
      if (K & 1) {
        uint32_t t = pk[j] ^ ws.value0[K - 1];
        if (MASKED) t &= ws.mask0[K - 1];
        accs[j & 3] = accs[j & 3] < t ? accs[j & 3] : t;
      }
    }
    const uint32_t m01 = accs[0] < accs[1] ? accs[0] : accs[1];
    const uint32_t m23 = accs[2] < accs[3] ? accs[2] : accs[3];
    const uint32_t acc = m01 < m23 ? m01 : m23;
    if (__ballot(acc == 0) == 0) return;
    uint32_t hm = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      uint32_t best = 0xFFFFFFFFu;
#pragma unroll
      for (int k = 0; k < K; k++) {
        uint32_t t = pk[j] ^ ws.value0[k];
        if (MASKED) t &= ws.mask0[k];
        best = best < t ? best : t;
      }
      hm |= static_cast<uint32_t>(best == 0) << j;
    }
    const uint64_t chunk_base = at - static_cast<uint64_t>(lane_id()) * 16;
    if (chunk_base < a.wlo || chunk_base + kChunk > a.whi) {
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const uint64_t w = at + j;
        if (w < a.wlo || w >= a.whi) hm &= ~(1u << j);
      }
    }
    hits.push_bits(hm, at, ws.offset);
    return;
  }
  constexpr int NX = TWO ? 20 : 16;  // windows needed: 16 positions (+4 for the second dword)
  // the unaligned 4-byte windows of this lane: x[j] = bytes [at+j, at+j+4)
  uint32_t x[NX];
#pragma unroll
  for (int q = 0; q < NX / 4; q++) {
    x[4 * q] = d[q];
    x[4 * q + 1] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], 1);
    x[4 * q + 2] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], 2);
    x[4 * q + 3] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], 3);
  }
  // Streaming test, VALU only.  For window k at position j
  //     t = ((x[j] ^ value0[k]) & mask0[k]) | ((x[j+4] ^ value1[k]) & mask1[k])
  // is zero iff the window matches; the minimum over all (j,k) is zero iff the lane has a
  // hit.  (The obvious form -- v_cmp per dword and s_and/s_or of the lane masks -- put ~130
  // scalar instructions per chunk on the CU's single scalar unit and ran at 2.9 TB/s.)
  if (TWO && TWOLEVEL) {
    // Two-level test for 5..8-byte windows over a large alphabet: the first dword alone is
    // already a strong filter (e.g. 74^-4 on random ASCII), so test it for all 16 positions
    // first and leave, wave-uniformly, when no lane has a first-dword hit.  Over a small
    // alphabet (DNA) some lane always has one and this level would be pure overhead, which is
    // why the host enables it only when the window bytes span more than 4 distinct values.
    uint32_t acc1 = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < 16; j++) {
#pragma unroll
      for (int k = 0; k < K; k++) {
        uint32_t t = x[j] ^ ws.value0[k];
        if (MASKED) t &= ws.mask0[k];
        acc1 = acc1 < t ? acc1 : t;
      }
    }
    if (__ballot(acc1 == 0) == 0) return;
  }
  // four independent min chains (the single chain of 16*K dependent v_min was latency-bound:
  // 2 chains 0.144 -> 0.139 ms although they cost 16 more VGPRs and one wave of occupancy)
  uint32_t accs[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
#pragma unroll
  for (int j = 0; j < 16; j++) {
#pragma unroll
    for (int k = 0; k < K; k++) {
      uint32_t t = x[j] ^ ws.value0[k];
      if (MASKED) t &= ws.mask0[k];
      if (TWO) {
        uint32_t u = x[j + 4] ^ ws.value1[k];
        t = MASKED ? ((u & ws.mask1[k]) | t) : (u | t);  // v_and_or_b32
      }
      accs[j & 3] = accs[j & 3] < t ? accs[j & 3] : t;
    }
  }
  const uint32_t m01 = accs[0] < accs[1] ? accs[0] : accs[1];
  const uint32_t m23 = accs[2] < accs[3] ? accs[2] : accs[3];
  const uint32_t acc = m01 < m23 ? m01 : m23;
  if (__ballot(acc == 0) == 0) return;  // wave-uniform: the common case leaves here

  // rare path: per-lane 16-bit hit mask -> the wave's region of the hit list
  uint32_t hm = 0;
#pragma unroll
  for (int j = 0; j < 16; j++) {
    uint32_t best = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < K; k++) {
      uint32_t t = x[j] ^ ws.value0[k];
      if (MASKED) t &= ws.mask0[k];
      if (TWO) {
        uint32_t u = x[j + 4] ^ ws.value1[k];
        t = MASKED ? ((u & ws.mask1[k]) | t) : (u | t);
      }
      best = best < t ? best : t;
    }
    hm |= static_cast<uint32_t>(best == 0) << j;
  }
  // positions outside [wlo, whi) only exist in the first / last chunk of the range
  const uint64_t chunk_base = at - static_cast<uint64_t>(lane_id()) * 16;
  if (chunk_base < a.wlo || chunk_base + kChunk > a.whi) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint64_t w = at + j;
      if (w < a.wlo || w >= a.whi) hm &= ~(1u << j);
    }
  }
  hits.push_bits(hm, at, ws.offset);
}

// lane i <- lane i + 1 (wave_shl:1); lane 63 keeps `last`
__device__ __forceinline__ uint32_t lane_above_or(uint32_t v, uint32_t last) {
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(last), static_cast<int>(v), 0x130, 0xF, 0xF, false));
}

template <bool TWO>
__device__ __forceinline__ void load_chunk(const uint8_t* text, uint64_t at, uint32_t (&d)[6]) {
  const uint4 v = *reinterpret_cast<const uint4*>(text + at);   // (default policy: the halo load below asks for the same lines)
  d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  if (TWO) {  // the neighbour's first 8 bytes (same cache lines: L1 hits, no extra HBM traffic)
    const uint2 h = *reinterpret_cast<const uint2*>(text + at + 16);
    d[4] = h.x; d[5] = h.y;
  } else {
    d[4] = *reinterpret_cast<const uint32_t*>(text + at + 16);
    d[5] = 0;
  }
}

// NT: the text is loaded with the non-temporal policy (stream_load.h) -- a kernel that reads its text ONCE.  The train reads a
// wave's span once per pattern and lives on passes 2..P finding it in the Infinity Cache: default policy there (measured with nt:
// 0.72 -> 0.83 ms per nine-pattern launch over 500 MB).
template <int K, bool TWO, bool MASKED, bool TWOLEVEL, bool NIB, bool NT = true>
__device__ __forceinline__ void scan_windows_body(const ScanParams& a, const WindowSet& ws) {
  const int lane = lane_id();
  const uint64_t wave = scalar_wave_index();
  if (a.zero_counters != nullptr && wave == 0 && lane < kCntSize) a.zero_counters[lane] = 0;
  RegionHits hits{a.hits + wave * a.region_cap, a.region_cap, 0u};
  const uint64_t first_chunk = a.wlo / kChunk;
  const uint64_t end_chunk = (a.whi + kChunk - 1) / kChunk;
  const WaveSpan span = wave_span(a, wave, first_chunk, end_chunk);
  // chunks below fast_end can be loaded without guards (16 B + 8 B halo stay < n)
  uint64_t fast_end = a.n >= kChunk + 8 ? (a.n - 8) / kChunk : 0;
  if (fast_end > span.c1) fast_end = span.c1;
  if (fast_end < span.c0) fast_end = span.c0;
  const uint64_t lane_off = static_cast<uint64_t>(lane) * 16;

  // Software-pipelined streaming loop, FOUR register buffers of 16 bytes per lane: while chunk c is compared the loads of
  // chunks c+2 and c+3 are in flight (2 KiB per wave), and chunk c+1 has landed -- lane 63 takes the bytes that follow its
  // own 16 from there (lane 0's first dwords, a v_readfirstlane), every other lane from the lane above it (DPP): NO halo load.
  // Round 6: with the halo loaded (rounds 1-5: 8 bytes at +16, "the same cache lines: L1 hits") every line of the text was
  // asked for by two instructions, and a line loaded with the non-temporal policy does not wait in the cache for the second
  // one -- the nt loads (stream_load.h) bought nothing until the halo load was gone.  The prologue and the steady loop run
  // only when all their loads exist, so every load is unconditional and the compiler can count them (it waits for exactly the
  // buffer it needs); a conditional prologue made the count at the loop head ambiguous and the compiler waited for ALL loads
  // there.  Short spans and the last chunks of a span take the plain loop below (lane 63 loads its 8 bytes); no chunk is
  // loaded twice.
  {
    auto ld = [&](uint64_t c) { return NT ? stream_load16(a.text + c * kChunk + lane_off) : *reinterpret_cast<const uint4*>(a.text + c * kChunk + lane_off); };
    // chunk c = q, the chunk behind it begins with the dwords (nx, ny) (wave-uniform)
    auto proc = [&](const uint4& q, uint32_t nx, uint32_t ny, uint64_t c) __attribute__((always_inline)) {
      uint32_t d[6];
      d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
      d[4] = lane_above_or(q.x, nx);
      d[5] = TWO ? lane_above_or(q.y, ny) : 0u;
      windows_chunk<K, TWO, MASKED, TWOLEVEL, NIB>(d, c * kChunk + lane_off, a, ws, hits);
    };
    auto first_of = [](uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); };
    uint4 q0, q1, q2, q3;
    uint64_t c = span.c0;
    if (c + 7 < fast_end) {
      q0 = ld(c);
      q1 = ld(c + 1);
      q2 = ld(c + 2);
      q3 = ld(c + 3);
      while (c + 7 < fast_end) {
        proc(q0, first_of(q1.x), first_of(q1.y), c);
        q0 = ld(c + 4);
        __builtin_amdgcn_sched_barrier(0);
        proc(q1, first_of(q2.x), first_of(q2.y), c + 1);
        q1 = ld(c + 5);
        __builtin_amdgcn_sched_barrier(0);
        proc(q2, first_of(q3.x), first_of(q3.y), c + 2);
        q2 = ld(c + 6);
        __builtin_amdgcn_sched_barrier(0);
        proc(q3, first_of(q0.x), first_of(q0.y), c + 3);
        q3 = ld(c + 7);
        __builtin_amdgcn_sched_barrier(0);
        c += 4;
      }
      // q0..q3 hold chunks c .. c+3 (the loads of the last iteration, all inside the span); what follows c+3 is not loaded
      proc(q0, first_of(q1.x), first_of(q1.y), c);
      proc(q1, first_of(q2.x), first_of(q2.y), c + 1);
      proc(q2, first_of(q3.x), first_of(q3.y), c + 2);
      q0 = q3;
      c += 3;
    } else if (c < fast_end) {
      q0 = ld(c);
    }
    // chunk c in q0; the 8 bytes behind it from one load of lane 63's (chunks below fast_end: they lie inside the text)
    for (; c < fast_end; c++) {
      uint2 h = make_uint2(0, 0);
      if (lane == kWave - 1) h = *reinterpret_cast<const uint2*>(a.text + (c + 1) * kChunk);
      const uint4 cur = q0;
      if (c + 1 < fast_end) q0 = ld(c + 1);
      proc(cur, h.x, h.y, c);   // (lane 63 keeps its own h: lane_above_or's `last` is per lane)
    }
  }
  // tail: the chunk(s) that touch the end of the text use guarded byte loads
  for (uint64_t t = fast_end; t < span.c1; t++) {
    uint32_t d[6];
    load_guarded(a.text, a.n, t * kChunk + lane_off, d);
    windows_chunk<K, TWO, MASKED, TWOLEVEL, NIB>(d, t * kChunk + lane_off, a, ws, hits);
  }
  if (lane == 0) a.hit_counts[wave] = hits.count;
}

template <int K, bool TWO, bool MASKED, bool TWOLEVEL, bool NIB>
__global__ __launch_bounds__(256) void scan_windows(ScanParams a, WindowSet ws) {
  scan_windows_body<K, TWO, MASKED, TWOLEVEL, NIB>(a, ws);
}

// A TRAIN of scans in one launch: the same streaming scan for several patterns, one after the other, every
// wave over its own span -- pattern p + 1 starts in a wave as soon as that wave is done with pattern p.
// Launched one by one (rj_multi mode 1 of round 1) every kernel boundary cost the drain of the last
// workgroups and the ramp-up of the next grid: 94-96 us per 500 MB pattern against 88 us in the
// kernel's steady state.  More than the boundaries is saved: a wave's passes 2..P run over the 32 KB span it
// has just read, and the spans of all resident waves (~150 MB) fit the 256 MiB Infinity Cache, so only the
// first pass of a span comes from HBM -- 77 us per 500 MB pattern.  (On a text several times the cache --
// bench.py `hbm_not_cache`, 2.5 GB -- the spans are 5 x larger and the passes stream from HBM again.)  The
// launch's algorithmic bytes are patterns x text bytes.  Two 5..8-byte nibble-form windows per pattern
// (regexdna's shape); every other set of patterns gets one launch per pattern.
template <bool MASKED>
__global__ __launch_bounds__(256) void scan_windows_train(TrainParams t) {
  for (uint32_t p = 0; p < t.n_patterns; p++) {
    ScanParams a;
    a.text = t.text;
    a.n = t.n;
    a.sb = t.sb;
    a.se = t.se;
    a.wlo = t.wlo[p];
    a.whi = t.whi[p];
    a.span_chunks = t.span_chunks;
    a.hits = t.hits[p];
    a.region_cap = t.region_cap[p];
    a.hit_counts = t.hit_counts[p];
    a.zero_counters = t.zero_counters[p];
    WindowSet ws;
    ws.value0[0] = t.value[p][0];
    ws.value0[1] = t.value[p][1];
    ws.mask0[0] = t.mask[p][0];
    ws.mask0[1] = t.mask[p][1];
    ws.offset = t.offset[p];
    ws.len = t.len[p];
    scan_windows_body<2, true, MASKED, false, true, false>(a, ws);
  }
}

// ---------------------------------------------------------------------------------------
// Fused fast-forward scan: P patterns in ONE pass over the text (regexdna's nine patterns read the
// same 500 MB nine times otherwise).  The nibble-packed dwords pk[j] of a chunk are built once;
// each pattern then costs two v_bitop3 + one v_min3 per position.  29 VALU per text byte for 9
// patterns makes this kernel VALU-bound (~1.3 TB/s of text, i.e. ~12 TB/s of "pattern-bytes"),
// but it moves 1/9 of the HBM bytes of nine separate scans.  Hits go to per-pattern regions, so
// everything downstream is the single-pattern pipeline (its kernels take grid.y = pattern).
__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) {
  const uint32_t ab = a < b ? a : b;
  return ab < c ? ab : c;
}

__device__ __forceinline__ void fused_chunk(const uint32_t (&d)[6], uint64_t at, const FusedParams& a, uint32_t* counts,
                                            uint64_t wave) {
  uint32_t nib[6], z[5], pk[16];  // packed on the aligned dwords first, see windows_chunk
#pragma unroll
  for (int q = 0; q < 6; q++) nib[q] = d[q] & 0x0F0F0F0Fu;
#pragma unroll
  for (int q = 0; q < 5; q++) z[q] = nib[q] | (nib[q + 1] << 4);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    pk[4 * q] = z[q];
    pk[4 * q + 1] = __builtin_amdgcn_alignbyte(z[q + 1], z[q], 1);
    pk[4 * q + 2] = __builtin_amdgcn_alignbyte(z[q + 1], z[q], 2);
    pk[4 * q + 3] = __builtin_amdgcn_alignbyte(z[q + 1], z[q], 3);
  }
  for (uint32_t g = 0; g < a.n_patterns; g += kFuseGroup) {
    uint32_t acc[kFuseGroup];
#pragma unroll
    for (int u = 0; u < kFuseGroup; u++) {
      const uint32_t v0 = a.value[g + u][0], m0 = a.mask[g + u][0], v1 = a.value[g + u][1], m1 = a.mask[g + u][1];
      uint32_t c[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};  // independent chains
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const uint32_t t0 = (pk[j] ^ v0) & m0, t1 = (pk[j] ^ v1) & m1;
        c[j & 3] = umin3(c[j & 3], t0, t1);  // one v_min3_u32 per position
      }
      acc[u] = umin3(c[0], c[1], c[2] < c[3] ? c[2] : c[3]);
    }
    uint32_t any = acc[0];
#pragma unroll
    for (int u = 1; u < kFuseGroup; u++) any = any < acc[u] ? any : acc[u];
    if (__ballot(any == 0) == 0) continue;  // wave-uniform: no pattern of the group hits in this chunk
    // rare path, pattern by pattern
#pragma unroll
    for (int u = 0; u < kFuseGroup; u++) {
      if (__ballot(acc[u] == 0) == 0) continue;  // this pattern has no hit in the chunk
      const uint32_t p = g + u;
      const uint32_t v0 = a.value[p][0], m0 = a.mask[p][0], v1 = a.value[p][1], m1 = a.mask[p][1];
      uint32_t hm = 0;
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const uint32_t t0 = (pk[j] ^ v0) & m0, t1 = (pk[j] ^ v1) & m1;
        hm |= static_cast<uint32_t>((t0 < t1 ? t0 : t1) == 0) << j;
      }
      // window positions of pattern p: w = s + offset, sb <= s < se, and the window must fit
      const uint64_t wlo = a.sb + a.offset[p];
      const uint64_t last_w = a.n >= a.len[p] ? a.n - a.len[p] + 1 : 0;
      uint64_t whi = a.se + a.offset[p];
      if (whi > last_w) whi = last_w;
      const uint64_t chunk_base = at - static_cast<uint64_t>(lane_id()) * 16;
      if (chunk_base < wlo || chunk_base + kChunk > whi) {  // only the first / last chunks of the range
#pragma unroll
        for (int j = 0; j < 16; j++) {
          const uint64_t w = at + j;
          if (w < wlo || w >= whi) hm &= ~(1u << j);
        }
      }
      RegionHits hits{a.hits[p] + wave * a.region_cap[p], a.region_cap[p], counts[p]};
      hits.push_bits(hm, at, a.offset[p]);
      counts[p] = hits.count;  // wave-uniform, every lane stores the same value
    }
  }
}

// The same with a SHARED prefilter (FusedParams::n_bases > 0): when every window of every pattern is
// within one nibble of one of NB base windows -- regexdna: all 18 windows are `agggtaaa` or `tttaccct`
// with at most one position turned into a class -- a text position can only hit if it differs from a
// base in at most ONE nibble.  That test is shared by all patterns and costs, per position and base,
//     u = (pk ^ base) + 0x77777777      bit 3 of a nibble <=> that nibble differs       (v_xad_u32)
//     c = popcount(u & 0x88888888)      differing nibbles                              (v_and, v_bcnt)
// plus one v_min3 for both bases: 7 VALU per position instead of 3 per position AND PATTERN (27 for
// regexdna's nine).  Nibbles are compared on their low 3 bits (bit 3 must be free for the carry-less
// add): one more superset step, removed like every alias by the exact verification downstream.
// Only chunks in which some lane passes the prefilter (about every second one on DNA: the true match
// density is one per 2.4 KiB) run the exact per-pattern tests, and only for the hit positions' chains.
template <int NB>
__device__ __forceinline__ void fused_chunk_d1(const uint32_t (&d)[6], uint64_t at, const FusedParams& a, uint32_t* counts,
                                               uint64_t wave) {
  uint32_t nib[6], z[5], pk[16];
#pragma unroll
  for (int q = 0; q < 6; q++) nib[q] = d[q] & 0x07070707u;
#pragma unroll
  for (int q = 0; q < 5; q++) z[q] = nib[q] | (nib[q + 1] << 4);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    pk[4 * q] = z[q];
    pk[4 * q + 1] = __builtin_amdgcn_alignbyte(z[q + 1], z[q], 1);
    pk[4 * q + 2] = __builtin_amdgcn_alignbyte(z[q + 1], z[q], 2);
    pk[4 * q + 3] = __builtin_amdgcn_alignbyte(z[q + 1], z[q], 3);
  }
  const uint32_t b0 = a.base[0], b1 = a.base[NB > 1 ? 1 : 0];
  const uint32_t c77 = 0x77777777u;
  // (pk ^ base) + 0x77777777 in ONE instruction: the compiler emits v_xor + v_add for the C expression
  auto xad = [&](uint32_t x, uint32_t base) -> uint32_t {
    uint32_t u;
    asm("v_xad_u32 %0, %1, %2, %3" : "=v"(u) : "v"(x), "s"(base), "v"(c77));
    return u;
  };
  uint32_t acc[8];  // chain q holds the positions q and q + 8
#pragma unroll
  for (int q = 0; q < 8; q++) acc[q] = 8;
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const uint32_t c0 = __builtin_popcount(xad(pk[j], b0) & 0x88888888u);
    if (NB > 1) {
      const uint32_t c1 = __builtin_popcount(xad(pk[j], b1) & 0x88888888u);
      acc[j & 7] = umin3(acc[j & 7], c0, c1);
    } else {
      acc[j & 7] = acc[j & 7] < c0 ? acc[j & 7] : c0;
    }
  }
  const uint32_t m0 = umin3(acc[0], acc[1], acc[2]), m1 = umin3(acc[3], acc[4], acc[5]);
  const uint32_t best = umin3(m0, m1, acc[6] < acc[7] ? acc[6] : acc[7]);
  if (__ballot(best <= 1) == 0) return;  // wave-uniform: no position of the chunk is near a base
  // which chains hold a hit (wave-uniform mask)
  uint32_t chains = 0;
#pragma unroll
  for (int q = 0; q < 8; q++) chains |= (__ballot(acc[q] <= 1) != 0 ? 1u : 0u) << q;
  const uint64_t chunk_base = at - static_cast<uint64_t>(lane_id()) * 16;
  for (uint32_t p = 0; p < a.n_patterns; p++) {
    if (a.region_cap[p] == 0) continue;  // padding entry
    const uint32_t v0 = a.value[p][0], k0 = a.mask[p][0], v1 = a.value[p][1], k1 = a.mask[p][1];
    uint32_t hm = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      if (((chains >> q) & 1u) == 0) continue;  // uniform
      {
        const uint32_t t0 = (pk[q] ^ v0) & k0, t1 = (pk[q] ^ v1) & k1;
        hm |= static_cast<uint32_t>((t0 < t1 ? t0 : t1) == 0) << q;
      }
      {
        const uint32_t t0 = (pk[q + 8] ^ v0) & k0, t1 = (pk[q + 8] ^ v1) & k1;
        hm |= static_cast<uint32_t>((t0 < t1 ? t0 : t1) == 0) << (q + 8);
      }
    }
    if (__ballot(hm != 0) == 0) continue;
    const uint64_t wlo = a.sb + a.offset[p];
    const uint64_t last_w = a.n >= a.len[p] ? a.n - a.len[p] + 1 : 0;
    uint64_t whi = a.se + a.offset[p];
    if (whi > last_w) whi = last_w;
    if (chunk_base < wlo || chunk_base + kChunk > whi) {  // only the first / last chunks of the range
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const uint64_t w = at + j;
        if (w < wlo || w >= whi) hm &= ~(1u << j);
      }
    }
    RegionHits hits{a.hits[p] + wave * a.region_cap[p], a.region_cap[p], counts[p]};
    hits.push_bits(hm, at, a.offset[p]);
    counts[p] = hits.count;
  }
}

template <int NB>
__device__ __forceinline__ void fused_any(const uint32_t (&d)[6], uint64_t at, const FusedParams& a, uint32_t* counts, uint64_t wave) {
  if (NB == 0) fused_chunk(d, at, a, counts, wave);
  else fused_chunk_d1<NB>(d, at, a, counts, wave);
}

template <int NB>
__global__ __launch_bounds__(256) void scan_windows_fused(FusedParams a) {
  __shared__ uint32_t region_count[4][kMaxFused];
  const int lane = lane_id();
  const uint64_t wave = scalar_wave_index();
  uint32_t* counts = region_count[threadIdx.x >> 6];
  if (lane < kMaxFused) counts[lane] = 0;
  if (wave == 0 && lane < kCntSize)
    for (uint32_t p = 0; p < a.n_patterns; p++)
      if (a.zero_counters[p] != nullptr) a.zero_counters[p][lane] = 0;
  const uint64_t first_chunk = a.sb / kChunk;
  // a window may begin up to 7 bytes after its start
  const uint64_t end_byte = a.se + 8 < a.n ? a.se + 8 : a.n;
  const uint64_t end_chunk = (end_byte + kChunk - 1) / kChunk;
  WaveSpan span;
  span.c0 = first_chunk + wave * a.span_chunks;
  span.c1 = span.c0 + a.span_chunks;
  if (span.c0 > end_chunk) span.c0 = end_chunk;
  if (span.c1 > end_chunk) span.c1 = end_chunk;
  uint64_t fast_end = a.n >= kChunk + 8 ? (a.n - 8) / kChunk : 0;
  if (fast_end > span.c1) fast_end = span.c1;
  if (fast_end < span.c0) fast_end = span.c0;
  const uint64_t lane_off = static_cast<uint64_t>(lane) * 16;
  {
    // the same 3-deep register pipeline as scan_windows (unconditional prologue, see there)
    uint32_t b0[6], b1[6], b2[6];
    uint64_t c = span.c0;
    if (c + 5 < fast_end) {
      load_chunk<true>(a.text, c * kChunk + lane_off, b0);
      load_chunk<true>(a.text, (c + 1) * kChunk + lane_off, b1);
      load_chunk<true>(a.text, (c + 2) * kChunk + lane_off, b2);
      while (c + 5 < fast_end) {
        fused_any<NB>(b0, c * kChunk + lane_off, a, counts, wave);
        load_chunk<true>(a.text, (c + 3) * kChunk + lane_off, b0);
        __builtin_amdgcn_sched_barrier(0);
        fused_any<NB>(b1, (c + 1) * kChunk + lane_off, a, counts, wave);
        load_chunk<true>(a.text, (c + 4) * kChunk + lane_off, b1);
        __builtin_amdgcn_sched_barrier(0);
        fused_any<NB>(b2, (c + 2) * kChunk + lane_off, a, counts, wave);
        load_chunk<true>(a.text, (c + 5) * kChunk + lane_off, b2);
        __builtin_amdgcn_sched_barrier(0);
        c += 3;
      }
      fused_any<NB>(b0, c * kChunk + lane_off, a, counts, wave);
      fused_any<NB>(b1, (c + 1) * kChunk + lane_off, a, counts, wave);
      fused_any<NB>(b2, (c + 2) * kChunk + lane_off, a, counts, wave);
      c += 3;
    }
    for (; c < fast_end; c++) {
      load_chunk<true>(a.text, c * kChunk + lane_off, b0);
      fused_any<NB>(b0, c * kChunk + lane_off, a, counts, wave);
    }
  }
  for (uint64_t t = fast_end; t < span.c1; t++) {
    uint32_t d[6];
    load_guarded(a.text, a.n, t * kChunk + lane_off, d);
    fused_any<NB>(d, t * kChunk + lane_off, a, counts, wave);
  }
  if (lane < static_cast<int>(a.n_patterns) && a.region_cap[lane] != 0) a.hit_counts[lane][wave] = counts[lane];
}

// ---------------------------------------------------------------------------------------
// Dense scan: every position s in [sb, se) that can start a match goes to the hit list.
__global__ __launch_bounds__(256) void scan_dense(ScanParams a, DevProgram P) {
  __shared__ uint32_t fb[8];
  if (threadIdx.x < 8) fb[threadIdx.x] = P.first_bytes[threadIdx.x];
  __syncthreads();
  const int lane = lane_id();
  const uint64_t wave = scalar_wave_index();
  RegionHits hits{a.hits + wave * a.region_cap, a.region_cap, 0u};
  const uint64_t first_chunk = a.sb / kChunk;
  const uint64_t end_chunk = (a.se + kChunk - 1) / kChunk;  // se <= n + 1
  const WaveSpan span = wave_span(a, wave, first_chunk, end_chunk);
  const bool ctxed = P.n_ctx > 1;

  for (uint64_t c = span.c0; c < span.c1; c++) {
    const uint64_t base = c * kChunk;
    const uint64_t at = base + static_cast<uint64_t>(lane) * 16;
    uint32_t d[6];
    if (base + kChunk <= a.n) {
      const uint4 v = *reinterpret_cast<const uint4*>(a.text + at);
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    } else {
      load_guarded(a.text, a.n, at, d);
    }
    uint32_t prev = '\n';  // byte before the lane's first byte ('\n' stands for "start of text")
    if (at > 0 && at <= a.n) prev = a.text[at - 1];
    uint32_t cand = 0;  // bit j: position at + j is a candidate start
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint32_t cur = (d[j >> 2] >> (8 * (j & 3))) & 0xFFu;
      const uint64_t s = at + j;
      bool ok = false;
      if (s < a.n) ok = (fb[cur >> 5] >> (cur & 31)) & 1u;
      if (P.nullable && s <= a.n) {
        int ctx = 0;
        if (ctxed) {
          if (s == 0 || rj_line_break(prev)) ctx |= 1;
          if (s == a.n || rj_line_break(cur)) ctx |= 2;
        }
        ok = ok || ((P.nullable >> ctx) & 1u);
      }
      ok = ok && s >= a.sb && s < a.se;
      cand |= static_cast<uint32_t>(ok) << j;
      prev = cur;
    }
    if (__ballot(cand != 0) == 0) continue;
    hits.push_bits(cand, at, 0);
  }
  if (lane == 0) a.hit_counts[wave] = hits.count;
}


// ---------------------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------------------
// Launchers (host side of the <<< >>> syntax lives here so engine.cc stays plain C++).
ScanGeometry scan_geometry(uint64_t chunks, uint64_t chunks_per_block) {
  // Measured on MI355X (tools/ab_probe.py): a grid of exactly the resident workgroups loses
  // ~12% to the partially filled last round; large texts stream best with >= 16 Ki workgroups
  // (6.2-6.4 TB/s) while tiny spans waste the pipeline prologue, so aim at >= 32 chunks per
  // wave and cap at 16 Ki workgroups.
  uint64_t blocks = chunks / chunks_per_block;
  if (blocks > 16384) blocks = 16384;
  if (blocks < 256) blocks = (chunks + 3) / 4 < 256 ? (chunks + 3) / 4 : 256;
  if (blocks == 0) blocks = 1;
  static const char* env_grid = getenv("RJ_SCAN_GRID");  // measurement override
  if (env_grid && atoi(env_grid) > 0) blocks = static_cast<uint64_t>(atoi(env_grid));
  ScanGeometry g;
  g.grid = static_cast<int>(blocks);
  g.n_regions = static_cast<uint32_t>(blocks * 4);
  g.span_chunks = (chunks + g.n_regions - 1) / g.n_regions;
  if (g.span_chunks == 0) g.span_chunks = 1;
  return g;
}

template <bool TWO, bool MASKED, bool TWOLEVEL, bool NIB>
static void launch_windows_k(int k, const ScanParams& a, const WindowSet& ws, int grid, hipEvent_t t0, hipEvent_t t1,
                             hipStream_t st) {
  // K is rounded up to an instantiated size; the host pads the window set with copies
  if (k <= 1) hipExtLaunchKernelGGL((scan_windows<1, TWO, MASKED, TWOLEVEL, NIB>), dim3(grid), dim3(256), 0, st, t0, t1, 0, a, ws);
  else if (k == 2) hipExtLaunchKernelGGL((scan_windows<2, TWO, MASKED, TWOLEVEL, NIB>), dim3(grid), dim3(256), 0, st, t0, t1, 0, a, ws);
  else if (k == 3) hipExtLaunchKernelGGL((scan_windows<3, TWO, MASKED, TWOLEVEL, NIB>), dim3(grid), dim3(256), 0, st, t0, t1, 0, a, ws);
  else if (k == 4) hipExtLaunchKernelGGL((scan_windows<4, TWO, MASKED, TWOLEVEL, NIB>), dim3(grid), dim3(256), 0, st, t0, t1, 0, a, ws);
  else if (k <= 6) hipExtLaunchKernelGGL((scan_windows<6, TWO, MASKED, TWOLEVEL, NIB>), dim3(grid), dim3(256), 0, st, t0, t1, 0, a, ws);
  else hipExtLaunchKernelGGL((scan_windows<8, TWO, MASKED, TWOLEVEL, NIB>), dim3(grid), dim3(256), 0, st, t0, t1, 0, a, ws);
}

void launch_scan_windows(const ScanParams& a, const WindowSet& ws, int n_windows, int grid, hipEvent_t t0, hipEvent_t t1,
                         hipStream_t st) {
  const bool two = ws.len > 4;
  if (two) {
    if (ws.two_level) {
      if (ws.masked) launch_windows_k<true, true, true, false>(n_windows, a, ws, grid, t0, t1, st);
      else launch_windows_k<true, false, true, false>(n_windows, a, ws, grid, t0, t1, st);
    } else if (ws.nibble) {
      if (ws.masked) launch_windows_k<true, true, false, true>(n_windows, a, ws, grid, t0, t1, st);
      else launch_windows_k<true, false, false, true>(n_windows, a, ws, grid, t0, t1, st);
    } else {
      if (ws.masked) launch_windows_k<true, true, false, false>(n_windows, a, ws, grid, t0, t1, st);
      else launch_windows_k<true, false, false, false>(n_windows, a, ws, grid, t0, t1, st);
    }
  } else {
    if (ws.masked) launch_windows_k<false, true, false, false>(n_windows, a, ws, grid, t0, t1, st);
    else launch_windows_k<false, false, false, false>(n_windows, a, ws, grid, t0, t1, st);
  }
}

void launch_scan_windows_train(const TrainParams& t, bool masked, int grid, hipEvent_t t0, hipEvent_t t1, hipStream_t st) {
  if (masked) hipExtLaunchKernelGGL((scan_windows_train<true>), dim3(grid), dim3(256), 0, st, t0, t1, 0, t);
  else hipExtLaunchKernelGGL((scan_windows_train<false>), dim3(grid), dim3(256), 0, st, t0, t1, 0, t);
}

void launch_scan_windows_fused(const FusedParams& a, int grid, hipEvent_t t0, hipEvent_t t1, hipStream_t st) {
  if (a.n_bases == 0) hipExtLaunchKernelGGL((scan_windows_fused<0>), dim3(grid), dim3(256), 0, st, t0, t1, 0, a);
  else if (a.n_bases == 1) hipExtLaunchKernelGGL((scan_windows_fused<1>), dim3(grid), dim3(256), 0, st, t0, t1, 0, a);
  else hipExtLaunchKernelGGL((scan_windows_fused<2>), dim3(grid), dim3(256), 0, st, t0, t1, 0, a);
}

void launch_scan_dense(const ScanParams& a, const DevProgram& P, int grid, hipEvent_t t0, hipEvent_t t1, hipStream_t st) {
  hipExtLaunchKernelGGL(scan_dense, dim3(grid), dim3(256), 0, st, t0, t1, 0, a, P);
}
}  // namespace rejit_amd
