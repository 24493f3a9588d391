// rejit_amd/csrc/emit_scan.hip -- assertion-only patterns (`^`, `$`, `^$`: DevProgram::n_pos == 0) in ONE pass
// with the matches written ONCE, at their final place.
//
// The line table of a grep-like caller is a MatchAll of "^" (reference sample/jrep.cc:239,294: the second pass over
// every file with a match; the reference scans for line breaks with pcmpistri, FastForwardGen::
// VisitSingleStartOrEndOfLine, src/x64/codegen-x64.cc:1568-1677).  Its OUTPUT is the traffic: 16 bytes per line
// start next to 1 byte read per text byte.  The dense kernel (scan_dense_walk, kernels.hip) serves it like every
// pattern: matches go to the wave's region, offsets_gather_check lays the regions out, gather_regions_by_wave
// copies them -- every pair written twice and read once in between, 8.9 GB moved for 6.3 GB of algorithmic bytes
// over a 5 GB text, 2.47 ms.
//
// Here the final position of a tile's matches comes from a single-pass prefix scan over the tiles before it (after
// Merrill & Garland's decoupled look-back; tile_lookback.h): a wave takes a tile of 32 KiB, finds the line breaks of its
// 32 chunks (16 bits per lane and chunk: the masks of the whole tile are 16 registers), counts, and the workgroup publishes
// the count of its four tiles.  The masks then wait in LDS while the workgroup does its NEXT four tiles; only after that
// is the count of everything before them looked up -- it was published long ago by then, nothing waits -- and the pairs are
// written where they belong.  (Resolving at once, as round 3 did and the textbook does, cost a quarter of the kernel: the
// ~1800 resident workgroups finish a round nearly in lock-step, and every one of them waited for the slowest.)  Tiles are
// handed out in the order in which workgroups ARRIVE (one ticket per workgroup and round, a tile per wave), so everything a
// wave waits for is owned by a wave that has started -- no assumption about dispatch order or residency.  The words of the
// scan are written and read with relaxed agent-scope atomics (the data is the flag; MI355X guide, "R2"); every spin is
// bounded: on a time-out the kernel flags the run and the engine repeats it on the dense kernel.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "dense_swar.h"
#include "device_program.h"
#include "kernels.h"
#include "tile_lookback.h"

namespace rejit_amd {

namespace {

constexpr int kWave = 64;
constexpr uint64_t kChunk = 1024;
constexpr int kTileChunks = 32;                        // chunks per tile (four tiles, one per wave, are a unit of the prefix scan)
constexpr uint64_t kTile = kChunk * kTileChunks;       // 32 KiB
constexpr int kDepth = 8;                              // chunk loads in flight per wave

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
__device__ __forceinline__ uint32_t wave_from_lane_below(uint32_t x) { return dpp_or_zero<0x138, 0xF>(x); }  // wave_shr:1
__device__ __forceinline__ uint32_t wave_last_lane(uint32_t x) { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(x), kWave - 1)); }

// Line breaks among the lane's 16 bytes.  0x0a and 0x0d differ in exactly their low three bits (010 / 101), so with
// y = byte ^ 0x0a a line break is y == 0 or y == 7: bits 3..7 clear and bits 0, 1, 2 all equal.  Per dword:
//   w     = y in bits 3..7 and bit 0, y ^ (y << 1) in bits 1, 2    (one v_bitop3: the constant selects)
//   zero  = no bit of w other than bit 0 set, per byte             (and, add, one v_bitop3: the SWAR zero test)
// and the four 0x80 flags of a dword become four mask bits with ONE v_dot4_u32_u8 (weights 1, 2, 4, 8 -- or 16 ..
// 128 for the next dword, accumulating): 6 + 1 instructions per dword where two zero tests and a shift cascade
// took 19.
__device__ __forceinline__ uint32_t break_flags(uint32_t d) {
  const uint32_t y = d ^ 0x0a0a0a0au;
  const uint32_t s = y << 1;
  const uint32_t w = ((y ^ s) & 0x06060606u) | (y & ~0x06060606u);
  const uint32_t t = (w & 0x7e7e7e7eu) + 0x7f7f7f7fu;
  return ~(t | w) & 0x80808080u;  // 0x80 in every byte that is a line break
}
__device__ __forceinline__ uint32_t line_breaks16(const uint4& v) {
  const uint32_t lo = __builtin_amdgcn_udot4(break_flags(v.x), 0x08040201u, __builtin_amdgcn_udot4(break_flags(v.y), 0x80402010u, 0u, false), false);
  const uint32_t hi = __builtin_amdgcn_udot4(break_flags(v.z), 0x08040201u, __builtin_amdgcn_udot4(break_flags(v.w), 0x80402010u, 0u, false), false);
  return (lo >> 7) | ((hi >> 7) << 8);  // (the flags weigh 128 each)
}

// 16 bytes of the lane, zeros beyond the end of the text
__device__ __forceinline__ uint4 load16_guarded(const uint8_t* text, uint64_t n, uint64_t at) {
  if (at + 16 <= n) return *reinterpret_cast<const uint4*>(text + at);
  uint32_t d[4] = {0, 0, 0, 0};
#pragma unroll 1
  for (int k = 0; k < 16; k++)
    if (at + k < n) d[k >> 2] |= static_cast<uint32_t>(text[at + k]) << (8 * (k & 3));
  return make_uint4(d[0], d[1], d[2], d[3]);
}

// the candidate mask of one chunk: bit j = position at + j is a match
__device__ __forceinline__ uint32_t chunk_mask(const uint4& v, uint64_t at, uint64_t n, uint64_t sb, uint64_t se, uint32_t nullable, bool inner,
                                               uint32_t* carry_lb) {
  const int lane = lane_id();
  uint32_t lb = line_breaks16(v);
  uint32_t lt_n = 0xFFFFu, le_n = 0xFFFFu;
  if (!inner) {
    // positions with a byte (s < n) / positions at all (s <= n)
    lt_n = n > at ? (n - at < 16 ? (1u << (n - at)) - 1u : 0xFFFFu) : 0u;
    le_n = n >= at ? (n - at < 15 ? (2u << (n - at)) - 1u : 0xFFFFu) : 0u;
    lb &= lt_n;
  }
  uint32_t prev = wave_from_lane_below(lb >> 15);
  if (lane == 0) prev = *carry_lb;
  *carry_lb = wave_last_lane(lb >> 15);
  const uint32_t sol = ((lb << 1) | prev) & 0xFFFFu;
  uint32_t eol = lb;
  if (!inner && n >= at && n - at < 16) eol |= 1u << (n - at);  // the end of the text
  uint32_t m = 0;
  if (nullable & 1u) m |= ~sol & ~eol;
  if (nullable & 2u) m |= sol & ~eol;
  if (nullable & 4u) m |= ~sol & eol;
  if (nullable & 8u) m |= sol & eol;
  m &= le_n;
  if (!inner && (at < sb || at + 16 > se)) {  // the ends of the own range
    const uint32_t hi = se > at ? (se - at < 16 ? static_cast<uint32_t>(se - at) : 16u) : 0u;
    const uint32_t lo = sb > at ? (sb - at < 16 ? static_cast<uint32_t>(sb - at) : 16u) : 0u;
    m &= ((1u << hi) - 1u) & ~((1u << lo) - 1u);
  }
  return m & 0xFFFFu;
}

}  // namespace

// nullable: bit c = the empty string matches in context c (bit 0 of c: start of line, bit 1: end of line).
// Matches are the positions s in [sb, se) (se <= n + 1) whose context is in `nullable`; out receives (s, s) pairs in
// position order, counters[kCntFinal] their number (also when it exceeds out_cap: the host grows and runs again),
// counters[kCntOverrun] = 1 when a spin timed out.
namespace {

// One tile, first half: its candidate masks (bit j of chunk c's mask = position base + 1024 c + 16 lane + j is a match; two
// chunks per register) and its count (wave-uniform).
__device__ __forceinline__ uint32_t tile_masks(const uint8_t* __restrict__ text, uint64_t n, uint64_t sb, uint64_t se, uint32_t nullable,
                                               bool live, uint64_t base, uint32_t (&mask)[kTileChunks / 2]) {
  const int lane = lane_id();
  uint32_t mine = 0;
  uint32_t carry_lb = 1u;          // was the byte before the chunk a line break (the start of the text counts as one)
  if (live && base > 0) {
    const uint8_t c = text[base - 1];
    carry_lb = (c == '\n' || c == '\r') ? 1u : 0u;
  }
  // a tile strictly inside the text and the own range needs no guards (wave-uniform)
  const bool inner = live && base >= sb && base + kTile + 16 <= n && base + kTile <= se;
  const uint8_t* lane_text = text + base + static_cast<uint64_t>(lane) * 16;
  if (inner) {
#pragma unroll
    for (int g = 0; g < kTileChunks / kDepth; g++) {   // kDepth chunks of loads in flight at a time
      uint4 v[kDepth];
#pragma unroll
      for (int c = 0; c < kDepth; c++) v[c] = *reinterpret_cast<const uint4*>(lane_text + static_cast<uint64_t>(kDepth * g + c) * kChunk);
#pragma unroll
      for (int c = 0; c < kDepth; c++) {
        const int idx = kDepth * g + c;
        const uint64_t at = base + static_cast<uint64_t>(idx) * kChunk + static_cast<uint64_t>(lane) * 16;
        const uint32_t m = chunk_mask(v[c], at, n, sb, se, nullable, true, &carry_lb);
        mine += __popc(m);
        if (idx & 1) mask[idx >> 1] |= m << 16;
        else mask[idx >> 1] = m;
      }
    }
  } else {
#pragma unroll
    for (int h = 0; h < kTileChunks / 2; h++) {
      uint32_t both = 0;
#pragma unroll 1
      for (int q = 0; q < 2; q++) {
        const uint64_t at = base + static_cast<uint64_t>(2 * h + q) * kChunk + static_cast<uint64_t>(lane) * 16;
        const uint64_t chunk_at = base + static_cast<uint64_t>(2 * h + q) * kChunk;
        uint32_t m = 0;
        if (live && chunk_at <= n && chunk_at < se) {  // (wave-uniform; chunks behind the end of the text or the range hold nothing)
          const uint4 v = load16_guarded(text, n, at);
          m = chunk_mask(v, at, n, sb, se, nullable, false, &carry_lb);
        } else {
          carry_lb = 0;
        }
        mine += __popc(m);
        both |= m << (16 * q);
      }
      mask[h] = both;
    }
  }
  return wave_last_lane(wave_inclusive_sum(mine));
}

// Second half, a round later: the pairs of the tile at `base`, whose masks wait in LDS (masks[h * 64 + lane]), at their
// final place: `pos` = the count of everything before the tile.
__device__ __forceinline__ void tile_pairs(const uint32_t* masks, uint64_t base, uint64_t pos, uint64_t* out, uint64_t out_cap) {
  const int lane = lane_id();
#pragma unroll 4
  for (int c = 0; c < kTileChunks; c++) {
    uint32_t m = (masks[(c >> 1) * kWave + lane] >> (16 * (c & 1))) & 0xFFFFu;
    const uint32_t cnt = __popc(m);
    const uint32_t inc = wave_inclusive_sum(cnt);
    uint64_t at_out = pos + inc - cnt;
    const uint64_t at = base + static_cast<uint64_t>(c) * kChunk + static_cast<uint64_t>(lane) * 16;
    while (m) {
      const int j = __ffs(static_cast<int>(m)) - 1;
      m &= m - 1;
      if (at_out < out_cap) *reinterpret_cast<ulonglong2*>(out + 2 * at_out) = make_ulonglong2(at + j, at + j);
      at_out++;
    }
    pos += wave_last_lane(inc);
  }
}

}  // namespace

// Four waves per workgroup share a ticket (four tiles, one each): 38 K tickets over a 5 GB text, well below the
// ~90 atomics per microsecond one address sustains; a ticket per wave would be at that limit.
// Variants measured over 5 GB with a line start every 61 bytes (this form: 1.65 ms, the dense kernel + gather 2.58):
//   * two tiles per wave and ticket: a CONVOY -- a wave waiting in the look-back of its first tile sits on the
//     unpublished count of its second one, the waves behind wait for that count, and the run resolves tile by tile
//     from the front: 310 ms.  A wave must never sit on an unpublished tile.
//   * six waves per SIMD (80 registers, four chunk loads in flight): 1.73 ms -- not latency hiding;
//   * one wave per workgroup, tiles of 128 KiB with the masks in LDS, eight loads in flight, no barrier: 1.83 ms --
//     ten waves per CU leave the ~100 VALU instructions per chunk latency-bound.
// What is left is instruction count (6.4 VALU operations per text byte at ~25 % VALU utilisation).
constexpr int kTilesPerTicket = 4;

__global__ __launch_bounds__(256) void emit_assertions(const uint8_t* __restrict__ text, uint64_t n, uint64_t sb, uint64_t se, uint32_t nullable,
                                                       unsigned long long* granules, unsigned long long* ticket, uint64_t n_tiles,
                                                       uint64_t* out, uint64_t out_cap, unsigned long long* counters,
                                                       unsigned long long* host_counters) {
  __shared__ unsigned long long s_ticket, s_before;
  __shared__ unsigned long long s_count[2][kTilesPerTicket];
  __shared__ uint32_t s_bad;
  __shared__ uint32_t s_masks[kTilesPerTicket][(kTileChunks / 2) * kWave];  // the masks of the round that waits for its prefix: 16 KiB
  const int wv = static_cast<int>(threadIdx.x) >> 6;
  const int lane = lane_id();
  const uint64_t first_tile = sb / kTile;
  const uint64_t n_tickets = (n_tiles + kTilesPerTicket - 1) / kTilesPerTicket;
  if (threadIdx.x == 0) s_bad = 0;
  // One unit of the prefix scan per workgroup and round (four tiles, one per wave), resolved ONE ROUND LATE
  // (tile_lookback.h): a round finds its tiles' masks and publishes their count; then the count of everything before the
  // PREVIOUS round's tiles is looked up -- published long ago by then: no waiting -- and that round's pairs are written from
  // the masks it left in LDS; then this round's masks take their place.
  uint64_t prev_tk = ~0ull;
  int cur = 0;
  for (;;) {
    // ---- a ticket per workgroup and round, in arrival order
    if (threadIdx.x == 0) s_ticket = atomicAdd(ticket, 1ull);
    __syncthreads();
    const uint64_t tk = s_ticket;
    const bool have = tk < n_tickets;
    const uint64_t t = tk * kTilesPerTicket + static_cast<uint64_t>(wv);
    uint32_t mask[kTileChunks / 2];
    if (have) {
      const uint32_t k = tile_masks(text, n, sb, se, nullable, t < n_tiles, (first_tile + t) * kTile, mask);
      if (lane == 0) s_count[cur][wv] = k;
    }
    // wave 0, its own tile done, looks up the count of everything before the PREVIOUS round (the other waves are still at
    // their tiles); behind the barrier it publishes this round's count
    if (wv == 0 && prev_tk != ~0ull) {
      unsigned long long b = 0;
      const bool ok = lookback::resolve(granules, n_tickets, prev_tk, &b);
      if (lane == 0) {
        unsigned long long total = 0;
#pragma unroll
        for (int w = 0; w < kTilesPerTicket; w++) total += s_count[cur ^ 1][w];
        s_before = b;
        if (!ok) {
          s_bad = 1;
          counters[kCntOverrun] = 1;
          if (host_counters) host_counters[kCntOverrun] = 1;
        } else if (prev_tk == n_tickets - 1) {
          counters[kCntFinal] = b + total;
          counters[kCntCands] = b + total;
          counters[kCntHits] = b + total;
          if (host_counters) {
            host_counters[kCntFinal] = b + total;
            host_counters[kCntCands] = b + total;
            host_counters[kCntHits] = b + total;
          }
        }
      }
    }
    __syncthreads();
    if (have && threadIdx.x == 0) {
      unsigned long long total = 0;
#pragma unroll
      for (int w = 0; w < kTilesPerTicket; w++) total += s_count[cur][w];
      lookback::publish(granules, n_tickets, tk, total);
    }
    if (prev_tk != ~0ull && s_bad == 0) {
      const uint64_t pt = prev_tk * kTilesPerTicket + static_cast<uint64_t>(wv);
      unsigned long long b = s_before;
      for (int w = 0; w < wv; w++) b += s_count[cur ^ 1][w];
      if (pt < n_tiles) tile_pairs(s_masks[wv], (first_tile + pt) * kTile, b, out, out_cap);
    }
    if (!have) return;
    // (a wave reads and writes only its own s_masks row; the other shared words of this round are rewritten behind the
    // next round's barriers)
    if (t < n_tiles) {
#pragma unroll
      for (int h = 0; h < kTileChunks / 2; h++) s_masks[wv][h * kWave + lane] = mask[h];
    }
    prev_tk = tk;
    cur ^= 1;
  }
}

void launch_emit_assertions(const uint8_t* text, uint64_t n, uint64_t sb, uint64_t se, uint32_t nullable, unsigned long long* scratch,
                            uint64_t* out, uint64_t out_cap, unsigned long long* counters, unsigned long long* host_counters, hipEvent_t t0,
                            hipEvent_t t1, hipStream_t st) {
  const uint64_t n_tiles = emit_tiles(sb, se);
  (void)hipMemsetAsync(scratch, 0, emit_scratch_bytes(sb, se), st);
  uint64_t blocks = (n_tiles + 3) / 4;
  blocks = blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks;   // persistent: workgroups take tiles until none is left
  hipExtLaunchKernelGGL(emit_assertions, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, t0, t1, 0, text, n, sb, se, nullable,
                        scratch + 1, scratch, n_tiles, out, out_cap, counters, host_counters);
}

uint64_t emit_tiles(uint64_t sb, uint64_t se) {
  if (se <= sb) return 0;
  return (se + kTile - 1) / kTile - sb / kTile;  // tiles that hold a start in [sb, se)
}

size_t emit_scratch_bytes(uint64_t sb, uint64_t se) { return (lookback::granule_words((emit_tiles(sb, se) + 3) / 4) + 1) * sizeof(unsigned long long); }

}  // namespace rejit_amd
