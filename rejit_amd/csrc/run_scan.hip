// rejit_amd/csrc/run_scan.hip -- the kernels of run_scan.h: MatchAll of `X+` / `A L*` / `A L* B` / `X+ B` (/ `A L+` / `A L+ B`: lag_marks) patterns in one or two
// passes over the text and one small scan over tile summaries, whatever the length of the runs (reference: the NFA loop's
// one long-lived thread, src/x64/codegen-x64.cc:535-581, its last accepting position :426-461, the restart behind the
// match :487-503).
//
//   run_summary  a wave per tile of 8 KiB (four iterations of 2 KiB: a lane owns 32 bytes, the next iteration's in flight):
//                the class streams of A, B and the breaks (dense_streams.h: rj_stream_range); the wave's state is the open
//                segment's (s1, q) in scalar registers.  An iteration without a break costs two ballots; one with breaks is
//                settled by all lanes at once (run_iteration_par: segments inside a lane's word by the lane, the segment a
//                lane's first break closes through two prefix maxima over the wave).  Leaves the tile's summary: its first break, the first A /
//                last B before it, the matches closed by its other breaks (they depend on nothing outside the tile), the
//                segment open at its end.
//   run_resolve  the summaries composed (tiles without a break hand the pending thread on, tiles with one replace it:
//                associative), every tile's incoming state and the number of its first output pair: ONE workgroup up to
//                32 MiB of text, beyond that two levels over blocks of 1024 tiles (run_reduce, run_resolve_blocks, run_apply).
//   run_emit     run_summary's walk again with the incoming state known, over the tiles that close a match: the pairs at
//                their final place.
// Cost: the streams (~36 VALU per range and 32 bytes) once or twice; FETCH_SIZE 1-2 x the text; 1.1 TB/s (a break every 33
// bytes) to 3.9 TB/s (none) for a whole call.  dense_streams (one pass) is tried first for the patterns it takes.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdlib.h>

#include "kernels.h"
#include "run_scan.h"

namespace rejit_amd {

namespace {

constexpr int kWave = 64;
constexpr uint32_t kIterBytes = 2048;
constexpr unsigned long long kNone = ~0ull, kBlocked = ~0ull - 1;

__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x) & (kWave - 1); }

// which class reads which range, as AND masks in scalar registers (0 / ~0); NR = the compile-time bound of the loop over the
// plan's ranges (ranges beyond n_ranges are computed and dropped: at most NR / 2 - 1 of them)
template <int NR>
struct RunMasks {
  uint32_t a[NR], l[NR], b[NR];
};
template <int NR>
__device__ __forceinline__ RunMasks<NR> run_masks(const RunPlan& pl) {
  RunMasks<NR> m;
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const bool in = static_cast<uint32_t>(r) < pl.n_ranges;
    m.a[r] = in ? 0u - ((pl.a_ranges >> r) & 1u) : 0u;
    m.l[r] = in ? 0u - ((pl.l_ranges >> r) & 1u) : 0u;
    m.b[r] = in ? 0u - ((pl.b_ranges >> r) & 1u) : 0u;
  }
  return m;
}

// the three streams of the lane's 32 bytes at `at`: A, B, breaks (a byte outside L; the text's end is one)
// SNL (RunPlan::bol: `^` in front of the shape): the stream of the line breaks \n / \r
template <int NR, bool HAS_B>
__device__ __forceinline__ void run_streams_of(const RunParams& a, const RunMasks<NR>& mk, uint64_t at, const uint4& v0, const uint4& v1, bool loaded,
                                               uint32_t* SA, uint32_t* SB, uint32_t* BR, uint32_t* SNL = nullptr) {
  uint32_t x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
  uint32_t valid = ~0u;
  if (!loaded) {   // (an iteration that touches the end of the text: byte by byte)
    valid = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) x[q] = 0;
#pragma unroll
    for (int q = 0; q < 8; q++)   // (constant indices: a run-time index would put x[] in scratch memory)
#pragma unroll
      for (int b = 0; b < 4; b++)
        if (at + static_cast<uint64_t>(4 * q + b) < a.n) {
          x[q] |= static_cast<uint32_t>(a.text[at + static_cast<uint64_t>(4 * q + b)]) << (8 * b);
          valid |= 1u << (4 * q + b);
        }
  }
  uint32_t x7[8], lowh[8], highh[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    x7[i] = x[i] & 0x7f7f7f7fu;
    lowh[i] = ~x[i] & 0x80808080u;
    highh[i] = x[i] & 0x80808080u;
  }
  const RunPlan& pl = a.plan;
  uint32_t sa = 0, sl = 0, sb = 0;
#pragma unroll
  for (int r = 0; r < NR; r++) {
    uint32_t T;
    if ((pl.high_half >> r) & 1u) T = rj_stream_range(x7, highh, pl.add_lo[r], pl.add_hi[r]);   // (wave-uniform)
    else T = rj_stream_range(x7, lowh, pl.add_lo[r], pl.add_hi[r]);
    sa |= T & mk.a[r];
    sl |= T & mk.l[r];
    if (HAS_B) sb |= T & mk.b[r];
  }
  sa = (pl.a_neg ? ~sa : sa) & valid;
  sl = (pl.l_neg ? ~sl : sl) & valid;
  sb = HAS_B ? ((pl.b_neg ? ~sb : sb) & valid) : 0u;
  uint32_t br = ~sl & valid;
  if (a.n >= at && a.n - at < 32) br |= 1u << static_cast<uint32_t>(a.n - at);   // the end of the text closes the last segment
  *SA = sa;
  *SB = sb;
  *BR = br;
  if (SNL) {   // (0x0a / 0x0d as ranges of one byte: add constants as run_detail::add_class makes them)
    // (RunPlan::eol reads it at the closing breaks: the text's end closes a line too)
    uint32_t snl = (a.plan.bol | a.plan.eol) ? (rj_stream_range(x7, lowh, 0x76767676u, 0x75757575u) | rj_stream_range(x7, lowh, 0x73737373u, 0x72727272u)) & valid : 0u;
    if (a.plan.eol && a.n >= at && a.n - at < 32) snl |= 1u << static_cast<uint32_t>(a.n - at);
    *SNL = snl;
  }
}

// bits of the lane's word whose index lane * 32 + bit lies in [lo, hi)
__device__ __forceinline__ uint32_t clip(uint32_t m, int lane, uint32_t lo, uint32_t hi) {
  const uint32_t base = static_cast<uint32_t>(lane) * 32u;
  const uint32_t l = lo > base ? lo - base : 0u, h = hi > base ? hi - base : 0u;
  const uint32_t below_h = h >= 32u ? ~0u : ((1u << h) - 1u), below_l = l >= 32u ? ~0u : ((1u << l) - 1u);
  return m & below_h & ~below_l;
}
// SA with the starts below min_start taken out
__device__ __forceinline__ uint32_t clip_starts(const RunParams& a, uint64_t it_base, uint32_t SA) {
  if (a.min_start <= it_base) return SA;   // (wave-uniform: all but the run's first iteration)
  const uint32_t a_lo = a.min_start - it_base < kIterBytes ? static_cast<uint32_t>(a.min_start - it_base) : kIterBytes;
  return clip(SA, lane_id(), a_lo, kIterBytes);
}

struct Open {
  unsigned long long s, q;   // the open segment's first A (kNone / kBlocked) and last B behind it (kNone)
};

__device__ __forceinline__ bool real(unsigned long long s) { return s < kBlocked; }
// the segment closed in state o holds a match that counts
template <bool HAS_B>
__device__ __forceinline__ bool counted(const RunParams& a, const Open& o) {
  return real(o.s) && (!HAS_B || o.q != kNone) && o.s >= a.sb && o.s < a.se;
}

// An iteration WITHOUT a break (the usual one in a text of long runs): the open segment takes the iteration's first A if it
// has no start yet, and the iteration's last B if that lies behind its start -- two ballots, the rest is scalar.
// SA: clipped.  any_b (may be null): the last B met so far.
template <bool HAS_B>
__device__ __forceinline__ void run_iteration_quiet(uint64_t it_base, uint32_t SA, uint32_t SB, Open& st, unsigned long long* any_b) {
  unsigned long long last_b = kNone;
  if (HAS_B) {
    const uint64_t bm = __ballot(SB != 0);
    if (bm != 0) {
      const int l = 63 - __builtin_clzll(bm);
      const uint32_t w = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(SB), l));
      last_b = it_base + static_cast<uint32_t>(l) * 32u + 31u - static_cast<uint32_t>(__builtin_clz(w));
      if (any_b) *any_b = last_b;
    }
  }
  if (st.s == kNone) {
    const uint64_t am = __ballot(SA != 0);
    if (am != 0) {
      const int l = __builtin_ctzll(am);
      const uint32_t w = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(SA), l));
      st.s = it_base + static_cast<uint32_t>(l) * 32u + static_cast<uint32_t>(__builtin_ctz(w));
    }
  }
  if (HAS_B && last_b != kNone && real(st.s) && last_b > st.s) st.q = last_b;
}

// ---- An iteration WITH breaks, settled by all lanes at once.  (The first form walked the breaks one after the other, the wave
// as ONE sequential machine, ~100 instructions a trip: a text with a break every 33 bytes ran at 236 GB/s.)  Every lane
// settles what lies between the breaks of its own 32 bytes by itself, and what crosses lanes comes from two prefix maxima
// over the wave:
//   * the segment closed by a lane's FIRST break began in a lane below (or before the iteration): its first A is the
//     smallest "A at or behind the last break" of the lanes since the last lane with a break, its last B the largest "B
//     behind the last break" of those lanes -- keys (breaks so far << 12 | position) make a plain prefix maximum of both;
//   * the segments between two breaks of one lane (for_inner) need nothing from outside.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t x) {
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), CTRL, ROW_MASK, 0xF, true));
}
__device__ __forceinline__ uint32_t umax(uint32_t x, uint32_t y) { return x > y ? x : y; }
__device__ __forceinline__ uint32_t wave_prefix_max(uint32_t x) {   // (inclusive; keys are >= 0: the zero a DPP move leaves is neutral)
  x = umax(x, dpp0<0x111, 0xF>(x));  // row_shr:1
  x = umax(x, dpp0<0x112, 0xF>(x));  // row_shr:2
  x = umax(x, dpp0<0x114, 0xF>(x));  // row_shr:4
  x = umax(x, dpp0<0x118, 0xF>(x));  // row_shr:8
  x = umax(x, dpp0<0x142, 0xA>(x));  // row_bcast:15 into rows 1 and 3
  x = umax(x, dpp0<0x143, 0xC>(x));  // row_bcast:31 into rows 2 and 3
  return x;
}
__device__ __forceinline__ uint32_t wave_prefix_sum(uint32_t x) {
  x += dpp0<0x111, 0xF>(x);
  x += dpp0<0x112, 0xF>(x);
  x += dpp0<0x114, 0xF>(x);
  x += dpp0<0x118, 0xF>(x);
  x += dpp0<0x142, 0xA>(x);
  x += dpp0<0x143, 0xC>(x);
  return x;
}
__device__ __forceinline__ uint32_t lane_below(uint32_t x) { return dpp0<0x138, 0xF>(x); }   // wave_shr:1, lane 0 gets 0
__device__ __forceinline__ uint32_t last_lane(uint32_t x) { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(x), kWave - 1)); }
// RunPlan::lag (`A L+`, `A L+ B`): the start stream becomes the marks "A at p - 1 and L at p"; prev_a = the A bit of the byte before
// the iteration (in: of this one, out: of the next)
__device__ __forceinline__ uint32_t lag_marks(uint32_t SA, uint32_t BR, uint32_t& prev_a) {
  const uint32_t below = lane_below(SA) >> 31;
  const uint32_t cin = lane_id() == 0 ? prev_a : below;
  prev_a = last_lane(SA) >> 31;
  return ((SA << 1) | cin) & ~BR;   // (BR holds the text's end: no mark there or beyond)
}
// RunPlan::bol (`^` in front): "the byte before p is a line break, or p is the text's begin"; prev_nl as prev_a
__device__ __forceinline__ uint32_t bol_stream(uint32_t SNL, uint32_t& prev_nl) {
  const uint32_t below = lane_below(SNL) >> 31;
  const uint32_t cin = lane_id() == 0 ? prev_nl : below;
  prev_nl = last_lane(SNL) >> 31;
  return (SNL << 1) | cin;
}
// ... and at a tile's begin: the two carries from the two bytes before the tile (every lane reads the same 32 bytes, once per tile;
// tiles begin at multiples of 2 KiB: base >= 2 unless it is 0)
template <int NR, bool HAS_B>
__device__ __forceinline__ void run_entry(const RunParams& a, const RunMasks<NR>& mk, uint64_t base, uint32_t& prev_a, uint32_t& prev_nl) {
  prev_a = 0u;
  prev_nl = base == 0 ? 1u : 0u;   // (the text's begin is a line start)
  if ((!a.plan.lag && !a.plan.bol) || base == 0) return;
  uint32_t sa, sb, br, snl;
  const uint4 none = make_uint4(0, 0, 0, 0);
  run_streams_of<NR, HAS_B>(a, mk, base - 2, none, none, false, &sa, &sb, &br, &snl);
  sa = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(sa)));
  snl = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(snl)));
  uint32_t a1 = (sa >> 1) & 1u;                 // A at base - 1 ...
  if (a.plan.bol) a1 &= snl & 1u;               // ... a start only behind a line break (base - 2)
  prev_a = a.plan.lag ? a1 : 0u;
  prev_nl = a.plan.bol ? (snl >> 1) & 1u : 0u;
}
__device__ __forceinline__ unsigned long long lane_value(unsigned long long x, int l) {
  const uint32_t lo = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(x)), l));
  const uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(x >> 32)), l));
  return (static_cast<unsigned long long>(hi) << 32) | lo;
}
__device__ __forceinline__ uint32_t bits_below(uint32_t b) { return (1u << b) - 1u; }        // b in 0 .. 31
__device__ __forceinline__ uint32_t bits_upto(uint32_t b) { return (2u << b) - 1u; }         // bits 0 .. b, b in 0 .. 31

struct LaneClose {
  unsigned long long s, q;   // the state the lane's first break closes (lanes with a break)
  unsigned long long b_any;  // the last B of the iteration at or before that break (kNone: none)
};

// brm = ballot(BR != 0) != 0.  SA: clipped.  Leaves what every lane's first break closes and the state behind the iteration's
// last break in st.
template <bool HAS_B>
__device__ __forceinline__ LaneClose run_iteration_par(uint64_t it_base, uint32_t SA, uint32_t SB, uint32_t BR, uint64_t brm, Open& st) {
  const int lane = lane_id();
  const uint32_t pbase = static_cast<uint32_t>(lane) * 32u;
  const bool hb = BR != 0;
  const uint32_t fb = hb ? static_cast<uint32_t>(__builtin_ctz(BR)) : 0u, lb = hb ? 31u - static_cast<uint32_t>(__builtin_clz(BR)) : 0u;
  const uint32_t headA = hb ? SA & bits_below(fb) : SA;      // starts before the first break
  const uint32_t headB = hb ? SB & bits_upto(fb) : SB;       // (B may be the break itself)
  const uint32_t tailA = hb ? SA & ~bits_below(lb) : SA;     // (a start may sit ON the break that opens its segment)
  const uint32_t tailB = hb ? SB & ~bits_upto(lb) : SB;
  const uint32_t seg_excl = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(brm >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(brm), 0u));
  const uint32_t seg_incl = seg_excl + (hb ? 1u : 0u);
  const uint32_t keyA = (seg_incl << 12) | (tailA ? 0xfffu - (pbase + static_cast<uint32_t>(__builtin_ctz(tailA))) : 0u);
  const uint32_t inclA = wave_prefix_max(keyA);
  const uint32_t exclA = lane_below(inclA) & 0xfffu;
  uint32_t inclB = 0, exclB = 0;
  if (HAS_B) {
    const uint32_t keyB = (seg_incl << 12) | (tailB ? pbase + 32u - static_cast<uint32_t>(__builtin_clz(tailB)) : 0u);   // position + 1
    inclB = wave_prefix_max(keyB);
    exclB = lane_below(inclB) & 0xfffu;
  }
  LaneClose c;
  c.s = c.q = c.b_any = kNone;
  const bool carried = seg_excl == 0 && st.s != kNone;   // the segment was open, with its start, when the iteration began
  bool s_here = false;
  if (carried) {
    c.s = st.s;
  } else if (exclA != 0) {
    c.s = it_base + (0xfffu - exclA);
  } else if (headA != 0) {
    c.s = it_base + pbase + static_cast<uint32_t>(__builtin_ctz(headA));
    s_here = true;
  }
  if (HAS_B) {
    if (headB != 0) c.b_any = it_base + pbase + 31u - static_cast<uint32_t>(__builtin_clz(headB));
    else if (exclB != 0 && seg_excl == 0) c.b_any = it_base + (exclB - 1u);
    if (real(c.s)) {
      const uint32_t hbm = s_here ? headB & ~bits_upto(static_cast<uint32_t>(__builtin_ctz(headA))) : headB;
      if (hbm != 0) {
        c.q = it_base + pbase + 31u - static_cast<uint32_t>(__builtin_clz(hbm));
      } else if (!s_here) {
        if (exclB != 0 && (carried || it_base + (exclB - 1u) > c.s)) c.q = it_base + (exclB - 1u);
        else if (carried) c.q = st.q;
      }
    }
  }
  // behind the iteration's last break
  const uint32_t endA = last_lane(inclA) & 0xfffu, endB = last_lane(inclB) & 0xfffu;
  st.s = endA != 0 ? it_base + (0xfffu - endA) : kNone;
  st.q = (endA != 0 && endB != 0 && it_base + (endB - 1u) > st.s) ? it_base + (endB - 1u) : kNone;
  return c;
}

// the segments between two breaks of the lane's own word, in text order: f(s, q, r) -- positions relative to the word; q = 32: no B
template <bool HAS_B, class F>
__device__ __forceinline__ void for_inner(uint32_t SA, uint32_t SB, uint32_t BR, F f) {
  if (BR == 0) return;
  uint32_t prev = static_cast<uint32_t>(__builtin_ctz(BR));
  uint32_t rest = BR & (BR - 1u);
  while (rest != 0) {
    const uint32_t r = static_cast<uint32_t>(__builtin_ctz(rest));
    rest &= rest - 1u;
    const uint32_t am = SA & bits_below(r) & ~bits_below(prev);
    if (am != 0) {
      const uint32_t s = static_cast<uint32_t>(__builtin_ctz(am));
      const uint32_t bm = SB & bits_upto(r) & ~bits_upto(s);
      if (!HAS_B) f(s, 32u, r);
      else if (bm != 0) f(s, 31u - static_cast<uint32_t>(__builtin_clz(bm)), r);
    }
    prev = r;
  }
}

}  // namespace

template <int NR, bool HAS_B>
__global__ __launch_bounds__(256) void run_summary(RunParams a) {
  const int lane = lane_id();
  const uint64_t tile = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  if (tile >= a.n_tiles) return;
  const RunMasks<NR> mk = run_masks<NR>(a.plan);
  const uint64_t base = (a.first_tile + tile) * a.tile_bytes;
  RunSummary sum;
  sum.r1 = kNone;
  sum.a1 = sum.b1 = sum.b1a = kNone;
  sum.open_s = sum.open_q = kNone;
  sum.cnt = 0;
  sum.pad = 0;
  Open st{kNone, kNone};
  unsigned long long any_b = kNone;
  bool before_first = true;
  unsigned long long cnt = 0;
  const int kIters = static_cast<int>(a.tile_bytes / kIterBytes);
  uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
  bool loaded = false;
  auto fetch = [&](int it) {
    const uint64_t at = base + static_cast<uint64_t>(it) * kIterBytes + static_cast<uint64_t>(lane) * 32;
    loaded = base + static_cast<uint64_t>(it + 1) * kIterBytes <= a.n;   // (wave-uniform)
    if (loaded) {
      v0 = *reinterpret_cast<const uint4*>(a.text + at);
      v1 = *reinterpret_cast<const uint4*>(a.text + at + 16);
    }
  };
  fetch(0);
  uint32_t prev_a, prev_nl;
  run_entry<NR, HAS_B>(a, mk, base, prev_a, prev_nl);
#pragma unroll 1
  for (int it = 0; it < kIters; it++) {
    const uint64_t it_base = base + static_cast<uint64_t>(it) * kIterBytes;
    if (it_base > a.n) break;   // (nothing here, not even the text's end)
    uint32_t SA, SB, BR, SNL;
    run_streams_of<NR, HAS_B>(a, mk, it_base + static_cast<uint64_t>(lane) * 32, v0, v1, loaded, &SA, &SB, &BR, &SNL);
    if (it + 1 < kIters) fetch(it + 1);
    if (a.plan.bol) SA &= bol_stream(SNL, prev_nl);   // `^`: an A counts only at a line start (before the shift of `A L+`)
    if (a.plan.lag) SA = lag_marks(SA, BR, prev_a);
    SA = clip_starts(a, it_base, SA);
    const uint64_t brm = __ballot(BR != 0);
    if (brm == 0) {
      run_iteration_quiet<HAS_B>(it_base, SA, SB, st, before_first ? &any_b : nullptr);
      continue;
    }
    const LaneClose c = run_iteration_par<HAS_B>(it_base, SA, SB, BR, brm, st);
    // RunPlan::eol (`$` behind a shape without B): a match ends at its segment's closing break, and counts when that break is a line end
    const bool eol = a.plan.eol != 0;
    const bool first_at_eol = !eol || (BR != 0 && ((SNL >> static_cast<uint32_t>(__builtin_ctz(BR | 0x80000000u))) & 1u) != 0);
    bool first = BR != 0 && first_at_eol && counted<HAS_B>(a, Open{c.s, c.q});
    if (before_first) {   // the tile's first break: what it closes depends on the tiles before
      const int l1 = __builtin_ctzll(brm);
      sum.pad = __builtin_amdgcn_readlane(static_cast<int>(first_at_eol ? 0 : 1), l1) != 0 ? 1ull : 0ull;   // (1: whatever it closes does not count)
      sum.r1 = it_base + static_cast<uint64_t>(l1) * 32u + static_cast<uint32_t>(__builtin_ctz(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(BR), l1))));
      sum.a1 = lane_value(c.s, l1);
      sum.b1a = lane_value(c.q, l1);
      const unsigned long long b = lane_value(c.b_any, l1);
      sum.b1 = b != kNone ? b : any_b;
      before_first = false;
      if (lane == l1) first = false;
    }
    cnt += static_cast<unsigned long long>(__popcll(__ballot(first)));
    if (__ballot((BR & (BR - 1u)) != 0) != 0) {   // (some lane holds two breaks)
      uint32_t inner = 0;
      for_inner<HAS_B>(SA, SB, BR, [&](uint32_t s, uint32_t, uint32_t r) {
        const uint64_t at = it_base + static_cast<uint64_t>(lane) * 32u + s;
        if (at >= a.sb && at < a.se && (!eol || ((SNL >> r) & 1u) != 0)) inner++;
      });
      cnt += last_lane(wave_prefix_sum(inner));
    }
  }
  if (before_first) {   // no break in the tile
    sum.a1 = st.s;
    sum.b1a = st.q;
    sum.b1 = any_b;
  } else {
    sum.open_s = st.s;
    sum.open_q = st.q;
  }
  sum.cnt = cnt;
  if (lane == 0) a.summaries[tile] = sum;
}

namespace {

// (the resolve kernels are not templated: the plan says whether a B exists)
__device__ __forceinline__ bool counted(const RunParams& a, const Open& o) { return a.plan.has_b ? counted<true>(a, o) : counted<false>(a, o); }

// a tile (or a run of tiles) as a function on the open segment's state.  cnt: the matches closed by its breaks other than the
// first (what the first closes depends on the state that comes in)
struct Elem {
  unsigned long long hb;   // holds a break
  unsigned long long a1, b1, b1a, open_s, open_q, cnt;
};
// hb: 0 = no break; 1 = the first break closes what came in; 2 (RunPlan::eol) = it is no line end: whatever it closes does not count
__device__ __forceinline__ Elem elem_of(const RunSummary& s) { return Elem{s.r1 != kNone ? (s.pad ? 2ull : 1ull) : 0ull, s.a1, s.b1, s.b1a, s.open_s, s.open_q, s.cnt}; }
__device__ __forceinline__ RunSummary summary_of(const Elem& e) {   // (a block of tiles in the tiles' own format: r1 is a flag only)
  RunSummary s;
  s.r1 = e.hb ? 0ull : kNone;
  s.a1 = e.a1;
  s.b1 = e.b1;
  s.b1a = e.b1a;
  s.open_s = e.open_s;
  s.open_q = e.open_q;
  s.cnt = e.cnt;
  s.pad = e.hb == 2 ? 1ull : 0ull;
  return s;
}
// the state behind a stretch without a break
__device__ __forceinline__ Open pass_on(const Elem& g, Open in) {
  Open o;
  if (in.s == kNone) {
    o.s = g.a1;
    o.q = g.a1 != kNone ? g.b1a : kNone;
  } else {
    o.s = in.s;
    o.q = (real(in.s) && g.b1 != kNone) ? g.b1 : in.q;
  }
  return o;
}
// f, then g
__device__ __forceinline__ Elem compose(const RunParams& a, const Elem& f, const Elem& g) {
  Elem e;
  if (!f.hb) {   // the stretch before the composite's first break: f's whole, then g's beginning
    e.a1 = f.a1 != kNone ? f.a1 : g.a1;
    e.b1 = g.b1 != kNone ? g.b1 : f.b1;
    e.b1a = f.a1 != kNone ? (g.b1 != kNone ? g.b1 : f.b1a) : g.b1a;
  } else {
    e.a1 = f.a1;
    e.b1 = f.b1;
    e.b1a = f.b1a;
  }
  e.hb = f.hb ? f.hb : g.hb;   // (the composite's first break)
  e.cnt = f.cnt + g.cnt;
  if (g.hb) {
    if (f.hb && g.hb == 1 && counted(a, pass_on(g, Open{f.open_s, f.open_q}))) e.cnt++;   // g's first break is not the composite's first
    e.open_s = g.open_s;
    e.open_q = g.open_q;
  } else if (f.hb) {
    const Open o = pass_on(g, Open{f.open_s, f.open_q});
    e.open_s = o.s;
    e.open_q = o.q;
  } else {
    e.open_s = e.open_q = kNone;
  }
  return e;
}
// the state behind an element; *emits: its first break closes a match that counts
__device__ __forceinline__ Open apply(const RunParams& a, const Elem& g, Open in, bool* emits) {
  *emits = false;
  if (!g.hb) return pass_on(g, in);
  *emits = g.hb == 1 && counted(a, pass_on(g, in));
  return Open{g.open_s, g.open_q};
}

constexpr uint64_t kBlockTiles = 1024;     // tiles per workgroup of the two-level resolve (RunParams::block_tiles)
constexpr uint64_t kOneLevelTiles = 4096;  // up to here ONE workgroup resolves everything

// T threads over the n elements src[0 .. n): thread t owns [t C, (t + 1) C).  Its elements composed into one; an inclusive scan
// of the T composites in LDS (log2 T rounds: the composition is associative -- the first version let thread 0 walk them one
// after the other: 197 us of the 280 us a 64 MiB text took).  Then either the whole span's composite goes to *whole (the
// reduce step of the two-level resolve), or: the composite BEFORE a thread's chunk applied to the span's incoming state
// (initial, off0) is the chunk's incoming state and first output pair; the chunk walked again with it leaves every element's
// incoming state and offset in dst; *total (thread T - 1) = off0 + the span's matches.
// (The loops over a thread's elements load four at a time: one thread's loads, issued one after the other and each waited
// for, were most of this kernel.)
template <int T>
__device__ __forceinline__ void resolve_span(const RunParams& a, const RunSummary* src, RunTileIn* dst, uint64_t n, Open initial, unsigned long long off0,
                                             Elem (*chunk)[T], RunSummary* whole, unsigned long long* total) {
  const uint32_t t = threadIdx.x;
  const uint64_t C = (n + T - 1) / T;
  const uint64_t lo = static_cast<uint64_t>(t) * C < n ? static_cast<uint64_t>(t) * C : n, hi = lo + C < n ? lo + C : n;
  const Elem identity{0, kNone, kNone, kNone, kNone, kNone, 0};   // (a stretch without a break, an A or a B hands every state on)
  constexpr int kBatch = 4;
  Elem e = identity;
  for (uint64_t i = lo; i < hi; i += kBatch) {
    RunSummary sm[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; k++) sm[k] = src[i + k < hi ? i + k : i];
#pragma unroll
    for (int k = 0; k < kBatch; k++)
      if (i + k < hi) e = (i + k == lo) ? elem_of(sm[k]) : compose(a, e, elem_of(sm[k]));
  }
  int cur = 0;
  chunk[0][t] = e;
  __syncthreads();
  for (uint32_t d = 1; d < T; d <<= 1) {
    Elem v = chunk[cur][t];
    if (t >= d) v = compose(a, chunk[cur][t - d], v);
    chunk[cur ^ 1][t] = v;
    cur ^= 1;
    __syncthreads();
  }
  if (whole) {
    if (t == T - 1) *whole = summary_of(chunk[cur][T - 1]);
    return;
  }
  bool emits = false;
  Open st = initial;
  unsigned long long off = off0;
  if (t != 0) {
    const Elem before = chunk[cur][t - 1];
    st = apply(a, before, initial, &emits);
    off += before.cnt + (emits ? 1ull : 0ull);
  }
  for (uint64_t i = lo; i < hi; i += kBatch) {
    RunSummary sm[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; k++) sm[k] = src[i + k < hi ? i + k : i];
#pragma unroll
    for (int k = 0; k < kBatch; k++) {
      if (i + k >= hi) break;
      const Open next = apply(a, elem_of(sm[k]), st, &emits);
      RunTileIn ti;
      ti.s = st.s;
      ti.q = st.q;
      ti.off = off;
      ti.cnt = sm[k].cnt + (emits ? 1ull : 0ull);
      dst[i + k] = ti;
      off += ti.cnt;
      st = next;
    }
  }
  if (t == T - 1) *total = off;
}

__device__ __forceinline__ void leave_total(const RunParams& a, unsigned long long run) {
  a.counters[kCntFinal] = run;
  a.counters[kCntCands] = run;
  a.counters[kCntHits] = run;
  if (a.host_counters) a.host_counters[kCntFinal] = run;
}

}  // namespace

// Texts of up to kOneLevelTiles tiles (32 MiB): ONE workgroup composes the tiles' summaries -- every tile's incoming state and
// the number of its first output pair.
__global__ __launch_bounds__(1024) void run_resolve(RunParams a) {
  __shared__ Elem chunk[2][1024];
  unsigned long long total = 0;
  resolve_span<1024>(a, a.summaries, a.tile_in, a.n_tiles, Open{a.blocked_in ? kBlocked : kNone, kNone}, 0, chunk, nullptr, &total);
  if (threadIdx.x == 1023) leave_total(a, total);
}
// Longer texts, two levels (one workgroup alone took half of a 4 GiB call: 512 tiles per thread, walked twice): a workgroup
// per block of kBlockTiles tiles composes its block into ONE element behind the tiles' summaries (run_reduce); one workgroup
// resolves the blocks (run_resolve_blocks: their incoming states and offsets behind the tiles' own); a workgroup per block
// resolves its tiles from there (run_apply).
__global__ __launch_bounds__(256) void run_reduce(RunParams a) {
  __shared__ Elem chunk[2][256];
  const uint64_t b = blockIdx.x, first = b * a.block_tiles;
  const uint64_t n = a.n_tiles - first < a.block_tiles ? a.n_tiles - first : a.block_tiles;
  resolve_span<256>(a, a.summaries + first, nullptr, n, Open{kNone, kNone}, 0, chunk, a.summaries + a.n_tiles + b, nullptr);
}
__global__ __launch_bounds__(1024) void run_resolve_blocks(RunParams a) {
  __shared__ Elem chunk[2][1024];
  const uint64_t n_blocks = (a.n_tiles + a.block_tiles - 1) / a.block_tiles;
  unsigned long long total = 0;
  resolve_span<1024>(a, a.summaries + a.n_tiles, a.tile_in + a.n_tiles, n_blocks, Open{a.blocked_in ? kBlocked : kNone, kNone}, 0, chunk, nullptr, &total);
  if (threadIdx.x == 1023) leave_total(a, total);
}
__global__ __launch_bounds__(256) void run_apply(RunParams a) {
  __shared__ Elem chunk[2][256];
  const uint64_t b = blockIdx.x, first = b * a.block_tiles;
  const uint64_t n = a.n_tiles - first < a.block_tiles ? a.n_tiles - first : a.block_tiles;
  const RunTileIn in = a.tile_in[a.n_tiles + b];
  unsigned long long total = 0;
  resolve_span<256>(a, a.summaries + first, a.tile_in + first, n, Open{in.s, in.q}, in.off, chunk, nullptr, &total);
}

template <int NR, bool HAS_B>
__global__ __launch_bounds__(256) void run_emit(RunParams a) {
  const int lane = lane_id();
  const uint64_t tile = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  if (tile >= a.n_tiles) return;
  const RunTileIn in = a.tile_in[tile];
  if (in.cnt == 0) return;   // (no pair ends here: a text of long runs is read ONCE)
  const RunMasks<NR> mk = run_masks<NR>(a.plan);
  const uint64_t base = (a.first_tile + tile) * a.tile_bytes;
  Open st{in.s, in.q};
  unsigned long long pos = in.off;
  const int kIters = static_cast<int>(a.tile_bytes / kIterBytes);
  uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
  bool loaded = false;
  auto fetch = [&](int it) {
    const uint64_t at = base + static_cast<uint64_t>(it) * kIterBytes + static_cast<uint64_t>(lane) * 32;
    loaded = base + static_cast<uint64_t>(it + 1) * kIterBytes <= a.n;
    if (loaded) {
      v0 = *reinterpret_cast<const uint4*>(a.text + at);
      v1 = *reinterpret_cast<const uint4*>(a.text + at + 16);
    }
  };
  fetch(0);
  uint32_t prev_a, prev_nl;
  run_entry<NR, HAS_B>(a, mk, base, prev_a, prev_nl);
#pragma unroll 1
  for (int it = 0; it < kIters; it++) {
    const uint64_t it_base = base + static_cast<uint64_t>(it) * kIterBytes;
    if (it_base > a.n) break;
    uint32_t SA, SB, BR, SNL;
    run_streams_of<NR, HAS_B>(a, mk, it_base + static_cast<uint64_t>(lane) * 32, v0, v1, loaded, &SA, &SB, &BR, &SNL);
    if (it + 1 < kIters) fetch(it + 1);
    if (a.plan.bol) SA &= bol_stream(SNL, prev_nl);   // `^`: an A counts only at a line start (before the shift of `A L+`)
    if (a.plan.lag) SA = lag_marks(SA, BR, prev_a);
    SA = clip_starts(a, it_base, SA);
    const uint64_t brm = __ballot(BR != 0);
    if (brm == 0) {
      run_iteration_quiet<HAS_B>(it_base, SA, SB, st, nullptr);
      continue;
    }
    const LaneClose c = run_iteration_par<HAS_B>(it_base, SA, SB, BR, brm, st);
    const uint64_t word = it_base + static_cast<uint64_t>(lane) * 32u;
    const bool eol = a.plan.eol != 0;   // (`$`: the closing break must be a line end -- as run_summary counted)
    const bool first = BR != 0 && (!eol || ((SNL >> static_cast<uint32_t>(__builtin_ctz(BR | 0x80000000u))) & 1u) != 0) && counted<HAS_B>(a, Open{c.s, c.q});
    const bool multi = __ballot((BR & (BR - 1u)) != 0) != 0;
    uint32_t mine = first ? 1u : 0u;
    if (multi)
      for_inner<HAS_B>(SA, SB, BR, [&](uint32_t s, uint32_t, uint32_t r) {
        if (word + s >= a.sb && word + s < a.se && (!eol || ((SNL >> r) & 1u) != 0)) mine++;
      });
    if (__ballot(mine != 0) == 0) continue;
    const uint32_t inc = wave_prefix_sum(mine);
    unsigned long long idx = pos + inc - mine;
    if (first) {
      if (idx < a.out_cap) *reinterpret_cast<ulonglong2*>(a.out + 2 * idx) = make_ulonglong2(c.s - a.plan.lag, HAS_B ? c.q + 1 : word + static_cast<uint32_t>(__builtin_ctz(BR)));
      idx++;
    }
    if (multi)
      for_inner<HAS_B>(SA, SB, BR, [&](uint32_t s, uint32_t q, uint32_t r) {
        if (word + s >= a.sb && word + s < a.se && (!eol || ((SNL >> r) & 1u) != 0)) {
          if (idx < a.out_cap) *reinterpret_cast<ulonglong2*>(a.out + 2 * idx) = make_ulonglong2(word + s - a.plan.lag, word + (HAS_B ? q + 1u : r));
          idx++;
        }
      });
    pos += last_lane(inc);
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// The PAIR shape (run_scan.h: `Q L* Q`, the same class at both ends, no Q inside L -- `"[^"]*"`, `'[^'\n]*'`): the matches are
// the pairs (1st, 2nd), (3rd, 4th) ... of the Q bytes since the last RESET (a break that is no Q; the text's end).  Reference:
// the thread a Q opens lives until the next break (src/x64/codegen-x64.cc:535-581); when that break is a Q the match ends behind
// it (:426-461) and the scan restarts behind the match (:487-503) -- so that Q opens nothing --, when it is not the thread dies.
//   pair_summary  a wave per tile: the Q stream and the reset stream; the tile as a FUNCTION on one bit (is a Q open when the
//                 tile begins?): the matches it closes, whether a Q is open behind it and where that Q sits.  Everything
//                 behind the tile's first reset does not depend on the bit; before it, c + k1 Q bytes make (c + k1) / 2 pairs.
//   pair_resolve  the functions composed (associative): every tile's incoming bit, open position and first output pair.
//   pair_emit     the tiles that close a match once more, the incoming state known: a Q is a CLOSER when an odd number of Q
//                 bytes lies between the last reset (or the tile's begin, the incoming bit counted) and itself -- a prefix
//                 parity inside the lane's word, the lane's own incoming parity from two ballots --; its opener is the Q before it.
// Whole texts only (own ranges and carried-in states keep the other paths).
namespace {

struct alignas(64) PairElem {    // 64 bytes: the slot of a RunSummary.  (Scalars, no arrays: an array member put the element in scratch memory
                                 // -- the first version's resolve took 120 us per GiB of text)
  unsigned long long m0, m1;     // matches closed, by the incoming bit
  unsigned long long pos0, pos1; // where the open Q sits behind the element (kind bit set)
  uint32_t oo;                   // bit c: a Q is open behind the element
  uint32_t kind;                 // bit c: 1 = that Q lies inside the element (pos), 0 = it is the one that came in
};
static_assert(sizeof(PairElem) == sizeof(RunSummary), "the pair kernels use the run kernels' buffers");

__device__ __forceinline__ PairElem pair_compose(const PairElem& f, const PairElem& g) {   // f, then g
  PairElem e;
  const bool a0 = (f.oo & 1u) != 0, a1 = (f.oo & 2u) != 0;            // the bit g sees, by the bit f saw
  const bool in0 = ((a0 ? g.kind >> 1 : g.kind) & 1u) != 0, in1 = ((a1 ? g.kind >> 1 : g.kind) & 1u) != 0;
  e.m0 = f.m0 + (a0 ? g.m1 : g.m0);
  e.m1 = f.m1 + (a1 ? g.m1 : g.m0);
  e.pos0 = in0 ? (a0 ? g.pos1 : g.pos0) : f.pos0;
  e.pos1 = in1 ? (a1 ? g.pos1 : g.pos0) : f.pos1;
  e.oo = ((a0 ? g.oo >> 1 : g.oo) & 1u) | (((a1 ? g.oo >> 1 : g.oo) & 1u) << 1);
  e.kind = (in0 ? 1u : (f.kind & 1u)) | ((in1 ? 1u : ((f.kind >> 1) & 1u)) << 1);
  return e;
}

struct PairState {
  uint32_t open;
  unsigned long long at;
};
__device__ __forceinline__ PairState pair_apply(const PairElem& g, PairState in, unsigned long long* closes) {
  const bool c = in.open != 0;
  *closes = c ? g.m1 : g.m0;
  PairState o;
  o.open = (c ? g.oo >> 1 : g.oo) & 1u;
  o.at = ((c ? g.kind >> 1 : g.kind) & 1u) ? (c ? g.pos1 : g.pos0) : in.at;
  return o;
}

// resolve_span for PairElem (see there): T threads over n elements; `whole`: only the span's composite (the reduce step)
template <int T>
__device__ __forceinline__ void pair_resolve_span(const PairElem* src, RunTileIn* dst, uint64_t n, PairState initial, unsigned long long off0,
                                                  PairElem (*chunk)[T], PairElem* whole, unsigned long long* total) {
  const uint32_t t = threadIdx.x;
  const uint64_t C = (n + T - 1) / T;
  const uint64_t lo = static_cast<uint64_t>(t) * C < n ? static_cast<uint64_t>(t) * C : n, hi = lo + C < n ? lo + C : n;
  PairElem identity;
  identity.m0 = identity.m1 = 0;
  identity.pos0 = identity.pos1 = kNone;
  identity.oo = 2u;    // (the bit is handed on)
  identity.kind = 0u;
  constexpr int kBatch = 4;
  PairElem e = identity;
  for (uint64_t i = lo; i < hi; i += kBatch) {
    PairElem sm[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; k++) sm[k] = src[i + k < hi ? i + k : i];
#pragma unroll
    for (int k = 0; k < kBatch; k++)
      if (i + k < hi) e = (i + k == lo) ? sm[k] : pair_compose(e, sm[k]);
  }
  int cur = 0;
  chunk[0][t] = e;
  __syncthreads();
  for (uint32_t d = 1; d < T; d <<= 1) {
    PairElem v = chunk[cur][t];
    if (t >= d) v = pair_compose(chunk[cur][t - d], v);
    chunk[cur ^ 1][t] = v;
    cur ^= 1;
    __syncthreads();
  }
  if (whole) {
    if (t == T - 1) *whole = chunk[cur][T - 1];
    return;
  }
  PairState st = initial;
  unsigned long long off = off0, closes = 0;
  if (t != 0) {
    st = pair_apply(chunk[cur][t - 1], initial, &closes);
    off += closes;
  }
  for (uint64_t i = lo; i < hi; i += kBatch) {
    PairElem sm[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; k++) sm[k] = src[i + k < hi ? i + k : i];
#pragma unroll
    for (int k = 0; k < kBatch; k++) {
      if (i + k >= hi) break;
      const PairState next = pair_apply(sm[k], st, &closes);
      RunTileIn ti;
      ti.s = st.open ? st.at : kNone;
      ti.q = kNone;
      ti.off = off;
      ti.cnt = closes;
      dst[i + k] = ti;
      off += closes;
      st = next;
    }
  }
  if (t == T - 1) *total = off;
}

// the Q stream and the reset stream of the lane's 32 bytes
template <int NR>
__device__ __forceinline__ void pair_streams_of(const RunParams& a, const RunMasks<NR>& mk, uint64_t at, const uint4& v0, const uint4& v1, bool loaded,
                                                uint32_t* SQ, uint32_t* RS) {
  uint32_t sa, sb, br;
  run_streams_of<NR, false>(a, mk, at, v0, v1, loaded, &sa, &sb, &br);
  *SQ = sa;
  *RS = br & ~sa;   // (every Q is a break: no Q lies inside L)
}
__device__ __forceinline__ uint64_t lanes_below(int lane) { return (1ull << lane) - 1ull; }
__device__ __forceinline__ uint32_t wave_total(uint32_t x) { return last_lane(wave_prefix_sum(x)); }

}  // namespace

template <int NR>
__global__ __launch_bounds__(256) void pair_summary(RunParams a) {
  const int lane = lane_id();
  const uint64_t tile = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  if (tile >= a.n_tiles) return;
  const RunMasks<NR> mk = run_masks<NR>(a.plan);
  const uint64_t base = (a.first_tile + tile) * a.tile_bytes;
  bool seen = false;                    // a reset has been met
  unsigned long long cur = 0;           // Q bytes since the last reset (or the tile's begin)
  unsigned long long k1 = 0, fixed = 0; // Q bytes before the first reset; pairs closed behind it
  unsigned long long last_q = kNone;
  const int kIters = static_cast<int>(a.tile_bytes / kIterBytes);
  uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
  bool loaded = false;
  auto fetch = [&](int it) {
    const uint64_t at = base + static_cast<uint64_t>(it) * kIterBytes + static_cast<uint64_t>(lane) * 32;
    loaded = base + static_cast<uint64_t>(it + 1) * kIterBytes <= a.n;   // (wave-uniform)
    if (loaded) {
      v0 = *reinterpret_cast<const uint4*>(a.text + at);
      v1 = *reinterpret_cast<const uint4*>(a.text + at + 16);
    }
  };
  fetch(0);
#pragma unroll 1
  for (int it = 0; it < kIters; it++) {
    const uint64_t it_base = base + static_cast<uint64_t>(it) * kIterBytes;
    if (it_base > a.n) break;
    uint32_t SQ, RS;
    pair_streams_of<NR>(a, mk, it_base + static_cast<uint64_t>(lane) * 32, v0, v1, loaded, &SQ, &RS);
    if (it + 1 < kIters) fetch(it + 1);
    const uint64_t qm = __ballot(SQ != 0), rm = __ballot(RS != 0);
    if ((qm | rm) == 0) continue;
    const uint32_t nq = static_cast<uint32_t>(__popc(SQ));
    if (qm != 0) {
      const int l = 63 - __builtin_clzll(qm);
      const uint32_t w = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(SQ), l));
      last_q = it_base + static_cast<uint32_t>(l) * 32u + 31u - static_cast<uint32_t>(__builtin_clz(w));
    }
    if (rm == 0) {
      cur += wave_total(nq);
      continue;
    }
    // resets: F = the Q bytes of the iteration below its first reset, Lq = below its last one; between the two every
    // segment (from one reset to the next) closes n / 2 pairs: (Lq - F - the number of odd segments) / 2 in all
    const uint32_t inc = wave_prefix_sum(nq);
    const uint32_t T = last_lane(inc);
    const int j1 = __builtin_ctzll(rm), jl = 63 - __builtin_clzll(rm);
    const bool hr = RS != 0;
    const uint32_t fr = hr ? static_cast<uint32_t>(__builtin_ctz(RS)) : 0u, lr = hr ? 31u - static_cast<uint32_t>(__builtin_clz(RS)) : 0u;
    const uint32_t qb = hr ? static_cast<uint32_t>(__popc(SQ & bits_below(fr))) : 0u;   // before the lane's first reset
    const uint32_t qa = hr ? static_cast<uint32_t>(__popc(SQ & ~bits_upto(lr))) : nq;   // behind its last one
    const uint32_t ev = inc - nq + qb, sv = inc - qa;
    const uint32_t F = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(ev), j1));
    const uint32_t Lq = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(sv), jl));
    const uint64_t S = __ballot(hr && (sv & 1u) != 0);
    const uint64_t lower = rm & lanes_below(lane);
    bool odd = false;
    if (hr && lower != 0) {   // the segment the lane's first reset closes began behind the last reset of lane i
      const int i = 63 - __builtin_clzll(lower);
      odd = ((ev ^ static_cast<uint32_t>(S >> i)) & 1u) != 0;
    }
    uint32_t n_odd = static_cast<uint32_t>(__popcll(__ballot(odd)));
    if (__ballot((RS & (RS - 1u)) != 0) != 0) {   // (some lane holds two resets: the segments inside its word)
      uint32_t inner = 0;
      if (hr) {
        uint32_t prev = fr, rest = RS & (RS - 1u);
        while (rest != 0) {
          const uint32_t r = static_cast<uint32_t>(__builtin_ctz(rest));
          rest &= rest - 1u;
          inner += static_cast<uint32_t>(__popc(SQ & bits_below(r) & ~bits_upto(prev))) & 1u;
          prev = r;
        }
      }
      n_odd += wave_total(inner);
    }
    const unsigned long long first_n = cur + F;
    if (!seen) {
      k1 = first_n;
      seen = true;
    } else {
      fixed += first_n / 2;
    }
    fixed += (Lq - F - n_odd) / 2;
    cur = T - Lq;
  }
  PairElem e;
  e.pos0 = e.pos1 = last_q;
  if (!seen) {   // no reset: the tile's Q bytes pair up from the incoming bit on
    k1 = cur;
    e.m0 = k1 / 2;
    e.m1 = (k1 + 1) / 2;
    e.oo = static_cast<uint32_t>(k1 & 1u) | (static_cast<uint32_t>((k1 + 1) & 1u) << 1);
    e.kind = k1 != 0 ? 3u : 0u;
  } else {
    fixed += cur / 2;
    e.m0 = k1 / 2 + fixed;
    e.m1 = (k1 + 1) / 2 + fixed;
    e.oo = (cur & 1u) ? 3u : 0u;
    e.kind = 3u;
  }
  if (lane == 0) reinterpret_cast<PairElem*>(a.summaries)[tile] = e;
}

__global__ __launch_bounds__(1024) void pair_resolve(RunParams a) {
  __shared__ PairElem chunk[2][1024];
  unsigned long long total = 0;
  pair_resolve_span<1024>(reinterpret_cast<const PairElem*>(a.summaries), a.tile_in, a.n_tiles, PairState{0u, kNone}, 0, chunk, nullptr, &total);
  if (threadIdx.x == 1023) leave_total(a, total);
}
__global__ __launch_bounds__(256) void pair_reduce(RunParams a) {
  __shared__ PairElem chunk[2][256];
  const uint64_t b = blockIdx.x, first = b * a.block_tiles;
  const uint64_t n = a.n_tiles - first < a.block_tiles ? a.n_tiles - first : a.block_tiles;
  PairElem* all = reinterpret_cast<PairElem*>(a.summaries);
  pair_resolve_span<256>(all + first, nullptr, n, PairState{0u, kNone}, 0, chunk, all + a.n_tiles + b, nullptr);
}
// (T = 256 up to 1024 blocks -- 32 GiB of text --: the scan's rounds over 1024 mostly idle threads were 35 us of a 1 GiB call)
template <int T>
__global__ __launch_bounds__(T) void pair_resolve_blocks(RunParams a) {
  __shared__ PairElem chunk[2][T];
  const uint64_t n_blocks = (a.n_tiles + a.block_tiles - 1) / a.block_tiles;
  unsigned long long total = 0;
  pair_resolve_span<T>(reinterpret_cast<const PairElem*>(a.summaries) + a.n_tiles, a.tile_in + a.n_tiles, n_blocks, PairState{0u, kNone}, 0, chunk, nullptr, &total);
  if (threadIdx.x == T - 1) leave_total(a, total);
}
__global__ __launch_bounds__(256) void pair_apply_blocks(RunParams a) {
  __shared__ PairElem chunk[2][256];
  const uint64_t b = blockIdx.x, first = b * a.block_tiles;
  const uint64_t n = a.n_tiles - first < a.block_tiles ? a.n_tiles - first : a.block_tiles;
  const RunTileIn in = a.tile_in[a.n_tiles + b];
  unsigned long long total = 0;
  pair_resolve_span<256>(reinterpret_cast<const PairElem*>(a.summaries) + first, a.tile_in + first, n, PairState{in.s != kNone ? 1u : 0u, in.s}, in.off, chunk, nullptr, &total);
}

template <int NR>
__global__ __launch_bounds__(256) void pair_emit(RunParams a) {
  const int lane = lane_id();
  const uint64_t tile = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  if (tile >= a.n_tiles) return;
  const RunTileIn in = a.tile_in[tile];
  if (in.cnt == 0) return;
  const RunMasks<NR> mk = run_masks<NR>(a.plan);
  const uint64_t base = (a.first_tile + tile) * a.tile_bytes;
  bool open = in.s != kNone;            // a Q is open; it sits at open_at
  unsigned long long open_at = in.s;
  unsigned long long pos = in.off;
  const int kIters = static_cast<int>(a.tile_bytes / kIterBytes);
  uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
  bool loaded = false;
  auto fetch = [&](int it) {
    const uint64_t at = base + static_cast<uint64_t>(it) * kIterBytes + static_cast<uint64_t>(lane) * 32;
    loaded = base + static_cast<uint64_t>(it + 1) * kIterBytes <= a.n;
    if (loaded) {
      v0 = *reinterpret_cast<const uint4*>(a.text + at);
      v1 = *reinterpret_cast<const uint4*>(a.text + at + 16);
    }
  };
  fetch(0);
#pragma unroll 1
  for (int it = 0; it < kIters; it++) {
    const uint64_t it_base = base + static_cast<uint64_t>(it) * kIterBytes;
    if (it_base > a.n) break;
    uint32_t SQ, RS;
    pair_streams_of<NR>(a, mk, it_base + static_cast<uint64_t>(lane) * 32, v0, v1, loaded, &SQ, &RS);
    if (it + 1 < kIters) fetch(it + 1);
    const uint64_t qm = __ballot(SQ != 0), rm = __ballot(RS != 0);
    if (qm == 0) {
      if (rm != 0) open = false;
      continue;
    }
    const uint32_t nq = static_cast<uint32_t>(__popc(SQ));
    const uint32_t inc = wave_prefix_sum(nq);
    const uint32_t T = last_lane(inc);
    // the lane's incoming parity: the Q bytes between the last reset below its word (or the iteration's begin, the carried
    // bit counted) and its word
    uint32_t ip = ((inc - nq) ^ (open ? 1u : 0u)) & 1u;
    uint32_t Lq = 0;
    if (rm != 0) {
      const bool hr = RS != 0;
      const uint32_t lr = hr ? 31u - static_cast<uint32_t>(__builtin_clz(RS)) : 0u;
      const uint32_t qa = hr ? static_cast<uint32_t>(__popc(SQ & ~bits_upto(lr))) : nq;
      const uint32_t sv = inc - qa;
      const uint64_t S = __ballot(hr && (sv & 1u) != 0);
      const uint64_t lower = rm & lanes_below(lane);
      if (lower != 0) {
        const int i = 63 - __builtin_clzll(lower);
        ip = ((inc - nq) ^ static_cast<uint32_t>(S >> i)) & 1u;
      }
      Lq = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(sv), 63 - __builtin_clzll(rm)));
    }
    // inside the word: the parity of the Q bytes below every bit; a reset makes what follows it start even
    uint32_t px = SQ;
    px ^= px << 1;
    px ^= px << 2;
    px ^= px << 4;
    px ^= px << 8;
    px ^= px << 16;
    const uint32_t before = px << 1;
    uint32_t flip = ip ? ~0u : 0u;
    for (uint32_t rs = RS; rs != 0; rs &= rs - 1u) {
      const uint32_t r = static_cast<uint32_t>(__builtin_ctz(rs));
      if (((before ^ flip) >> r) & 1u) flip ^= ~0u << r;
    }
    const uint32_t closers = SQ & (before ^ flip);
    const uint32_t mine = static_cast<uint32_t>(__popc(closers));
    // behind the iteration
    const unsigned long long carried_at = open_at;
    {
      const int l = 63 - __builtin_clzll(qm);
      const uint32_t w = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(SQ), l));
      open_at = it_base + static_cast<uint32_t>(l) * 32u + 31u - static_cast<uint32_t>(__builtin_clz(w));   // (read only while `open`)
    }
    open = rm != 0 ? ((T - Lq) & 1u) != 0 : (open != ((T & 1u) != 0));
    if (__ballot(mine != 0) == 0) continue;
    // the opener of a lane's first closer, when it lies below the lane's word: the last Q of the lanes below, else the carried one
    const uint32_t topq = SQ != 0 ? 31u - static_cast<uint32_t>(__builtin_clz(SQ)) : 0u;
    const uint64_t lowerq = qm & lanes_below(lane);
    const int pl = lowerq != 0 ? 63 - __builtin_clzll(lowerq) : 0;
    const uint32_t tq = static_cast<uint32_t>(__shfl(static_cast<int>(topq), pl));
    const unsigned long long below_at = lowerq != 0 ? it_base + static_cast<uint32_t>(pl) * 32u + tq : carried_at;
    const uint64_t word = it_base + static_cast<uint64_t>(lane) * 32u;
    const uint32_t incm = wave_prefix_sum(mine);
    unsigned long long idx = pos + incm - mine;
    for (uint32_t m = closers; m != 0; m &= m - 1u) {
      const uint32_t b = static_cast<uint32_t>(__builtin_ctz(m));
      const uint32_t under = SQ & bits_below(b);
      const unsigned long long begin = under != 0 ? word + 31u - static_cast<uint32_t>(__builtin_clz(under)) : below_at;
      if (idx < a.out_cap) *reinterpret_cast<ulonglong2*>(a.out + 2 * idx) = make_ulonglong2(begin, word + b + 1u);
      idx++;
    }
    pos += last_lane(incm);
  }
}

// Bytes per tile (a wave's share, a multiple of the 2-KiB iteration).  A wave pays one exposed trip to memory for its first
// iteration whatever its tile: 8 KiB keeps enough waves for texts of a few MiB, longer texts take 32 KiB (RJ_RUN_TILE_KB: measurements).
uint64_t run_tile_bytes(uint64_t span) {
  static const uint64_t forced = [] {
    const char* e = getenv("RJ_RUN_TILE_KB");
    const long x = e ? atol(e) : 0;
    return static_cast<uint64_t>(x >= 2 ? (x / 2) * 2048 : 0);
  }();
  if (forced) return forced;
  return span <= (256ull << 20) ? kRunTile : 4 * kRunTile;
}
uint64_t run_tiles(uint64_t sb, uint64_t n, uint64_t tile_bytes, uint64_t* first_tile) {
  *first_tile = sb / tile_bytes;
  return n / tile_bytes - *first_tile + 1;   // (the tile that holds position n -- the text's end -- is the last)
}

namespace {
template <int NR>
void launch_summary_nr(const RunParams& a, unsigned grid, hipEvent_t t0, hipEvent_t t1, hipStream_t st) {
  if (a.plan.has_b) hipExtLaunchKernelGGL((run_summary<NR, true>), dim3(grid), dim3(256), 0, st, t0, t1, 0, a);
  else hipExtLaunchKernelGGL((run_summary<NR, false>), dim3(grid), dim3(256), 0, st, t0, t1, 0, a);
}
template <int NR>
void launch_emit_nr(const RunParams& a, unsigned grid, hipEvent_t t1, hipStream_t st) {
  if (a.plan.has_b) hipExtLaunchKernelGGL((run_emit<NR, true>), dim3(grid), dim3(256), 0, st, nullptr, t1, 0, a);
  else hipExtLaunchKernelGGL((run_emit<NR, false>), dim3(grid), dim3(256), 0, st, nullptr, t1, 0, a);
}
}  // namespace

// RJ_RUN_TWO_LEVELS=<tiles per block>: every text through the two-level kernels, in blocks of that many tiles (tests)
static uint64_t forced_block_tiles() {
  static const uint64_t v = [] {
    const char* e = getenv("RJ_RUN_TWO_LEVELS");
    const long x = e ? atol(e) : 0;
    return static_cast<uint64_t>(x > 0 ? x : 0);
  }();
  return v;
}
uint64_t run_resolve_slots(uint64_t n_tiles) {
  const uint64_t bt = forced_block_tiles() ? forced_block_tiles() : kBlockTiles;
  return n_tiles + (n_tiles + bt - 1) / bt;
}
void launch_run_resolve(const RunParams& a0, hipStream_t st) {
  RunParams a = a0;
  a.block_tiles = forced_block_tiles() ? forced_block_tiles() : kBlockTiles;
  if (a.n_tiles <= kOneLevelTiles && !forced_block_tiles()) {
    hipLaunchKernelGGL(run_resolve, dim3(1), dim3(1024), 0, st, a);
    return;
  }
  const unsigned blocks = static_cast<unsigned>((a.n_tiles + a.block_tiles - 1) / a.block_tiles);
  hipLaunchKernelGGL(run_reduce, dim3(blocks), dim3(256), 0, st, a);
  hipLaunchKernelGGL(run_resolve_blocks, dim3(1), dim3(1024), 0, st, a);
  hipLaunchKernelGGL(run_apply, dim3(blocks), dim3(256), 0, st, a);
}

// (eight kernels each: the ranges rounded up to 1 / 2 / 4 / 8, with and without a B position)
void launch_run_summary(const RunParams& a, hipEvent_t t0, hipEvent_t t1, hipStream_t st) {
  const unsigned grid = static_cast<unsigned>((a.n_tiles + 3) / 4);
  const uint32_t nr = a.plan.n_ranges;
  if (nr <= 1) launch_summary_nr<1>(a, grid, t0, t1, st);
  else if (nr <= 2) launch_summary_nr<2>(a, grid, t0, t1, st);
  else if (nr <= 4) launch_summary_nr<4>(a, grid, t0, t1, st);
  else launch_summary_nr<8>(a, grid, t0, t1, st);
}
void launch_run_emit(const RunParams& a, hipEvent_t t1, hipStream_t st) {
  const unsigned grid = static_cast<unsigned>((a.n_tiles + 3) / 4);
  const uint32_t nr = a.plan.n_ranges;
  if (nr <= 1) launch_emit_nr<1>(a, grid, t1, st);
  else if (nr <= 2) launch_emit_nr<2>(a, grid, t1, st);
  else if (nr <= 4) launch_emit_nr<4>(a, grid, t1, st);
  else launch_emit_nr<8>(a, grid, t1, st);
}

// ---- the pair shape
void launch_pair_summary(const RunParams& a, hipEvent_t t0, hipEvent_t t1, hipStream_t st) {
  const unsigned grid = static_cast<unsigned>((a.n_tiles + 3) / 4);
  const uint32_t nr = a.plan.n_ranges;
  if (nr <= 1) hipExtLaunchKernelGGL((pair_summary<1>), dim3(grid), dim3(256), 0, st, t0, t1, 0, a);
  else if (nr <= 2) hipExtLaunchKernelGGL((pair_summary<2>), dim3(grid), dim3(256), 0, st, t0, t1, 0, a);
  else if (nr <= 4) hipExtLaunchKernelGGL((pair_summary<4>), dim3(grid), dim3(256), 0, st, t0, t1, 0, a);
  else hipExtLaunchKernelGGL((pair_summary<8>), dim3(grid), dim3(256), 0, st, t0, t1, 0, a);
}
void launch_pair_resolve(const RunParams& a0, hipStream_t st) {
  RunParams a = a0;
  a.block_tiles = forced_block_tiles() ? forced_block_tiles() : kBlockTiles;
  if (a.n_tiles <= kOneLevelTiles && !forced_block_tiles()) {
    hipLaunchKernelGGL(pair_resolve, dim3(1), dim3(1024), 0, st, a);
    return;
  }
  const unsigned blocks = static_cast<unsigned>((a.n_tiles + a.block_tiles - 1) / a.block_tiles);
  hipLaunchKernelGGL(pair_reduce, dim3(blocks), dim3(256), 0, st, a);
  if (blocks <= 1024) hipLaunchKernelGGL(pair_resolve_blocks<256>, dim3(1), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(pair_resolve_blocks<1024>, dim3(1), dim3(1024), 0, st, a);
  hipLaunchKernelGGL(pair_apply_blocks, dim3(blocks), dim3(256), 0, st, a);
}
void launch_pair_emit(const RunParams& a, hipEvent_t t1, hipStream_t st) {
  const unsigned grid = static_cast<unsigned>((a.n_tiles + 3) / 4);
  const uint32_t nr = a.plan.n_ranges;
  if (nr <= 1) hipExtLaunchKernelGGL((pair_emit<1>), dim3(grid), dim3(256), 0, st, nullptr, t1, 0, a);
  else if (nr <= 2) hipExtLaunchKernelGGL((pair_emit<2>), dim3(grid), dim3(256), 0, st, nullptr, t1, 0, a);
  else if (nr <= 4) hipExtLaunchKernelGGL((pair_emit<4>), dim3(grid), dim3(256), 0, st, nullptr, t1, 0, a);
  else hipExtLaunchKernelGGL((pair_emit<8>), dim3(grid), dim3(256), 0, st, nullptr, t1, 0, a);
}

}  // namespace rejit_amd
