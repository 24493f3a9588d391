// rejit_amd/csrc/device_program.h -- the lowered program as the kernels see it (plain
// data in HBM, passed to every kernel by value as a small descriptor of pointers), and
// the per-lane automaton step shared by the verify / dense kernels.
//
// This header is deliberately free of HIP runtime calls so that the lane simulator can
// also be compiled by g++ in the CPU unit tests (tests/support/) -- the SAME source the
// kernels run, which is how the automaton logic is checked without a GPU.
#ifndef REJIT_AMD_DEVICE_PROGRAM_H_
#define REJIT_AMD_DEVICE_PROGRAM_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define RJ_HD __host__ __device__ __forceinline__
#else
#define RJ_HD inline
#endif

namespace rejit_amd {

constexpr int kDevMaxWindows = 8;

// Lane-packed pre-steps of the dense kernel (dense_swar.h): automata of at most 8 positions without
// assertions whose classes are a few byte ranges.  Four starts share a register, one state BYTE each; the
// class rows of four text bytes are computed with byte-parallel range tests instead of table lookups.
constexpr int kSwarMaxRanges = 12;
struct SwarPlan {
  uint32_t n_ranges;  // 0: not usable
  uint32_t n_low;     // ranges [0, n_low) lie in 0x00..0x7f, the rest in 0x80..0xff
  uint32_t depth;     // pre-steps: 1, 2 or 4
  // range r = [lo, hi] within one half of the byte values: a byte b (low 7 bits b7) is inside when
  // b7 + add_lo carries into bit 7 (b7 >= lo) and b7 + add_hi does not (b7 <= hi); replicated x4
  uint32_t add_lo[kSwarMaxRanges], add_hi[kSwarMaxRanges];
  uint32_t shift[kSwarMaxRanges];  // 7 - position: bit 7 of the test goes to the position's bit
  // position masks replicated into all four bytes: the first set, positions that pass to i + 1, loops
  // (stay), positions whose follow set is neither (their starts go to the walkers)
  uint32_t first, step1, loopm, gen;
  uint32_t last_shift[2];  // the (one or two) accepting positions
  uint32_t first_shift;    // the first position when there is exactly one (DevProgram::loop_first)
};

// Table layout (all uint32 words, W = n_words, C = n_ctx in {1,4}):
//   first [C][W]   last [C][W]   linear [W]   row_of [P] (int32)   rows [C][n_rows][W]
//   cls [256][W]
struct DevProgram {
  int32_t n_pos;
  int32_t n_words;
  int32_t n_ctx;        // 1 when the pattern has no ^ / $, else 4
  int32_t n_rows;
  uint32_t nullable;    // bit c: the empty string matches in context c (bits replicated when n_ctx == 1)
  int32_t mode;         // 0 dense, 1 windows
  int32_t n_windows;
  uint32_t win_offset;
  uint32_t win_len;     // pattern bytes covered by a window (1..8)
  uint32_t win_value0[kDevMaxWindows], win_mask0[kDevMaxWindows];  // first dword of the window
  uint32_t win_value1[kDevMaxWindows], win_mask1[kDevMaxWindows];  // second dword (win_len > 4)
  uint32_t float_range;  // candidate starts per window hit: 1 = fixed offset, else float_max-float_min+1
  uint32_t float_max;    // floating windows: start s = w - float_max + delta, delta < float_range
  uint32_t first_bytes[8];
  uint64_t min_len;
  const uint32_t* first;
  const uint32_t* last;
  const uint32_t* linear;
  const int32_t* row_of;
  const uint32_t* rows;
  const uint32_t* cls;
  uint32_t table_words;  // the tables are one contiguous blob of this many words starting at `first`
  // lane-sized automata (n_words <= 4) only: non-linear positions whose follow set is, in every
  // context, exactly {i, i+1} (x+ / x*) or {i+1, i+2} (the position before an optional one);
  // the dense walker steps them with shifts instead of fetching their rows
  uint32_t loop_mask[4];
  uint32_t skip_mask[4];
  // A walk from one start is cut after this many bytes and the run flagged (kCntOverrun); the engine
  // then repeats the run on the linear-time carry scan (carry_scan.h).  <= kMaxSimSteps.
  uint32_t max_walk;
  // Windows behind an unbounded prefix (lowering.h: Program::behind): a hit of window k at w makes the
  // positions cut_fwd[k] live at w (they consume text[w]); cut_rev[k] = the same set in the reverse
  // automaton's numbering.  Lane-sized automata only (n_words <= 4).
  uint32_t behind;
  // `X+ rest`: in every context a match can only begin at ONE position, and that position follows itself
  // (no assertions, not nullable).  A start s whose previous byte is in X as well is then never selected:
  // the thread of s - 1 passes through the same position at s, so it reaches every end s reaches (s - 1
  // is a candidate whenever s is, with an end at least as far), and no selected match can END at s -- its
  // own thread would still be alive in X at s and run on to whatever s reaches.  The dense kernel
  // therefore takes only the FIRST byte of every run of X as a candidate: `[a-z]+` over 1 GB has 338 M
  // starts with a match but 35 M runs.
  uint32_t loop_first;
  uint32_t cut_fwd[kDevMaxWindows][4];
  uint32_t cut_rev[kDevMaxWindows][4];
  SwarPlan swar;
  // Bounded patterns of at most 16 bytes without assertions and <= 64 positions (the nine regexdna
  // patterns: 8): the longest possible match in bytes, else 0.  rj_lane_longest_short then reads a
  // candidate's text with two loads and fetches the class rows of its bytes together, instead of a
  // byte load and a row load per step, each waiting for the one before.
  uint32_t short_max;
};

// The NFA graph for the exact sequential kernel (reference ring semantics).
//   byte edge e: src/dst state, len > 0: literal bytes lit[off .. off+len); len == 0: one byte
//   out of the 256-bit class cls[off*8 .. off*8+8)
//   control edge: kind 0 epsilon, 1 start-of-line, 2 end-of-line
struct DevGraph {
  int32_t n_states, entry, exit;
  int32_t n_byte_edges, n_control_edges;
  int32_t times;  // 1 + min(longest literal edge, 64)
  const int32_t* be_src;
  const int32_t* be_dst;
  const int32_t* be_len;
  const int32_t* be_off;
  const uint8_t* lit;
  const uint32_t* cls;
  const int32_t* ce_src;
  const int32_t* ce_dst;
  const int32_t* ce_kind;
};

RJ_HD bool rj_line_break(uint32_t c) { return c == '\n' || c == '\r'; }

// The text as the walkers read it: `t[p]` for a plain pointer, or through RjCachedText -- 16 aligned bytes
// of the text held in registers, reloaded when a walk leaves them.  A walk's step is then a chain of
// register operations and table lookups; with a plain pointer every step waits for a byte load, which for
// the text around a fast-forward hit (streamed past minutes of kernel time ago, long out of the caches) is
// a trip to HBM.  The text must be 16-byte aligned (the engine's device texts are) unless built for the host.
struct RjCachedText {
  const uint8_t* t;
  uint64_t n;
  mutable uint64_t base, lo, hi;
  RJ_HD RjCachedText(const uint8_t* text, uint64_t len) : t(text), n(len), base(~0ull), lo(0), hi(0) {}
  RJ_HD uint8_t operator[](uint64_t p) const {
    const uint64_t b = p & ~15ull;
    if (b != base) {
      base = b;
      if (b + 16 <= n) {
#if defined(__HIP_DEVICE_COMPILE__)
        const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(t + b);
        lo = v.x;
        hi = v.y;
#else
        __builtin_memcpy(&lo, t + b, 8);
        __builtin_memcpy(&hi, t + b + 8, 8);
#endif
      } else {
        lo = hi = 0;
        for (uint64_t k = 0; k < 16 && b + k < n; k++) {
          const uint64_t c = t[b + k];
          if (k < 8) lo |= c << (8 * k);
          else hi |= c << (8 * (k - 8));
        }
      }
    }
    // (an arithmetic select: `(p & 8) ? hi : lo` becomes an indexed load from the object, which then lives
    // in scratch memory instead of registers)
    const uint64_t m = 0ull - ((p >> 3) & 1ull);
    return static_cast<uint8_t>(((lo & ~m) | (hi & m)) >> (8 * (p & 7)));
  }
};

// Context at text position p: bit0 start-of-line, bit1 end-of-line
// (MatchStartOrEndOfLine, reference src/x64/codegen-x64.cc:686-708).  Text: const uint8_t* or RjCachedText.
template <class Text>
RJ_HD int rj_context(const Text& t, uint64_t n, uint64_t p) {
  int ctx = 0;
  if (p == 0 || rj_line_break(t[p - 1])) ctx |= 1;
  if (p == n || rj_line_break(t[p])) ctx |= 2;
  return ctx;
}

// Longest match that starts exactly at s, automaton state held in NQ 64-bit words per
// lane (P <= 64 * NQ).  Returns false when no match starts at s.
// One step is S' = follow_ctx(S) & cls[byte]: the linear part is a shift, positions with
// a non-trivial follow set OR in their row.
// P.max_walk bounds the walk: a start that is still alive after that many bytes sets *overrun (per-start
// walks are quadratic when many starts live long; the engine then repeats the run on the linear-time
// carry scan).  `abort` (may be null) is polled now and then: once any walk of the run has overrun
// the run is void, and the others need not finish.
constexpr uint64_t kMaxSimSteps = 1ull << 20;
// A walk that passes kLongWalk bytes reports itself (once) in the run's long-walk counter, kLongWalksAfterOverrun
// slots behind the overrun flag `abort` points at; the kLongWalkBudget-th such walk of a run voids the run like one
// that reaches P.max_walk.  A FEW long candidates -- a window hit on a very long line -- are cheaper to walk than to
// hand the text to the carry scan, which is why max_walk is 64 Ki bytes in windows mode; but `a.*b` over a text
// without line breaks has a long candidate at every `a`, and all of them walked 64 Ki dependent steps (~0.8 us each
// from device memory: 54 ms) before the first one reached the limit.
constexpr uint32_t kLongWalk = 4096;
constexpr unsigned long long kLongWalkBudget = 64;
constexpr int kLongWalksAfterOverrun = 4;

template <int NQ, class Text>
RJ_HD bool rj_lane_longest(const DevProgram& P, const Text& t, uint64_t n, uint64_t s, uint64_t* end,
                           bool* overrun, const volatile unsigned long long* abort = nullptr) {
  const int W = P.n_words;  // 32-bit words; NQ*2 >= W
  const bool ctxed = P.n_ctx > 1;
  int ctx = ctxed ? rj_context(t, n, s) : 0;
  bool found = false;
  if ((P.nullable >> ctx) & 1u) {
    found = true;
    *end = s;
  }
  if (s >= n || P.n_pos == 0) return found;

  uint64_t S[NQ], lin[NQ];
  {
    const uint32_t* fr = P.first + ctx * W;
    const uint32_t* cr = P.cls + (uint32_t)t[s] * W;
    for (int q = 0; q < NQ; q++) {
      uint64_t f = 0, c = 0, l = 0;
      if (2 * q < W) { f = fr[2 * q]; c = cr[2 * q]; l = P.linear[2 * q]; }
      if (2 * q + 1 < W) {
        f |= (uint64_t)fr[2 * q + 1] << 32;
        c |= (uint64_t)cr[2 * q + 1] << 32;
        l |= (uint64_t)P.linear[2 * q + 1] << 32;
      }
      S[q] = f & c;
      lin[q] = l;
    }
  }
  uint64_t p = s + 1;
  for (;;) {
    uint64_t alive = 0;
    for (int q = 0; q < NQ; q++) alive |= S[q];
    if (!alive) break;
    ctx = ctxed ? rj_context(t, n, p) : 0;
    {
      const uint32_t* lr = P.last + ctx * W;
      uint64_t acc = 0;
      for (int q = 0; q < NQ; q++) {
        uint64_t l = 0;
        if (2 * q < W) l = lr[2 * q];
        if (2 * q + 1 < W) l |= (uint64_t)lr[2 * q + 1] << 32;
        acc |= S[q] & l;
      }
      if (acc) {
        found = true;
        *end = p;
      }
    }
    if (p == n) break;
    if (p - s >= P.max_walk) {
      *overrun = true;
      break;
    }
    if (abort != nullptr && ((p - s) & 255u) == 0 && *abort != 0) break;  // the run is void already
#if defined(__HIP_DEVICE_COMPILE__)
    if (abort != nullptr && p - s == kLongWalk && P.max_walk > kLongWalk) {
      unsigned long long* long_walks = const_cast<unsigned long long*>(abort) + kLongWalksAfterOverrun;
      if (atomicAdd(long_walks, 1ull) + 1 >= kLongWalkBudget) {
        *overrun = true;
        break;
      }
    }
#endif
    uint64_t T[NQ];
    uint64_t carry = 0;
    for (int q = 0; q < NQ; q++) {
      uint64_t x = S[q] & lin[q];
      T[q] = (x << 1) | carry;
      carry = x >> 63;
    }
    for (int q = 0; q < NQ; q++) {
      uint64_t sp = S[q] & ~lin[q];
      while (sp) {
        int b = __builtin_ctzll(sp);
        sp &= sp - 1;
        const uint32_t* row = P.rows + ((size_t)ctx * P.n_rows + P.row_of[q * 64 + b]) * W;
        for (int j = 0; j < NQ; j++) {
          uint64_t r = 0;
          if (2 * j < W) r = row[2 * j];
          if (2 * j + 1 < W) r |= (uint64_t)row[2 * j + 1] << 32;
          T[j] |= r;
        }
      }
    }
    const uint32_t* cr = P.cls + (uint32_t)t[p] * W;
    for (int q = 0; q < NQ; q++) {
      uint64_t c = 0;
      if (2 * q < W) c = cr[2 * q];
      if (2 * q + 1 < W) c |= (uint64_t)cr[2 * q + 1] << 32;
      S[q] = T[q] & c;
    }
    p++;
  }
  return found;
}

// rj_lane_longest for DevProgram::short_max != 0 (n_ctx == 1, n_words <= 2, every match at most short_max
// <= 16 bytes): same result, but the dependent chain is  text (two 8-byte loads) -> the class rows of all
// bytes (independent loads) -> register arithmetic,  where the general walk has a byte load and a row load
// per step, each waiting for the previous step.  The tails of the headline run are made of such chains.
//
// rj_load16: the 16 text bytes from s on (zeros beyond the end of the text)
RJ_HD void rj_load16(const uint8_t* t, uint64_t n, uint64_t s, uint64_t* lo_out, uint64_t* hi_out) {
  uint64_t lo = 0, hi = 0;
  if (s + 16 <= n) {
    __builtin_memcpy(&lo, t + s, 8);
    __builtin_memcpy(&hi, t + s + 8, 8);
  } else {
    for (uint32_t k = 0; k < 16 && s + k < n; k++) {
      const uint64_t c = t[s + k];
      if (k < 8) lo |= c << (8 * k);
      else hi |= c << (8 * (k - 8));
    }
  }
  *lo_out = lo;
  *hi_out = hi;
}

// (lo, hi) = rj_load16(t, n, s): callers that test several patterns at one start load the text once
RJ_HD bool rj_lane_longest_short_at(const DevProgram& P, uint64_t lo, uint64_t hi, uint64_t n, uint64_t s, uint64_t* end) {
  const int W = P.n_words;
  bool found = false;
  if (P.nullable & 1u) {
    found = true;
    *end = s;
  }
  if (s >= n || P.n_pos == 0) return found;
  const uint32_t L = P.short_max;
  const uint32_t avail = n - s < 16 ? static_cast<uint32_t>(n - s) : 16u;
  const uint32_t steps = L < avail ? L : avail;  // bytes a match can consume here
  uint64_t row[16];
#pragma unroll
  for (uint32_t k = 0; k < 16; k++) {
    row[k] = 0;
    if (k < steps) {
      const uint32_t c = static_cast<uint32_t>(((k < 8 ? lo : hi) >> (8 * (k & 7))) & 0xFFu);
      const uint32_t* cr = P.cls + c * W;
      row[k] = cr[0];
      if (W > 1) row[k] |= static_cast<uint64_t>(cr[1]) << 32;
    }
  }
  uint64_t first = P.first[0], last = P.last[0], lin = P.linear[0];
  if (W > 1) {
    first |= static_cast<uint64_t>(P.first[1]) << 32;
    last |= static_cast<uint64_t>(P.last[1]) << 32;
    lin |= static_cast<uint64_t>(P.linear[1]) << 32;
  }
  uint64_t S = first & row[0];
#pragma unroll
  for (uint32_t k = 1; k <= 16; k++) {
    if (S == 0 || k > steps) break;
    if (S & last) {
      found = true;
      *end = s + k;
    }
    if (k == steps) break;
    uint64_t T = (S & lin) << 1;
    for (uint64_t sp = S & ~lin; sp; sp &= sp - 1) {
      const uint32_t* r = P.rows + static_cast<size_t>(P.row_of[__builtin_ctzll(sp)]) * W;
      T |= r[0];
      if (W > 1) T |= static_cast<uint64_t>(r[1]) << 32;
    }
    S = T & row[k];
  }
  return found;
}

RJ_HD bool rj_lane_longest_short(const DevProgram& P, const uint8_t* t, uint64_t n, uint64_t s, uint64_t* end) {
  uint64_t lo, hi;
  rj_load16(t, n, s, &lo, &hi);
  return rj_lane_longest_short_at(P, lo, hi, n, s, end);
}

// Candidate test used by the dense scan: can a match start at s at all?
RJ_HD bool rj_dense_candidate(const DevProgram& P, const uint8_t* t, uint64_t n, uint64_t s) {
  if (P.nullable) {
    int ctx = P.n_ctx > 1 ? rj_context(t, n, s) : 0;
    if ((P.nullable >> ctx) & 1u) return true;
  }
  if (s >= n) return false;
  uint32_t c = t[s];
  return (P.first_bytes[c >> 5] >> (c & 31)) & 1u;
}

// Left-most-longest selection over candidates sorted by begin (one entry per begin):
// the sequential definition (MatchAllAppendFilter + the non-overlap rule, reference
// src/codegen.cc:36-86, codegen-x64.cc:448-460), used by the single-lane cluster walk.
struct RjSelectState {
  uint64_t cur;       // smallest begin the next match may have
  uint64_t prev_end;
  bool have_prev;
};

// returns true when (b,e) is to be emitted
RJ_HD bool rj_select_step(RjSelectState* st, uint64_t b, uint64_t e, bool* taken) {
  *taken = false;
  if (b < st->cur) return false;
  *taken = true;
  st->cur = e > b ? e : b + 1;
  bool drop = (e == b) && st->have_prev && st->prev_end == b;  // zero-length rule
  st->have_prev = true;
  st->prev_end = e;
  return !drop;
}

}  // namespace rejit_amd
#endif
