// rejit_amd/csrc/lds_walk.h -- the per-start automaton walks of the verify kernels over PADDED tables
// (verify_lds.hip), shared with the CPU unit tests (tests/support/carry_exec.cc: every walk here against the
// one it replaces).
//
// The walkers of device_program.h / behind_walk.h read 32-bit table words through generic pointers, each
// word behind its own `word < n_words` test: on the GPU a step became a chain of 10-20 flat loads, every one
// waiting for the one before (0.6-0.9 us per text byte measured -- a hit of the complex benchmark regex,
// 42 bytes long, cost 35 us; tools/verify_trace.py).  Here a table row is NQ 64-bit words whatever the
// automaton's width (NQ = 1: <= 64 positions, 2: <= 128), rows are indexed by POSITION (no row_of
// indirection; row n_pos is all zero) and the blob is copied to LDS as it is, so a step is
//     one row read per non-linear live position (two positions per round, independent reads)
//   + one class row read,
// each a single ds_read of 8 or 16 bytes.  Results are identical to the old walkers by construction
// (same recurrence S' = follow_ctx(S) & cls[byte]); tests/test_carry_scan.py::test_lds_walkers_* checks it.
//
// Blob layout (uint64 words):  first [C][NQ]  last [C][NQ]  linear [NQ]  loops [NQ]  rows [C][n_pos + 1][NQ]  cls [256][NQ]
// (loops: non-linear positions whose follow set is {i, i + 1} in every context -- x+, x*, .* -- stepped with two
// shifts instead of a row read: the commonest kind, and the read was the one LDS trip left in a step's chain)
#ifndef REJIT_AMD_LDS_WALK_H_
#define REJIT_AMD_LDS_WALK_H_

#include <stdint.h>

#include "device_program.h"

#ifndef RJ_STAMP  // (kernels.hip, -DRJ_TRACE_VERIFY: phase time stamps of a debug build)
#define RJ_STAMP(i) ((void)0)
#endif

namespace rejit_amd {

RJ_HD uint32_t lw_blob_words(int nq, int n_ctx, int n_pos) {
  return static_cast<uint32_t>(nq) * static_cast<uint32_t>(2 * n_ctx + 2 + n_ctx * (n_pos + 1) + 256);
}

// The text as the walks read it: `t[p]` for single bytes, and for the tight loops
//   span(p, &ptr)       k >= 0 bytes text[p .. p + k) readable as ptr[0 .. k)   (never beyond the end of the text)
//   span_back(p, &ptr)  k >= 0 bytes text[p - 1], text[p - 2], ... readable as ptr[0], ptr[-1], ...
// This one is plain memory (CPU tests; verify_lds.hip has the LDS-window versions).
struct PlainText {
  const uint8_t* t;
  uint64_t n;
  RJ_HD PlainText(const uint8_t* text, uint64_t len) : t(text), n(len) {}
  RJ_HD uint8_t operator[](uint64_t p) const { return t[p]; }
  RJ_HD uint32_t span(uint64_t p, const uint8_t** ptr) const {
    *ptr = t + p;
    const uint64_t k = p < n ? n - p : 0;
    return k > 0x40000000u ? 0x40000000u : static_cast<uint32_t>(k);
  }
  RJ_HD uint32_t span_back(uint64_t p, const uint8_t** ptr) const {
    *ptr = t + p - 1;
    return p > 0x40000000u ? 0x40000000u : static_cast<uint32_t>(p);
  }
};

// The two LDS-window texts of verify_lds.hip.  `Loader::block16(dst, text, n, at)` puts text[at .. at + 16) (bytes at
// or beyond n as 0) at dst; on the GPU dst is LDS, the CPU tests run the same index arithmetic over plain arrays
// with every access checked (tests/support/carry_exec.cc).
//
// Floating windows: a window shared by the wave (every lane walks a start of the same hit); a walk that leaves it
// -- a match of several hundred bytes -- goes on through 16 bytes of its own, refilled as it moves.
template <class Loader>
struct WaveWindowText {
  const uint8_t* win;
  uint8_t* slot;  // 16 bytes of this lane
  uint64_t base;
  uint32_t len;
  const uint8_t* text;
  uint64_t n;
  mutable uint64_t slot_base;
  RJ_HD WaveWindowText(const uint8_t* w, uint8_t* own, uint64_t b, uint32_t l, const uint8_t* t, uint64_t tn)
      : win(w), slot(own), base(b), len(l), text(t), n(tn), slot_base(~0ull) {}
  RJ_HD const uint8_t* far(uint64_t p) const {
    const uint64_t b = p & ~15ull;
    if (b != slot_base) {
      slot_base = b;
      Loader::block16(slot, text, n, b);
    }
    return slot + (static_cast<uint32_t>(p) & 15u);
  }
  // (plain 64-bit comparisons: the wrap-around forms `uint32(p - base) < len`, `d - 1u < len` went wrong in the
  // device build of LaneWindowText below -- a backward walk read on past the window's first byte and past text[0])
  RJ_HD bool holds(uint64_t p) const { return p >= base && p - base < len; }
  RJ_HD uint8_t operator[](uint64_t p) const { return holds(p) ? win[static_cast<uint32_t>(p - base)] : *far(p); }
  RJ_HD uint32_t span(uint64_t p, const uint8_t** ptr) const {
    if (holds(p)) {
      const uint32_t d = static_cast<uint32_t>(p - base);
      *ptr = win + d;
      return len - d;
    }
    if (p >= n) return 0;
    *ptr = far(p);
    const uint32_t k = 16u - (static_cast<uint32_t>(p) & 15u);
    return n - p < k ? static_cast<uint32_t>(n - p) : k;
  }
  RJ_HD uint32_t span_back(uint64_t p, const uint8_t** ptr) const {
    if (p > base && p - base <= len) {
      const uint32_t d = static_cast<uint32_t>(p - base);
      *ptr = win + d - 1;
      return d;
    }
    if (p == 0) return 0;
    *ptr = far(p - 1);
    return (static_cast<uint32_t>(p - 1) & 15u) + 1u;
  }
};

// Behind an unbounded prefix: a lane per hit, so the window (WIN bytes, a multiple of 16) is the lane's own and
// simply moves when a walk leaves it.
template <class Loader, uint32_t WIN>
struct LaneWindowText {
  uint8_t* win;
  mutable uint64_t base;
  mutable uint32_t len;
  const uint8_t* text;
  uint64_t n;
  RJ_HD LaneWindowText(uint8_t* w, uint64_t b, uint32_t l, const uint8_t* t, uint64_t tn) : win(w), base(b), len(l), text(t), n(tn) {}
  RJ_HD void move(uint64_t nb) const {  // (rare: block by block)
    base = nb;
    len = n - nb < WIN ? static_cast<uint32_t>(n - nb) : WIN;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (uint32_t k = 0; k < WIN / 16; k++) Loader::block16(win + 16 * k, text, n, nb + 16 * k);
  }
  RJ_HD bool holds(uint64_t p) const { return p >= base && p - base < len; }
  RJ_HD uint8_t operator[](uint64_t p) const {
    if (!holds(p)) move(p & ~15ull);
    return win[static_cast<uint32_t>(p - base)];
  }
  RJ_HD uint32_t span(uint64_t p, const uint8_t** ptr) const {
    if (p >= n) return 0;
    if (!holds(p)) move(p & ~15ull);
    const uint32_t d = static_cast<uint32_t>(p - base);
    *ptr = win + d;
    return len - d;
  }
  RJ_HD uint32_t span_back(uint64_t p, const uint8_t** ptr) const {
    if (p == 0) return 0;
    if (!holds(p - 1)) {  // put text[p - 1] into the window's last block
      const uint64_t blk = (p - 1) & ~15ull;
      move(blk >= WIN - 16 ? blk - (WIN - 16) : 0ull);
    }
    const uint32_t d = static_cast<uint32_t>(p - base);
    *ptr = win + d - 1;
    return d;
  }
};

template <int NQ>
struct WalkTab {
  const uint64_t* first;
  const uint64_t* last;
  const uint64_t* rows;
  const uint64_t* cls;
  uint64_t linear[NQ];
  uint64_t loops[NQ];
  int32_t n_ctx, n_pos;
  uint32_t nullable;  // DevProgram::nullable
  uint32_t max_walk;  // DevProgram::max_walk
};

template <int NQ>
RJ_HD WalkTab<NQ> lw_point(const uint64_t* blob, int n_ctx, int n_pos, uint32_t nullable, uint32_t max_walk) {
  WalkTab<NQ> T;
  T.first = blob;
  T.last = blob + n_ctx * NQ;
  const uint64_t* lin = blob + 2 * n_ctx * NQ;
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    T.linear[q] = lin[q];
    T.loops[q] = lin[NQ + q];
  }
  T.rows = lin + 2 * NQ;
  T.cls = T.rows + static_cast<uint32_t>(n_ctx * (n_pos + 1)) * NQ;
  T.n_ctx = n_ctx;
  T.n_pos = n_pos;
  T.nullable = nullable;
  T.max_walk = max_walk;
  return T;
}

// out = follow_ctx(S): linear positions shift, the others OR their rows in -- two rows per round, the second
// one the zero row when only one position is left, so both reads are issued together
template <int NQ>
RJ_HD void lw_follow(const WalkTab<NQ>& T, const uint64_t (&S)[NQ], int ctx, uint64_t (&out)[NQ]) {
  uint64_t carry = 0;
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    const uint64_t l = S[q] & T.loops[q], x = (S[q] & T.linear[q]) | l;  // (disjoint masks)
    out[q] = (x << 1) | carry | l;
    carry = x >> 63;
  }
  const uint64_t* rows = T.rows + static_cast<uint32_t>(ctx * (T.n_pos + 1)) * NQ;
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    uint64_t sp = S[q] & ~(T.linear[q] | T.loops[q]);
    while (sp) {
      const int i0 = q * 64 + __builtin_ctzll(sp);
      sp &= sp - 1;
      int i1 = T.n_pos;
      if (sp) {
        i1 = q * 64 + __builtin_ctzll(sp);
        sp &= sp - 1;
      }
      const uint64_t* r0 = rows + static_cast<uint32_t>(i0) * NQ;
      const uint64_t* r1 = rows + static_cast<uint32_t>(i1) * NQ;
#pragma unroll
      for (int j = 0; j < NQ; j++) out[j] |= r0[j] | r1[j];
    }
  }
}

// context bits at boundary p given the bytes on both sides (prev = text[p - 1], cur = text[p])
RJ_HD int lw_ctx(bool at_begin, uint32_t prev, bool at_end, uint32_t cur) {
  return ((at_begin || rj_line_break(prev)) ? 1 : 0) | ((at_end || rj_line_break(cur)) ? 2 : 0);
}

// CTX (the automaton has ^ / $ contexts, WalkTab::n_ctx > 1) is a template parameter of every walk on purpose.  As a
// run-time flag it is uniform, and the compiler kept it as a lane mask computed INSIDE the loop of one walk (bits
// only for the lanes still walking there) and tested it with `s_and vcc, exec, mask` in the next walk: lanes that
// had left the first loop early then took the context path of an automaton without contexts, read the wrong
// `last` row and lost the last byte of a match followed by a line break or the end of the text (temporal
// divergence; found with tools/dbg_behind.py, one text in 750).
//
// = rj_lane_longest (device_program.h): the longest match that starts exactly at s
template <int NQ, bool CTX, class Text>
RJ_HD bool lw_longest(const WalkTab<NQ>& T, const Text& t, uint64_t n, uint64_t s, uint64_t* end, bool* overrun,
                      const volatile unsigned long long* abort = nullptr) {
  constexpr bool ctxed = CTX;
  uint32_t cur = s < n ? t[s] : 0u;
  int ctx = ctxed ? lw_ctx(s == 0, s == 0 ? 0u : t[s - 1], s == n, cur) : 0;
  bool found = false;
  if ((T.nullable >> ctx) & 1u) {
    found = true;
    *end = s;
  }
  if (s >= n || T.n_pos == 0) return found;
  uint64_t S[NQ];
  {
    const uint64_t* fr = T.first + ctx * NQ;
    const uint64_t* cr = T.cls + cur * NQ;
#pragma unroll
    for (int q = 0; q < NQ; q++) S[q] = fr[q] & cr[q];
  }
  uint64_t p = s + 1;
  for (;;) {
    uint64_t alive = 0;
#pragma unroll
    for (int q = 0; q < NQ; q++) alive |= S[q];
    if (!alive) break;
    {
      // The boundaries at which nothing but a step can happen -- inside the text, below the walk limit, not a
      // poll boundary, bytes at hand -- in a loop with ONE exit: the general iteration below spends most of its
      // ~150 instructions on the masks of its five exits, and a lone wave pays every instruction in full.
      const uint64_t k = p - s;
      uint64_t m = (k & 255u) != 0 ? 256u - (k & 255u) : 0u;
      if (k >= T.max_walk) m = 0;
      else if (T.max_walk - k < m) m = T.max_walk - k;
      const uint8_t* ptr = nullptr;
      const uint32_t at_hand = m != 0 ? t.span(p, &ptr) : 0u;  // (never beyond the end of the text)
      if (at_hand < m) m = at_hand;
      if (m != 0) {
        // The byte two steps ahead and the class row one step ahead are fetched while the current step runs:
        // neither depends on the automaton state, and read on demand they were two LDS round trips in every
        // step's dependent chain.  (Indices are clamped to the span: the extra reads repeat its last byte.)
        uint32_t i = 0, hit = ~0u, prev = cur;
        uint64_t live;
        const uint32_t last_i = static_cast<uint32_t>(m) - 1u;
        uint32_t c = ptr[0], c1 = ptr[last_i < 1u ? last_i : 1u];
        uint64_t row[NQ];
        {
          const uint64_t* cr = T.cls + c * NQ;
#pragma unroll
          for (int q = 0; q < NQ; q++) row[q] = cr[q];
        }
        do {
          const uint32_t c2 = ptr[i + 2u < last_i ? i + 2u : last_i];
          uint64_t row1[NQ];
          {
            const uint64_t* cr = T.cls + c1 * NQ;
#pragma unroll
            for (int q = 0; q < NQ; q++) row1[q] = cr[q];
          }
          const int cx = ctxed ? lw_ctx(false, prev, false, c) : 0;
          const uint64_t* lr = T.last + cx * NQ;
          uint64_t acc = 0;
#pragma unroll
          for (int q = 0; q < NQ; q++) acc |= S[q] & lr[q];
          hit = acc != 0 ? i : hit;
          uint64_t N[NQ];
          lw_follow<NQ>(T, S, cx, N);
          live = 0;
#pragma unroll
          for (int q = 0; q < NQ; q++) {
            S[q] = N[q] & row[q];
            live |= S[q];
            row[q] = row1[q];
          }
          prev = c;
          c = c1;
          c1 = c2;
          i++;
        } while (i < m && live != 0);
        if (hit != ~0u) {
          found = true;
          *end = p + hit;
        }
        cur = prev;
        p += i;
        continue;
      }
    }
    const uint32_t prev = cur;
    cur = p < n ? t[p] : 0u;
    ctx = ctxed ? lw_ctx(false, prev, p == n, cur) : 0;
    {
      const uint64_t* lr = T.last + ctx * NQ;
      uint64_t acc = 0;
#pragma unroll
      for (int q = 0; q < NQ; q++) acc |= S[q] & lr[q];
      if (acc) {
        found = true;
        *end = p;
      }
    }
    if (p == n) break;
    if (p - s >= T.max_walk) {
      *overrun = true;
      break;
    }
    if (abort != nullptr && ((p - s) & 255u) == 0 && *abort != 0) break;  // the run is void already
#if defined(__HIP_DEVICE_COMPILE__)
    if (abort != nullptr && p - s == kLongWalk && T.max_walk > kLongWalk) {  // (rj_lane_longest: the long-walk budget)
      unsigned long long* long_walks = const_cast<unsigned long long*>(abort) + kLongWalksAfterOverrun;
      if (atomicAdd(long_walks, 1ull) + 1 >= kLongWalkBudget) {
        *overrun = true;
        break;
      }
    }
#endif
    uint64_t N[NQ];
    lw_follow<NQ>(T, S, ctx, N);
    const uint64_t* cr = T.cls + cur * NQ;
#pragma unroll
    for (int q = 0; q < NQ; q++) S[q] = N[q] & cr[q];
    p++;
  }
  return found;
}

// = rj_reaches_accept (behind_walk.h): a thread that has consumed text[p] at forward position q -- does it
// reach an accepting boundary?
template <int NQ, bool CTX, class Text>
RJ_HD bool lw_reaches_accept(const WalkTab<NQ>& T, const Text& t, uint64_t n, uint64_t p, int q, bool* overrun,
                             const volatile unsigned long long* abort = nullptr) {
  constexpr bool ctxed = CTX;
  uint64_t S[NQ];
#pragma unroll
  for (int k = 0; k < NQ; k++) S[k] = 0;
  S[q >> 6] = 1ull << (q & 63);
  uint32_t cur = t[p];
  uint64_t at = p + 1;
  for (;;) {  // S = positions that have consumed text[at - 1]
    {
      // (the plain steps in a tight loop, see lw_longest)
      const uint64_t k = at - p;
      uint64_t m = (k & 255u) != 0 ? 256u - (k & 255u) : 0u;
      if (k >= T.max_walk) m = 0;
      else if (T.max_walk - k < m) m = T.max_walk - k;
      const uint8_t* ptr = nullptr;
      const uint32_t at_hand = m != 0 ? t.span(at, &ptr) : 0u;
      if (at_hand < m) m = at_hand;
      if (m != 0) {
        uint32_t i = 0, prev = cur;
        int state = 0;  // 1: accepted, 2: dead
        do {
          const uint32_t c = ptr[i];
          const int cx = ctxed ? lw_ctx(false, prev, false, c) : 0;
          const uint64_t* lr = T.last + cx * NQ;
          uint64_t acc = 0, live = 0;
#pragma unroll
          for (int j = 0; j < NQ; j++) acc |= S[j] & lr[j];
          uint64_t N[NQ];
          lw_follow<NQ>(T, S, cx, N);
          const uint64_t* cr = T.cls + c * NQ;
#pragma unroll
          for (int j = 0; j < NQ; j++) {
            S[j] = N[j] & cr[j];
            live |= S[j];
          }
          state = acc != 0 ? 1 : live == 0 ? 2 : 0;
          prev = c;
          i++;
        } while (i < m && state == 0);
        if (state == 1) return true;
        if (state == 2) return false;
        cur = prev;
        at += i;
        continue;
      }
    }
    const uint32_t prev = cur;
    cur = at < n ? t[at] : 0u;
    const int ctx = ctxed ? lw_ctx(false, prev, at == n, cur) : 0;
    const uint64_t* lr = T.last + ctx * NQ;
    uint64_t acc = 0, alive = 0;
#pragma unroll
    for (int k = 0; k < NQ; k++) acc |= S[k] & lr[k];
    if (acc) return true;
    if (at == n) return false;
    if (at - p >= T.max_walk) {
      *overrun = true;
      return false;
    }
    if (abort != nullptr && ((at - p) & 255u) == 0 && *abort != 0) return false;
    uint64_t N[NQ];
    lw_follow<NQ>(T, S, ctx, N);
    const uint64_t* cr = T.cls + cur * NQ;
#pragma unroll
    for (int k = 0; k < NQ; k++) {
      S[k] = N[k] & cr[k];
      alive |= S[k];
    }
    if (!alive) return false;
    at++;
  }
}

// = rj_leftmost_start (behind_walk.h): S = reverse-automaton positions that have consumed text[p]; the
// left-most boundary at which one of their threads can begin a match
template <int NQ, bool CTX, class Text>
RJ_HD bool lw_leftmost_start(const WalkTab<NQ>& R, const Text& t, uint64_t n, uint64_t p, uint64_t (&S)[NQ], uint32_t max_walk,
                             uint64_t* start, bool* overrun, const volatile unsigned long long* abort = nullptr) {
  constexpr bool ctxed = CTX;
  bool found = false;
  uint32_t cur = t[p];  // text[at]
  uint64_t at = p;
  for (;;) {  // S = reverse positions that have consumed text[at]
    {
      // (the plain steps in a tight loop, see lw_longest: boundaries at, at - 1, ... that are >= 1, below the walk
      // limit, not a poll boundary, their byte before at hand)
      const uint64_t k = p - at;
      uint64_t m = (k & 255u) != 0 ? 256u - (k & 255u) : 0u;
      if (k >= max_walk) m = 0;
      else if (max_walk - k < m) m = max_walk - k;
      const uint8_t* ptr = nullptr;
      const uint32_t at_hand = m != 0 ? t.span_back(at, &ptr) : 0u;  // text[at - 1], text[at - 2], ... = ptr[0], ptr[-1], ...
      if (at_hand < m) m = at_hand;
      if (m != 0) {
        // (bytes two steps ahead, class rows one step ahead: see lw_longest)
        uint32_t i = 0, hit = ~0u;
        uint64_t live;
        const uint32_t last_i = static_cast<uint32_t>(m) - 1u;
        uint32_t before = *ptr, before1 = *(ptr - (last_i < 1u ? last_i : 1u));
        uint64_t row[NQ];
        {
          const uint64_t* cr = R.cls + before * NQ;
#pragma unroll
          for (int j = 0; j < NQ; j++) row[j] = cr[j];
        }
        do {
          const uint32_t before2 = *(ptr - (i + 2u < last_i ? i + 2u : last_i));
          uint64_t row1[NQ];
          {
            const uint64_t* cr = R.cls + before1 * NQ;
#pragma unroll
            for (int j = 0; j < NQ; j++) row1[j] = cr[j];
          }
          const int cx = ctxed ? lw_ctx(false, before, false, cur) : 0;
          const uint64_t* lr = R.last + cx * NQ;
          uint64_t acc = 0;
#pragma unroll
          for (int j = 0; j < NQ; j++) acc |= S[j] & lr[j];
          hit = acc != 0 ? i : hit;
          uint64_t N[NQ];
          lw_follow<NQ>(R, S, cx, N);
          live = 0;
#pragma unroll
          for (int j = 0; j < NQ; j++) {
            S[j] = N[j] & row[j];
            live |= S[j];
            row[j] = row1[j];
          }
          cur = before;
          before = before1;
          before1 = before2;
          i++;
        } while (i < m && live != 0);
        if (hit != ~0u) {
          found = true;
          *start = at - hit;
        }
        if (live == 0) break;
        at -= i;
        continue;
      }
    }
    const uint32_t before = at > 0 ? t[at - 1] : 0u;
    const int ctx = ctxed ? lw_ctx(at == 0, before, at == n, cur) : 0;
    const uint64_t* lr = R.last + ctx * NQ;
    uint64_t acc = 0, alive = 0;
#pragma unroll
    for (int k = 0; k < NQ; k++) acc |= S[k] & lr[k];
    if (acc) {
      found = true;
      *start = at;
    }
    if (at == 0) break;
    if (p - at >= max_walk) {
      *overrun = true;
      break;
    }
    if (abort != nullptr && ((p - at) & 255u) == 0 && *abort != 0) break;
    uint64_t N[NQ];
    lw_follow<NQ>(R, S, ctx, N);
    const uint64_t* cr = R.cls + before * NQ;
#pragma unroll
    for (int k = 0; k < NQ; k++) {
      S[k] = N[k] & cr[k];
      alive |= S[k];
    }
    if (!alive) break;
    cur = before;
    at--;
  }
  return found;
}

// does window k of P occur at text position w?  (= rj_window_at, behind_walk.h)
template <class Text>
RJ_HD bool lw_window_at(const DevProgram& P, int k, const Text& t, uint64_t n, uint64_t w) {
  if (w + P.win_len > n) return false;
  uint32_t v0 = 0, v1 = 0;
  for (uint32_t i = 0; i < P.win_len; i++) {
    const uint32_t c = t[w + i];
    if (i < 4) v0 |= c << (8 * i);
    else v1 |= c << (8 * (i - 4));
  }
  return (v0 & P.win_mask0[k]) == P.win_value0[k] && (v1 & P.win_mask1[k]) == P.win_value1[k];
}

// = rj_behind_candidate (behind_walk.h): the candidate of the hit at w.  P supplies the window constants and
// the cut sets (kernel arguments: scalar registers), F / R the forward / reverse tables.
template <int NQ, bool CTX, class Text>
RJ_HD bool lw_behind_candidate(const DevProgram& P, const WalkTab<NQ>& F, const WalkTab<NQ>& R, const Text& t, uint64_t n, uint64_t w,
                               uint64_t* begin, uint64_t* end, bool* overrun, const volatile unsigned long long* abort = nullptr) {
  uint64_t cut[NQ], ok[NQ];
#pragma unroll
  for (int j = 0; j < NQ; j++) cut[j] = ok[j] = 0;
  // (constant indices into P: see behind_walk.h -- a run-time index puts the descriptor into scratch memory)
#pragma unroll
  for (int k = 0; k < kDevMaxWindows; k++) {
    if (k >= P.n_windows || !lw_window_at(P, k, t, n, w)) continue;
#pragma unroll
    for (int j = 0; j < NQ; j++) cut[j] |= static_cast<uint64_t>(P.cut_fwd[k][2 * j]) | (static_cast<uint64_t>(P.cut_fwd[k][2 * j + 1]) << 32);
  }
  RJ_STAMP(4);
  bool any = false;
#pragma unroll
  for (int j = 0; j < NQ; j++) {
    uint64_t bits = cut[j];
    while (bits) {
      const int b = __builtin_ctzll(bits);
      bits &= bits - 1;
      const int q = j * 64 + b;
      if (lw_reaches_accept<NQ, CTX>(F, t, n, w, q, overrun, abort)) {
        const int r = F.n_pos - 1 - q;
        ok[r >> 6] |= 1ull << (r & 63);
        any = true;
      }
    }
  }
  if (!any) return false;
  RJ_STAMP(5);
  uint64_t s = 0;
  if (*overrun || !lw_leftmost_start<NQ, CTX>(R, t, n, w, ok, F.max_walk, &s, overrun, abort) || *overrun) return false;
  RJ_STAMP(6);
  *begin = s;
  const bool found = lw_longest<NQ, CTX>(F, t, n, s, end, overrun, abort);
  RJ_STAMP(7);
  return found;
}

}  // namespace rejit_amd
#endif
