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
// Blob layout (uint64 words):  first [C][NQ]  last [C][NQ]  linear [NQ]  rows [C][n_pos + 1][NQ]  cls [256][NQ]
#ifndef REJIT_AMD_LDS_WALK_H_
#define REJIT_AMD_LDS_WALK_H_

#include <stdint.h>

#include "device_program.h"

#ifndef RJ_STAMP  // (kernels.hip, -DRJ_TRACE_VERIFY: phase time stamps of a debug build)
#define RJ_STAMP(i) ((void)0)
#endif

namespace rejit_amd {

RJ_HD uint32_t lw_blob_words(int nq, int n_ctx, int n_pos) {
  return static_cast<uint32_t>(nq) * static_cast<uint32_t>(2 * n_ctx + 1 + n_ctx * (n_pos + 1) + 256);
}

template <int NQ>
struct WalkTab {
  const uint64_t* first;
  const uint64_t* last;
  const uint64_t* rows;
  const uint64_t* cls;
  uint64_t linear[NQ];
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
  for (int q = 0; q < NQ; q++) T.linear[q] = lin[q];
  T.rows = lin + NQ;
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
    const uint64_t x = S[q] & T.linear[q];
    out[q] = (x << 1) | carry;
    carry = x >> 63;
  }
  const uint64_t* rows = T.rows + static_cast<uint32_t>(ctx * (T.n_pos + 1)) * NQ;
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    uint64_t sp = S[q] & ~T.linear[q];
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
  for (uint64_t at = p + 1;; at++) {  // S = positions that have consumed text[at - 1]
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
  for (uint64_t at = p;; at--) {  // S = reverse positions that have consumed text[at]
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
