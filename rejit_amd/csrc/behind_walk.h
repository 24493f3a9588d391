// rejit_amd/csrc/behind_walk.h -- the per-hit procedure of the "windows behind an unbounded prefix"
// mode (lowering.h: Program::behind), shared by the HIP kernel verify_behind_in_regions and by the CPU
// unit tests (tests/support/carry_exec.cc).
//
// The reference fast-forwards on any literal of the pattern and, when that literal is not at the
// start of the match, runs its NFA BACKWARDS from the hit to find the start, then forwards to find the
// end (FF_finder, src/codegen.cc:352-383; GenerateMatchBackward, src/x64/codegen-x64.cc:643-650; the
// dispatch loop :166-189).  Here a hit of window k at text position w means: some thread may sit at one
// of the positions cut_fwd[k] having consumed text[w].  Per hit:
//   1. forward from every such position on its own: does it reach an accepting boundary at all?
//      (a position that does not would give starts that have no match through this hit);
//   2. the REVERSE automaton from the surviving positions, backwards from w: the LEFT-MOST text
//      position at which one of its threads can begin a match (rev.last = forward first);
//   3. the ordinary forward walk from that start: the longest end.
// One candidate (start, end) per hit.  A match contains a hit (the windows' literals are a cut of the
// NFA graph), so the left-most-longest selection over these candidates is the reference's result --
// except when a candidate that an earlier match hides ends AFTER that match (its hit may still serve a
// later start): the selection flags that conflict and the engine repeats the run in dense mode.
#ifndef REJIT_AMD_BEHIND_WALK_H_
#define REJIT_AMD_BEHIND_WALK_H_

#include <stdint.h>

#include "carry_scan.h"
#include "device_program.h"

#ifndef RJ_STAMP  // (kernels.hip, -DRJ_TRACE_VERIFY: phase time stamps of a debug build)
#define RJ_STAMP(i) ((void)0)
#endif

namespace rejit_amd {

// does window k of P (<= 8 bytes, per-byte masks) occur at text position w?
template <class Text>
RJ_HD bool rj_window_at(const DevProgram& P, int k, const Text& t, uint64_t n, uint64_t w) {
  if (w + P.win_len > n) return false;
  uint32_t v0 = 0, v1 = 0;
  for (uint32_t i = 0; i < P.win_len; i++) {
    const uint32_t c = t[w + i];
    if (i < 4) v0 |= c << (8 * i);
    else v1 |= c << (8 * (i - 4));
  }
  return (v0 & P.win_mask0[k]) == P.win_value0[k] && (v1 & P.win_mask1[k]) == P.win_value1[k];
}

// step 1: a thread that has consumed text[p] at forward position q -- does it reach an accepting boundary?
// (`abort`, may be null: polled every 256 steps -- once any walk of the run has hit the limit the run is
// void and the others need not finish; a step is a chain of dependent loads, ~1 us)
template <int NW, class Text>
RJ_HD bool rj_reaches_accept(const DevProgram& P, const Text& t, uint64_t n, uint64_t p, int q, bool* overrun,
                             const volatile unsigned long long* abort = nullptr) {
  const int W = P.n_words;
  uint32_t S[NW], T[NW];
#pragma unroll
  for (int k = 0; k < NW; k++) S[k] = 0;
  S[q >> 5] = 1u << (q & 31);
  for (uint64_t at = p + 1;; at++) {  // S = positions that have consumed text[at - 1]
    const int ctx = cs_context(P, t, n, at);
    const uint32_t* lr = P.last + static_cast<size_t>(ctx) * W;
    uint32_t acc = 0, alive = 0;
#pragma unroll
    for (int k = 0; k < NW; k++)
      if (k < W) acc |= S[k] & lr[k];
    if (acc) return true;
    if (at == n) return false;
    if (at - p >= P.max_walk) {
      *overrun = true;
      return false;
    }
    if (abort != nullptr && ((at - p) & 255u) == 0 && *abort != 0) return false;
    cs_follow<NW>(P, S, ctx, T);
    const uint32_t* cr = P.cls + static_cast<size_t>(t[at]) * W;
#pragma unroll
    for (int k = 0; k < NW; k++) {
      S[k] = k < W ? (T[k] & cr[k]) : 0u;
      alive |= S[k];
    }
    if (!alive) return false;
  }
}

// step 2: S = reverse-automaton positions that have consumed text[p]; the left-most boundary at which one
// of their threads can begin a match.  false: there is none.
template <int NW, class Text>
RJ_HD bool rj_leftmost_start(const DevProgram& R, const Text& t, uint64_t n, uint64_t p, uint32_t (&S)[NW], uint32_t max_walk,
                             uint64_t* start, bool* overrun, const volatile unsigned long long* abort = nullptr) {
  const int W = R.n_words;
  uint32_t T[NW];
  bool found = false;
  for (uint64_t at = p;; at--) {  // S = reverse positions that have consumed text[at]
    const int ctx = cs_context(R, t, n, at);
    const uint32_t* lr = R.last + static_cast<size_t>(ctx) * W;
    uint32_t acc = 0, alive = 0;
#pragma unroll
    for (int k = 0; k < NW; k++)
      if (k < W) acc |= S[k] & lr[k];
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
    cs_follow<NW>(R, S, ctx, T);
    const uint32_t* cr = R.cls + static_cast<size_t>(t[at - 1]) * W;
#pragma unroll
    for (int k = 0; k < NW; k++) {
      S[k] = k < W ? (T[k] & cr[k]) : 0u;
      alive |= S[k];
    }
    if (!alive) break;
  }
  return found;
}

// the candidate of the hit at w: true and (*begin, *end) when there is one.  NQ = 64-bit state words of
// the forward walk (rj_lane_longest), NW = 32-bit words: NQ = (NW + 1) / 2.
template <int NW, int NQ>
RJ_HD bool rj_behind_candidate(const DevProgram& P, const DevProgram& R, const uint8_t* text, uint64_t n, uint64_t w, uint64_t* begin,
                               uint64_t* end, bool* overrun, const volatile unsigned long long* abort = nullptr) {
  // (the three walks read the text around the hit through 16 bytes in registers, device_program.h)
  const RjCachedText t(text, n);
  const int W = P.n_words;
  uint32_t ok[NW];
#pragma unroll
  for (int k = 0; k < NW; k++) ok[k] = 0;
  bool any = false;
  // The positions that may have consumed text[w]: those of every window that occurs at w.  (Constant
  // indices into P -- the loop is unrolled over the maximum count: an index that is only known at run time
  // makes the compiler keep the whole descriptor, two of them here, in scratch memory, which every lane of
  // every workgroup then fills: 800 MB of stores per launch, the kernel's whole 180 us.)
  uint32_t cut[NW];
#pragma unroll
  for (int j = 0; j < NW; j++) cut[j] = 0;
#pragma unroll
  for (int k = 0; k < kDevMaxWindows; k++) {
    if (k >= P.n_windows || !rj_window_at(P, k, t, n, w)) continue;
#pragma unroll
    for (int j = 0; j < NW; j++)
      if (j < W) cut[j] |= P.cut_fwd[k][j];
  }
  if (cut[0] != 0) RJ_STAMP(4);
#pragma unroll
  for (int j = 0; j < NW; j++) {
    uint32_t bits = cut[j];
    while (bits) {
      const int b = __builtin_ctz(bits);
      bits &= bits - 1;
      const int q = j * 32 + b;
      if (rj_reaches_accept<NW>(P, t, n, w, q, overrun, abort)) {
        const int r = P.n_pos - 1 - q;
        ok[r >> 5] |= 1u << (r & 31);
        any = true;
      }
    }
  }
  if (!any) return false;
  RJ_STAMP(5);
  uint64_t s = 0;
  if (*overrun || !rj_leftmost_start<NW>(R, t, n, w, ok, P.max_walk, &s, overrun, abort) || *overrun) return false;
  RJ_STAMP(6);
  *begin = s;
  const bool found = rj_lane_longest<NQ>(P, t, n, s, end, overrun, abort);
  RJ_STAMP(7);
  return found;
}

}  // namespace rejit_amd
#endif
