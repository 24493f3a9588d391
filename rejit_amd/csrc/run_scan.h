// rejit_amd/csrc/run_scan.h -- MatchAll for patterns whose match is ONE long-lived thread in ONE loop position: `X+`
// (`[acgt]+`, `[^>]+`), `A L*` (`a[bc]*`), `A L* B` (`a.*b`, `<[^>]*>`), `X+ B` (`[a-f]+[0-9]`) -- in time linear in the text
// however long the loop runs (round 6; kernels: run_scan.hip).
//
// The reference walks such a text with one thread in the loop's state, one byte per iteration of its NFA loop
// (src/x64/codegen-x64.cc:535-581), keeps the last accepting position (:426-461) and, when the thread dies, restarts behind
// the match (:487-503).  Here the same answer comes from the CLASS STREAMS of the three positions, segment by segment:
//   * a BREAK is a byte outside L (for `a.*b`: a line break; for `[acgt]+`: any other byte); the text end counts as one;
//   * the starts of a segment are the positions [r', r) between two consecutive breaks r' < r (a start may sit ON the
//     break that opens its segment: A is consumed first, L* begins behind it);
//   * the left-most match of a segment begins at its first A position s1, and the longest match from s1 ends behind the
//     LAST B position q in (s1, r] (without B: at r) -- the thread lives until the break whatever it meets;
//   * behind that match nothing else of the segment can match: a later start s has no B in (s, r] left (without B: lies
//     inside the match).  So every segment holds at most ONE match, (s1, q + 1) or (s1, r), and segments do not interact --
//     provided no byte is in A, in B and a break at once (then `B` of one match could be `A` of the next; such patterns
//     keep the general paths).
// What crosses a tile boundary is therefore a pair (s1, q) of the segment that is still open -- the "pending thread" --,
// which tiles without a break hand on and tiles with a break replace: an associative scan over tile summaries
// (run_scan.hip: run_summary -> run_resolve -> run_emit; the text is read twice, nothing else is).
// Before this the paths for such texts were the scalar walk of dense_streams (void beyond max_walk), scan_dense_walk
// (quadratic in the run length) and the carry scan of linear.hip: six phases, 12-16 GB/s, 70-140 x the text in FETCH_SIZE
// (profiles/r05_pmc_hbm_traffic.txt).
#ifndef REJIT_AMD_RUN_SCAN_H_
#define REJIT_AMD_RUN_SCAN_H_

#include <stdint.h>

#include "dense_streams.h"
#include "device_program.h"

namespace rejit_amd {

constexpr int kRunMaxRanges = 8;

// A class is the union of some of the plan's byte ranges, or the complement of that union (whichever takes fewer ranges:
// `.` is "not \n, not \r").  Ranges as in StreamPlan (dense_streams.h: rj_stream_range).
struct RunPlan {
  uint32_t ok;                  // 0: the pattern does not have the shape
  uint32_t has_b;               // a B position exists (the match ends behind the last B; else at the break)
  uint32_t n_ranges;
  uint32_t add_lo[kRunMaxRanges], add_hi[kRunMaxRanges];
  uint32_t high_half;           // bit r: the range lies in 0x80..0xff
  uint32_t a_ranges, l_ranges, b_ranges;   // bit r: the class (or its complement) holds range r
  uint32_t a_neg, l_neg, b_neg;            // 1: the class is the complement of its ranges
  uint32_t lag;                 // 1: `A L+` / `A L+ B` -- the kernels' start stream is "A at p - 1 and L at p" (below), a match begins one byte before its start
  uint32_t bol, eol;            // `^...` / `...$` around a shape: a start counts at a line start only (a mask on the start stream) / a match when its
                                // closing break is a line end (below; run_scan.hip: bol_stream, the SNL bit at the break)
  uint32_t pair;                // the PAIR shape (`"[^"]*"`; below): ok stays 0, the pair kernels of run_scan.hip take the pattern
};

}  // namespace rejit_amd

// ----------------------------------------------------------------------------------------------- host part
#include <vector>

#include "lowering.h"

namespace rejit_amd {

namespace run_detail {

// the byte ranges of `set` within the two halves of the byte values
inline std::vector<std::pair<int, int>> ranges_of(const bool (&set)[256]) {
  std::vector<std::pair<int, int>> out;
  for (int half = 0; half < 2; half++) {
    int b = 0;
    while (b < 128) {
      while (b < 128 && !set[half * 128 + b]) b++;
      if (b >= 128) break;
      int e = b;
      while (e + 1 < 128 && set[half * 128 + e + 1]) e++;
      out.emplace_back(half * 128 + b, half * 128 + e);
      b = e + 1;
    }
  }
  return out;
}

inline bool add_class(RunPlan* pl, const bool (&set)[256], uint32_t* mask, uint32_t* neg) {
  bool comp[256];
  for (int c = 0; c < 256; c++) comp[c] = !set[c];
  const auto direct = ranges_of(set), inverse = ranges_of(comp);
  const bool use_neg = inverse.size() < direct.size();
  *neg = use_neg ? 1u : 0u;
  *mask = 0;
  for (const auto& r : use_neg ? inverse : direct) {
    const int half = r.first >> 7, b = r.first & 127, e = r.second & 127;
    const uint32_t lo = static_cast<uint32_t>(0x80 - b) * 0x01010101u, hi = static_cast<uint32_t>(0x7f - e) * 0x01010101u;
    uint32_t k = 0;
    while (k < pl->n_ranges && !(pl->add_lo[k] == lo && pl->add_hi[k] == hi && ((pl->high_half >> k) & 1u) == static_cast<uint32_t>(half))) k++;
    if (k == pl->n_ranges) {
      if (pl->n_ranges >= static_cast<uint32_t>(kRunMaxRanges)) return false;
      pl->add_lo[k] = lo;
      pl->add_hi[k] = hi;
      pl->high_half |= static_cast<uint32_t>(half) << k;
      pl->n_ranges++;
    }
    *mask |= 1u << k;
  }
  return true;
}

}  // namespace run_detail

// The position automaton's shapes (positions in pattern order; follow sets from Program::linear / rows):
//   X+      1 position   first {0}  last {0}    0 -> {0}
//   A L*    2 positions  first {0}  last {0,1}  0 -> {1}    1 -> {1}
//   X+ B    2 positions  first {0}  last {1}    0 -> {0,1}  1 -> {}
//   A L* B  3 positions  first {0}  last {2}    0 -> {1,2}  1 -> {1,2}  2 -> {}
//   A L+    2 positions  first {0}  last {1}    0 -> {1}    1 -> {1}              (lag: see below)
//   A L+ B  3 positions  first {0}  last {2}    0 -> {1}    1 -> {1,2}  2 -> {}   (lag)
inline RunPlan make_run_plan(const Program& P) {
  RunPlan pl{};
  if (P.n_pos < 1 || P.n_pos > 3 || P.n_words != 1 || P.any_nullable) return pl;
  // (a pattern at risk of the reference's ring artefact -- Program::q8_risk -- keeps the paths that test for it, with one exception below:
  // `X+` between `^` / `$`)
  if (P.q8_risk && !(P.has_assertions && P.n_pos == 1)) return pl;
  const uint32_t all = (1u << P.n_pos) - 1u;
  // `^` in front / `$` behind (contexts: lowering.h -- bit 0 = the boundary is a line start, bit 1 = a line end): the first set
  // exists only at a line start / the last set only at a line end, nothing else depends on the context.  Then the matches are
  // those of the pattern WITHOUT the assertions that begin at a line start / end at a line end, provided no line break is in A
  // or L (and, for `$`, there is no B position): only a segment's FIRST A can follow a line break, and a match without B ends
  // at its segment's break -- the filter cannot turn the selection of another start into a match.
  uint32_t bol = 0, eol = 0;
  const int cx = P.has_assertions ? 3 : 0;
  if (P.has_assertions) {
    auto f = [&](int c) { return P.first[c][0] & all; };
    auto la = [&](int c) { return P.last[c][0] & all; };
    if (f(1) != f(3) || f(0) != f(2) || la(2) != la(3) || la(0) != la(1)) return pl;
    if (f(0) != f(3)) {
      if (f(0) != 0) return pl;
      bol = 1;
    }
    if (la(0) != la(3)) {
      if (la(0) != 0) return pl;
      eol = 1;
    }
    for (int c = 0; c < 3; c++)
      if (P.rows[c] != P.rows[3]) return pl;
    if (!bol && !eol) return pl;
  }
  const uint32_t first = P.first[cx][0] & all, last = P.last[cx][0] & all;
  uint32_t F[3] = {0, 0, 0};
  for (int k = 0; k < P.n_pos; k++) {
    const int r = P.row_of[static_cast<size_t>(k)];
    F[k] = (r < 0 ? (1u << (k + 1)) : P.rows[cx][static_cast<size_t>(r)]) & all;   // (no row: the follow set is {k + 1})
  }
  int a = -1, l = -1, b = -1;
  uint32_t lag = 0;
  if (P.n_pos == 1 && first == 1 && last == 1 && F[0] == 1) {
    a = l = 0;
  } else if (P.n_pos == 2 && first == 1 && last == 3 && F[0] == 2 && F[1] == 2) {
    a = 0;
    l = 1;
  } else if (P.n_pos == 2 && first == 1 && last == 2 && F[0] == 3 && F[1] == 0) {
    a = l = 0;
    b = 1;
  } else if (P.n_pos == 3 && first == 1 && last == 4 && F[0] == 6 && F[1] == 6 && F[2] == 0) {
    a = 0;
    l = 1;
    b = 2;
  } else if (P.n_pos == 2 && first == 1 && last == 2 && F[0] == 2 && F[1] == 2) {
    // `A L+` (`[A-Z][a-z]+`, `#.+`, `@[a-z]+`): `A L*` whose loop runs at least once.  The kernels take as START stream the marks
    // "A at p - 1 and L at p" (one shift of the A stream, the carry from the lane below / the iteration before): the left-most
    // mark of a segment sits one byte behind the left-most A that has an L byte behind it, the thread lives until the break as
    // before, and with a B position "the last B BEHIND the start" is exactly "at least one L byte between A and B".  A match
    // begins one byte before its mark (lag).
    a = 0;
    l = 1;
    lag = 1;
  } else if (P.n_pos == 3 && first == 1 && last == 4 && F[0] == 2 && F[1] == 6 && F[2] == 0) {
    a = 0;     // `A L+ B` (`a.+b`, `<[^>]+>`)
    l = 1;
    b = 2;
    lag = 1;
  } else {
    return pl;
  }
  bool A[256], L[256], B[256], clash = false;
  for (int c = 0; c < 256; c++) {
    const uint32_t cls = P.cls[static_cast<size_t>(c)];
    A[c] = ((cls >> a) & 1u) != 0;
    L[c] = ((cls >> l) & 1u) != 0;
    B[c] = b >= 0 && ((cls >> b) & 1u) != 0;
    if (A[c] && B[c] && !L[c]) clash = true;   // (the B of one match could be the A of the next)
  }
  if (bol || eol) {
    if ((eol && b >= 0) || A['\n'] || A['\r'] || L['\n'] || L['\r']) return pl;
  }
  // `^X+`, `X+$`, `^X+$` (`[a-z]+$`, ` +$`: at risk of the ring artefact by the static analysis).  The artefact needs a candidate that begins
  // exactly where another one ends (engine.hip: detect_adjacent): a candidate of `X+$` ends at a line break, which is no X
  // byte and begins nothing; a candidate of `^X+` begins behind a line break, and what ends there would have to end ON that line break's
  // successor -- behind an X byte, not behind a line break.  With no line break in X the condition cannot arise: the documented
  // semantics are the reference's (checked against the oracle, which restates the artefact: tests/test_run_plan.py).
  if (P.q8_risk && !((bol || eol) && a == l && b < 0)) return pl;
  if (clash) {
    // The PAIR shape: `Q L* Q` with the same class Q at both ends and no byte of Q inside L -- `"[^"]*"`, `%[a-z]*%`, `"[^"<LF>]*"` (the
    // line-break byte itself: this dialect has no escapes inside brackets).
    // Every Q byte is a break, and whether it OPENS a match depends on the match before it: a Q closes the match its predecessor
    // opened, the next Q opens again; a break that is no Q (a RESET: `\n` for `"[^"\n]*"`, the text's end) drops an open Q.  So
    // the matches are the pairs (1st, 2nd), (3rd, 4th) ... of the Q bytes since the last reset: a parity per segment, carried
    // from tile to tile as a function on one bit (run_scan.hip: pair_summary / pair_resolve / pair_emit).
    if (b < 0 || a == l || lag || bol || eol) return pl;
    for (int c = 0; c < 256; c++)
      if (A[c] != B[c] || (A[c] && L[c])) return pl;
    if (!run_detail::add_class(&pl, A, &pl.a_ranges, &pl.a_neg) || !run_detail::add_class(&pl, L, &pl.l_ranges, &pl.l_neg)) return RunPlan{};
    pl.b_ranges = pl.a_ranges;
    pl.b_neg = pl.a_neg;
    pl.has_b = 1;
    pl.pair = 1;
    return pl;   // (ok == 0: the run kernels' segment rule does not hold)
  }
  if (!run_detail::add_class(&pl, A, &pl.a_ranges, &pl.a_neg) || !run_detail::add_class(&pl, L, &pl.l_ranges, &pl.l_neg)) return RunPlan{};
  if (b >= 0 && !run_detail::add_class(&pl, B, &pl.b_ranges, &pl.b_neg)) return RunPlan{};
  pl.has_b = b >= 0 ? 1u : 0u;
  pl.lag = lag;
  pl.bol = bol;
  pl.eol = eol;
  pl.ok = 1;
  return pl;
}

}  // namespace rejit_amd
#endif
