// rejit_amd/csrc/exact_replay.h -- bit-exactness with the reference's ring artefact ("Q8", DESIGN.md
// section 6) on texts of any size: the reference's own no-fast-forward loop (GenerateMatchDirection,
// reference src/x64/codegen-x64.cc:535-640; SetState :951-987; CheckMatch + ClearStates :401-466,
// :1075-1097; the sink MatchAllAppendFilter, src/codegen.cc:36-86) replayed segment by segment, one
// lane per segment, all segments in parallel.
//
// What makes that possible: a SYNCHRONISATION POINT is a text position p at which no thread of the
// reference's ring is alive before the thread of p is seeded.  There the loop's state is the initial one
// (every pending match has been emitted, no match reaches p, no later match can begin before p), so the
// text between two synchronisation points is an independent run of the loop and the concatenation of the
// segments' outputs is the reference's output.  Synchronisation points are found with the position
// automaton the rest of the engine uses: A(p) = the positions some start s < p has alive after consuming
// text[p-1], i.e. A(p+1) = (follow(A(p)) | first(ctx(p))) & cls[text[p]].  Every thread of the ring was
// seeded at some start and advanced over NFA edges -- the ring only ever LOSES threads to ClearStates --
// so A(p) == 0 implies an empty ring: every such p is a synchronisation point (the converse need not
// hold; missing one only makes a segment longer).
//
// RJ_HD only, no HIP runtime calls: tests/support/exact_exec.cc compiles these bodies with g++.
#ifndef REJIT_AMD_EXACT_REPLAY_H_
#define REJIT_AMD_EXACT_REPLAY_H_

#include <stdint.h>

#include "device_program.h"

namespace rejit_amd {

constexpr uint64_t kNoSync = ~0ull;

template <int NQ>
struct RjAlive {
  uint64_t S[NQ];
  uint64_t lin[NQ];
};

template <int NQ>
RJ_HD void rj_alive_init(const DevProgram& P, RjAlive<NQ>* A, bool full) {
  const int W = P.n_words;
  for (int q = 0; q < NQ; q++) {
    uint64_t l = 0;
    if (2 * q < W) l = P.linear[2 * q];
    if (2 * q + 1 < W) l |= (uint64_t)P.linear[2 * q + 1] << 32;
    A->lin[q] = l;
    // `full`: every position alive -- a superset of whatever really is alive at an arbitrary position
    const int lo = q * 64, hi = lo + 64;
    uint64_t m = 0;
    if (full && P.n_pos > lo) m = P.n_pos >= hi ? ~0ull : ((1ull << (P.n_pos - lo)) - 1);
    A->S[q] = m;
  }
}

template <int NQ>
RJ_HD bool rj_alive_empty(const RjAlive<NQ>& A) {
  uint64_t any = 0;
  for (int q = 0; q < NQ; q++) any |= A.S[q];
  return any == 0;
}

// A(p) -> A(p+1); p < n
template <int NQ>
RJ_HD void rj_alive_step(const DevProgram& P, const uint8_t* t, uint64_t n, uint64_t p, RjAlive<NQ>* A) {
  const int W = P.n_words;
  const int ctx = P.n_ctx > 1 ? rj_context(t, n, p) : 0;
  uint64_t T[NQ];
  uint64_t carry = 0;
  for (int q = 0; q < NQ; q++) {
    const uint64_t x = A->S[q] & A->lin[q];
    T[q] = (x << 1) | carry;
    carry = x >> 63;
  }
  for (int q = 0; q < NQ; q++) {
    uint64_t sp = A->S[q] & ~A->lin[q];
    while (sp) {
      const int b = __builtin_ctzll(sp);
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
  const uint32_t* fr = P.first + ctx * W;
  const uint32_t* cr = P.cls + (uint32_t)t[p] * W;
  for (int q = 0; q < NQ; q++) {
    uint64_t f = 0, c = 0;
    if (2 * q < W) { f = fr[2 * q]; c = cr[2 * q]; }
    if (2 * q + 1 < W) {
      f |= (uint64_t)fr[2 * q + 1] << 32;
      c |= (uint64_t)cr[2 * q + 1] << 32;
    }
    A->S[q] = (T[q] | f) & c;
  }
}

// The first synchronisation point at or after x that A() PROVES (A(p) == 0), or n + 1 when there is none
// up to and including n.  A(x) depends on the text before x: the walk starts `back` bytes earlier with
// every position alive (a superset), and is exact from the first position at which that superset dies
// out; `back` grows until that happens before x (position 0 is exact by definition: nothing is alive at
// the start of the text).  The answer does not depend on `back` -- two callers with different ranges
// agree on it, which is what lets neighbouring shards split the text at these points.
template <int NQ>
RJ_HD uint64_t rj_first_sync(const DevProgram& P, const uint8_t* t, uint64_t n, uint64_t x) {
  if (x == 0) return 0;
  if (x > n) return n + 1;
  for (uint64_t back = 256;; back *= 4) {
    const uint64_t start = x > back ? x - back : 0;
    RjAlive<NQ> A;
    rj_alive_init<NQ>(P, &A, start != 0);
    bool exact = start == 0;
    bool widen = false;
    for (uint64_t p = start;; p++) {
      if (rj_alive_empty<NQ>(A)) {
        exact = true;
        if (p >= x) return p;
      }
      if (p >= x && !exact) {
        widen = true;
        break;
      }
      if (p == n) break;
      rj_alive_step<NQ>(P, t, n, p, &A);
    }
    if (!widen) return n + 1;
  }
}

// First proven synchronisation point in [c0, c1) (c1 <= n + 1), or kNoSync.  `exact_start`: c0 is known to
// be a synchronisation point itself; otherwise the walk starts with every position alive, which can only
// hide synchronisation points near the beginning of the chunk.
// (from: only a point at or after it counts -- the ends of an ownership range, exact_replay.hip: xr_lookup)
template <int NQ>
RJ_HD uint64_t rj_chunk_first_sync(const DevProgram& P, const uint8_t* t, uint64_t n, uint64_t c0, uint64_t c1,
                                   bool exact_start, uint64_t from = 0) {
  RjAlive<NQ> A;
  rj_alive_init<NQ>(P, &A, !exact_start);
  for (uint64_t p = c0; p < c1; p++) {
    if (p >= from && rj_alive_empty<NQ>(A)) return p;
    if (p >= n) break;
    rj_alive_step<NQ>(P, t, n, p, &A);
  }
  return kNoSync;
}

// ---- the reference's loop, in pieces (rj_replay_segment puts them together; rj_replay_raw does so for a part of a segment)
//
// `ring(i)` is slot i of times x states start offsets (int64_t&, -1 = free); `base` = the time row of the current position.

// Position p, first half: seed p's thread, close over the control edges; when the exit state is occupied report the match
// (*pb .. p) and clear the threads that began inside it (CheckMatch + ClearStates, src/x64/codegen-x64.cc:401-466,
// 1075-1097).
template <class Ring>
RJ_HD bool rj_ring_match(const DevGraph& G, const uint8_t* t, uint64_t n, uint64_t p, Ring ring, int base, int64_t* pb) {
  const int S = G.n_states, slots = S * G.times;
  const int t0 = base * S;
  ring(t0 + G.entry) = static_cast<int64_t>(p);
  bool changed = true;
  while (changed) {
    changed = false;
    for (int i = 0; i < G.n_control_edges; i++) {
      const int64_t v = ring(t0 + G.ce_src[i]);
      if (v < 0) continue;
      const int kind = G.ce_kind[i];
      bool ok = true;
      if (kind == 1) ok = p == 0 || rj_line_break(t[p - 1]);
      else if (kind == 2) ok = p == n || rj_line_break(t[p]);
      if (!ok) continue;
      const int d = t0 + G.ce_dst[i];
      const int64_t cur = ring(d);
      if (cur < 0 || v < cur) {
        ring(d) = v;
        changed = true;
      }
    }
  }
  const int64_t xs = ring(t0 + G.exit);
  if (xs < 0) return false;
  const int64_t pe = static_cast<int64_t>(p);
  for (int i = 0; i < slots; i++) {
    const int64_t v = ring(i);
    if (v > xs && v < pe) ring(i) = -1;
  }
  *pb = xs;
  return true;
}

// Position p < n, second half: the byte edges out of the current row (SetState, :951-987), then the row is freed; returns
// the next position's row.
template <class Ring>
RJ_HD int rj_ring_consume(const DevGraph& G, const uint8_t* t, uint64_t n, uint64_t p, Ring ring, int base) {
  const int S = G.n_states, T = G.times;
  const int t0 = base * S;
  for (int i = 0; i < G.n_byte_edges; i++) {
    const int64_t v = ring(t0 + G.be_src[i]);
    if (v < 0) continue;
    const int len = G.be_len[i];
    int land = 0;
    if (len > 0) {
      if (p + static_cast<uint64_t>(len) <= n) {
        const uint8_t* lit = G.lit + G.be_off[i];
        bool eq = true;
        for (int k = 0; k < len && eq; k++) eq = t[p + k] == lit[k];
        if (eq) land = len;
      }
    } else {
      const uint32_t c = t[p];
      if ((G.cls[G.be_off[i] * 8 + (c >> 5)] >> (c & 31)) & 1u) land = 1;
    }
    if (land) {
      int tt = base + land;
      if (tt >= T) tt -= T;
      const int d = tt * S + G.be_dst[i];
      const int64_t cur = ring(d);
      if (cur < 0 || v < cur) ring(d) = v;
    }
  }
  for (int s = 0; s < S; s++) ring(t0 + s) = -1;
  base++;
  if (base >= T) base -= T;
  return base;
}

// MatchAllAppendFilter (src/codegen.cc:36-86) for one reported match: it replaces those that begin at or after its begin;
// an empty match right at the end of the previous one is dropped.  Returns the new number of pairs in out.
RJ_HD uint64_t rj_sink_append(uint64_t* out, uint64_t out_n, int64_t pb, int64_t pe) {
  while (out_n > 0 && static_cast<int64_t>(out[2 * (out_n - 1)]) >= pb) out_n--;
  if (!(pb == pe && out_n > 0 && static_cast<int64_t>(out[2 * (out_n - 1) + 1]) == pb)) {
    out[2 * out_n] = static_cast<uint64_t>(pb);
    out[2 * out_n + 1] = static_cast<uint64_t>(pe);
    out_n++;
  }
  return out_n;
}

// The reference's loop over one segment [a, b): a is a synchronisation point (the ring starts empty),
// b the next one (b <= n), or n + 1 for the segment that runs to the end of the text.  Matches go to out[2*k],
// out[2*k+1]; begins are strictly increasing and lie in [a, b), so b - a pairs of room suffice.  Returns the number
// of matches.
//
// No state crosses b: nothing is alive there, a match found at b - 1 is emitted before the loop stops,
// the sink can neither pop (begins before b are smaller than any later begin) nor filter (a match that
// ENDS at b would mean a thread alive at b) anything of this segment on behalf of a later one.
template <class Ring>
RJ_HD uint64_t rj_replay_segment(const DevGraph& G, const uint8_t* t, uint64_t n, uint64_t a, uint64_t b, Ring ring,
                                 uint64_t* out) {
  const int slots = G.n_states * G.times;
  for (int i = 0; i < slots; i++) ring(i) = -1;
  int base = 0;
  uint64_t out_n = 0;
  for (uint64_t p = a; p != b; p++) {
    int64_t pb = 0;
    if (rj_ring_match(G, t, n, p, ring, base, &pb)) out_n = rj_sink_append(out, out_n, pb, static_cast<int64_t>(p));
    if (p == n) break;
    base = rj_ring_consume(G, t, n, p, ring, base);
  }
  return out_n;
}

// ---- long segments: speculate and verify (round 4).
//
// One lane replays a segment at 2-6 us per byte, so a stretch of megabytes without a synchronisation point took minutes.
// Two facts make parts of it independent.  (1) The ring's evolution does not depend on the sink.  (2) It depends on the
// start offsets in the ring only through their ORDER: "left-most start wins" is a minimum (SetState), ClearStates frees the
// starts between a match's begin and its end -- comparisons, never arithmetic; the values themselves only come out as the
// begins of matches.  So two runs of the loop that stand at the same position with rings of the same ORDER PATTERN -- the
// same slots occupied, their distinct starts ranked the same way -- go through the same steps from there on, thread for
// thread, and report the same matches, except that a match of a thread that was already there reports ITS start.
// A long segment is cut into parts at c_1 < c_2 < ...:
//   round 0      every part i is replayed from c_i - W bytes with a free ring (part 0 from the segment's beginning: exact),
//                in parallel.  On arriving at c_i it notes the ring's order pattern (E_i) and RELABELS the threads that
//                are there, oldest first, c_i - cnt .. c_i - 1 (order kept; a start below c_i from then on means "the
//                thread of that rank at the part's entry"); on leaving the part it notes the ring as it is (X_i^0);
//   the walk     goes over the parts in order with the TRUE ring T (free at the segment's beginning, real start offsets): a
//                part that has been replayed from T's order pattern -- it equals E_i, or a candidate pattern C_k the part
//                was given in a later round -- hands on its exit ring, the inherited ranks replaced by T's real starts;
//                otherwise the walk stops, T's pattern becomes a new candidate, and
//   round k      every part from there on is replayed from C_k, in parallel, noting X_i^k; the walk carries on.
// The number of rounds is the number of different ORDER PATTERNS met at the cuts that the warm-up did not produce by
// itself: usually none -- also for a thread that lives for megabytes (`[xy]+z` inside a run of x: it holds its slot in
// every part's warm-up just as in the true run, only under another start) --; a text that keeps a PHASE (`.{0,2}.` over a
// run without line breaks: matches tile it in threes) needs one per phase.  After kReplayMaxRounds the segment is given up
// (the caller keeps the documented semantics).  Then every part is replayed once more from its verified pattern,
// reporting its matches RAW with the inherited begins replaced by the true starts the walk left for it, and the sink
// (rj_sink_append) is applied to them: cheap next to the ring.

constexpr int kReplayMaxRounds = 12;

// The ring's order pattern, slots in time order from the current row: pat[d * S + s] = the rank of the slot's start among
// the ring's distinct starts (0 = oldest), -1 for a free slot; sorted[k] = the k-th smallest distinct start (room for one
// per slot).  Returns their number.
template <class Ring>
RJ_HD int rj_ring_pattern(const DevGraph& G, Ring ring, int base, int64_t* pat, int64_t* sorted) {
  const int S = G.n_states, T = G.times, slots = S * T;
  int cnt = 0;
  for (int i = 0; i < slots; i++) {  // the distinct starts, ascending (insertion: a few dozen at most)
    const int64_t v = ring(i);
    if (v < 0) continue;
    int k = 0;
    while (k < cnt && sorted[k] < v) k++;
    if (k < cnt && sorted[k] == v) continue;
    for (int j = cnt; j > k; j--) sorted[j] = sorted[j - 1];
    sorted[k] = v;
    cnt++;
  }
  for (int d = 0; d < T; d++) {
    int row = base + d;
    if (row >= T) row -= T;
    for (int s = 0; s < S; s++) {
      const int64_t v = ring(row * S + s);
      int64_t rank = -1;
      if (v >= 0) {
        int k = 0;
        while (sorted[k] != v) k++;
        rank = k;
      }
      pat[d * S + s] = rank;
    }
  }
  return cnt;
}

// the ring as it is, slots in time order from the current row (start offsets, -1 free)
template <class Ring>
RJ_HD void rj_ring_values(const DevGraph& G, Ring ring, int base, int64_t* snap) {
  const int S = G.n_states, T = G.times;
  for (int d = 0; d < T; d++) {
    int row = base + d;
    if (row >= T) row -= T;
    for (int s = 0; s < S; s++) snap[d * S + s] = ring(row * S + s);
  }
}

// Positions [start, stop) of a segment (stop <= n + 1; the loop also ends behind position n).
//   init == null: from a free ring at `start` <= emit_from (the warm-up).  On arriving at emit_from the ring's order pattern
//                 goes to entry_pat (if not null) and the threads that are there are relabelled emit_from - cnt + rank.
//   init != null: start == emit_from, the ring is built from the order pattern `init` with the same labels.
// A start below emit_from then means "the thread of rank start - (emit_from - cnt) at the part's entry".  On reaching
// `stop` the ring's values go to exit_values (if not null).  Matches that END at or after emit_from go to out as found
// (no sink; out may be null: counted only), an inherited begin replaced by true_starts[rank] (the real start offsets of
// the entry's threads, oldest first) when that is given.  Returns the number of raw matches.
template <class Ring>
RJ_HD uint64_t rj_replay_raw(const DevGraph& G, const uint8_t* t, uint64_t n, uint64_t start, const int64_t* init, uint64_t emit_from,
                             uint64_t stop, Ring ring, int64_t* entry_pat, int64_t* exit_values, uint64_t* out, const int64_t* true_starts,
                             int64_t* pattern_scratch) {
  const int slots = G.n_states * G.times;
  int64_t cnt = 0;
  if (init) {
    for (int i = 0; i < slots; i++) cnt = init[i] + 1 > cnt ? init[i] + 1 : cnt;
    for (int i = 0; i < slots; i++) ring(i) = init[i] >= 0 ? static_cast<int64_t>(emit_from) - cnt + init[i] : -1;
  } else {
    for (int i = 0; i < slots; i++) ring(i) = -1;
  }
  int base = 0;
  uint64_t out_n = 0;
  for (uint64_t p = start;; p++) {
    if (p == emit_from && !init) {
      // (base rotates with p - start: the pattern is taken in time order, the relabelling done slot by slot)
      // (pattern_scratch: 2 x slots words -- the pattern when the caller does not want it, and the distinct starts)
      int64_t* pat = entry_pat ? entry_pat : pattern_scratch;
      cnt = rj_ring_pattern(G, ring, base, pat, pattern_scratch + slots);
      const int S = G.n_states, T = G.times;
      for (int d = 0; d < T; d++) {
        int row = base + d;
        if (row >= T) row -= T;
        for (int s = 0; s < S; s++)
          if (pat[d * S + s] >= 0) ring(row * S + s) = static_cast<int64_t>(emit_from) - cnt + pat[d * S + s];
      }
    }
    if (p == stop) {
      if (exit_values) rj_ring_values(G, ring, base, exit_values);
      break;
    }
    int64_t pb = 0;
    if (rj_ring_match(G, t, n, p, ring, base, &pb) && p >= emit_from) {
      if (out) {
        if (true_starts && pb < static_cast<int64_t>(emit_from)) pb = true_starts[pb - (static_cast<int64_t>(emit_from) - cnt)];
        out[2 * out_n] = static_cast<uint64_t>(pb);
        out[2 * out_n + 1] = p;
      }
      out_n++;
    }
    if (p == n) break;
    base = rj_ring_consume(G, t, n, p, ring, base);
  }
  return out_n;
}

}  // namespace rejit_amd
#endif
