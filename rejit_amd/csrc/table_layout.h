// rejit_amd/csrc/table_layout.h -- the automaton tables as ONE contiguous blob of 32-bit words, the
// form the kernels read (device_program.h):
//   first [C][W]   last [C][W]   linear [W]   row_of [P]   rows [C][n_rows][W]   cls [256][W]
// Shared by the engine (upload to HBM; forward and reverse automaton) and by the CPU unit tests,
// which point a DevProgram at the host copy.
#ifndef REJIT_AMD_TABLE_LAYOUT_H_
#define REJIT_AMD_TABLE_LAYOUT_H_

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "dense_streams.h"
#include "device_program.h"
#include "lds_walk.h"
#include "lowering.h"

namespace rejit_amd {

struct TableBlob {
  std::vector<uint32_t> words;
  size_t off_first = 0, off_last = 0, off_linear = 0, off_rowof = 0, off_rows = 0, off_cls = 0;
  int W = 1, C = 1, R = 1, Pn = 1;
};

// `Tables`: Program (forward) or Program::Reverse -- same member names
template <class Tables>
TableBlob make_table_blob(const Tables& T, int n_pos, int n_words, bool has_assertions) {
  TableBlob b;
  b.W = n_words;
  b.C = has_assertions ? kNumCtx : 1;
  b.R = std::max(T.n_rows, 1);
  b.Pn = std::max(n_pos, 1);
  const size_t W = static_cast<size_t>(b.W), C = static_cast<size_t>(b.C), R = static_cast<size_t>(b.R);
  b.off_first = 0;
  b.off_last = b.off_first + C * W;
  b.off_linear = b.off_last + C * W;
  b.off_rowof = b.off_linear + W;
  b.off_rows = b.off_rowof + static_cast<size_t>(b.Pn);
  b.off_cls = b.off_rows + C * R * W;
  b.words.assign(b.off_cls + 256 * W, 0u);
  for (size_t c = 0; c < C; c++) {
    std::copy(T.first[c].begin(), T.first[c].end(), b.words.begin() + static_cast<long>(b.off_first + c * W));
    std::copy(T.last[c].begin(), T.last[c].end(), b.words.begin() + static_cast<long>(b.off_last + c * W));
    for (int r = 0; r < T.n_rows; r++)
      std::copy(T.rows[c].begin() + static_cast<long>(r) * b.W, T.rows[c].begin() + static_cast<long>(r + 1) * b.W,
                b.words.begin() + static_cast<long>(b.off_rows + (c * R + static_cast<size_t>(r)) * W));
  }
  std::copy(T.linear.begin(), T.linear.end(), b.words.begin() + static_cast<long>(b.off_linear));
  for (int i = 0; i < n_pos; i++) b.words[b.off_rowof + static_cast<size_t>(i)] = static_cast<uint32_t>(T.row_of[static_cast<size_t>(i)]);
  std::copy(T.cls.begin(), T.cls.end(), b.words.begin() + static_cast<long>(b.off_cls));
  return b;
}

// The same tables padded for the LDS walkers (lds_walk.h): NQ 64-bit words per row, rows by position (row n_pos
// all zero), first / last / linear / rows / cls in this order.
inline std::vector<uint64_t> make_walk_blob(const TableBlob& b, int n_pos, int nq) {
  const size_t NQ = static_cast<size_t>(nq), C = static_cast<size_t>(b.C), W = static_cast<size_t>(b.W), P = static_cast<size_t>(std::max(n_pos, 0));
  std::vector<uint64_t> out(static_cast<size_t>(lw_blob_words(nq, b.C, static_cast<int>(P))), 0ull);
  auto pack = [&](size_t dst, size_t src) {  // W 32-bit words at b.words[src] -> NQ 64-bit words at out[dst]
    for (size_t k = 0; k < W && k < 2 * NQ; k++) out[dst + k / 2] |= static_cast<uint64_t>(b.words[src + k]) << (32 * (k & 1));
  };
  const size_t o_first = 0, o_last = C * NQ, o_lin = 2 * C * NQ, o_loop = o_lin + NQ, o_rows = o_loop + NQ, o_cls = o_rows + C * (P + 1) * NQ;
  for (size_t c = 0; c < C; c++) {
    pack(o_first + c * NQ, b.off_first + c * W);
    pack(o_last + c * NQ, b.off_last + c * W);
    for (size_t i = 0; i < P; i++) {
      const int32_t r = static_cast<int32_t>(b.words[b.off_rowof + i]);
      if (r >= 0) pack(o_rows + (c * (P + 1) + i) * NQ, b.off_rows + (c * static_cast<size_t>(b.R) + static_cast<size_t>(r)) * W);
    }
  }
  pack(o_lin, b.off_linear);
  for (size_t v = 0; v < 256; v++) pack(o_cls + v * NQ, b.off_cls + v * W);
  // loops: rows that are exactly {i, i + 1} in every context (position i + 1 exists)
  for (size_t i = 0; i + 1 < P; i++) {
    if (static_cast<int32_t>(b.words[b.off_rowof + i]) < 0) continue;
    bool loop = true;
    for (size_t c = 0; c < C && loop; c++)
      for (size_t q = 0; q < NQ && loop; q++) {
        uint64_t want = 0;
        if (i / 64 == q) want |= 1ull << (i % 64);
        if ((i + 1) / 64 == q) want |= 1ull << ((i + 1) % 64);
        loop = out[o_rows + (c * (P + 1) + i) * NQ + q] == want;
      }
    if (loop) out[o_loop + i / 64] |= 1ull << (i % 64);
  }
  return out;
}

// table pointers / sizes of D for a blob that lives at `base` (host or device memory)
inline void point_tables(DevProgram* D, const uint32_t* base, const TableBlob& b, int n_pos) {
  D->n_pos = n_pos;
  D->n_words = b.W;
  D->n_ctx = b.C;
  D->n_rows = b.R;
  D->table_words = static_cast<uint32_t>(b.words.size());
  D->first = base + b.off_first;
  D->last = base + b.off_last;
  D->linear = base + b.off_linear;
  D->row_of = reinterpret_cast<const int32_t*>(base + b.off_rowof);
  D->rows = base + b.off_rows;
  D->cls = base + b.off_cls;
}

// the window constants and the behind-mode cut sets of D (unused window slots repeat the last window,
// so kernels instantiated for a larger K stay exact)
inline void fill_windows(DevProgram* D, const Program& P) {
  D->mode = P.mode == ScanMode::Windows ? 1 : 0;
  D->n_windows = static_cast<int>(P.windows.size());
  D->win_offset = P.windows.empty() ? 0 : P.windows[0].offset;
  D->win_len = P.windows.empty() ? 0 : P.windows[0].len;
  for (int k = 0; k < kDevMaxWindows; k++) {
    const FFWindow w = P.windows.empty() ? FFWindow{} : P.windows[std::min<size_t>(static_cast<size_t>(k), P.windows.size() - 1)];
    D->win_value0[k] = w.value0;
    D->win_mask0[k] = w.mask0;
    D->win_value1[k] = w.value1;
    D->win_mask1[k] = w.mask1;
  }
  D->behind = P.behind ? 1u : 0u;
  for (int k = 0; k < kDevMaxWindows; k++)
    for (int j = 0; j < 4; j++) D->cut_fwd[k][j] = D->cut_rev[k][j] = 0;
  if (P.behind)
    for (size_t k = 0; k < P.cut_positions.size() && k < static_cast<size_t>(kDevMaxWindows); k++)
      for (int q = 0; q < P.n_pos; q++)
        if ((P.cut_positions[k][static_cast<size_t>(q) >> 5] >> (q & 31)) & 1u) {
          const int r = P.n_pos - 1 - q;
          D->cut_fwd[k][q >> 5] |= 1u << (q & 31);
          D->cut_rev[k][r >> 5] |= 1u << (r & 31);
        }
}

inline uint32_t nullable_bits(const Program& P) {
  uint32_t bits = 0;
  for (int c = 0; c < kNumCtx; c++)
    if (P.nullable[P.has_assertions ? c : 0]) bits |= 1u << c;
  return bits;
}


// Positions of a lane-sized automaton whose follow set is, in every context, exactly {i, i+1} (a loop) or
// {i+1, i+2} (a skip): the dense kernel steps them with shifts (DevProgram::loop_mask / skip_mask).
inline void loop_skip_masks(const Program& P, uint32_t (&loop)[4], uint32_t (&skip)[4]) {
  const int W = P.n_words, C = P.has_assertions ? kNumCtx : 1;
  for (int k = 0; k < 4; k++) loop[k] = skip[k] = 0;
  if (W > 4) return;
  for (int i = 0; i < P.n_pos; i++) {
    const int r = P.row_of[static_cast<size_t>(i)];
    if (r < 0) continue;  // linear
    bool is_loop = i + 1 < P.n_pos, is_skip = i + 2 < P.n_pos;
    for (int c = 0; c < C; c++)
      for (int k = 0; k < W; k++) {
        const uint32_t row = P.rows[c][static_cast<size_t>(r) * W + k];
        uint32_t want_loop = 0, want_skip = 0;
        for (int d = 0; d < 3; d++) {
          const int j = i + d;
          if ((j >> 5) != k) continue;
          if (d <= 1) want_loop |= 1u << (j & 31);
          if (d >= 1) want_skip |= 1u << (j & 31);
        }
        is_loop = is_loop && row == want_loop;
        is_skip = is_skip && row == want_skip;
      }
    if (is_loop) loop[i >> 5] |= 1u << (i & 31);
    else if (is_skip) skip[i >> 5] |= 1u << (i & 31);
  }
}

// DevProgram::loop_first -- `X+ rest`: only the FIRST byte of a run of X can begin a selected match.
// Holds when (1) in the only context a match can begin at ONE position p0, which follows itself, the
// pattern has no assertions and is not nullable: the thread of start s - 1 then passes through p0 at s, so
// s - 1 reaches every end s reaches, and by induction so does the run's first byte; a start inside the run
// could then only be selected as the END of an earlier match; and (2) no longest match ends inside a run:
// an accepting position q that can consume a byte of X is followed by accepting positions that consume
// every byte of X, so a match whose last byte is in X goes on while the text stays in X
// (`[a-z]+@[a-z]+` qualifies, `[a-f]+[0-9][a-f]` does not: "ab1c|d2e").
inline bool run_start_rule(const Program& P) {
  if (P.has_assertions || P.any_nullable || P.n_words > 4 || P.n_pos == 0) return false;
  const int W = P.n_words;
  int n_first = 0, p0 = -1;
  for (int i = 0; i < P.n_pos; i++)
    if ((P.first[0][static_cast<size_t>(i) >> 5] >> (i & 31)) & 1u) {
      n_first++;
      p0 = i;
    }
  if (n_first != 1) return false;
  const int r0 = P.row_of[static_cast<size_t>(p0)];
  if (r0 < 0 || !((P.rows[0][static_cast<size_t>(r0) * W + (p0 >> 5)] >> (p0 & 31)) & 1u)) return false;
  auto in_cls = [&](int b, int q) { return (P.cls[static_cast<size_t>(b) * W + (q >> 5)] >> (q & 31)) & 1u; };
  auto is_last = [&](int q) { return (P.last[0][static_cast<size_t>(q) >> 5] >> (q & 31)) & 1u; };
  for (int q = 0; q < P.n_pos; q++) {
    if (!is_last(q)) continue;
    bool eats_x = false;
    for (int b = 0; b < 256 && !eats_x; b++) eats_x = in_cls(b, p0) && in_cls(b, q);
    if (!eats_x) continue;
    // follow(q) & last must cover X
    std::vector<uint32_t> fol(static_cast<size_t>(W), 0);
    const int rq = P.row_of[static_cast<size_t>(q)];
    if (rq < 0) {
      if (q + 1 < P.n_pos) fol[static_cast<size_t>(q + 1) >> 5] |= 1u << ((q + 1) & 31);
    } else {
      for (int k = 0; k < W; k++) fol[static_cast<size_t>(k)] = P.rows[0][static_cast<size_t>(rq) * W + k];
    }
    for (int b = 0; b < 256; b++) {
      if (!in_cls(b, p0)) continue;
      bool goes_on = false;
      for (int k = 0; k < W && !goes_on; k++)
        goes_on = (fol[static_cast<size_t>(k)] & P.last[0][static_cast<size_t>(k)] & P.cls[static_cast<size_t>(b) * W + k]) != 0;
      if (!goes_on) return false;
    }
  }
  return true;
}

// DevProgram::short_max: the longest match of a bounded pattern that rj_lane_longest_short can take, else 0
inline uint32_t short_match_bound(const Program& P) {
  if (P.has_assertions || P.n_words > 2 || P.n_pos == 0 || P.max_len == Program::kUnboundedLen || P.max_len > 16 || P.max_len == 0) return 0;
  return static_cast<uint32_t>(P.max_len);
}

// The plan of the lane-packed pre-steps (device_program.h: SwarPlan), or n_ranges = 0 when the pattern
// does not qualify: <= 8 positions, no assertions, not nullable, at most two accepting positions, every
// class a few ranges.
inline SwarPlan make_swar_plan(const Program& P) {
  SwarPlan pl{};
  if (P.n_pos < 1 || P.n_pos > 8 || P.has_assertions || P.any_nullable || P.n_words != 1) return pl;
  uint32_t loop[4], skip[4];
  loop_skip_masks(P, loop, skip);
  uint32_t n = 0;
  for (int half = 0; half < 2; half++) {
    for (int p = 0; p < P.n_pos; p++) {
      int b = 0;
      while (b < 128) {
        while (b < 128 && !((P.cls[static_cast<size_t>(half * 128 + b)] >> p) & 1u)) b++;
        if (b >= 128) break;
        int e = b;
        while (e + 1 < 128 && ((P.cls[static_cast<size_t>(half * 128 + e + 1)] >> p) & 1u)) e++;
        if (n >= static_cast<uint32_t>(kSwarMaxRanges)) return SwarPlan{};
        pl.add_lo[n] = static_cast<uint32_t>(0x80 - b) * 0x01010101u;
        pl.add_hi[n] = static_cast<uint32_t>(0x7f - e) * 0x01010101u;
        pl.shift[n] = static_cast<uint32_t>(7 - p);
        n++;
        b = e + 1;
      }
    }
    if (half == 0) pl.n_low = n;
  }
  if (n == 0) return pl;
  const uint32_t all = (1u << P.n_pos) - 1u;
  uint32_t gen = 0;
  for (int p = 0; p < P.n_pos; p++) {
    const int r = P.row_of[static_cast<size_t>(p)];
    if (r < 0 || (loop[0] >> p) & 1u) continue;
    if (P.rows[0][static_cast<size_t>(r)] != 0) gen |= 1u << p;  // (an empty row: the position is a dead end)
  }
  const uint32_t last = P.last[0][0] & all, first = P.first[0][0] & all;
  if (last == 0 || __builtin_popcount(last) > 2) return SwarPlan{};
  pl.last_shift[0] = static_cast<uint32_t>(__builtin_ctz(last));
  pl.last_shift[1] = static_cast<uint32_t>(31 - __builtin_clz(last));
  pl.first_shift = __builtin_popcount(first) == 1 ? static_cast<uint32_t>(__builtin_ctz(first)) : 0u;
  const uint32_t rep = 0x01010101u;
  pl.first = first * rep;
  pl.loopm = (loop[0] & all) * rep;
  pl.step1 = ((P.linear[0] | loop[0]) & all) * rep;
  pl.gen = gen * rep;
  pl.depth = P.max_len <= 1 ? 1 : P.max_len <= 2 ? 2 : 4;
  pl.n_ranges = n;
  return pl;
}

// The plan of the bit-stream dense kernel (dense_streams.h: StreamPlan), n_pos = 0 when the pattern does not qualify:
// <= 8 positions in one word, no assertions, not nullable, every follow set a subset of {i, i + 1}, <= 8 byte ranges in
// all, not at risk of the ring artefact, and -- decided here on the automaton -- NO TWO CANDIDATES CAN OVERLAP, so that
// the candidates are the reference's selection (src/codegen.cc:36-86) and can be written once, in place.
//   Candidates: every start with a match; under `loop_first` (DevProgram::loop_first, `X+ rest`) the first byte of every
//   run of X.  Two candidates s < s' overlap when the thread of s consumes the byte at s'.  Sufficient for "never": over
//   all thread states reachable after >= 1 consumed bytes (<= 256 position sets x "was the last byte in X"), no state can
//   consume a byte at which a candidate may begin (a byte of a first position's class; under loop_first: a byte of X
//   that follows a byte outside X).
inline StreamPlan make_stream_plan(const Program& P, bool loop_first, bool q8_risk) {
  StreamPlan pl{};
  if (P.n_pos < 1 || P.n_pos > kStreamMaxPos || P.has_assertions || P.any_nullable || P.n_words != 1 || q8_risk) return pl;
  const uint32_t all = (1u << P.n_pos) - 1u;
  const uint32_t first = P.first[0][0] & all, last = P.last[0][0] & all;
  if (first == 0 || last == 0) return pl;
  uint32_t step = 0, loop = 0;
  for (int k = 0; k < P.n_pos; k++) {
    const int r = P.row_of[static_cast<size_t>(k)];
    const uint32_t F = (r < 0 ? (1u << (k + 1)) : P.rows[0][static_cast<size_t>(r)]) & all;
    const uint32_t allowed = (1u << k) | (k + 1 < P.n_pos ? 1u << (k + 1) : 0u);
    if (F & ~allowed) return pl;  // a general follow set (alternation inside a repetition, optional positions)
    if ((F >> k) & 1u) loop |= 1u << k;
    if (k + 1 < P.n_pos && ((F >> (k + 1)) & 1u)) step |= 1u << k;
  }
  // the byte ranges of all positions, identical ranges shared
  uint32_t n = 0;
  for (int half = 0; half < 2; half++)
    for (int k = 0; k < P.n_pos; k++) {
      int b = 0;
      while (b < 128) {
        while (b < 128 && !((P.cls[static_cast<size_t>(half * 128 + b)] >> k) & 1u)) b++;
        if (b >= 128) break;
        int e = b;
        while (e + 1 < 128 && ((P.cls[static_cast<size_t>(half * 128 + e + 1)] >> k) & 1u)) e++;
        const uint32_t lo = static_cast<uint32_t>(0x80 - b) * 0x01010101u, hi = static_cast<uint32_t>(0x7f - e) * 0x01010101u;
        uint32_t r = 0;
        while (r < n && !(pl.add_lo[r] == lo && pl.add_hi[r] == hi && ((pl.high_half >> r) & 1u) == static_cast<uint32_t>(half))) r++;
        if (r == n) {
          if (n >= static_cast<uint32_t>(kStreamMaxRanges)) return StreamPlan{};
          pl.add_lo[n] = lo;
          pl.add_hi[n] = hi;
          pl.high_half |= static_cast<uint32_t>(half) << n;
          n++;
        }
        pl.range_pos[r] |= 1u << k;
        b = e + 1;
      }
    }
  if (n == 0) return StreamPlan{};
  const uint32_t first_pos = static_cast<uint32_t>(__builtin_ctz(first));
  if (loop_first && (__builtin_popcount(first) != 1 || !((loop >> first_pos) & 1u))) loop_first = false;
  // can two candidates overlap?
  auto follow = [&](uint32_t S) {
    uint32_t T = 0;
    for (int k = 0; k < P.n_pos; k++)
      if ((S >> k) & 1u) T |= (((step >> k) & 1u) << (k + 1)) | (((loop >> k) & 1u) << k);
    return T;
  };
  std::vector<uint8_t> seen(512, 0);
  std::vector<uint32_t> todo;
  for (int b = 0; b < 256; b++) {
    const uint32_t S = first & P.cls[static_cast<size_t>(b)] & all;
    if (S == 0) continue;
    const uint32_t in_x = loop_first ? 1u : 0u;  // (under loop_first the first byte is in X by definition)
    if (!seen[S * 2 + in_x]) {
      seen[S * 2 + in_x] = 1;
      todo.push_back(S * 2 + in_x);
    }
  }
  while (!todo.empty()) {
    const uint32_t st = todo.back();
    todo.pop_back();
    const uint32_t S = st >> 1, prev_in_x = st & 1u, T = follow(S);
    for (int c = 0; c < 256; c++) {
      const uint32_t cls = P.cls[static_cast<size_t>(c)] & all;
      const uint32_t S2 = T & cls;
      if (S2 == 0) continue;  // the thread does not consume this byte
      const bool in_x = loop_first && ((cls >> first_pos) & 1u);
      const bool may_begin = loop_first ? (in_x && !prev_in_x) : (first & cls) != 0;
      if (may_begin) {
        // two candidates can overlap: the kernel makes the selection itself when every match ends within the register
        // steps (StreamPlan::select, round 5); else this is scan_dense_walk's pattern
        if (loop_first || P.max_len == Program::kUnboundedLen || P.max_len > kStreamShift) return StreamPlan{};
        pl.select = 1;
        todo.clear();
        break;
      }
      const uint32_t nx = S2 * 2 + (in_x ? 1u : 0u);
      if (!seen[nx]) {
        seen[nx] = 1;
        todo.push_back(nx);
      }
    }
  }
  pl.n_pos = static_cast<uint32_t>(P.n_pos);
  pl.n_ranges = n;
  pl.depth = P.max_len == Program::kUnboundedLen || P.max_len > kStreamShift ? kStreamShift : static_cast<uint32_t>(std::max<uint64_t>(P.max_len, 1));
  pl.loop_first = loop_first ? 1u : 0u;
  pl.first_pos = first_pos;
  pl.first = first;
  pl.last = last;
  pl.step = step;
  pl.loop = loop;
  // the run form of the steps (dense_streams.h: rj_stream_runs): `X+`, or `X+ Y` with no byte in both classes
  if (pl.loop_first && !pl.select && first == 1u) {
    if (P.n_pos == 1 && last == 1u && loop == 1u) {
      pl.run_shape = 1;
    } else if (P.n_pos == 2 && last == 2u && loop == 1u && step == 1u) {
      bool disjoint = true;
      for (int c = 0; c < 256; c++) disjoint = disjoint && (P.cls[static_cast<size_t>(c)] & 3u) != 3u;
      if (disjoint) pl.run_shape = 2;
    }
  }
  if (getenv("RJ_NO_RUN_STEPS") != nullptr) pl.run_shape = 0;   // measurement override
  return pl;
}

// The NFA graph for the exact replay (DevGraph): seven int32 arrays (byte edges: src dst len off; control
// edges: src dst kind), the 256-bit classes of the class edges, then the bytes of the literal edges.
struct GraphBlob {
  std::vector<uint8_t> bytes;
  size_t o_src = 0, o_dst = 0, o_len = 0, o_off = 0, o_cs = 0, o_cd = 0, o_ck = 0, o_cls = 0, o_lit = 0;  // in words
  int32_t n_states = 0, entry = 0, exit = 0, n_byte_edges = 0, n_control_edges = 0, times = 1;
};

inline GraphBlob make_graph_blob(const Graph& g) {
  GraphBlob b;
  std::vector<uint32_t> classes;
  std::string lits;
  std::vector<int32_t> be_src, be_dst, be_len, be_off, ce_src, ce_dst, ce_kind;
  size_t longest = 1;
  for (const ByteEdge& e : g.byte_edges) {
    be_src.push_back(e.src);
    be_dst.push_back(e.dst);
    if (!e.bytes.empty()) {
      be_len.push_back(static_cast<int32_t>(e.bytes.size()));
      be_off.push_back(static_cast<int32_t>(lits.size()));
      lits += e.bytes;
      longest = std::max(longest, e.bytes.size());
    } else {
      be_len.push_back(0);
      be_off.push_back(static_cast<int32_t>(classes.size() / 8));
      for (int k = 0; k < 8; k++) classes.push_back(e.cls.w[k]);
    }
  }
  for (const ControlEdge& c : g.control_edges) {
    ce_src.push_back(c.src);
    ce_dst.push_back(c.dst);
    ce_kind.push_back(c.kind == ControlKind::Epsilon ? 0 : c.kind == ControlKind::StartOfLine ? 1 : 2);
  }
  const size_t nb = be_src.size(), nc = ce_src.size();
  const size_t words = 4 * nb + 3 * nc + classes.size();
  b.bytes.assign(words * 4 + lits.size() + 16, 0);
  uint32_t* w = reinterpret_cast<uint32_t*>(b.bytes.data());
  size_t o = 0;
  auto put = [&](const std::vector<int32_t>& v) {
    const size_t at = o;
    if (!v.empty()) memcpy(w + o, v.data(), v.size() * 4);
    o += v.size();
    return at;
  };
  b.o_src = put(be_src);
  b.o_dst = put(be_dst);
  b.o_len = put(be_len);
  b.o_off = put(be_off);
  b.o_cs = put(ce_src);
  b.o_cd = put(ce_dst);
  b.o_ck = put(ce_kind);
  b.o_cls = o;
  if (!classes.empty()) memcpy(w + o, classes.data(), classes.size() * 4);
  o += classes.size();
  b.o_lit = o;
  if (!lits.empty()) memcpy(b.bytes.data() + o * 4, lits.data(), lits.size());
  b.n_states = g.n_states;
  b.entry = g.entry;
  b.exit = g.exit;
  b.n_byte_edges = static_cast<int32_t>(nb);
  b.n_control_edges = static_cast<int32_t>(nc);
  b.times = 1 + static_cast<int32_t>(std::min<size_t>(longest, 64));
  return b;
}

// pointers of G for a blob that lives at `base` (host or device memory)
inline void point_graph(DevGraph* G, const uint8_t* base, const GraphBlob& b) {
  const uint32_t* w = reinterpret_cast<const uint32_t*>(base);
  G->n_states = b.n_states;
  G->entry = b.entry;
  G->exit = b.exit;
  G->n_byte_edges = b.n_byte_edges;
  G->n_control_edges = b.n_control_edges;
  G->times = b.times;
  G->be_src = reinterpret_cast<const int32_t*>(w + b.o_src);
  G->be_dst = reinterpret_cast<const int32_t*>(w + b.o_dst);
  G->be_len = reinterpret_cast<const int32_t*>(w + b.o_len);
  G->be_off = reinterpret_cast<const int32_t*>(w + b.o_off);
  G->ce_src = reinterpret_cast<const int32_t*>(w + b.o_cs);
  G->ce_dst = reinterpret_cast<const int32_t*>(w + b.o_cd);
  G->ce_kind = reinterpret_cast<const int32_t*>(w + b.o_ck);
  G->cls = w + b.o_cls;
  G->lit = reinterpret_cast<const uint8_t*>(w + b.o_lit);
}

}  // namespace rejit_amd
#endif
