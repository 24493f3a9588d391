// tests/support/program_exec.cc -- TEST-ONLY scalar executor of a lowered Program.
//
// Compiled by tests/test_lowering.py together with rejit_amd/csrc/{parser,lowering}.cc
// into tests/support/libprogram_exec.so (g++, no HIP).  It restates, one position at a
// time, exactly what the HIP kernels compute from the same tables (candidate filter ->
// forward automaton -> left-most-longest selection), so the lowering can be checked
// against the oracle on a machine without a GPU.  It is never linked into the product
// library and the product never falls back to it.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../rejit_amd/csrc/exact_count.h"
#include "../../rejit_amd/csrc/lowering.h"
#include "../../rejit_amd/csrc/run_scan.h"
#include "../../rejit_amd/csrc/table_layout.h"

using namespace rejit_amd;

namespace {

struct Span { uint64_t begin, end; };

inline bool line_break(uint8_t c) { return c == '\n' || c == '\r'; }

inline int context_at(const uint8_t* t, uint64_t n, uint64_t p) {
  int ctx = 0;
  if (p == 0 || line_break(t[p - 1])) ctx |= 1;
  if (p == n || line_break(t[p])) ctx |= 2;
  return ctx;
}

bool longest_at(const Program& P, const uint8_t* t, uint64_t n, uint64_t s, uint64_t* end) {
  const int W = P.n_words;
  bool found = false;
  int ctx = context_at(t, n, s);
  if (P.nullable[ctx]) {
    found = true;
    *end = s;
  }
  if (s >= n || P.n_pos == 0) return found;
  std::vector<uint32_t> S(W), T(W);
  const uint32_t* row = &P.cls[(size_t)t[s] * W];
  bool any = false;
  for (int k = 0; k < W; k++) {
    S[k] = P.first[ctx][k] & row[k];
    any |= S[k] != 0;
  }
  uint64_t p = s + 1;
  while (any) {
    ctx = context_at(t, n, p);
    for (int k = 0; k < W; k++)
      if (S[k] & P.last[ctx][k]) {
        found = true;
        *end = p;
        break;
      }
    if (p == n) break;
    // follow
    uint32_t carry = 0;
    for (int k = 0; k < W; k++) {
      uint32_t lin = S[k] & P.linear[k];
      T[k] = (lin << 1) | carry;
      carry = lin >> 31;
    }
    for (int k = 0; k < W; k++) {
      uint32_t sp = S[k] & ~P.linear[k];
      while (sp) {
        int b = __builtin_ctz(sp);
        sp &= sp - 1;
        int r = P.row_of[(size_t)k * 32 + b];
        const uint32_t* fr = &P.rows[ctx][(size_t)r * W];
        for (int j = 0; j < W; j++) T[j] |= fr[j];
      }
    }
    row = &P.cls[(size_t)t[p] * W];
    any = false;
    for (int k = 0; k < W; k++) {
      S[k] = T[k] & row[k];
      any |= S[k] != 0;
    }
    p++;
  }
  return found;
}

bool candidate(const Program& P, const uint8_t* t, uint64_t n, uint64_t s) {
  if (P.mode == ScanMode::Windows && P.behind) return s < n && P.first_bytes.has(t[s]);  // (every start; the hits are checked by pe_match_all_behind)
  if (P.mode == ScanMode::Windows && P.floating) {
    // some window must occur at w in [s + float_min, s + float_max]
    for (uint64_t w0 = s + P.float_min; w0 <= s + P.float_max; w0++)
      for (const FFWindow& w : P.windows) {
        if (w0 + w.len > n) continue;
        uint32_t v0 = 0, v1 = 0;
        for (uint32_t k = 0; k < w.len; k++) {
          uint32_t c = t[w0 + k];
          if (k < 4) v0 |= c << (8 * k);
          else v1 |= c << (8 * (k - 4));
        }
        if ((v0 & w.mask0) == w.value0 && (v1 & w.mask1) == w.value1) return true;
      }
    return false;
  }
  if (P.mode == ScanMode::Windows) {
    for (const FFWindow& w : P.windows) {
      if (s + w.offset + w.len > n) continue;
      uint32_t v0 = 0, v1 = 0;
      for (uint32_t k = 0; k < w.len; k++) {
        uint32_t c = t[s + w.offset + k];
        if (k < 4) v0 |= c << (8 * k);
        else v1 |= c << (8 * (k - 4));
      }
      if ((v0 & w.mask0) == w.value0 && (v1 & w.mask1) == w.value1) return true;
    }
    return false;
  }
  if (P.any_nullable) {
    if (P.nullable[context_at(t, n, s)]) return true;
  }
  if (!(s < n && P.first_bytes.has(t[s]))) return false;
  // `X+ rest` (DevProgram::loop_first, as the engine derives it): a start whose previous byte is in X
  // too is never selected, so the dense kernel does not take it as a candidate
  if (s > 0 && P.first_bytes.has(t[s - 1]) && run_start_rule(P)) return false;
  return true;
}

// ---- windows behind an unbounded prefix (Program::behind): the per-hit procedure of
// verify_behind_in_regions restated on the host tables.
struct Tables {
  const std::vector<uint32_t>* first;   // [ctx]
  const std::vector<uint32_t>* last;    // [ctx]
  const std::vector<uint32_t>* linear;
  const std::vector<int32_t>* row_of;
  const std::vector<uint32_t>* rows;    // [ctx]
  const std::vector<uint32_t>* cls;
};

void follow(const Tables& T, int W, const std::vector<uint32_t>& S, int ctx, std::vector<uint32_t>* out) {
  uint32_t carry = 0;
  for (int k = 0; k < W; k++) {
    uint32_t lin = S[k] & (*T.linear)[k];
    (*out)[k] = (lin << 1) | carry;
    carry = lin >> 31;
  }
  for (int k = 0; k < W; k++) {
    uint32_t sp = S[k] & ~(*T.linear)[k];
    while (sp) {
      int b = __builtin_ctz(sp);
      sp &= sp - 1;
      const uint32_t* fr = &T.rows[ctx][(size_t)(*T.row_of)[(size_t)k * 32 + b] * W];
      for (int j = 0; j < W; j++) (*out)[j] |= fr[j];
    }
  }
}

bool any(const std::vector<uint32_t>& S) {
  for (uint32_t x : S)
    if (x) return true;
  return false;
}

// does a thread that consumes text[p] at (forward) position q reach an accepting boundary?
bool reaches_accept(const Program& P, const uint8_t* t, uint64_t n, uint64_t p, int q) {
  const int W = P.n_words;
  const Tables F{P.first, P.last, &P.linear, &P.row_of, P.rows, &P.cls};
  std::vector<uint32_t> S(W, 0u), T(W);
  S[q >> 5] = 1u << (q & 31);
  for (uint64_t at = p + 1;; at++) {  // S = positions that consumed text[at - 1]
    const int ctx = context_at(t, n, at);
    for (int k = 0; k < W; k++)
      if (S[k] & P.last[ctx][k]) return true;
    if (at == n) return false;
    follow(F, W, S, ctx, &T);
    const uint32_t* row = &P.cls[(size_t)t[at] * W];
    for (int k = 0; k < W; k++) S[k] = T[k] & row[k];
    if (!any(S)) return false;
  }
}

// left-most s <= p such that a thread started at s sits at one of `ok_rev` (reverse numbering) after
// consuming text[p]; returns false when there is none
bool leftmost_start(const Program& P, const uint8_t* t, uint64_t n, uint64_t p, std::vector<uint32_t> S, uint64_t* start) {
  const int W = P.n_words;
  const Program::Reverse& R = P.rev;
  const Tables B{R.first, R.last, &R.linear, &R.row_of, R.rows, &R.cls};
  std::vector<uint32_t> T(W);
  bool found = false;
  for (uint64_t at = p;; at--) {  // S = reverse positions that consumed text[at]
    const int ctx = context_at(t, n, at);
    for (int k = 0; k < W; k++)
      if (S[k] & R.last[ctx][k]) {
        found = true;
        *start = at;
        break;
      }
    if (at == 0) break;
    follow(B, W, S, ctx, &T);
    const uint32_t* row = &R.cls[(size_t)t[at - 1] * W];
    for (int k = 0; k < W; k++) S[k] = T[k] & row[k];
    if (!any(S)) break;
  }
  return found;
}

// candidates (begin, end) of the behind mode, one per hit at most
void behind_candidates(const Program& P, const uint8_t* t, uint64_t n, std::vector<Span>* out) {
  const int W = P.n_words;
  for (uint64_t p = 0; p < n; p++) {
    std::vector<uint32_t> ok(W, 0u);
    bool hit = false;
    for (size_t k = 0; k < P.windows.size(); k++) {
      const FFWindow& w = P.windows[k];
      if (p + w.len > n) continue;
      uint32_t v0 = 0, v1 = 0;
      for (uint32_t i = 0; i < w.len; i++) {
        uint32_t c = t[p + i];
        if (i < 4) v0 |= c << (8 * i);
        else v1 |= c << (8 * (i - 4));
      }
      if ((v0 & w.mask0) != w.value0 || (v1 & w.mask1) != w.value1) continue;
      for (int q = 0; q < P.n_pos; q++)
        if ((P.cut_positions[k][(size_t)q >> 5] >> (q & 31)) & 1u)
          if (reaches_accept(P, t, n, p, q)) {
            const int r = P.n_pos - 1 - q;
            ok[(size_t)r >> 5] |= 1u << (r & 31);
            hit = true;
          }
    }
    if (!hit) continue;
    uint64_t s, e;
    if (!leftmost_start(P, t, n, p, ok, &s)) continue;
    if (longest_at(P, t, n, s, &e)) out->push_back({s, e});
  }
}

}  // namespace

extern "C" {

// behind mode only: number of matches, -100 when the candidates conflict (a skipped candidate ends
// after the match that hides it: the engine then repeats the run in dense mode), -101 when the
// pattern's plan is not `behind`
long pe_match_all_behind(const char* re, const uint8_t* text, uint64_t n, uint64_t* out, uint64_t cap) {
  LowerResult lr = lower(re);
  if (lr.status != 0) return lr.status;
  const Program& P = *lr.program;
  if (!P.behind) return -101;
  std::vector<Span> cands;
  behind_candidates(P, text, n, &cands);
  std::stable_sort(cands.begin(), cands.end(), [](const Span& a, const Span& b) { return a.begin < b.begin; });
  std::vector<Span> sel;
  uint64_t cur = 0;
  for (size_t i = 0; i < cands.size(); i++) {
    const Span& c = cands[i];
    if (i > 0 && c.begin == cands[i - 1].begin) continue;
    if (c.begin < cur) {
      if (c.end > cur) return -100;
      continue;
    }
    sel.push_back(c);
    cur = c.end > c.begin ? c.end : c.begin + 1;
  }
  for (size_t i = 0; i < sel.size() && i < cap; i++) {
    out[2 * i] = sel[i].begin;
    out[2 * i + 1] = sel[i].end;
  }
  return (long)sel.size();
}

// returns count, or a negative status
long pe_match_all(const char* re, const uint8_t* text, uint64_t n, uint64_t* out, uint64_t cap) {
  LowerResult lr = lower(re);
  if (lr.status != 0) return lr.status;
  const Program& P = *lr.program;
  std::vector<Span> sel;
  uint64_t cur = 0;  // smallest allowed start
  bool have_prev = false;
  uint64_t prev_end = 0;
  for (uint64_t s = 0; s <= n; s++) {
    if (s < cur) continue;
    if (!candidate(P, text, n, s)) continue;
    uint64_t e;
    if (!longest_at(P, text, n, s, &e)) continue;
    cur = e > s ? e : s + 1;
    if (e == s && have_prev && prev_end == s) {
      // zero-length rule, src/codegen.cc:65-73
    } else {
      sel.push_back({s, e});
    }
    have_prev = true;
    prev_end = e;
  }
  for (size_t i = 0; i < sel.size() && i < cap; i++) {
    out[2 * i] = sel[i].begin;
    out[2 * i + 1] = sel[i].end;
  }
  return (long)sel.size();
}

int pe_match_full(const char* re, const uint8_t* text, uint64_t n) {
  LowerResult lr = lower(re);
  if (lr.status != 0) return lr.status;
  // anchored at 0, must end at n: simulate keeping only "accept at n"
  const Program& P = *lr.program;
  const int W = P.n_words;
  int ctx = context_at(text, n, 0);
  if (n == 0) return P.nullable[ctx] ? 1 : 0;
  if (P.n_pos == 0) return 0;
  std::vector<uint32_t> S(W), T(W);
  const uint32_t* row = &P.cls[(size_t)text[0] * W];
  bool any = false;
  for (int k = 0; k < W; k++) {
    S[k] = P.first[ctx][k] & row[k];
    any |= S[k] != 0;
  }
  for (uint64_t p = 1; any; p++) {
    ctx = context_at(text, n, p);
    if (p == n) {
      for (int k = 0; k < W; k++)
        if (S[k] & P.last[ctx][k]) return 1;
      return 0;
    }
    uint32_t carry = 0;
    for (int k = 0; k < W; k++) {
      uint32_t lin = S[k] & P.linear[k];
      T[k] = (lin << 1) | carry;
      carry = lin >> 31;
    }
    for (int k = 0; k < W; k++) {
      uint32_t sp = S[k] & ~P.linear[k];
      while (sp) {
        int b = __builtin_ctz(sp);
        sp &= sp - 1;
        const uint32_t* fr = &P.rows[ctx][(size_t)P.row_of[(size_t)k * 32 + b] * W];
        for (int j = 0; j < W; j++) T[j] |= fr[j];
      }
    }
    row = &P.cls[(size_t)text[p] * W];
    any = false;
    for (int k = 0; k < W; k++) {
      S[k] = T[k] & row[k];
      any |= S[k] != 0;
    }
  }
  return 0;
}

// plan introspection: mode (0 dense, 1 windows), window count, offset, P, min_len, max_len
int pe_plan(const char* re, uint64_t* info, uint32_t* window_values) {
  LowerResult lr = lower(re);
  if (lr.status != 0) return lr.status;
  const Program& P = *lr.program;
  info[0] = P.mode == ScanMode::Windows;
  info[1] = P.windows.size();
  info[2] = P.windows.empty() ? 0 : P.windows[0].offset;
  info[3] = (uint64_t)P.n_pos;
  info[4] = P.min_len;
  info[5] = P.max_len;
  info[6] = (uint64_t)P.n_rows;
  info[7] = P.literal.size();
  info[8] = P.floating;
  info[11] = P.behind;
  info[9] = P.float_min;
  info[10] = P.float_max;
  for (size_t i = 0; i < P.windows.size(); i++) {
    window_values[4 * i] = P.windows[i].value0;
    window_values[4 * i + 1] = P.windows[i].mask0;
    window_values[4 * i + 2] = P.windows[i].value1;
    window_values[4 * i + 3] = P.windows[i].mask1;
  }
  return 0;
}


// exact_count.h: the plan of MatchAllCount-in-one-kernel for `n_rx` patterns and `n_bases` base windows (8 bytes each).
// 1 = the set has the shape (table_out: kExactTabWords words, base_out: lo[2], hi[2]), 0 = refused, < 0 = a pattern
// does not compile.
int pe_exact_plan(const char* const* rxs, int n_rx, const uint8_t* bases, int n_bases, uint32_t* table_out, uint32_t* base_out) {
  std::vector<LowerResult> lowered;
  std::vector<const Program*> progs;
  for (int i = 0; i < n_rx; i++) {
    lowered.push_back(lower(rxs[i]));
    if (lowered.back().status != 0) return -1 - i;
  }
  for (auto& l : lowered) progs.push_back(l.program.get());
  uint8_t b[2][8] = {};
  for (int k = 0; k < n_bases && k < 2; k++) memcpy(b[k], bases + 8 * k, 8);
  ExactCountPlan plan;
  make_exact_count_plan(progs, b, static_cast<uint32_t>(n_bases), &plan);
  if (!plan.ok) return 0;
  memcpy(table_out, plan.table, sizeof(plan.table));
  base_out[0] = plan.base_lo[0];
  base_out[1] = plan.base_lo[1];
  base_out[2] = plan.base_hi[0];
  base_out[3] = plan.base_hi[1];
  return 1;
}

// the kernel's per-candidate code (exact_classify) on 8 bytes: bit p = pattern p matches them
uint32_t pe_exact_classify(const uint32_t* table, const uint32_t* base, int n_bases, const uint8_t* bytes8) {
  uint32_t lo, hi;
  memcpy(&lo, bytes8, 4);
  memcpy(&hi, bytes8 + 4, 4);
  return n_bases > 1 ? exact_classify<2>(table, base, base + 2, lo, hi) : exact_classify<1>(table, base, base + 2, lo, hi);
}

// run_scan.h on the CPU: the plan (make_run_plan: which patterns have ONE long-lived thread in one loop position; its classes as
// byte ranges or complements) and the segment rule the kernels implement -- per segment between two breaks the first A and the
// last B behind it -- walked byte by byte.  Returns the number of matches, -101 when the pattern does not have the shape, a
// negative lowering status; shape[0..3]: has_b, n_ranges, the three classes' complement flags packed, 0.
long pe_run_match_all(const char* re, const uint8_t* text, uint64_t n, uint64_t* out, uint64_t cap, uint32_t* shape) {
  LowerResult lr = lower(re);
  if (lr.status != 0) return lr.status;
  const RunPlan pl = make_run_plan(*lr.program);
  if (!pl.ok) return -101;
  shape[0] = pl.has_b;
  shape[1] = pl.n_ranges;
  shape[2] = pl.a_neg | (pl.l_neg << 1) | (pl.b_neg << 2);
  shape[3] = 0;
  // a byte's membership from the plan's encoding (what rj_stream_range computes in the kernel)
  auto in_class = [&](uint8_t c, uint32_t ranges, uint32_t neg) {
    bool in = false;
    for (uint32_t r = 0; r < pl.n_ranges; r++) {
      if (!((ranges >> r) & 1u)) continue;
      const uint32_t lo = 0x80u - (pl.add_lo[r] & 0xFFu), hi = 0x7fu - (pl.add_hi[r] & 0xFFu);
      const uint32_t half = (pl.high_half >> r) & 1u;
      if ((static_cast<uint32_t>(c) >> 7) == half && (c & 0x7fu) >= lo && (c & 0x7fu) <= hi) in = true;
    }
    return in != (neg != 0);
  };
  uint64_t k = 0;
  const uint64_t none = ~0ull;
  uint64_t s1 = none, q = none;
  bool prev_a = false;   // lag plans (`A L+`): a start is the MARK "A at p - 1 and L at p"; the match begins one byte before it
  shape[3] = pl.lag;
  for (uint64_t p = 0; p <= n; p++) {
    const bool at_end = p == n;
    const uint8_t c = at_end ? 0 : text[p];
    const bool brk = at_end || !in_class(c, pl.l_ranges, pl.l_neg);
    const bool is_a = !at_end && in_class(c, pl.a_ranges, pl.a_neg);
    const bool start_here = pl.lag ? (prev_a && !brk) : is_a;
    prev_a = is_a;
    if (brk) {
      // B may be the break itself; a start lies before it
      if (!at_end && pl.has_b && s1 != none && in_class(c, pl.b_ranges, pl.b_neg)) q = p;
      if (s1 != none && (!pl.has_b || q != none)) {
        const uint64_t mb = s1 - pl.lag, me = pl.has_b ? q + 1 : p;
        // `^` / `$` (RunPlan::bol / eol): the match stays when it begins at a line start / ends at a line end
        const bool at_bol = mb == 0 || text[mb - 1] == '\n' || text[mb - 1] == '\r';
        const bool at_eol = me == n || text[me] == '\n' || text[me] == '\r';
        if ((!pl.bol || at_bol) && (!pl.eol || at_eol)) {
          if (k < cap) {
            out[2 * k] = mb;
            out[2 * k + 1] = me;
          }
          k++;
        }
      }
      s1 = none;
      q = none;
      if (start_here) s1 = p;   // the next segment's starts begin AT the break (never a mark: a mark is an L byte)
      continue;
    }
    if (pl.has_b && s1 != none && in_class(c, pl.b_ranges, pl.b_neg)) q = p;
    if (s1 == none && start_here) s1 = p;
  }
  return static_cast<long>(k);
}

// The PAIR shape of run_scan.h (`"[^"]*"`: the same class Q at both ends, no Q inside L) on the CPU: the plan's class encoding and the
// rule the pair kernels implement -- the matches are the pairs (1st, 2nd), (3rd, 4th) ... of the Q bytes since the last RESET (a
// break that is no Q; the text's end) -- walked byte by byte.  -101: the pattern does not have the shape.
long pe_pair_match_all(const char* re, const uint8_t* text, uint64_t n, uint64_t* out, uint64_t cap) {
  LowerResult lr = lower(re);
  if (lr.status != 0) return lr.status;
  const RunPlan pl = make_run_plan(*lr.program);
  if (!pl.pair || pl.ok) return -101;
  auto in_class = [&](uint8_t c, uint32_t ranges, uint32_t neg) {
    bool in = false;
    for (uint32_t r = 0; r < pl.n_ranges; r++) {
      if (!((ranges >> r) & 1u)) continue;
      const uint32_t lo = 0x80u - (pl.add_lo[r] & 0xFFu), hi = 0x7fu - (pl.add_hi[r] & 0xFFu);
      const uint32_t half = (pl.high_half >> r) & 1u;
      if ((static_cast<uint32_t>(c) >> 7) == half && (c & 0x7fu) >= lo && (c & 0x7fu) <= hi) in = true;
    }
    return in != (neg != 0);
  };
  uint64_t k = 0;
  bool open = false;
  uint64_t at = 0;
  for (uint64_t p = 0; p < n; p++) {
    const uint8_t c = text[p];
    if (in_class(c, pl.a_ranges, pl.a_neg)) {
      if (open) {
        if (k < cap) {
          out[2 * k] = at;
          out[2 * k + 1] = p + 1;
        }
        k++;
        open = false;
      } else {
        open = true;
        at = p;
      }
    } else if (!in_class(c, pl.l_ranges, pl.l_neg)) {
      open = false;
    }
  }
  return static_cast<long>(k);
}

}  // extern "C"
