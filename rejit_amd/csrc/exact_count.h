// rejit_amd/csrc/exact_count.h -- MatchAllCount (reference src/rejit.cc:203-208: the caller wants the NUMBER of
// matches; sample/regexdna.cc:65 asks nothing else of its nine patterns) for the pattern sets the bit-plane scan
// serves, decided by ONE table lookup per candidate instead of an automaton walk.
//
// Shape (checked on the host, make_exact_count_plan): every pattern matches exactly 8 bytes, has no assertions and
// is not at risk of the ring artefact, and its language lies entirely within ONE byte of one of the scan's <= 2
// base windows -- `agggtaaa|tttaccct`, `[cgt]gggtaaa|tttaccc[acg]`, ...: k-mers with one degenerate position, both
// strands.  The plane scan (plane_count.hip) finds the text positions whose 8 bytes differ from a base in at most
// one 2-bit symbol code; a candidate is then a match of pattern p iff its 8 bytes differ from base b in at most one
// BYTE -- position j, byte c -- and the string "base b with c at j" is in p's language.  That predicate is a table:
//     tab[b][j][cid[c]] = bit mask of the patterns that accept it      (j = 8: the base itself)
// where cid folds the 256 byte values into <= 16 classes that no pattern tells apart.  The table is built by running
// every pattern's position automaton (lowering.h: Program) over the 2 x 8 x 256 one-off strings, and the plan is only
// valid when that enumerates each language COMPLETELY: the number of accepted one-off strings must equal the size of
// the language, counted by a subset walk over the automaton.  So the counts are exact by construction, not a filter.
//
// The reference's answer for such a pattern is the left-most-longest, non-overlapping selection
// (src/x64/codegen-x64.cc:401-466,469-522).  With every match 8 bytes long that selection drops a match only when
// another match OF THE SAME PATTERN begins fewer than 8 bytes before it; the kernel detects that case (it needs two
// candidates within 8 bytes of each other) and voids the run, which the host then repeats with the span pipeline.
//
// Host part: plain C++ (g++ compiles it for the CPU tests, tests/support/program_exec.cc).  exact_classify is the code
// the kernel runs per lane (RJ_HD).
#ifndef REJIT_AMD_EXACT_COUNT_H_
#define REJIT_AMD_EXACT_COUNT_H_

#include <stdint.h>

#include "device_program.h"

namespace rejit_amd {

constexpr int kExactMaxCid = 16;
constexpr int kExactMaxPatterns = 32;                                   // bits of a table entry
constexpr uint32_t kExactCidWords = 64;                                 // cid[256], one byte each
constexpr uint32_t kExactTabWords = kExactCidWords + 2 * 9 * kExactMaxCid;  // + tab[2][9][16]

// The patterns (bit p: pattern p) that match the 8 bytes (lo, hi) -- little endian, lo = bytes 0..3.
// table: cid bytes, then tab[b][j][cid] (LDS in the kernel).
template <int NB>
RJ_HD uint32_t exact_classify(const uint32_t* table, const uint32_t* base_lo, const uint32_t* base_hi, uint32_t lo, uint32_t hi) {
  const uint8_t* cid = reinterpret_cast<const uint8_t*>(table);
  const uint32_t* tab = table + kExactCidWords;
  const uint64_t w = (static_cast<uint64_t>(hi) << 32) | lo;
  uint32_t mask = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int b = 0; b < NB; b++) {
    const uint64_t x = w ^ ((static_cast<uint64_t>(base_hi[b]) << 32) | base_lo[b]);
    const uint32_t tz = x ? (static_cast<uint32_t>(__builtin_ctzll(x)) & 56u) : 0u;  // bit offset of the lowest differing byte
    const bool at_most_one = (x >> tz) < 256u;                                      // ... which is the only one
    const uint32_t c = static_cast<uint32_t>(w >> tz) & 0xFFu;
    const uint32_t j = x ? (tz >> 3) : 8u;
    const uint32_t e = tab[(static_cast<uint32_t>(b) * 9u + j) * kExactMaxCid + cid[c]];
    mask |= at_most_one ? e : 0u;
  }
  return mask;
}

}  // namespace rejit_amd

// ----------------------------------------------------------------------------------------------- host part
#include <map>
#include <vector>

#include "lowering.h"

namespace rejit_amd {

struct ExactCountPlan {
  bool ok = false;
  uint32_t n_bases = 0, n_patterns = 0;
  uint32_t base_lo[2] = {0, 0}, base_hi[2] = {0, 0};
  uint32_t table[kExactTabWords] = {};
};

namespace exact_detail {

using Set = std::vector<uint32_t>;

inline Set follow(const Program& P, const Set& S) {
  const size_t W = static_cast<size_t>(P.n_words);
  Set T(W, 0u);
  for (size_t k = 0; k < W; k++) {  // linear positions pass to i + 1
    const uint32_t l = S[k] & P.linear[k];
    T[k] |= l << 1;
    if (k + 1 < W) T[k + 1] |= l >> 31;
  }
  for (int i = 0; i < P.n_pos; i++) {
    if (!((S[static_cast<size_t>(i) >> 5] >> (i & 31)) & 1u)) continue;
    const int r = P.row_of[static_cast<size_t>(i)];
    if (r < 0) continue;
    for (size_t k = 0; k < W; k++) T[k] |= P.rows[0][static_cast<size_t>(r) * W + k];
  }
  return T;
}

inline bool empty(const Set& S) {
  for (uint32_t v : S)
    if (v) return false;
  return true;
}

inline Set consume(const Program& P, const Set& from, uint8_t c) {
  const size_t W = static_cast<size_t>(P.n_words);
  Set S(W);
  for (size_t k = 0; k < W; k++) S[k] = from[k] & P.cls[static_cast<size_t>(c) * W + k];
  return S;
}

inline bool accepting(const Program& P, const Set& S) {
  for (size_t k = 0; k < S.size(); k++)
    if (S[k] & P.last[0][k]) return true;
  return false;
}

// does the automaton accept exactly these 8 bytes (context 0: the patterns have no assertions)
inline bool accepts8(const Program& P, const uint8_t* s) {
  Set S = consume(P, P.first[0], s[0]);
  for (int k = 1; k < 8 && !empty(S); k++) S = consume(P, follow(P, S), s[k]);
  return !empty(S) && accepting(P, S);
}

// number of 8-byte strings the automaton accepts (subset walk; bytes with the same class column walk together);
// false when the walk grows beyond `limit` distinct position sets
inline bool language_size8(const Program& P, size_t limit, unsigned __int128* out) {
  const size_t W = static_cast<size_t>(P.n_words);
  std::map<std::vector<uint32_t>, int> column_id;   // class column -> group
  std::vector<uint8_t> rep;                         // a byte of every group
  std::vector<uint32_t> mult;                       // bytes per group
  for (int c = 0; c < 256; c++) {
    std::vector<uint32_t> col(P.cls.begin() + static_cast<long>(static_cast<size_t>(c) * W), P.cls.begin() + static_cast<long>((static_cast<size_t>(c) + 1) * W));
    auto it = column_id.find(col);
    if (it == column_id.end()) {
      column_id.emplace(col, static_cast<int>(rep.size()));
      rep.push_back(static_cast<uint8_t>(c));
      mult.push_back(1);
    } else {
      mult[static_cast<size_t>(it->second)]++;
    }
  }
  std::map<Set, unsigned __int128> cur;
  cur[P.first[0]] = 1;   // the set the first byte is taken from
  for (int k = 0; k < 8; k++) {
    std::map<Set, unsigned __int128> nxt;
    for (const auto& kv : cur) {
      const Set from = k == 0 ? kv.first : follow(P, kv.first);
      for (size_t g = 0; g < rep.size(); g++) {
        Set S = consume(P, from, rep[g]);
        if (empty(S)) continue;
        nxt[S] += kv.second * mult[g];
      }
    }
    if (nxt.size() > limit) return false;
    cur.swap(nxt);
  }
  unsigned __int128 total = 0;
  for (const auto& kv : cur)
    if (accepting(P, kv.first)) total += kv.second;
  *out = total;
  return true;
}

}  // namespace exact_detail

// progs: the patterns of the set; bases: the plane scan's base windows (8 bytes each).  plan->ok == false: the set
// does not have the shape (the caller keeps the span pipeline).
inline void make_exact_count_plan(const std::vector<const Program*>& progs, const uint8_t (*bases)[8], uint32_t n_bases, ExactCountPlan* plan) {
  using namespace exact_detail;
  *plan = ExactCountPlan{};
  const size_t P = progs.size();
  if (n_bases < 1 || n_bases > 2 || P < 1 || P > static_cast<size_t>(kExactMaxPatterns)) return;
  if (n_bases == 2) {  // the one-off neighbourhoods of the bases must be disjoint (a string is counted once)
    int d = 0;
    for (int i = 0; i < 8; i++) d += bases[0][i] != bases[1][i];
    if (d < 3) return;
  }
  for (const Program* p : progs)
    if (!p || p->has_assertions || p->q8_risk || p->min_len != 8 || p->max_len != 8 || p->n_pos < 1 || p->any_nullable) return;
  // accepted[b][j][c]: patterns that accept base b with byte c at position j; base_mask[b]: ... the base itself
  std::vector<uint32_t> accepted(static_cast<size_t>(n_bases) * 8 * 256, 0u);
  uint32_t base_mask[2] = {0, 0};
  std::vector<unsigned __int128> one_off(P, 0);
  for (size_t p = 0; p < P; p++) {
    for (uint32_t b = 0; b < n_bases; b++) {
      uint8_t s[8];
      for (int i = 0; i < 8; i++) s[i] = bases[b][i];
      if (accepts8(*progs[p], s)) {
        base_mask[b] |= 1u << p;
        one_off[p] += 1;
      }
      for (int j = 0; j < 8; j++) {
        for (int c = 0; c < 256; c++) {
          if (c == bases[b][j]) continue;
          s[j] = static_cast<uint8_t>(c);
          if (accepts8(*progs[p], s)) {
            accepted[(static_cast<size_t>(b) * 8 + static_cast<size_t>(j)) * 256 + static_cast<size_t>(c)] |= 1u << p;
            one_off[p] += 1;
          }
        }
        s[j] = bases[b][j];
      }
    }
    unsigned __int128 size = 0;
    if (!language_size8(*progs[p], 4096, &size) || size != one_off[p]) return;  // some match lies further from the bases
  }
  // byte classes: bytes no (base, position) tells apart
  std::map<std::vector<uint32_t>, int> sig_id;
  uint8_t* cid = reinterpret_cast<uint8_t*>(plan->table);
  uint32_t* tab = plan->table + kExactCidWords;
  for (int c = 0; c < 256; c++) {
    std::vector<uint32_t> sig;
    for (uint32_t b = 0; b < n_bases; b++)
      for (int j = 0; j < 8; j++) {
        // (the base's own byte at j: the string is the base, which row 8 answers -- any entry will do, take row 8's)
        sig.push_back(c == bases[b][j] ? base_mask[b] : accepted[(static_cast<size_t>(b) * 8 + static_cast<size_t>(j)) * 256 + static_cast<size_t>(c)]);
      }
    auto it = sig_id.find(sig);
    if (it == sig_id.end()) {
      if (sig_id.size() >= static_cast<size_t>(kExactMaxCid)) return;
      const int id = static_cast<int>(sig_id.size());
      sig_id.emplace(sig, id);
      for (uint32_t b = 0; b < n_bases; b++)
        for (int j = 0; j < 8; j++) tab[(b * 9 + static_cast<uint32_t>(j)) * kExactMaxCid + static_cast<uint32_t>(id)] = sig[b * 8 + static_cast<uint32_t>(j)];
      cid[c] = static_cast<uint8_t>(id);
    } else {
      cid[c] = static_cast<uint8_t>(it->second);
    }
  }
  for (uint32_t b = 0; b < n_bases; b++) {
    for (int id = 0; id < kExactMaxCid; id++) tab[(b * 9 + 8) * kExactMaxCid + static_cast<uint32_t>(id)] = base_mask[b];
    plan->base_lo[b] = static_cast<uint32_t>(bases[b][0]) | static_cast<uint32_t>(bases[b][1]) << 8 | static_cast<uint32_t>(bases[b][2]) << 16 |
                       static_cast<uint32_t>(bases[b][3]) << 24;
    plan->base_hi[b] = static_cast<uint32_t>(bases[b][4]) | static_cast<uint32_t>(bases[b][5]) << 8 | static_cast<uint32_t>(bases[b][6]) << 16 |
                       static_cast<uint32_t>(bases[b][7]) << 24;
  }
  plan->n_bases = n_bases;
  plan->n_patterns = static_cast<uint32_t>(P);
  plan->ok = true;
}

}  // namespace rejit_amd
#endif
