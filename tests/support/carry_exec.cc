// tests/support/carry_exec.cc -- TEST-ONLY driver of the linear-time carry scan on the CPU.
//
// Compiles rejit_amd/csrc/carry_scan.h -- the very bodies the HIP kernels call per sub-chunk -- with
// g++ and runs the phases one sub-chunk after the other, so the algorithm (reverse automaton,
// class stepping, symbolic summaries, chain selection) is checked against the oracle without a GPU.
// Never linked into the product library.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../rejit_amd/csrc/behind_walk.h"
#include "../../rejit_amd/csrc/carry_scan.h"
#include "../../rejit_amd/csrc/dense_streams.h"
#include "../../rejit_amd/csrc/dense_swar.h"
#include "../../rejit_amd/csrc/exact_replay.h"
#include "../../rejit_amd/csrc/lowering.h"
#include "../../rejit_amd/csrc/table_layout.h"

using namespace rejit_amd;

namespace {

template <int NW>
long run(const Program& P, const DevProgram& R, const uint8_t* t, uint64_t n, uint64_t sub, uint64_t sb, uint64_t se,
         uint64_t carry_cur, uint64_t carry_prev_end, int have_prev, uint64_t* out, uint64_t cap) {
  if (se > n + 1) se = n + 1;
  if (sb >= se) return 0;
  const int np = std::max(P.n_pos, 1), W = P.n_words;
  const uint64_t c_first = sb / sub, n_sub = n / sub + 1;  // sub-chunk c: bytes [c*sub, min((c+1)*sub, n))
  const uint64_t e_base = c_first * sub;
  std::vector<uint64_t> cval(static_cast<size_t>(np));
  std::vector<uint32_t> cset(static_cast<size_t>(np) * NW), src(static_cast<size_t>(np) * NW);
  CsClasses<NW> C{CsArr<uint64_t>{cval.data(), 1}, CsArr<uint32_t>{cset.data(), 1}, 0};
  const CsArr<uint32_t> srcarr{src.data(), 1};
  const uint64_t m = n_sub - c_first;
  std::vector<uint64_t> D(static_cast<size_t>(m + 1) * np, 0);  // [m] = beyond the text: zeros
  std::vector<uint32_t> Rm(static_cast<size_t>(m) * np * W, 0);
  for (uint64_t i = 0; i < m; i++) {
    const uint64_t a = (c_first + i) * sub, b = std::min(a + sub, n);
    cs_summarize<NW>(R, t, n, a, b, C, srcarr, &D[i * np], &Rm[i * np * W]);
  }
  {
    // two-level resolve as on the device: groups of `grp` sub-chunks composed into one transfer each,
    // a sequential pass over the groups, then every group's own sub-chunks
    const uint64_t grp = 3, ng = (m + grp - 1) / grp;
    std::vector<uint64_t> gD(static_cast<size_t>(ng + 1) * np, 0), la(np), lb(np);
    std::vector<uint32_t> gR(static_cast<size_t>(ng) * np * W, 0), ra(static_cast<size_t>(np) * W), rb(static_cast<size_t>(np) * W);
    for (uint64_t g = 0; g < ng; g++) {
      std::fill(la.begin(), la.end(), 0);
      std::fill(ra.begin(), ra.end(), 0u);
      for (int k = 0; k < P.n_pos; k++) ra[static_cast<size_t>(k) * W + (k >> 5)] = 1u << (k & 31);  // identity
      for (uint64_t i = std::min(m, (g + 1) * grp); i-- > g * grp;) {
        cs_compose(P.n_pos, W, &D[i * np], &Rm[i * np * W], la.data(), ra.data(), lb.data(), rb.data());
        la.swap(lb);
        ra.swap(rb);
      }
      std::copy(la.begin(), la.end(), gD.begin() + static_cast<long>(g * np));
      std::copy(ra.begin(), ra.end(), gR.begin() + static_cast<long>(g * np * W));
    }
    for (uint64_t g = ng; g-- > 0;) cs_resolve(P.n_pos, W, &gD[g * np], &gR[g * np * W], &gD[(g + 1) * np]);
    for (uint64_t g = 0; g < ng; g++) {
      const uint64_t last = std::min(m, (g + 1) * grp) - 1;
      for (uint64_t i = last + 1; i-- > g * grp;)
        cs_resolve(P.n_pos, W, &D[i * np], &Rm[i * np * W], i == last ? &gD[(g + 1) * np] : &D[(i + 1) * np]);
    }
  }
  const uint64_t n_own = (se - 1) / sub - c_first + 1;  // sub-chunks that hold own starts
  std::vector<uint64_t> E(static_cast<size_t>(n_own * sub), kCsNone), G(static_cast<size_t>(n_own * sub), 0);
  for (uint64_t i = 0; i < n_own; i++) {
    const uint64_t a = (c_first + i) * sub, b = std::min(a + sub, n);
    cs_emit<NW>(R, t, n, a, b, sb, se, i + 1 < m ? &D[(i + 1) * np] : nullptr, C, E.data(), e_base);
  }
  auto bounds = [&](uint64_t i, uint64_t* lo, uint64_t* hi) {
    *lo = std::max((c_first + i) * sub, sb);
    *hi = std::min((c_first + i + 1) * sub, se);
  };
  for (uint64_t i = 0; i < n_own; i++) {
    uint64_t lo, hi;
    bounds(i, &lo, &hi);
    cs_local_chain(E.data(), G.data(), e_base, lo, hi);
  }
  std::vector<uint64_t> entry(static_cast<size_t>(n_own), kCsNone);
  for (uint64_t cur = std::max(carry_cur, sb); cur < se;) {
    entry[cur / sub - c_first] = cur;
    cur = G[cur - e_base];
  }
  RjSelectState st{carry_cur, carry_prev_end, have_prev != 0};
  uint64_t out_n = 0;
  for (uint64_t i = 0; i < n_own; i++) {
    if (entry[i] == kCsNone) continue;
    uint64_t lo, hi;
    bounds(i, &lo, &hi);
    const uint32_t cnt = cs_take(E.data(), G.data(), e_base, i * sub, hi, entry[i]);
    for (uint32_t k = 0; k < cnt; k++) {
      const uint64_t bb = G[i * sub + k], ee = E[i * sub + k];
      bool taken;
      if (rj_select_step(&st, bb, ee, &taken)) {  // the zero-length rule (the device tail applies it)
        if (out_n < cap) {
          out[2 * out_n] = bb;
          out[2 * out_n + 1] = ee;
        }
        out_n++;
      }
      if (!taken) return -8;  // the chain must only hold takeable matches
    }
  }
  return static_cast<long>(out_n);
}

}  // namespace

// The candidates of the behind mode through rejit_amd/csrc/behind_walk.h (what verify_behind_in_regions
// runs per hit), then the selection with its conflict rule.  Returns the count, -100 on a conflict,
// -101 when the pattern's plan is not `behind`, -102 when a walk hit the limit.
template <int NW, int NQ>
static long behind_run(const DevProgram& P, const DevProgram& R, const uint8_t* t, uint64_t n, uint64_t* out, uint64_t cap) {
  struct Span { uint64_t b, e; };
  std::vector<Span> c;
  bool overrun = false;
  for (uint64_t w = 0; w < n; w++) {
    uint64_t b = 0, e = 0;
    if (rj_behind_candidate<NW, NQ>(P, R, t, n, w, &b, &e, &overrun)) c.push_back({b, e});
  }
  if (overrun) return -102;
  std::stable_sort(c.begin(), c.end(), [](const Span& x, const Span& y) { return x.b < y.b; });
  uint64_t cur = 0, k = 0;
  for (size_t i = 0; i < c.size(); i++) {
    if (i > 0 && c[i].b == c[i - 1].b) continue;
    if (c[i].b < cur) {
      if (c[i].e > cur) return -100;
      continue;
    }
    if (k < cap) {
      out[2 * k] = c[i].b;
      out[2 * k + 1] = c[i].e;
    }
    k++;
    cur = c[i].e > c[i].b ? c[i].e : c[i].b + 1;
  }
  return static_cast<long>(k);
}


// The exact replay (rejit_amd/csrc/exact_replay.h) as exact_replay.hip drives it: ownership
// [first sync >= sb, first sync >= se), chunks of `chunk` bytes each reporting its first proven
// synchronisation point, one replay of the reference's loop per segment.  Returns the count; -9 when the
// automaton is too wide; *n_segments = segments replayed, *longest = the longest of them.
template <int NQ>
static long exact_run(const Program& P, const DevProgram& F, const DevGraph& G, const uint8_t* t, uint64_t n, uint64_t chunk,
                      uint64_t sb, uint64_t se, uint64_t* out, uint64_t cap, uint64_t* n_segments, uint64_t* longest) {
  (void)P;
  if (se > n + 1) se = n + 1;
  const uint64_t y0 = rj_first_sync<NQ>(F, t, n, sb);
  const uint64_t y1 = se > n ? n + 1 : rj_first_sync<NQ>(F, t, n, se);
  std::vector<uint64_t> syncs;
  for (uint64_t c0 = y0; c0 < y1; c0 += chunk) {
    const uint64_t s = rj_chunk_first_sync<NQ>(F, t, n, c0, std::min(c0 + chunk, y1), c0 == y0);
    if (s != kNoSync) syncs.push_back(s);
  }
  std::vector<int64_t> ring(static_cast<size_t>(G.n_states) * G.times);
  uint64_t k = 0;
  *n_segments = syncs.size();
  *longest = 0;
  for (size_t i = 0; i < syncs.size(); i++) {
    const uint64_t a = syncs[i], b = i + 1 < syncs.size() ? syncs[i + 1] : y1;
    *longest = std::max(*longest, b - a);
    std::vector<uint64_t> seg(2 * (b - a) + 2);
    int64_t* r = ring.data();
    const uint64_t m = rj_replay_segment(G, t, n, a, b, [r](int i2) -> int64_t& { return r[i2]; }, seg.data());
    for (uint64_t j = 0; j < m; j++, k++)
      if (k < cap) {
        out[2 * k] = seg[2 * j];
        out[2 * k + 1] = seg[2 * j + 1];
      }
  }
  return static_cast<long>(k);
}


// One segment [a, b) the way exact_replay.hip takes a LONG one (exact_replay.h, "speculate and verify"): parts of `sub`
// bytes; round 0 replays each from `warm` bytes before its beginning with a free ring (the first from a) and notes the ring's
// ORDER PATTERN on entering the part; the walk carries the true ring over the parts, a pattern no part has been replayed from
// becomes a candidate and a new round; then the raw matches of every part from its verified pattern (inherited begins
// replaced by the true starts), every part's own sunk list, and the join.  *fixed = rounds beyond the first; returns ~0
// when there were more than kReplayMaxRounds (the device gives the segment up).
static uint64_t replay_segment_speculatively(const DevGraph& G, const uint8_t* t, uint64_t n, uint64_t a, uint64_t b, uint64_t sub,
                                             uint64_t warm, uint64_t* out, uint64_t* fixed, bool local_sink) {
  const size_t slots = static_cast<size_t>(G.n_states) * G.times;
  const uint64_t n_parts = (b - a + sub - 1) / sub;
  typedef std::vector<int64_t> Snap;
  std::vector<Snap> entry0(n_parts, Snap(slots, -1));                                      // order patterns
  std::vector<std::vector<Snap>> exits(1, std::vector<Snap>(n_parts, Snap(slots, -1)));   // [round][part]: ring values at the part's end
  std::vector<Snap> cand(1);                                                               // candidate patterns (0: the warm-up's own)
  std::vector<uint64_t> cand_from(1, 0);
  std::vector<int64_t> ring(slots), scratch(2 * slots);
  int64_t* r = ring.data();
  auto ring_fn = [r](int i) -> int64_t& { return r[i]; };
  auto part_start = [&](uint64_t i) { return a + i * sub; };
  auto part_stop = [&](uint64_t i) { return std::min(b, a + (i + 1) * sub); };
  auto warm_start = [&](uint64_t i) { const uint64_t c0 = part_start(i); return i == 0 ? a : std::max(a, c0 > warm ? c0 - warm : 0); };
  for (uint64_t i = 0; i < n_parts; i++)  // round 0 (in parallel on the device)
    rj_replay_raw(G, t, n, warm_start(i), static_cast<const int64_t*>(nullptr), part_start(i), part_stop(i), ring_fn, entry0[i].data(),
                  exits[0][i].data(), static_cast<uint64_t*>(nullptr), static_cast<const int64_t*>(nullptr), scratch.data());
  std::vector<int> chosen(n_parts, -1);
  std::vector<Snap> true_starts(n_parts);  // the real start offsets of the threads at each part's entry, oldest first
  Snap T(slots, -1);                       // the true ring at the beginning of the part the walk stands at (free at a)
  for (uint64_t i = 0; i < n_parts;) {
    // T's order pattern
    Snap sorted;
    for (size_t k = 0; k < slots; k++)
      if (T[k] >= 0) sorted.push_back(T[k]);
    std::sort(sorted.begin(), sorted.end());
    sorted.erase(std::unique(sorted.begin(), sorted.end()), sorted.end());
    Snap pat(slots, -1);
    for (size_t k = 0; k < slots; k++)
      if (T[k] >= 0) pat[k] = std::lower_bound(sorted.begin(), sorted.end(), T[k]) - sorted.begin();
    int k = -1;
    if (pat == entry0[i]) k = 0;
    for (size_t c = 1; c < cand.size() && k < 0; c++)
      if (i >= cand_from[c] && pat == cand[c]) k = static_cast<int>(c);
    if (k < 0) {  // a pattern nobody has replayed this part from: a new candidate, a new round from here on
      if (static_cast<int>(cand.size()) > kReplayMaxRounds) return ~0ull;
      cand.push_back(pat);
      cand_from.push_back(i);
      exits.push_back(std::vector<Snap>(n_parts, Snap(slots, -1)));
      for (uint64_t j = i; j < n_parts; j++)
        rj_replay_raw(G, t, n, part_start(j), pat.data(), part_start(j), part_stop(j), ring_fn, static_cast<int64_t*>(nullptr),
                      exits.back()[j].data(), static_cast<uint64_t*>(nullptr), static_cast<const int64_t*>(nullptr), scratch.data());
      (*fixed)++;
      continue;
    }
    chosen[i] = k;
    true_starts[i] = sorted;
    const int64_t c0 = static_cast<int64_t>(part_start(i)), cnt = static_cast<int64_t>(sorted.size());
    const Snap& x = exits[static_cast<size_t>(k)][i];
    for (size_t q = 0; q < slots; q++) T[q] = x[q] < 0 ? -1 : x[q] < c0 ? sorted[static_cast<size_t>(x[q] - (c0 - cnt))] : x[q];
    i++;
  }
  uint64_t out_n = 0;
  std::vector<uint64_t> raw(2 * sub + 2);
  std::vector<std::vector<uint64_t>> lists(n_parts);  // local_sink: every part's own list, then the join (xr_emit / xr_join)
  std::vector<uint64_t> min_begin(n_parts, ~0ull);
  for (uint64_t i = 0; i < n_parts; i++) {  // (in parallel on the device; then the sink, in order)
    const int k = chosen[i];
    const int64_t* ts = true_starts[i].empty() ? nullptr : true_starts[i].data();
    const uint64_t m = k == 0 ? rj_replay_raw(G, t, n, warm_start(i), static_cast<const int64_t*>(nullptr), part_start(i), part_stop(i), ring_fn,
                                              static_cast<int64_t*>(nullptr), static_cast<int64_t*>(nullptr), raw.data(), ts, scratch.data())
                              : rj_replay_raw(G, t, n, part_start(i), cand[static_cast<size_t>(k)].data(), part_start(i), part_stop(i), ring_fn,
                                              static_cast<int64_t*>(nullptr), static_cast<int64_t*>(nullptr), raw.data(), ts, scratch.data());
    if (!local_sink) {
      for (uint64_t j = 0; j < m; j++) out_n = rj_sink_append(out, out_n, static_cast<int64_t>(raw[2 * j]), static_cast<int64_t>(raw[2 * j + 1]));
      continue;
    }
    uint64_t kept = 0;
    for (uint64_t j = 0; j < m; j++) {
      const uint64_t pb = raw[2 * j], pe = raw[2 * j + 1];
      min_begin[i] = std::min(min_begin[i], pb);
      kept = rj_sink_append(raw.data(), kept, static_cast<int64_t>(pb), static_cast<int64_t>(pe));
    }
    lists[i].assign(raw.begin(), raw.begin() + static_cast<long>(2 * kept));
  }
  if (!local_sink) return out_n;
  std::vector<uint64_t> keep(n_parts, 0), first(n_parts, 0);
  std::vector<int64_t> prev(n_parts, -1);
  int64_t top_part = -1;
  for (uint64_t i = 0; i < n_parts; i++) {  // xr_join
    const uint64_t l = lists[i].size() / 2, mb = min_begin[i];
    while (top_part >= 0 && mb != ~0ull) {
      const size_t tp = static_cast<size_t>(top_part);
      uint64_t kk = keep[tp];
      while (kk > first[tp] && lists[tp][2 * (kk - 1)] >= mb) kk--;
      keep[tp] = kk;
      if (kk != first[tp]) break;
      top_part = prev[tp];
    }
    keep[i] = l;
    // the filter of the sink, for the one entry whose decision the part could not take on its own: its list's first, an empty
    // match right at the end of the entry before it (which belongs to an earlier part)
    if (l != 0 && top_part >= 0 && lists[i][0] == lists[i][1] && lists[i][0] == mb &&
        lists[static_cast<size_t>(top_part)][2 * (keep[static_cast<size_t>(top_part)] - 1) + 1] == lists[i][0])
      first[i] = 1;
    if (l != first[i]) {
      prev[i] = top_part;
      top_part = static_cast<int64_t>(i);
    }
  }
  for (uint64_t i = 0; i < n_parts; i++)
    for (uint64_t j = first[i]; j < keep[i]; j++) {
      out[2 * out_n] = lists[i][2 * j];
      out[2 * out_n + 1] = lists[i][2 * j + 1];
      out_n++;
    }
  return out_n;
}

static bool local_sink_always = true;  // (the device: xr_join for every pattern since the filter's one cross-part case is handled there)
template <int NQ>
static long exact_run_spec(const DevProgram& F, const DevGraph& G, const uint8_t* t, uint64_t n, uint64_t chunk, uint64_t sb, uint64_t se,
                           uint64_t sub, uint64_t warm, uint64_t* out, uint64_t cap, uint64_t* fixed) {
  if (se > n + 1) se = n + 1;
  const uint64_t y0 = rj_first_sync<NQ>(F, t, n, sb);
  const uint64_t y1 = se > n ? n + 1 : rj_first_sync<NQ>(F, t, n, se);
  std::vector<uint64_t> syncs;
  for (uint64_t c0 = y0; c0 < y1; c0 += chunk) {
    const uint64_t s = rj_chunk_first_sync<NQ>(F, t, n, c0, std::min(c0 + chunk, y1), c0 == y0);
    if (s != kNoSync) syncs.push_back(s);
  }
  uint64_t k = 0;
  for (size_t i = 0; i < syncs.size(); i++) {
    const uint64_t a = syncs[i], b = i + 1 < syncs.size() ? syncs[i + 1] : y1;
    std::vector<uint64_t> seg(2 * (b - a) + 2);
    uint64_t m = replay_segment_speculatively(G, t, n, a, b, sub, warm, seg.data(), fixed, local_sink_always || F.nullable == 0);
    if (m == ~0ull) {  // (given up: the device keeps the documented semantics; here the one sequential replay)
      std::vector<int64_t> ring(static_cast<size_t>(G.n_states) * G.times);
      int64_t* r = ring.data();
      m = rj_replay_segment(G, t, n, a, b, [r](int i2) -> int64_t& { return r[i2]; }, seg.data());
      *fixed += 1000000;
    }
    for (uint64_t j = 0; j < m; j++, k++)
      if (k < cap) {
        out[2 * k] = seg[2 * j];
        out[2 * k + 1] = seg[2 * j + 1];
      }
  }
  return static_cast<long>(k);
}

// dense_swar.h against the scalar automaton: every 16-byte block of the text (with the 4 bytes after it),
// every start of the block.  Returns the number of disagreements (0 expected), -101 when the pattern does
// not qualify for the packed pre-steps; stats: [0] starts checked, [1] sent to the walkers, [2] decided with
// a match, [3] depth.
template <int D>
static long swar_check(const Program& P, const SwarPlan& pl, const uint8_t* t, uint64_t n, uint64_t* stats) {
  long bad = 0;
  uint32_t loop[4], skip[4];
  loop_skip_masks(P, loop, skip);
  for (uint64_t o = 0; o + 20 <= n; o += 16) {
    uint32_t x[5], rows[5], H[4], walk, matched, in_first;
    memcpy(x, t + o, 20);
    rj_swar_rows5(pl, x, rows);
    const bool one_first = __builtin_popcount(P.first[0][0]) == 1;
    if (one_first) rj_swar_presteps<D, true>(pl, rows, &walk, &matched, H, &in_first);
    else rj_swar_presteps<D, false>(pl, rows, &walk, &matched, H, &in_first);
    {  // F layout -> one bit per start, and the layout helpers themselves
      const uint32_t w16 = rj_swar_f_to_starts(walk);
      if (rj_swar_f_from_starts(w16) != walk) bad++;
      if (rj_swar_f_to_starts(rj_swar_f_next(walk, 1)) != (((w16 << 1) | 1u) & 0xFFFFu)) bad++;
      if (rj_swar_f_last(walk) != ((w16 >> 15) & 1u)) bad++;
      walk = w16;
      matched = rj_swar_f_to_starts(matched);
      in_first = rj_swar_f_to_starts(in_first);
    }
    for (int j = 0; j < 16; j++) {
      // the scalar truth: D + 1 steps of the position automaton from start o + j
      uint32_t S = P.first[0][0] & P.cls[t[o + j]];
      const bool first_ok = S != 0;
      uint32_t hits = 0;
      bool gen = false;
      for (int k = 1; k <= D; k++) {
        if (S & P.last[0][0]) hits |= 1u << (k - 1);
        uint32_t T = 0;
        for (int p = 0; p < P.n_pos; p++) {
          if (!((S >> p) & 1u)) continue;
          const int r = P.row_of[static_cast<size_t>(p)];
          if (r < 0) T |= 1u << (p + 1);
          else {
            T |= P.rows[0][static_cast<size_t>(r)];
            if (!((loop[0] >> p) & 1u) && P.rows[0][static_cast<size_t>(r)] != 0) gen = true;
          }
        }
        S = T & P.cls[t[o + j + k]];
      }
      const uint32_t hb = (H[j >> 2] >> (8 * (j & 3))) & 0xFFu;
      stats[0]++;
      if (one_first && ((in_first >> j) & 1u) != (first_ok ? 1u : 0u)) bad++;
      if (gen) {
        if (!((walk >> j) & 1u)) bad++;
        stats[1]++;
        continue;
      }
      if (((walk >> j) & 1u) != (S != 0 ? 1u : 0u)) bad++;
      if (hb != hits || ((matched >> j) & 1u) != (hits != 0 ? 1u : 0u)) bad++;
      if ((walk >> j) & 1u) stats[1]++;
      else if (hits) stats[2]++;
    }
  }
  stats[3] = D;
  return bad;
}

// dense_streams.h, lane by lane on the CPU: MatchAll of the starts [sb, se) exactly as dense_streams.hip computes it --
// 32 bytes per lane, the start frame kStreamShift bytes back, class streams, position-major steps, the scalar walk for
// the starts still alive after the plan's depth -- the pairs in order.  Returns the count, -101 when the pattern does
// not qualify (make_stream_plan), a negative lowering status; stats: [0] starts decided in registers with a match,
// [1] starts handed to the scalar walk, [2] disagreements between the register steps and the scalar automaton
// (start by start: candidate flag, "alive after depth bytes", longest length), [3] the plan's depth, [4] loop_first.
template <int NP>
static long stream_run(const Program& P, const DevProgram& F, const StreamPlan& pl, const uint8_t* t, uint64_t n, uint64_t sb, uint64_t se,
                       uint64_t* out, uint64_t cap, uint64_t* stats) {
  const StreamMasks<NP> mk = rj_stream_masks<NP>(pl);
  const StreamRangeMasks<NP, kStreamMaxRanges> rm = rj_stream_range_masks<NP, kStreamMaxRanges>(pl);
  uint64_t k = 0;
  if (se > n + 1) se = n + 1;
  const uint64_t lim = se < n ? se : n;  // starts s < lim
  uint32_t Sb[NP];
  for (int q = 0; q < NP; q++) Sb[q] = 0;
  auto any = [](uint32_t x) { return x != 0; };
  // plans with `select`: the lanes' matches are kept and the kernel's selection is replayed behind the loop
  struct LaneMatches {
    uint32_t take, len[4];
  };
  std::vector<LaneMatches> lanes;
  for (uint64_t at = 0; lim > 0 && at <= lim - 1 + kStreamShift; at += 32) {
    uint32_t x[8] = {0, 0, 0, 0, 0, 0, 0, 0}, valid = 0, S[NP];
    for (int j = 0; j < 32; j++)
      if (at + j < n) {
        x[j >> 2] |= static_cast<uint32_t>(t[at + j]) << (8 * (j & 3));
        valid |= 1u << j;
      }
    if (pl.high_half) rj_stream_classes<NP, kStreamMaxRanges, true>(pl, rm, x, valid, S);
    else rj_stream_classes<NP, kStreamMaxRanges, false>(pl, rm, x, valid, S);
    uint32_t start_mask = 0;
    for (int j = 0; j < 32; j++) {
      const uint64_t p = at + j;  // start p - kStreamShift
      if (p >= kStreamShift && p - kStreamShift >= sb && p - kStreamShift < lim) start_mask |= 1u << j;
    }
    uint32_t matched, alive, len[4], cand;
    rj_stream_steps<NP>(pl, mk, S, Sb, start_mask, any, &matched, &alive, len, &cand);
    if (pl.run_shape != 0) {
      // the run form of the steps (rj_stream_runs, round 6) against the generic steps: the same candidates, the same undecided
      // starts, the same matches with the same lengths
      uint64_t starts = 0, ends = 0;
      uint32_t alive_r = 0;
      rj_stream_runs(pl.run_shape, S[0], Sb[0], NP > 1 ? S[NP > 1 ? 1 : 0] : 0u, NP > 1 ? Sb[NP > 1 ? 1 : 0] : 0u, start_mask, &starts, &ends, &alive_r);
      if (static_cast<uint32_t>(starts >> kStreamShift) != cand || (starts & ~(0xFFFFFFFFull << kStreamShift)) != 0) stats[2]++;
      // (a run of more than 16 bytes that ends inside the window is decided by the run form and undecided -- alive -- by the
      // sixteen steps: such starts are checked against the scalar walk)
      if ((alive_r & ~alive) != 0) stats[2]++;
      uint32_t matched_r = 0;
      while (ends != 0) {
        int j = 0;
        const uint32_t l = rj_stream_run_next(pl.run_shape, starts, &ends, &j);
        if (j < 0 || j > 31 || ((matched_r >> j) & 1u) || ((alive_r >> j) & 1u)) {
          stats[2]++;
          break;
        }
        matched_r |= 1u << j;
        bool ov = false;
        const uint32_t want = ((alive >> j) & 1u) ? rj_stream_walk(pl, t, n, at + static_cast<uint64_t>(j) - kStreamShift, 1u << 20, &ov)
                                                  : (((matched >> j) & 1u) ? rj_stream_len(len, j) : 0u);
        if (l != want) stats[2]++;
      }
      for (int j = 0; j < 32; j++) {
        if (((matched_r | alive_r) >> j) & 1u) continue;
        // not matched by the run form and decided: the steps (or the walk) must not have a match either
        bool ov = false;
        const bool has = ((alive >> j) & 1u) ? rj_stream_walk(pl, t, n, at + static_cast<uint64_t>(j) - kStreamShift, 1u << 20, &ov) != 0 : ((matched >> j) & 1u) != 0;
        if (has && ((start_mask >> j) & 1u)) stats[2]++;
      }
    }
    if (pl.select) {
      if (alive != 0) stats[2]++;   // (no match of such a plan outlives the register steps)
      lanes.push_back(LaneMatches{matched, {len[0], len[1], len[2], len[3]}});
    }
    for (int j = 0; j < 32; j++) {
      if (!((start_mask >> j) & 1u)) {
        if ((cand >> j) & 1u) stats[2]++;
        continue;
      }
      const uint64_t s = at + j - kStreamShift;
      // the scalar truth for this start
      uint32_t St = P.first[0][0] & P.cls[t[s]];
      bool is_cand = St != 0;
      if (pl.loop_first && s > 0 && (P.first[0][0] & P.cls[t[s - 1]]) != 0) is_cand = false;
      uint32_t longest = 0;
      for (uint32_t d = 1; d <= kStreamShift && St; d++) {
        if (St & P.last[0][0]) longest = d;
        uint32_t T = 0;
        for (int q = 0; q < P.n_pos; q++)
          if ((St >> q) & 1u) {
            const int r = P.row_of[static_cast<size_t>(q)];
            T |= r < 0 ? 1u << (q + 1) : P.rows[0][static_cast<size_t>(r)];
          }
        St = s + d < n ? (T & P.cls[t[s + d]]) : 0u;
      }
      if (((cand >> j) & 1u) != (is_cand ? 1u : 0u)) stats[2]++;
      if (!is_cand) continue;
      if (((alive >> j) & 1u) != (St != 0 ? 1u : 0u)) stats[2]++;
      if ((alive >> j) & 1u) {
        stats[1]++;
        uint64_t e = 0;
        bool overrun = false, ov2 = false;
        const bool f1 = rj_lane_longest<1>(F, t, n, s, &e, &overrun);
        const uint32_t l = rj_stream_walk(pl, t, n, s, 1u << 20, &ov2);   // the kernel's own walk == the general walker
        if ((l != 0) != f1 || (f1 && s + l != e)) stats[2]++;
        if (l != 0) {
          if (k < cap) {
            out[2 * k] = s;
            out[2 * k + 1] = s + l;
          }
          k++;
        }
        continue;
      }
      if (((matched >> j) & 1u) != (longest != 0 ? 1u : 0u)) stats[2]++;
      if ((matched >> j) & 1u) {
        const uint32_t l = rj_stream_len(len, j);
        if (l != longest) stats[2]++;
        stats[0]++;
        if (pl.select) continue;
        if (k < cap) {
          out[2 * k] = s;
          out[2 * k + 1] = s + l;
        }
        k++;
      }
    }
    for (int q = 0; q < NP; q++) Sb[q] = S[q];
  }
  if (!pl.select) return static_cast<long>(k);
  // The selection as dense_streams.hip makes it (stream_tile<.., SELECT>): tiles of 1024 lanes on their own, a tile's entry
  // state from the 64 lanes before it (the last lane without a match resets the chain; none: the run is void, -103), inside
  // an iteration of 64 lanes every lane assumes that nothing reaches into it and is corrected from the lane below until
  // nothing changes, from iteration to iteration the state is carried.
  const uint64_t n_lanes = lanes.size();
  auto lane_at = [&](uint64_t i) { return i < n_lanes ? lanes[static_cast<size_t>(i)] : LaneMatches{0, {0, 0, 0, 0}}; };
  bool unsure = false;
  uint64_t rounds_max = 0;
  auto resolve = [&](LaneMatches (&w)[64], uint32_t d_carry, uint32_t (&sel)[64]) {
    uint32_t d_in[64], d_out[64];
    for (int i = 0; i < 64; i++) {
      d_in[i] = i == 0 ? d_carry : 0u;
      sel[i] = rj_stream_select(w[i].take, w[i].len, d_in[i], &d_out[i]);
    }
    for (uint64_t rounds = 1;; rounds++) {
      uint32_t want[64];
      bool any_redo = false;
      for (int i = 0; i < 64; i++) {
        want[i] = i == 0 ? d_carry : d_out[i - 1];
        any_redo = any_redo || want[i] != d_in[i];
      }
      if (rounds > rounds_max) rounds_max = rounds;
      if (!any_redo) break;
      for (int i = 0; i < 64; i++)
        if (want[i] != d_in[i]) {
          d_in[i] = want[i];
          sel[i] = rj_stream_select(w[i].take, w[i].len, d_in[i], &d_out[i]);
        }
    }
    return d_out[63];
  };
  const uint64_t tile_lanes = 1024, first_tile = (sb + kStreamShift) / (tile_lanes * 32);
  for (uint64_t tile = first_tile; tile * tile_lanes < n_lanes; tile++) {
    const uint64_t l0 = tile * tile_lanes;
    uint32_t d_carry = 0;
    for (int it = l0 >= 64 ? -1 : 0; it < 16; it++) {
      const uint64_t i0 = l0 + static_cast<uint64_t>(static_cast<int64_t>(it) * 64);
      LaneMatches w[64];
      bool any_take = false;
      for (int i = 0; i < 64; i++) w[i] = lane_at(i0 + i);
      if (it < 0) {
        int last_quiet = -1;
        for (int i = 0; i < 64; i++)
          if (w[i].take == 0) last_quiet = i;
        if (last_quiet < 0) {
          unsure = true;
          last_quiet = 63;
        }
        for (int i = 0; i <= last_quiet; i++) w[i].take = 0;
        d_carry = 0;
      }
      for (int i = 0; i < 64; i++) any_take = any_take || w[i].take != 0;
      if (!any_take) {
        d_carry = 0;
        continue;
      }
      uint32_t sel[64];
      d_carry = resolve(w, d_carry, sel);
      if (it < 0) continue;
      for (int i = 0; i < 64; i++)
        for (uint32_t m = sel[i]; m; m &= m - 1) {
          const int j = __builtin_ctz(m);
          const uint64_t s0 = (i0 + i) * 32 + j - kStreamShift;
          if (k < cap) {
            out[2 * k] = s0;
            out[2 * k + 1] = s0 + rj_stream_len(w[i].len, j);
          }
          k++;
        }
    }
  }
  stats[5] = rounds_max;
  if (unsure) return -103;
  return static_cast<long>(k);
}

extern "C" long ce_stream_match_all(const char* re, const uint8_t* text, uint64_t n, uint64_t sb, uint64_t se, uint64_t* out, uint64_t cap,
                                    uint64_t* stats) {
  LowerResult lr = lower(re);
  if (lr.status != 0) return lr.status;
  const Program& P = *lr.program;
  const StreamPlan pl = make_stream_plan(P, run_start_rule(P) && !P.q8_risk, P.q8_risk);
  if (pl.n_pos == 0) return -101;
  const TableBlob fb = make_table_blob(P, P.n_pos, P.n_words, P.has_assertions);
  DevProgram F{};
  point_tables(&F, fb.words.data(), fb, P.n_pos);
  F.nullable = nullable_bits(P);
  F.max_walk = 1u << 20;
  stats[3] = pl.depth;
  stats[4] = pl.loop_first;
  switch (pl.n_pos) {
    case 1: return stream_run<1>(P, F, pl, text, n, sb, se, out, cap, stats);
    case 2: return stream_run<2>(P, F, pl, text, n, sb, se, out, cap, stats);
    case 3: return stream_run<3>(P, F, pl, text, n, sb, se, out, cap, stats);
    case 4: return stream_run<4>(P, F, pl, text, n, sb, se, out, cap, stats);
    case 5: case 6: return stream_run<6>(P, F, pl, text, n, sb, se, out, cap, stats);
    default: return stream_run<8>(P, F, pl, text, n, sb, se, out, cap, stats);
  }
}

extern "C" {

// MatchAll of the starts in [sb, se) through the carry scan with sub-chunks of `sub` bytes;
// returns the count, a negative lowering status, or -9 (automaton too wide for this driver)
long ce_match_range(const char* re, const uint8_t* text, uint64_t n, uint64_t sub, uint64_t sb, uint64_t se,
                    uint64_t carry_cur, uint64_t carry_prev_end, int have_prev, uint64_t* out, uint64_t cap) {
  LowerResult lr = lower(re);
  if (lr.status != 0) return lr.status;
  const Program& P = *lr.program;
  const TableBlob blob = make_table_blob(P.rev, P.n_pos, P.n_words, P.has_assertions);
  DevProgram R{};
  point_tables(&R, blob.words.data(), blob, P.n_pos);
  R.nullable = nullable_bits(P);
  if (P.n_words <= 1) return run<1>(P, R, text, n, sub, sb, se, carry_cur, carry_prev_end, have_prev, out, cap);
  if (P.n_words <= 2) return run<2>(P, R, text, n, sub, sb, se, carry_cur, carry_prev_end, have_prev, out, cap);
  if (P.n_words <= 4) return run<4>(P, R, text, n, sub, sb, se, carry_cur, carry_prev_end, have_prev, out, cap);
  if (P.n_words <= 8) return run<8>(P, R, text, n, sub, sb, se, carry_cur, carry_prev_end, have_prev, out, cap);
  if (P.n_words <= 16) return run<16>(P, R, text, n, sub, sb, se, carry_cur, carry_prev_end, have_prev, out, cap);
  if (P.n_words <= 32) return run<32>(P, R, text, n, sub, sb, se, carry_cur, carry_prev_end, have_prev, out, cap);
  if (P.n_words <= 64) return run<64>(P, R, text, n, sub, sb, se, carry_cur, carry_prev_end, have_prev, out, cap);
  if (P.n_words <= 128) return run<128>(P, R, text, n, sub, sb, se, carry_cur, carry_prev_end, have_prev, out, cap);
  if (P.n_words <= 256) return run<256>(P, R, text, n, sub, sb, se, carry_cur, carry_prev_end, have_prev, out, cap);
  return -9;
}

long ce_match_all_behind(const char* re, const uint8_t* text, uint64_t n, uint32_t max_walk, uint64_t* out, uint64_t cap) {
  LowerResult lr = lower(re);
  if (lr.status != 0) return lr.status;
  const Program& P = *lr.program;
  if (!P.behind) return -101;
  const TableBlob fb = make_table_blob(P, P.n_pos, P.n_words, P.has_assertions);
  const TableBlob rb = make_table_blob(P.rev, P.n_pos, P.n_words, P.has_assertions);
  DevProgram F{}, R{};
  point_tables(&F, fb.words.data(), fb, P.n_pos);
  point_tables(&R, rb.words.data(), rb, P.n_pos);
  F.nullable = R.nullable = nullable_bits(P);
  F.max_walk = R.max_walk = max_walk;
  fill_windows(&F, P);
  if (P.n_words <= 1) return behind_run<1, 1>(F, R, text, n, out, cap);
  if (P.n_words <= 2) return behind_run<2, 1>(F, R, text, n, out, cap);
  if (P.n_words <= 4) return behind_run<4, 2>(F, R, text, n, out, cap);
  return -9;
}

long ce_match_all(const char* re, const uint8_t* text, uint64_t n, uint64_t sub, uint64_t* out, uint64_t cap) {
  return ce_match_range(re, text, n, sub, 0, n + 1, 0, 0, 0, out, cap);
}

long ce_exact_range(const char* re, const uint8_t* text, uint64_t n, uint64_t chunk, uint64_t sb, uint64_t se, uint64_t* out,
                    uint64_t cap, uint64_t* n_segments, uint64_t* longest, int* q8_risk) {
  LowerResult lr = lower(re);
  if (lr.status != 0) return lr.status;
  const Program& P = *lr.program;
  *q8_risk = P.q8_risk ? 1 : 0;
  const TableBlob fb = make_table_blob(P, P.n_pos, P.n_words, P.has_assertions);
  DevProgram F{};
  point_tables(&F, fb.words.data(), fb, P.n_pos);
  F.nullable = nullable_bits(P);
  const GraphBlob gb = make_graph_blob(P.graph);
  DevGraph G{};
  point_graph(&G, gb.bytes.data(), gb);
  if (P.n_words <= 2) return exact_run<1>(P, F, G, text, n, chunk, sb, se, out, cap, n_segments, longest);
  if (P.n_words <= 4) return exact_run<2>(P, F, G, text, n, chunk, sb, se, out, cap, n_segments, longest);
  if (P.n_words <= 8) return exact_run<4>(P, F, G, text, n, chunk, sb, se, out, cap, n_segments, longest);
  if (P.n_words <= 16) return exact_run<8>(P, F, G, text, n, chunk, sb, se, out, cap, n_segments, longest);
  if (P.n_words <= 32) return exact_run<16>(P, F, G, text, n, chunk, sb, se, out, cap, n_segments, longest);
  return -9;
}

// ce_exact_range with every segment taken in parts of `sub` bytes, speculatively (see replay_segment_speculatively)
long ce_exact_range_spec(const char* re, const uint8_t* text, uint64_t n, uint64_t chunk, uint64_t sb, uint64_t se, uint64_t sub, uint64_t warm,
                         uint64_t* out, uint64_t cap, uint64_t* fixed) {
  LowerResult lr = lower(re);
  if (lr.status != 0) return lr.status;
  const Program& P = *lr.program;
  const TableBlob fb = make_table_blob(P, P.n_pos, P.n_words, P.has_assertions);
  DevProgram F{};
  point_tables(&F, fb.words.data(), fb, P.n_pos);
  F.nullable = nullable_bits(P);
  const GraphBlob gb = make_graph_blob(P.graph);
  DevGraph G{};
  point_graph(&G, gb.bytes.data(), gb);
  *fixed = 0;
  if (P.n_words <= 2) return exact_run_spec<1>(F, G, text, n, chunk, sb, se, sub, warm, out, cap, fixed);
  if (P.n_words <= 4) return exact_run_spec<2>(F, G, text, n, chunk, sb, se, sub, warm, out, cap, fixed);
  if (P.n_words <= 8) return exact_run_spec<4>(F, G, text, n, chunk, sb, se, sub, warm, out, cap, fixed);
  if (P.n_words <= 16) return exact_run_spec<8>(F, G, text, n, chunk, sb, se, sub, warm, out, cap, fixed);
  if (P.n_words <= 32) return exact_run_spec<16>(F, G, text, n, chunk, sb, se, sub, warm, out, cap, fixed);
  return -9;
}

long ce_swar_check(const char* re, const uint8_t* text, uint64_t n, uint64_t* stats) {
  LowerResult lr = lower(re);
  if (lr.status != 0) return lr.status;
  const Program& P = *lr.program;
  const SwarPlan pl = make_swar_plan(P);
  if (pl.n_ranges == 0) return -101;
  if (pl.depth == 1) return swar_check<1>(P, pl, text, n, stats);
  if (pl.depth == 2) return swar_check<2>(P, pl, text, n, stats);
  return swar_check<4>(P, pl, text, n, stats);
}

// rj_lane_longest_short against rj_lane_longest, every start of the text.  Returns disagreements, -101 when
// the pattern has no short bound; *checked = starts compared.
long ce_short_check(const char* re, const uint8_t* text, uint64_t n, uint64_t* checked) {
  LowerResult lr = lower(re);
  if (lr.status != 0) return lr.status;
  const Program& P = *lr.program;
  const TableBlob fb = make_table_blob(P, P.n_pos, P.n_words, P.has_assertions);
  DevProgram F{};
  point_tables(&F, fb.words.data(), fb, P.n_pos);
  F.nullable = nullable_bits(P);
  F.max_walk = 1u << 20;
  F.short_max = short_match_bound(P);
  if (F.short_max == 0) return -101;
  long bad = 0;
  for (uint64_t s0 = 0; s0 <= n; s0++) {
    uint64_t e1 = 0, e2 = 0;
    bool overrun = false;
    const bool f1 = rj_lane_longest<1>(F, text, n, s0, &e1, &overrun);
    const bool f2 = rj_lane_longest_short(F, text, n, s0, &e2);
    if (f1 != f2 || (f1 && e1 != e2)) bad++;
    (*checked)++;
  }
  return bad;
}

}  // extern "C"

// The padded-table walkers of lds_walk.h against the ones they replace (device_program.h, behind_walk.h):
//   every start: lw_longest == rj_lane_longest (found, end, overrun);
//   every (text position p, automaton position q): lw_reaches_accept == rj_reaches_accept, and
//   lw_leftmost_start == rj_leftmost_start from the single reverse position of q;
//   behind patterns, every text position: lw_behind_candidate == rj_behind_candidate.
// Returns the number of disagreements (-9: wider than 128 positions); *checked = comparisons made.
// text that hands out its bytes in aligned blocks of B at most: the tight loops of the walkers then run in many
// short pieces, like a walk that leaves its LDS window on the GPU
template <unsigned B>
struct BlockText {
  const uint8_t* t;
  uint64_t n;
  uint8_t operator[](uint64_t p) const { return t[p]; }
  uint32_t span(uint64_t p, const uint8_t** ptr) const {
    *ptr = t + p;
    if (p >= n) return 0;
    const uint64_t k = B - (p % B);
    return static_cast<uint32_t>(std::min<uint64_t>(k, n - p));
  }
  uint32_t span_back(uint64_t p, const uint8_t** ptr) const {
    *ptr = t + p - 1;
    if (p == 0) return 0;
    return static_cast<uint32_t>((p - 1) % B + 1);
  }
};

// The GPU's window texts (lds_walk.h) over plain arrays: the loader checks every block it is asked for, and the
// windows sit between guard bytes that must stay untouched.
static long g_loader_violations = 0;
struct CheckedLoader {
  static void block16(uint8_t* dst, const uint8_t* text, uint64_t n, uint64_t at) {
    if ((at & 15u) != 0 || at > n + 4096) g_loader_violations++;
    for (uint64_t k = 0; k < 16; k++) dst[k] = at + k < n ? text[at + k] : 0;
  }
};

template <int NQ, int NW, bool CTX, class WText>
static long lds_walk_check_with(const Program& P, const DevProgram& F, const DevProgram& R, const WalkTab<NQ>& WF, const WalkTab<NQ>& WR,
                                const uint8_t* text, const WText& wtext, uint64_t n, uint64_t* checked) {
  long bad = 0;
  for (uint64_t s0 = 0; s0 <= n; s0++) {
    uint64_t e1 = 0, e2 = 0;
    bool o1 = false, o2 = false;
    const bool f1 = rj_lane_longest<NQ>(F, text, n, s0, &e1, &o1);
    const bool f2 = lw_longest<NQ, CTX>(WF, wtext, n, s0, &e2, &o2);
    if (f1 != f2 || o1 != o2 || (f1 && e1 != e2)) bad++;
    (*checked)++;
  }
  for (uint64_t p = 0; p < n; p++) {
    for (int q = 0; q < P.n_pos; q++) {
      bool o1 = false, o2 = false;
      const bool r1 = rj_reaches_accept<NW>(F, text, n, p, q, &o1);
      const bool r2 = lw_reaches_accept<NQ, CTX>(WF, wtext, n, p, q, &o2);
      if (r1 != r2 || o1 != o2) bad++;
      uint32_t S1[NW];
      uint64_t S2[NQ];
      for (int k = 0; k < NW; k++) S1[k] = 0;
      for (int k = 0; k < NQ; k++) S2[k] = 0;
      S1[q >> 5] = 1u << (q & 31);
      S2[q >> 6] = 1ull << (q & 63);
      uint64_t b1 = 0, b2 = 0;
      o1 = o2 = false;
      const bool l1 = rj_leftmost_start<NW>(R, text, n, p, S1, F.max_walk, &b1, &o1);
      const bool l2 = lw_leftmost_start<NQ, CTX>(WR, wtext, n, p, S2, F.max_walk, &b2, &o2);
      if (l1 != l2 || o1 != o2 || (l1 && b1 != b2)) bad++;
      (*checked) += 2;
    }
    if (P.behind) {
      uint64_t b1 = 0, e1 = 0, b2 = 0, e2 = 0;
      bool o1 = false, o2 = false;
      const bool c1 = rj_behind_candidate<NW, NQ>(F, R, text, n, p, &b1, &e1, &o1);
      const bool c2 = lw_behind_candidate<NQ, CTX>(F, WF, WR, wtext, n, p, &b2, &e2, &o2);
      if (c1 != c2 || o1 != o2 || (c1 && (b1 != b2 || e1 != e2))) bad++;
      (*checked)++;
    }
  }
  return bad;
}

template <int NQ, int NW, bool CTX>
static long lds_walk_check(const Program& P, const DevProgram& F, const DevProgram& R, const WalkTab<NQ>& WF, const WalkTab<NQ>& WR,
                           const uint8_t* text, uint64_t n, uint64_t* checked) {
  long extra = 0;
  {
    // a lane window of 48 bytes and a wave window of 64 bytes + a 16-byte slot, placed around every 7th position:
    // walks from there leave the windows in both directions
    std::vector<uint8_t> arena(16 + 64 + 16 + 16 + 16, 0xEE);
    for (uint64_t c = 0; c <= n; c += 7) {
      const uint64_t wb = c >= 16 ? (c - 16) & ~15ull : 0;
      uint8_t* lane_win = arena.data() + 16;
      LaneWindowText<CheckedLoader, 48> lt(lane_win, wb, 0, text, n);
      lt.move(wb);
      uint64_t e1 = 0, e2 = 0, b1 = 0, b2 = 0;
      bool o1 = false, o2 = false;
      const bool f1 = rj_lane_longest<NQ>(F, text, n, c, &e1, &o1), f2 = lw_longest<NQ, CTX>(WF, lt, n, c, &e2, &o2);
      if (f1 != f2 || o1 != o2 || (f1 && e1 != e2)) extra++;
      for (int q = 0; q < P.n_pos && c < n; q += 3) {
        uint32_t S1[NW];
        uint64_t S2[NQ];
        for (int k = 0; k < NW; k++) S1[k] = 0;
        for (int k = 0; k < NQ; k++) S2[k] = 0;
        S1[q >> 5] = 1u << (q & 31);
        S2[q >> 6] = 1ull << (q & 63);
        o1 = o2 = false;
        const bool l1 = rj_leftmost_start<NW>(R, text, n, c, S1, F.max_walk, &b1, &o1);
        const bool l2 = lw_leftmost_start<NQ, CTX>(WR, lt, n, c, S2, F.max_walk, &b2, &o2);
        if (l1 != l2 || o1 != o2 || (l1 && b1 != b2)) extra++;
      }
      uint8_t* wave_win = arena.data() + 16;
      for (uint64_t k = 0; k < 64; k += 16) CheckedLoader::block16(wave_win + k, text, n, wb + k);
      const uint64_t avail = n - wb;
      WaveWindowText<CheckedLoader> wt(wave_win, arena.data() + 16 + 64 + 16, wb, avail < 64 ? static_cast<uint32_t>(avail) : 64u, text, n);
      o2 = false;
      const bool f3 = lw_longest<NQ, CTX>(WF, wt, n, c, &e2, &o2);
      if (f1 != f3 || (f1 && e1 != e2)) extra++;
      for (size_t g = 0; g < 16; g++)
        if (arena[g] != 0xEE || arena[16 + 64 + g] != 0xEE || arena[16 + 64 + 32 + g] != 0xEE) extra++;
      (*checked) += 3;
    }
    extra += g_loader_violations;
    g_loader_violations = 0;
  }
  return extra + lds_walk_check_with<NQ, NW, CTX>(P, F, R, WF, WR, text, PlainText(text, n), n, checked) +
         lds_walk_check_with<NQ, NW, CTX>(P, F, R, WF, WR, text, BlockText<16>{text, n}, n, checked) +
         lds_walk_check_with<NQ, NW, CTX>(P, F, R, WF, WR, text, BlockText<3>{text, n}, n, checked);
}

extern "C" {

long ce_lds_walk_check(const char* re, const uint8_t* text, uint64_t n, uint32_t max_walk, uint64_t* checked) {
  LowerResult lr = lower(re);
  if (lr.status != 0) return lr.status;
  const Program& P = *lr.program;
  if (P.n_words > 4) return -9;
  const TableBlob fb = make_table_blob(P, P.n_pos, P.n_words, P.has_assertions);
  const TableBlob rb = make_table_blob(P.rev, P.n_pos, P.n_words, P.has_assertions);
  DevProgram F{}, R{};
  point_tables(&F, fb.words.data(), fb, P.n_pos);
  point_tables(&R, rb.words.data(), rb, P.n_pos);
  F.nullable = R.nullable = nullable_bits(P);
  F.max_walk = R.max_walk = max_walk;
  fill_windows(&F, P);
  const int nq = P.n_words <= 2 ? 1 : 2;
  const std::vector<uint64_t> wf = make_walk_blob(fb, P.n_pos, nq), wr = make_walk_blob(rb, P.n_pos, nq);
  if (nq == 1) {
    const WalkTab<1> WF = lw_point<1>(wf.data(), fb.C, P.n_pos, F.nullable, max_walk), WR = lw_point<1>(wr.data(), rb.C, P.n_pos, R.nullable, max_walk);
    if (fb.C > 1) return P.n_words <= 1 ? lds_walk_check<1, 1, true>(P, F, R, WF, WR, text, n, checked) : lds_walk_check<1, 2, true>(P, F, R, WF, WR, text, n, checked);
    return P.n_words <= 1 ? lds_walk_check<1, 1, false>(P, F, R, WF, WR, text, n, checked) : lds_walk_check<1, 2, false>(P, F, R, WF, WR, text, n, checked);
  }
  const WalkTab<2> WF = lw_point<2>(wf.data(), fb.C, P.n_pos, F.nullable, max_walk), WR = lw_point<2>(wr.data(), rb.C, P.n_pos, R.nullable, max_walk);
  return fb.C > 1 ? lds_walk_check<2, 4, true>(P, F, R, WF, WR, text, n, checked) : lds_walk_check<2, 4, false>(P, F, R, WF, WR, text, n, checked);
}

}  // extern "C"
