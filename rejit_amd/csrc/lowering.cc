// rejit_amd/csrc/lowering.cc -- AST -> NFA graph -> position automaton + scan plan.
// See lowering.h for the map to the reference files this replaces.
#include "lowering.h"

#include <algorithm>
#include <cstring>
#include <functional>
#include <set>

namespace rejit_amd {

// =============================================================================
// NFA graph.  Same wiring as RegexpIndexer / RegexpLister (src/codegen.cc:91-324).
namespace {

class GraphBuilder {
 public:
  GraphBuilder(Graph* g, int max_states) : g_(g), max_states_(max_states) {}
  bool failed() const { return failed_; }

  int new_state() {
    if (g_->n_states >= max_states_) {
      failed_ = true;
      return 0;
    }
    return g_->n_states++;
  }

  void control(int src, int dst, ControlKind k) { g_->control_edges.push_back({src, dst, k}); }

  // `copied`: this sub-tree is one the reference obtains through DeepCopy()
  // (codegen.cc:228-241); Bracket::DeepCopy drops the non_matching flag.
  void emit(const Node& n, int entry, int exit, bool copied) {
    if (failed_) return;
    switch (n.kind) {
      case NodeKind::Literal: {
        ByteEdge e;
        e.src = entry;
        e.dst = exit;
        e.bytes = n.bytes;
        g_->byte_edges.push_back(std::move(e));
        break;
      }
      case NodeKind::Any: {
        ByteEdge e;
        e.src = entry;
        e.dst = exit;
        ByteSet lb;
        lb.add('\n');
        lb.add('\r');
        e.cls = lb.inverted();  // VisitPeriod, codegen-x64.cc:851-873
        g_->byte_edges.push_back(std::move(e));
        break;
      }
      case NodeKind::Class: {
        ByteEdge e;
        e.src = entry;
        e.dst = exit;
        e.cls = (n.negated && !copied) ? n.listed.inverted() : n.listed;
        g_->byte_edges.push_back(std::move(e));
        break;
      }
      case NodeKind::StartOfLine:
        control(entry, exit, ControlKind::StartOfLine);
        break;
      case NodeKind::EndOfLine:
        control(entry, exit, ControlKind::EndOfLine);
        break;
      case NodeKind::Concat: {
        int cur = entry;
        for (size_t i = 0; i < n.kids.size(); i++) {
          int nxt = (i + 1 == n.kids.size()) ? exit : new_state();
          emit(*n.kids[i], cur, nxt, copied);
          cur = nxt;
        }
        break;
      }
      case NodeKind::Alternate:
        for (const auto& k : n.kids) emit(*k, entry, exit, copied);
        break;
      case NodeKind::Repeat:
        repeat(n, entry, exit, copied);
        break;
    }
  }

 private:
  void repeat(const Node& n, int entry, int exit, bool copied) {
    const Node& base = *n.kids[0];
    const uint32_t lo = n.min, hi = n.max;
    const bool bounded = hi != kUnbounded;
    if (lo == 0 && hi == 0) {
      control(entry, exit, ControlKind::Epsilon);
      return;
    }
    const bool chain = lo > 1 || (hi > 1 && bounded);
    const uint32_t copies = chain ? (bounded ? hi : lo) : 1;
    int in = entry, out = exit;
    if (!bounded) {
      out = new_state();
      if (lo <= 1) in = new_state();
    }
    std::vector<int> from(copies), to(copies);
    int cur = in;
    for (uint32_t i = 0; i < copies && !failed_; i++) {
      int nxt = (i + 1 == copies) ? out : new_state();
      from[i] = cur;
      to[i] = nxt;
      emit(base, cur, nxt, copied || i > 0);
      cur = nxt;
    }
    if (failed_) return;
    if (lo == 0) control(entry, exit, ControlKind::Epsilon);
    if (bounded && hi > 1) {
      for (uint32_t i = std::max(1u, lo) - 1; i + 1 < copies; i++) control(to[i], exit, ControlKind::Epsilon);
    } else {
      // unbounded repetitions AND bounded ones with max == 1 (x?, x{1}, x{0,1}): the
      // reference wires a "repeat" epsilon from the last copy's exit back to its entry
      // (codegen.cc:266-312) -- for max == 1 those are the repetition's own entry/exit
      // states, shared with whatever surrounds it.
      if (lo <= 1) control(entry, in, ControlKind::Epsilon);
      control(out, exit, ControlKind::Epsilon);
      control(to[copies - 1], from[copies - 1], ControlKind::Epsilon);
    }
  }

  Graph* g_;
  int max_states_;
  bool failed_ = false;
};

}  // namespace

bool build_graph(const Node& root, Graph* g, std::string* message, int max_states) {
  g->n_states = 2;
  g->entry = 0;
  g->exit = 1;
  GraphBuilder b(g, max_states);
  b.emit(root, g->entry, g->exit, false);
  if (b.failed()) {
    if (message) *message = "pattern expands to too many states";
    return false;
  }
  return true;
}

// =============================================================================
// Position automaton.
namespace {

struct Bits {
  std::vector<uint32_t> w;
  explicit Bits(int words = 0) : w(static_cast<size_t>(words), 0u) {}
  void set(int i) { w[static_cast<size_t>(i) >> 5] |= 1u << (i & 31); }
  bool get(int i) const { return (w[static_cast<size_t>(i) >> 5] >> (i & 31)) & 1u; }
  bool any() const {
    for (uint32_t x : w)
      if (x) return true;
    return false;
  }
  void operator|=(const Bits& o) {
    for (size_t i = 0; i < w.size(); i++) w[i] |= o.w[i];
  }
  bool operator==(const Bits& o) const { return w == o.w; }
  int count() const {
    int n = 0;
    for (uint32_t x : w) n += __builtin_popcount(x);
    return n;
  }
  template <class F>
  void for_each(F f) const {
    for (size_t k = 0; k < w.size(); k++) {
      uint32_t x = w[k];
      while (x) {
        int b = __builtin_ctz(x);
        x &= x - 1;
        f(static_cast<int>(k * 32 + b));
      }
    }
  }
};

struct Closure {
  std::vector<int> states;
  bool reaches_exit = false;
};

class Automaton {
 public:
  Automaton(const Graph& g, Program* p) : g_(g), p_(p) {}

  bool build(std::string* message) {
    // positions
    edge_first_.resize(g_.byte_edges.size());
    int n = 0;
    for (size_t e = 0; e < g_.byte_edges.size(); e++) {
      edge_first_[e] = n;
      n += g_.byte_edges[e].bytes.empty() ? 1 : static_cast<int>(g_.byte_edges[e].bytes.size());
    }
    if (n > kMaxPositions) {
      if (message) *message = "pattern has too many positions for the device automaton";
      return false;
    }
    P_ = n;
    W_ = std::max(1, (n + 31) / 32);
    p_->n_pos = P_;
    p_->n_words = W_;
    out_edges_.assign(static_cast<size_t>(g_.n_states), {});
    ctrl_.assign(static_cast<size_t>(g_.n_states), {});
    for (size_t e = 0; e < g_.byte_edges.size(); e++) out_edges_[static_cast<size_t>(g_.byte_edges[e].src)].push_back(static_cast<int>(e));
    for (const ControlEdge& c : g_.control_edges) {
      ctrl_[static_cast<size_t>(c.src)].push_back(&c);
      if (c.kind != ControlKind::Epsilon) p_->has_assertions = true;
    }

    // byte classes
    p_->cls.assign(static_cast<size_t>(256) * W_, 0u);
    for (size_t e = 0; e < g_.byte_edges.size(); e++) {
      const ByteEdge& be = g_.byte_edges[e];
      if (be.bytes.empty()) {
        for (int b = 0; b < 256; b++)
          if (be.cls.has(static_cast<uint8_t>(b))) set_bit(&p_->cls[static_cast<size_t>(b) * W_], edge_first_[e]);
      } else {
        for (size_t k = 0; k < be.bytes.size(); k++)
          set_bit(&p_->cls[static_cast<size_t>(static_cast<uint8_t>(be.bytes[k])) * W_], edge_first_[e] + static_cast<int>(k));
      }
    }

    // first / nullable
    for (int ctx = 0; ctx < kNumCtx; ctx++) {
      Closure c = closure(g_.entry, ctx);
      p_->first[ctx].assign(static_cast<size_t>(W_), 0u);
      p_->last[ctx].assign(static_cast<size_t>(W_), 0u);
      starts_of(c, p_->first[ctx].data());
      p_->nullable[ctx] = c.reaches_exit;
      p_->any_nullable |= c.reaches_exit;
    }

    // follow of the LAST position of every edge; interior literal positions are linear
    p_->linear.assign(static_cast<size_t>(W_), 0u);
    p_->row_of.assign(static_cast<size_t>(std::max(P_, 1)), -1);
    std::vector<std::vector<uint32_t>> rows[kNumCtx];
    for (size_t e = 0; e < g_.byte_edges.size(); e++) {
      const ByteEdge& be = g_.byte_edges[e];
      int len = be.bytes.empty() ? 1 : static_cast<int>(be.bytes.size());
      for (int k = 0; k + 1 < len; k++) set_bit(p_->linear.data(), edge_first_[e] + k);
      int lastpos = edge_first_[e] + len - 1;
      std::vector<uint32_t> f[kNumCtx];
      bool lin = true;
      for (int ctx = 0; ctx < kNumCtx; ctx++) {
        Closure c = closure(be.dst, ctx);
        f[ctx].assign(static_cast<size_t>(W_), 0u);
        starts_of(c, f[ctx].data());
        if (c.reaches_exit) set_bit(p_->last[ctx].data(), lastpos);
        // linear iff follow == {lastpos + 1}
        std::vector<uint32_t> want(static_cast<size_t>(W_), 0u);
        if (lastpos + 1 < P_) set_bit(want.data(), lastpos + 1);
        if (lastpos + 1 >= P_ || f[ctx] != want) lin = false;
      }
      if (lin) {
        set_bit(p_->linear.data(), lastpos);
      } else {
        p_->row_of[static_cast<size_t>(lastpos)] = p_->n_rows++;
        for (int ctx = 0; ctx < kNumCtx; ctx++) rows[ctx].push_back(std::move(f[ctx]));
      }
    }
    for (int ctx = 0; ctx < kNumCtx; ctx++) {
      p_->rows[ctx].assign(static_cast<size_t>(std::max(p_->n_rows, 1)) * W_, 0u);
      for (int r = 0; r < p_->n_rows; r++)
        std::copy(rows[ctx][static_cast<size_t>(r)].begin(), rows[ctx][static_cast<size_t>(r)].end(),
                  p_->rows[ctx].begin() + static_cast<long>(r) * W_);
    }

    // bytes that can start a non-empty match
    Bits first_any(W_);
    for (int ctx = 0; ctx < kNumCtx; ctx++)
      for (int k = 0; k < W_; k++) first_any.w[static_cast<size_t>(k)] |= p_->first[ctx][static_cast<size_t>(k)];
    for (int b = 0; b < 256; b++) {
      const uint32_t* row = &p_->cls[static_cast<size_t>(b) * W_];
      for (int k = 0; k < W_; k++)
        if (row[k] & first_any.w[static_cast<size_t>(k)]) {
          p_->first_bytes.add(static_cast<uint8_t>(b));
          break;
        }
    }
    lengths(first_any);
    plan(first_any);
    build_reverse();
    return true;
  }

 private:
  static void set_bit(uint32_t* w, int i) { w[i >> 5] |= 1u << (i & 31); }

  // states reachable from q through control edges that hold in context ctx
  Closure closure(int q, int ctx) {
    Closure c;
    std::vector<int> stack{q};
    // (visited marks by epoch: a std::set here was 40 % of the compile time of automata with thousands of states)
    if (mark_.size() != static_cast<size_t>(g_.n_states)) mark_.assign(static_cast<size_t>(g_.n_states), 0u);
    if (++mark_epoch_ == 0) {
      std::fill(mark_.begin(), mark_.end(), 0u);
      mark_epoch_ = 1;
    }
    const auto first_visit = [&](int st) {
      if (mark_[static_cast<size_t>(st)] == mark_epoch_) return false;
      mark_[static_cast<size_t>(st)] = mark_epoch_;
      return true;
    };
    first_visit(q);
    while (!stack.empty()) {
      int s = stack.back();
      stack.pop_back();
      c.states.push_back(s);
      if (s == g_.exit) c.reaches_exit = true;
      for (const ControlEdge* e : ctrl_[static_cast<size_t>(s)]) {
        bool ok = e->kind == ControlKind::Epsilon || (e->kind == ControlKind::StartOfLine && (ctx & 1)) ||
                  (e->kind == ControlKind::EndOfLine && (ctx & 2));
        if (ok && first_visit(e->dst)) stack.push_back(e->dst);
      }
    }
    return c;
  }

  void starts_of(const Closure& c, uint32_t* out) {
    for (int s : c.states)
      for (int e : out_edges_[static_cast<size_t>(s)]) set_bit(out, edge_first_[static_cast<size_t>(e)]);
  }

  // follow set of position i, union over all contexts (an over-approximation used only
  // for length bounds and for planning the scan, never for matching)
  Bits follow_any(int i) const {
    Bits f(W_);
    if ((p_->linear[static_cast<size_t>(i) >> 5] >> (i & 31)) & 1u) {
      f.set(i + 1);
      return f;
    }
    int r = p_->row_of[static_cast<size_t>(i)];
    for (int ctx = 0; ctx < kNumCtx; ctx++)
      for (int k = 0; k < W_; k++) f.w[static_cast<size_t>(k)] |= p_->rows[ctx][static_cast<size_t>(r) * W_ + k];
    return f;
  }

  Bits step_any(const Bits& s) const {
    Bits out(W_);
    s.for_each([&](int i) {  // out |= follow_any(i), without a set per position
      if ((p_->linear[static_cast<size_t>(i) >> 5] >> (i & 31)) & 1u) {
        out.set(i + 1);
        return;
      }
      const int r = p_->row_of[static_cast<size_t>(i)];
      for (int ctx = 0; ctx < kNumCtx; ctx++)
        for (int k = 0; k < W_; k++) out.w[static_cast<size_t>(k)] |= p_->rows[ctx][static_cast<size_t>(r) * W_ + k];
    });
    return out;
  }

  bool is_last_any(int i) const {
    for (int ctx = 0; ctx < kNumCtx; ctx++)
      if ((p_->last[ctx][static_cast<size_t>(i) >> 5] >> (i & 31)) & 1u) return true;
    return false;
  }

  void lengths(const Bits& first_any) {
    // min_len: BFS by depth over position sets; max_len: longest path, unbounded on a cycle
    if (p_->any_nullable) {
      p_->min_len = 0;
    } else {
      Bits level = first_any, seen(W_);
      uint64_t depth = 1;
      bool found = false;
      while (level.any() && !found) {
        level.for_each([&](int i) { found |= is_last_any(i); });
        if (found) break;
        seen |= level;
        Bits next = step_any(level);
        for (int k = 0; k < W_; k++) next.w[static_cast<size_t>(k)] &= ~seen.w[static_cast<size_t>(k)];
        level = next;
        depth++;
      }
      p_->min_len = found ? depth : 0;
      if (!found) p_->min_len = Program::kUnboundedLen;  // the pattern can never match
    }
    // longest path with DFS colouring
    std::vector<int> colour(static_cast<size_t>(std::max(P_, 1)), 0);
    std::vector<uint64_t> best(static_cast<size_t>(std::max(P_, 1)), 0);
    bool cyclic = false;
    std::function<void(int)> dfs = [&](int i) {
      colour[static_cast<size_t>(i)] = 1;
      uint64_t b = 1;
      follow_any(i).for_each([&](int j) {
        if (cyclic) return;
        if (colour[static_cast<size_t>(j)] == 1) {
          cyclic = true;
          return;
        }
        if (colour[static_cast<size_t>(j)] == 0) dfs(j);
        b = std::max(b, 1 + best[static_cast<size_t>(j)]);
      });
      best[static_cast<size_t>(i)] = b;
      colour[static_cast<size_t>(i)] = 2;
    };
    uint64_t longest = 0;
    first_any.for_each([&](int i) {
      if (cyclic) return;
      if (colour[static_cast<size_t>(i)] == 0) dfs(i);
      longest = std::max(longest, best[static_cast<size_t>(i)]);
    });
    p_->max_len = cyclic ? Program::kUnboundedLen : longest;
  }

  // Scan plan (the GPU's answer to FF_finder, src/codegen.cc:327-557).  The reference
  // picks literal NODES of the tree and needs a backward NFA pass when they are not at
  // the start of the match.  Here: choose a byte offset d such that the set of up-to-8-byte
  // strings a match can contain at [d, d+len) is small; strings that differ in one position
  // (a class position) are merged into one window with a wildcard byte.  A candidate start
  // is any s with one of the <= kMaxWindows windows at s + d.  Every candidate is then
  // verified by the forward automaton from s, so the windows only have to be a necessary
  // condition.
  struct Pattern {
    uint8_t byte[8];
    bool wild[8];
  };

  static int distance(const Pattern& a, const Pattern& b, int len) {
    int d = 0;
    for (int k = 0; k < len; k++)
      if (a.wild[k] != b.wild[k] || (!a.wild[k] && a.byte[k] != b.byte[k])) d++;
    return d;
  }

  static int fixed_after_merge(const Pattern& a, const Pattern& b, int len) {
    int fixed = 0;
    for (int k = 0; k < len; k++)
      if (!a.wild[k] && !b.wild[k] && a.byte[k] == b.byte[k]) fixed++;
    return fixed;
  }

  // merge patterns that differ in few positions (wildcarding those positions) until at
  // most `target` remain; returns the number of wildcards introduced, or -1
  static int merge_patterns(std::vector<Pattern>* ps, int len, size_t target) {
    int wilds = 0;
    // 0: keep both, 1: the same pattern twice, 2: merge j into i
    const auto verdict = [&](size_t i, size_t j, int max_dist) {
      const int dist = distance((*ps)[i], (*ps)[j], len);
      if (dist == 0) return 1;
      return dist <= max_dist && fixed_after_merge((*ps)[i], (*ps)[j], len) >= std::max(1, len - 2) ? 2 : 0;
    };
    // distance-1 merges are always worth it (one class position)
    //
    // The rule: take the FIRST pair (i, j), i < j, in lexicographic order that can be merged, merge it, start
    // again -- until nothing merges (or, at distance 2, until `target` is reached).  Starting again from (0, 1)
    // made this cubic: 256 patterns ([ab]{8}) at 40 window offsets were half a second of compile time.  After a
    // merge at (i, j) the only pairs before (i, j) whose answer can have changed are (a, i), a < i -- every other
    // one holds the same two patterns as when it was refused -- so the next first pair is the smallest such a,
    // else whatever a scan resumed at (i, i + 1) finds.  Same merges in the same order, quadratic.
    for (int max_dist = 1; max_dist <= 2; max_dist++) {
      size_t i = 0, j = 1;  // the scan resumes here
      while (max_dist == 1 || ps->size() > target) {
        int what = 0;
        for (; i < ps->size(); i++, j = i + 1) {
          for (; j < ps->size(); j++)
            if ((what = verdict(i, j, max_dist)) != 0) break;
          if (what != 0) break;
        }
        if (what == 0) break;
        for (;;) {
          if (what == 2) {
            for (int k = 0; k < len; k++)
              if ((*ps)[i].wild[k] != (*ps)[j].wild[k] || (*ps)[i].byte[k] != (*ps)[j].byte[k]) {
                if (!(*ps)[i].wild[k]) wilds++;
                (*ps)[i].wild[k] = true;
                (*ps)[i].byte[k] = 0;
              }
          }
          ps->erase(ps->begin() + static_cast<long>(j));
          if (what == 1) break;  // (pattern i is what it was: no earlier pair has changed)
          if (!(max_dist == 1 || ps->size() > target)) break;
          // pattern i changed: an earlier pattern may merge with it now
          what = 0;
          for (size_t a = 0; a < i; a++)
            if ((what = verdict(a, i, max_dist)) != 0) {
              j = i;
              i = a;
              break;
            }
          if (what == 0) break;
        }
        j = what == 1 ? j : i + 1;  // a duplicate removed: go on where the scan was; else pattern i meets everyone again
      }
    }
    return ps->size() <= target ? wilds : -1;
  }

  void plan(const Bits& first_any) {
    p_->mode = ScanMode::Dense;
    p_->windows.clear();
    if (p_->any_nullable || p_->min_len == 0 || p_->min_len == Program::kUnboundedLen) return;
    const int wl = static_cast<int>(std::min<uint64_t>(8, p_->min_len));
    const uint64_t max_d = std::min<uint64_t>(p_->min_len - static_cast<uint64_t>(wl), 48);
    std::vector<Pattern> best;
    int best_fixed = -1;  // exact (non-wildcard) bytes of the weakest window
    uint64_t best_d = 0;
    Bits level = first_any;
    std::vector<Pattern> last_raw, last_merged;
    bool last_ok = false;
    for (uint64_t d = 0; d <= max_d; d++) {
      std::vector<Pattern> ps;
      Pattern cur{};
      bool overflow = false;
      enumerate(level, 0, wl, &cur, &ps, &overflow);
      if (!overflow && !ps.empty()) {
        // (a repetition offers the same strings at one offset after the other: merge them once)
        const bool same = ps.size() == last_raw.size() &&
                          std::memcmp(ps.data(), last_raw.data(), ps.size() * sizeof(Pattern)) == 0;
        if (same) {
          ps = last_merged;
        } else {
          last_raw = ps;
          last_ok = merge_patterns(&ps, wl, kMaxWindows) >= 0;
          last_merged = ps;
        }
        if (last_ok) {
          int weakest = wl;
          for (const Pattern& pt : ps) {
            int fixed = 0;
            for (int k = 0; k < wl; k++) fixed += !pt.wild[k];
            weakest = std::min(weakest, fixed);
          }
          // prefer: more exact bytes in the weakest window, then fewer windows, then smaller d
          bool better = best.empty() || weakest > best_fixed || (weakest == best_fixed && ps.size() < best.size());
          if (weakest >= 1 && better) {
            best = ps;
            best_fixed = weakest;
            best_d = d;
          }
        }
      }
      level = step_any(level);
      if (!level.any()) break;
    }
    if (best.empty()) {
      plan_floating(first_any);
      return;
    }
    p_->mode = ScanMode::Windows;
    for (const Pattern& pt : best) {
      FFWindow w{};
      w.offset = static_cast<uint32_t>(best_d);
      w.len = static_cast<uint32_t>(wl);
      for (int k = 0; k < wl; k++) {
        if (pt.wild[k]) continue;
        if (k < 4) {
          w.value0 |= static_cast<uint32_t>(pt.byte[k]) << (8 * k);
          w.mask0 |= 0xFFu << (8 * k);
        } else {
          w.value1 |= static_cast<uint32_t>(pt.byte[k]) << (8 * (k - 4));
          w.mask1 |= 0xFFu << (8 * (k - 4));
        }
      }
      p_->windows.push_back(w);
    }
  }

  // ---- floating windows: a set of literal edges that every match has to cross ----
  bool is_cut(const std::vector<char>& chosen) const {
    // can the exit be reached from the entry without using a chosen byte edge?
    std::vector<char> seen(static_cast<size_t>(g_.n_states), 0);
    std::vector<int> stack{g_.entry};
    seen[static_cast<size_t>(g_.entry)] = 1;
    while (!stack.empty()) {
      int q = stack.back();
      stack.pop_back();
      if (q == g_.exit) return false;
      for (const ControlEdge* c : ctrl_[static_cast<size_t>(q)])
        if (!seen[static_cast<size_t>(c->dst)]) {
          seen[static_cast<size_t>(c->dst)] = 1;
          stack.push_back(c->dst);
        }
      for (int e : out_edges_[static_cast<size_t>(q)]) {
        if (chosen[static_cast<size_t>(e)]) continue;
        int d = g_.byte_edges[static_cast<size_t>(e)].dst;
        if (!seen[static_cast<size_t>(d)]) {
          seen[static_cast<size_t>(d)] = 1;
          stack.push_back(d);
        }
      }
    }
    return true;
  }

  void plan_floating(const Bits& first_any) {
    const size_t ne = g_.byte_edges.size();
    // candidate cuts, strongest first: one literal edge alone (longest first), then all literal
    // edges of at least L bytes for L = 8..2
    std::vector<std::vector<char>> tries;
    std::vector<size_t> by_len;
    for (size_t e = 0; e < ne; e++)
      if (g_.byte_edges[e].bytes.size() >= 2) by_len.push_back(e);
    std::sort(by_len.begin(), by_len.end(), [&](size_t a, size_t b) {
      return g_.byte_edges[a].bytes.size() > g_.byte_edges[b].bytes.size();
    });
    for (size_t e : by_len) {
      std::vector<char> c(ne, 0);
      c[e] = 1;
      tries.push_back(std::move(c));
      if (tries.size() >= 64) break;
    }
    for (size_t L = 8; L >= 2; L--) {
      std::vector<char> c(ne, 0);
      int n = 0;
      for (size_t e = 0; e < ne; e++)
        if (g_.byte_edges[e].bytes.size() >= L) { c[e] = 1; n++; }
      if (n) tries.push_back(std::move(c));
    }
    // depth range of every position: dmin by BFS levels, dmax by longest path (inf on a cycle)
    std::vector<uint32_t> dmin(static_cast<size_t>(P_), 0xFFFFFFFFu);
    {
      Bits level = first_any, seen(W_);
      uint32_t depth = 0;
      while (level.any()) {
        level.for_each([&](int i) { dmin[static_cast<size_t>(i)] = depth; });
        seen |= level;
        Bits next = step_any(level);
        for (int k = 0; k < W_; k++) next.w[static_cast<size_t>(k)] &= ~seen.w[static_cast<size_t>(k)];
        level = next;
        depth++;
      }
    }
    std::vector<std::vector<int>> preds(static_cast<size_t>(P_));
    for (int i = 0; i < P_; i++) follow_any(i).for_each([&](int j) { preds[static_cast<size_t>(j)].push_back(i); });
    constexpr uint32_t kInf = 0xFFFFFFFFu;
    std::vector<int> colour(static_cast<size_t>(P_), 0);
    std::vector<uint32_t> dmax(static_cast<size_t>(P_), 0);
    std::function<uint32_t(int)> longest = [&](int i) -> uint32_t {
      if (colour[static_cast<size_t>(i)] == 2) return dmax[static_cast<size_t>(i)];
      if (colour[static_cast<size_t>(i)] == 1) return kInf;  // cycle among the ancestors
      colour[static_cast<size_t>(i)] = 1;
      uint32_t best = 0;  // a first position can be entered with nothing consumed
      bool is_first = first_any.get(i);
      bool any_pred = false;
      for (int q : preds[static_cast<size_t>(i)]) {
        uint32_t v = longest(q);
        if (v == kInf) { best = kInf; break; }
        best = std::max(best, v + 1);
        any_pred = true;
      }
      if (!is_first && !any_pred) best = kInf;  // unreachable: treat as unusable
      colour[static_cast<size_t>(i)] = 2;
      dmax[static_cast<size_t>(i)] = best;
      return best;
    };
    for (const std::vector<char>& chosen : tries) {
      if (!is_cut(chosen)) continue;
      // windows = the first min(8, len) bytes of the distinct literals; common length
      size_t wl = 8;
      std::set<std::string> lits;
      uint32_t lo = kInf, hi = 0;
      bool ok = true;
      for (size_t e = 0; e < ne && ok; e++) {
        if (!chosen[e]) continue;
        const std::string& b = g_.byte_edges[e].bytes;
        wl = std::min(wl, b.size());
        int pos = edge_first_[e];
        if (dmin[static_cast<size_t>(pos)] == 0xFFFFFFFFu) continue;  // unreachable copy
        uint32_t mx = longest(pos);
        if (mx == kInf) ok = false;
        lo = std::min(lo, dmin[static_cast<size_t>(pos)]);
        hi = std::max(hi, mx);
      }
      if (!ok || lo == kInf || hi - lo > 255) {
        // no bounded distance from the match start: remember the strongest such cut for plan_behind
        if (behind_try.empty()) behind_try = chosen;
        continue;
      }
      for (size_t e = 0; e < ne; e++)
        if (chosen[e]) lits.insert(g_.byte_edges[e].bytes.substr(0, wl));
      if (lits.empty() || lits.size() > static_cast<size_t>(kMaxWindows)) continue;
      p_->mode = ScanMode::Windows;
      p_->floating = true;
      p_->float_min = lo;
      p_->float_max = hi;
      for (const std::string& lit : lits) {
        FFWindow w{};
        w.offset = 0;
        w.len = static_cast<uint32_t>(wl);
        for (size_t k = 0; k < wl; k++) {
          const uint32_t c = static_cast<uint8_t>(lit[k]);
          if (k < 4) { w.value0 |= c << (8 * k); w.mask0 |= 0xFFu << (8 * k); }
          else { w.value1 |= c << (8 * (k - 4)); w.mask1 |= 0xFFu << (8 * (k - 4)); }
        }
        p_->windows.push_back(w);
      }
      return;
    }
    // single-byte literals were not tried above (a floating 1-byte window is too weak to pay for the
    // 256 starts it implies); behind an unbounded prefix they still beat walking every start
    if (behind_try.empty()) {
      for (size_t L = 8; L >= 1 && behind_try.empty(); L--) {
        std::vector<char> c(ne, 0);
        int n = 0;
        for (size_t e = 0; e < ne; e++)
          if (g_.byte_edges[e].bytes.size() >= L) { c[e] = 1; n++; }
        if (n && is_cut(c)) behind_try = c;
      }
    }
    if (!behind_try.empty()) plan_behind(behind_try);
  }

  // windows = the first min(8, len) bytes of the chosen literal edges (a cut); every window carries
  // the positions of the first byte of the edges it stands for
  void plan_behind(const std::vector<char>& chosen) {
    const size_t ne = g_.byte_edges.size();
    if (W_ > 4) return;  // the backward pass is a per-lane walk
    size_t wl = 8;
    for (size_t e = 0; e < ne; e++)
      if (chosen[e]) wl = std::min(wl, g_.byte_edges[e].bytes.size());
    std::vector<std::string> lits;
    std::vector<std::vector<uint32_t>> cuts;
    for (size_t e = 0; e < ne; e++) {
      if (!chosen[e]) continue;
      const std::string head = g_.byte_edges[e].bytes.substr(0, wl);
      size_t k = 0;
      while (k < lits.size() && lits[k] != head) k++;
      if (k == lits.size()) {
        lits.push_back(head);
        cuts.emplace_back(static_cast<size_t>(W_), 0u);
      }
      set_bit(cuts[k].data(), edge_first_[e]);
    }
    if (lits.empty() || lits.size() > static_cast<size_t>(kMaxWindows)) return;
    p_->mode = ScanMode::Windows;
    p_->behind = true;
    p_->floating = false;
    p_->windows.clear();
    p_->cut_positions = cuts;
    for (const std::string& lit : lits) {
      FFWindow w{};
      w.offset = 0;
      w.len = static_cast<uint32_t>(wl);
      for (size_t k = 0; k < wl; k++) {
        const uint32_t c = static_cast<uint8_t>(lit[k]);
        if (k < 4) { w.value0 |= c << (8 * k); w.mask0 |= 0xFFu << (8 * k); }
        else { w.value1 |= c << (8 * (k - 4)); w.mask1 |= 0xFFu << (8 * (k - 4)); }
      }
      p_->windows.push_back(w);
    }
  }

  std::vector<char> behind_try;

  // all byte patterns of `remaining` more positions readable from position set `level`
  // (capped); a depth with more than 16 possible byte values becomes a wildcard
  static constexpr size_t kMaxEnumerated = 256;

  void enumerate(const Bits& level, int depth, int len, Pattern* cur, std::vector<Pattern>* found, bool* overflow) {
    if (*overflow) return;
    if (depth == len) {
      found->push_back(*cur);
      if (found->size() > kMaxEnumerated) *overflow = true;
      return;
    }
    // the byte values some position of the level consumes (no set is built for the others, nor for any once there
    // are more than 16: this loop was most of the compile time of patterns with wide repetitions)
    int values[17];
    size_t n_values = 0;
    for (int b = 0; b < 256 && n_values <= 16; b++) {
      const uint32_t* row = &p_->cls[static_cast<size_t>(b) * W_];
      uint32_t any = 0;
      for (int k = 0; k < W_; k++) any |= level.w[static_cast<size_t>(k)] & row[k];
      if (any) values[n_values++] = b;
    }
    std::vector<std::pair<int, Bits>> branches;
    if (n_values <= 16) {
      branches.reserve(n_values);
      for (size_t v = 0; v < n_values; v++) {
        const uint32_t* row = &p_->cls[static_cast<size_t>(values[v]) * W_];
        Bits hit(W_);
        for (int k = 0; k < W_; k++) hit.w[static_cast<size_t>(k)] = level.w[static_cast<size_t>(k)] & row[k];
        branches.emplace_back(values[v], std::move(hit));
      }
    }
    if (n_values > 16) {  // a wide class ('.', [a-z], ...): do not split on it
      cur->byte[depth] = 0;
      cur->wild[depth] = true;
      Bits next = depth + 1 < len ? step_any(level) : level;
      if (depth + 1 < len && !next.any()) return;
      enumerate(next, depth + 1, len, cur, found, overflow);
      return;
    }
    for (auto& br : branches) {
      if (*overflow) return;
      cur->byte[depth] = static_cast<uint8_t>(br.first);
      cur->wild[depth] = false;
      Bits next = depth + 1 < len ? step_any(br.second) : br.second;
      // a match cannot be shorter than d + len, so every real path continues; an empty
      // continuation can only come from the context over-approximation -- drop it
      if (depth + 1 < len && !next.any()) continue;
      enumerate(next, depth + 1, len, cur, found, overflow);
    }
  }

  // follow set of position i in context ctx
  Bits follow_ctx(int i, int ctx) const {
    Bits f(W_);
    if ((p_->linear[static_cast<size_t>(i) >> 5] >> (i & 31)) & 1u) {
      f.set(i + 1);
      return f;
    }
    const int r = p_->row_of[static_cast<size_t>(i)];
    for (int k = 0; k < W_; k++) f.w[static_cast<size_t>(k)] = p_->rows[ctx][static_cast<size_t>(r) * W_ + k];
    return f;
  }

  // Program::rev: positions back to front, follow transposed per context
  void build_reverse() {
    Program::Reverse& R = p_->rev;
    const auto rv = [&](int i) { return P_ - 1 - i; };
    const auto flip = [&](const std::vector<uint32_t>& in) {
      std::vector<uint32_t> out(static_cast<size_t>(W_), 0u);
      for (int i = 0; i < P_; i++)
        if ((in[static_cast<size_t>(i) >> 5] >> (i & 31)) & 1u) set_bit(out.data(), rv(i));
      return out;
    };
    for (int ctx = 0; ctx < kNumCtx; ctx++) {
      R.first[ctx] = flip(p_->last[ctx]);
      R.last[ctx] = flip(p_->first[ctx]);
    }
    R.cls.assign(static_cast<size_t>(256) * W_, 0u);
    for (int b = 0; b < 256; b++) {
      std::vector<uint32_t> row(p_->cls.begin() + static_cast<long>(b) * W_, p_->cls.begin() + static_cast<long>(b + 1) * W_);
      std::vector<uint32_t> f = flip(row);
      std::copy(f.begin(), f.end(), R.cls.begin() + static_cast<long>(b) * W_);
    }
    // rfollow_ctx[r(k)] = { r(i) : k in follow_ctx(i) }
    std::vector<Bits> rf[kNumCtx];
    for (int ctx = 0; ctx < kNumCtx; ctx++) {
      rf[ctx].assign(static_cast<size_t>(std::max(P_, 1)), Bits(W_));
      for (int i = 0; i < P_; i++) follow_ctx(i, ctx).for_each([&](int k) { rf[ctx][static_cast<size_t>(rv(k))].set(rv(i)); });
    }
    R.linear.assign(static_cast<size_t>(W_), 0u);
    R.row_of.assign(static_cast<size_t>(std::max(P_, 1)), -1);
    R.n_rows = 0;
    std::vector<std::vector<uint32_t>> rows[kNumCtx];
    for (int j = 0; j < P_; j++) {
      bool lin = j + 1 < P_;
      for (int ctx = 0; ctx < kNumCtx && lin; ctx++) {
        Bits want(W_);
        want.set(j + 1);
        lin = rf[ctx][static_cast<size_t>(j)] == want;
      }
      if (lin) {
        set_bit(R.linear.data(), j);
      } else {
        R.row_of[static_cast<size_t>(j)] = R.n_rows++;
        for (int ctx = 0; ctx < kNumCtx; ctx++) rows[ctx].push_back(rf[ctx][static_cast<size_t>(j)].w);
      }
    }
    for (int ctx = 0; ctx < kNumCtx; ctx++) {
      R.rows[ctx].assign(static_cast<size_t>(std::max(R.n_rows, 1)) * W_, 0u);
      for (int r = 0; r < R.n_rows; r++)
        std::copy(rows[ctx][static_cast<size_t>(r)].begin(), rows[ctx][static_cast<size_t>(r)].end(),
                  R.rows[ctx].begin() + static_cast<long>(r) * W_);
    }
  }

  const Graph& g_;
  Program* p_;
  int P_ = 0, W_ = 1;
  std::vector<uint32_t> mark_;  // closure(): visited marks
  uint32_t mark_epoch_ = 0;
  std::vector<int> edge_first_;
  std::vector<std::vector<int>> out_edges_;
  std::vector<std::vector<const ControlEdge*>> ctrl_;
};

void find_literal(const Node& n, std::string* out, bool* ok) {
  if (!*ok) return;
  if (n.kind == NodeKind::Literal) {
    *out += n.bytes;
  } else if (n.kind == NodeKind::Concat) {
    for (const auto& k : n.kids) find_literal(*k, out, ok);
  } else {
    *ok = false;
  }
}

}  // namespace

// Two threads of the reference's ring, started at b < s and both alive at time t: when the positions the
// younger one holds are always a subset of the older one's, "left-most start wins" (SetState,
// reference src/x64/codegen-x64.cc:951-987) gives every slot the younger would occupy to the older, i.e.
// the younger thread does not exist.  Checked on the position automaton by exploring the pairs
// (older set, younger set) over the pattern's byte classes: the younger starts with first & cls[c] while the
// older, already alive, steps on the same byte; then both step together.  Conservative: assertions, more
// than 64 positions or too many pairs answer "no".
bool threads_always_nested(const Program& P) {
  if (P.has_assertions || P.n_pos == 0 || P.n_pos > 64) return false;
  const int W = P.n_words;
  auto word64 = [&](const std::vector<uint32_t>& v, size_t off) {
    uint64_t x = v[off];
    if (W > 1) x |= static_cast<uint64_t>(v[off + 1]) << 32;
    return x;
  };
  std::vector<uint64_t> fol(static_cast<size_t>(P.n_pos), 0);
  for (int p = 0; p < P.n_pos; p++) {
    const int r = P.row_of[static_cast<size_t>(p)];
    if (r < 0) {
      if (p + 1 < P.n_pos) fol[static_cast<size_t>(p)] = 1ull << (p + 1);
    } else {
      fol[static_cast<size_t>(p)] = word64(P.rows[0], static_cast<size_t>(r) * W);
    }
  }
  std::vector<uint64_t> classes;  // distinct, non-empty class rows
  for (int b = 0; b < 256; b++) {
    const uint64_t row = word64(P.cls, static_cast<size_t>(b) * W);
    if (row != 0 && std::find(classes.begin(), classes.end(), row) == classes.end()) classes.push_back(row);
  }
  const uint64_t F = word64(P.first[0], 0);
  auto step = [&](uint64_t S, uint64_t row) {
    uint64_t T = 0;
    for (uint64_t m = S; m; m &= m - 1) T |= fol[static_cast<size_t>(__builtin_ctzll(m))];
    return T & row;
  };
  // every set an alive thread can hold
  std::vector<uint64_t> singles;
  auto add_single = [&](uint64_t S) {
    if (S != 0 && std::find(singles.begin(), singles.end(), S) == singles.end()) singles.push_back(S);
  };
  for (uint64_t row : classes) add_single(F & row);
  for (size_t i = 0; i < singles.size(); i++) {
    if (singles.size() > 4096) return false;
    for (uint64_t row : classes) add_single(step(singles[i], row));
  }
  std::vector<std::pair<uint64_t, uint64_t>> pairs;
  auto add_pair = [&](uint64_t B, uint64_t S) {
    if (B == 0 || S == 0) return true;   // one of them is dead: nothing to absorb
    if (S & ~B) return false;            // the younger thread holds a position of its own
    const std::pair<uint64_t, uint64_t> q(B, S);
    if (std::find(pairs.begin(), pairs.end(), q) == pairs.end()) pairs.push_back(q);
    return true;
  };
  for (uint64_t B : singles)
    for (uint64_t row : classes)
      if (!add_pair(step(B, row), F & row)) return false;
  for (size_t i = 0; i < pairs.size(); i++) {
    if (pairs.size() > 8192) return false;
    for (uint64_t row : classes)
      if (!add_pair(step(pairs[i].first, row), step(pairs[i].second, row))) return false;
  }
  return true;
}

LowerResult lower(const char* regexp) {
  LowerResult r;
  ParseResult pr = parse(regexp);
  if (pr.status != kParseOk) {
    r.status = pr.status;
    r.message = pr.message;
    return r;
  }
  Graph g;
  if (!build_graph(*pr.root, &g, &r.message)) {
    r.status = -2;
    return r;
  }
  auto prog = std::make_unique<Program>();
  Automaton a(g, prog.get());
  if (!a.build(&r.message)) {
    r.status = -2;
    return r;
  }
  // Q8 risk: closure(entry) \ {entry} intersects the closure of some byte-edge destination
  {
    std::vector<std::vector<int>> adj(static_cast<size_t>(g.n_states));
    for (const ControlEdge& c : g.control_edges) adj[static_cast<size_t>(c.src)].push_back(c.dst);
    auto reach = [&](const std::vector<int>& seeds) {
      std::vector<char> seen(static_cast<size_t>(g.n_states), 0);
      std::vector<int> stack = seeds;
      for (int s0 : seeds) seen[static_cast<size_t>(s0)] = 1;
      while (!stack.empty()) {
        int s0 = stack.back();
        stack.pop_back();
        for (int d : adj[static_cast<size_t>(s0)])
          if (!seen[static_cast<size_t>(d)]) {
            seen[static_cast<size_t>(d)] = 1;
            stack.push_back(d);
          }
      }
      return seen;
    };
    std::vector<char> from_entry = reach({g.entry});
    std::vector<int> landings;
    for (const ByteEdge& e : g.byte_edges) landings.push_back(e.dst);
    std::vector<char> from_landing = reach(landings);
    for (int q = 0; q < g.n_states; q++)
      if (from_entry[static_cast<size_t>(q)] && from_landing[static_cast<size_t>(q)]) prog->q8_risk = true;  // q may be the entry itself
    // ... and the artefact needs a thread that started INSIDE a match and still owns a ring slot when the
    // match is found; where every younger thread is absorbed by the older one that is alive (`X+ rest`,
    // `x*`, `[acgt]+`: "left-most start wins" leaves one thread) there is no such thread
    if (prog->q8_risk && threads_always_nested(*prog)) prog->q8_risk = false;
  }
  prog->graph = g;
  bool lit = true;
  std::string bytes;
  find_literal(*pr.root, &bytes, &lit);
  if (lit) prog->literal = bytes;
  r.program = std::move(prog);
  return r;
}

}  // namespace rejit_amd
