// rejit_amd/csrc/carry_scan.h -- the linear-time matcher ("carry scan"): per-sub-chunk bodies shared
// by the HIP kernels (carry_kernels.hip) and by the CPU unit tests (tests/support/carry_exec.cc
// compiles this very header with g++), so the algorithm is checked against the oracle without a GPU.
//
// Why it exists.  The reference's NFA loop is linear in the text whatever the pattern: a state holds
// ONE thread, the left-most start wins (SetState, src/x64/codegen-x64.cc:951-987), dominated threads
// are killed (ClearStates, :1075-1097), one byte per iteration (GenerateMatchDirection, :535-640).
// The parallel verifier of this library walks every candidate start on its own, which is quadratic
// when many starts stay alive for long (`[acgt]+` over a 250 MB run of DNA).  This path computes the
// same result in O(n * live states), parallel over sub-chunks of the text:
//
//   E(s) = longest end of a match that begins at s, for EVERY s, in ONE backward pass with the
//   REVERSE automaton (lowering.h: Program::rev).  Going backwards the state is, per position k of
//   the automaton, D_k(p) = the largest end reachable by a thread that consumes text[p] at k; a step
//   only MOVES values (D'_k = max over the followers of k), so positions holding the same value move
//   together: the state is a short list of (value, position set) CLASSES in descending value order,
//   a step is the ordinary bit-parallel  S' = follow(S) & cls[byte]  per class, masked by the classes
//   before it ("the larger end wins" -- the mirror image of the reference's "left-most start wins").
//
//   The recurrence is linear over (max, select), so a sub-chunk's effect on the state entering at its
//   right edge is a transfer function:  D(a) = max(L, R x D(b))  with L the locally born values and
//   R a P x P reachability matrix (which entering positions survive the whole sub-chunk, and where
//   they arrive).  Three phases, like a decoupled scan:
//     1. summarize  (cs_summarize)   every sub-chunk on its own lane: L and R, the entering state
//                                    carried SYMBOLICALLY (one source per position)
//     2. resolve    (cs_resolve)     right to left over the summaries: the true D at every boundary
//     3. emit       (cs_emit)        every sub-chunk again, now with the true entering state: E(s)
//
//   Selection (left-most longest, non-overlapping; MatchAllAppendFilter + CheckMatch, reference
//   src/codegen.cc:36-86, codegen-x64.cc:401-466) is a chain  cur -> max(E(s), s + 1), s = first
//   start >= cur with a match; it is sequential, but a sub-chunk's part of it depends only on where
//   the chain ENTERS the sub-chunk:
//     4. cs_local_chain   per sub-chunk, right to left: G[p] = where a chain entering at p leaves
//     5. cs_global_chain  one thread hops from sub-chunk to sub-chunk through G: the entry points
//     6. cs_take          per entered sub-chunk: follow the chain inside, compact the taken
//                         (begin, end) pairs to the front of the sub-chunk's slab
//   and the existing offsets_gather_check / selection tail lays them out (regions = sub-chunks) and
//   applies the zero-length rule.
#ifndef REJIT_AMD_CARRY_SCAN_H_
#define REJIT_AMD_CARRY_SCAN_H_

#include <stdint.h>

#include "device_program.h"

namespace rejit_amd {

constexpr uint64_t kCsNone = ~0ull;  // E(s): no match begins at s;  entry[c]: the chain skips sub-chunk c

// Lane-private arrays live in memory shared by the 64 lanes of a wave (LDS or a global scratch
// slab): element i of this lane's array is base[i * stride].  stride = 1 on the CPU.
template <class T>
struct CsArr {
  T* base;
  int stride;
  RJ_HD T& operator[](int i) const { return base[static_cast<size_t>(i) * static_cast<size_t>(stride)]; }
};

// boundary context as in rj_context, 0 when the pattern has no assertions
template <class Text>
RJ_HD int cs_context(const DevProgram& R, const Text& t, uint64_t n, uint64_t q) {
  return R.n_ctx > 1 ? rj_context(t, n, q) : 0;
}

// T = follow_ctx(S): linear positions shift, the others OR their rows in (words beyond W stay 0)
template <int NW>
RJ_HD void cs_follow(const DevProgram& R, const uint32_t (&S)[NW], int ctx, uint32_t (&T)[NW]) {
  const int W = R.n_words;
  uint32_t carry = 0;
#pragma unroll
  for (int k = 0; k < NW; k++) {
    T[k] = 0;
    if (k < W) {
      const uint32_t x = S[k] & R.linear[k];
      T[k] = (x << 1) | carry;
      carry = x >> 31;
    }
  }
#pragma unroll
  for (int k = 0; k < NW; k++) {
    if (k >= W) continue;
    uint32_t sp = S[k] & ~R.linear[k];
    while (sp) {
      const int b = __builtin_ctz(sp);
      sp &= sp - 1;
      const uint32_t* row = R.rows + (static_cast<size_t>(ctx) * R.n_rows + R.row_of[k * 32 + b]) * W;
#pragma unroll
      for (int j = 0; j < NW; j++)
        if (j < W) T[j] |= row[j];
    }
  }
}

// State of the backward scan of one sub-chunk: `nc` classes (value descending, sets disjoint).
template <int NW>
struct CsClasses {
  CsArr<uint64_t> val;   // [P]
  CsArr<uint32_t> set;   // [P * NW], class i at [i * NW + k]
  int nc;
};

// One backward step over text[p] (boundary context ctx = context at p + 1): every class follows and
// is gated by the byte's class row and by the classes before it; threads that END at p + 1 are born
// as the last class.  Empty classes disappear.
template <int NW>
RJ_HD void cs_step_classes(const DevProgram& R, CsClasses<NW>& C, int ctx, const uint32_t* clsrow, uint64_t born_value) {
  const int W = R.n_words;
  uint32_t taken[NW];
#pragma unroll
  for (int k = 0; k < NW; k++) taken[k] = 0;
  int j = 0;
  for (int i = 0; i < C.nc; i++) {
    uint32_t S[NW], T[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) S[k] = k < W ? C.set[i * NW + k] : 0u;
    cs_follow<NW>(R, S, ctx, T);
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) {
      T[k] = k < W ? (T[k] & clsrow[k] & ~taken[k]) : 0u;
      any |= T[k];
    }
    if (any) {
      const uint64_t v = C.val[i];
#pragma unroll
      for (int k = 0; k < NW; k++) {
        if (k < W) C.set[j * NW + k] = T[k];
        taken[k] |= T[k];
      }
      C.val[j] = v;
      j++;
    }
  }
  uint32_t any = 0;
  uint32_t seed[NW];
  const uint32_t* fr = R.first + static_cast<size_t>(ctx) * W;
#pragma unroll
  for (int k = 0; k < NW; k++) {
    seed[k] = k < W ? (fr[k] & clsrow[k] & ~taken[k]) : 0u;
    any |= seed[k];
  }
  if (any) {
#pragma unroll
    for (int k = 0; k < NW; k++)
      if (k < W) C.set[j * NW + k] = seed[k];
    C.val[j] = born_value;
    j++;
  }
  C.nc = j;
}

// ---- phase 1: summary of the sub-chunk [a, b) (b <= n).
//   Lval[k]     largest end of a thread born INSIDE the sub-chunk that sits at position k at `a`
//               (has consumed text[a]); 0 = none
//   Rmat[m][W]  positions at `a` reached by the thread that enters at position m at `b` (has
//               consumed text[b]); all zero when it dies inside (the usual case)
// Scratch: C (classes), src (P * NW words).
template <int NW>
RJ_HD void cs_summarize(const DevProgram& R, const uint8_t* t, uint64_t n, uint64_t a, uint64_t b, CsClasses<NW>& C,
                        CsArr<uint32_t> src, uint64_t* Lval, uint32_t* Rmat) {
  const int W = R.n_words, P = R.n_pos;
  uint32_t alive[NW];
#pragma unroll
  for (int k = 0; k < NW; k++) alive[k] = 0;
  C.nc = 0;
  if (b < n) {
    for (int m = 0; m < P; m++) {
      for (int k = 0; k < W; k++) src[m * NW + k] = 0;
      src[m * NW + (m >> 5)] = 1u << (m & 31);
    }
#pragma unroll
    for (int k = 0; k < NW; k++) {
      if (k >= W) continue;
      const int left = P - 32 * k;
      alive[k] = left >= 32 ? 0xFFFFFFFFu : left > 0 ? (1u << left) - 1u : 0u;
    }
  }
  for (uint64_t p = b; p-- > a;) {
    const int ctx = cs_context(R, t, n, p + 1);
    const uint32_t* clsrow = R.cls + static_cast<size_t>(t[p]) * W;
    cs_step_classes<NW>(R, C, ctx, clsrow, p + 1);
#pragma unroll
    for (int k0 = 0; k0 < NW; k0++) {
      if (k0 >= W) continue;
      uint32_t todo = alive[k0];
      while (todo) {
        const int bit = __builtin_ctz(todo);
        todo &= todo - 1;
        const int m = k0 * 32 + bit;
        uint32_t S[NW], T[NW];
#pragma unroll
        for (int k = 0; k < NW; k++) S[k] = k < W ? src[m * NW + k] : 0u;
        cs_follow<NW>(R, S, ctx, T);
        uint32_t any = 0;
#pragma unroll
        for (int k = 0; k < NW; k++) {
          T[k] = k < W ? (T[k] & clsrow[k]) : 0u;
          any |= T[k];
        }
        if (any) {
#pragma unroll
          for (int k = 0; k < NW; k++)
            if (k < W) src[m * NW + k] = T[k];
        } else {
          alive[k0] &= ~(1u << bit);
        }
      }
    }
  }
  for (int k = 0; k < P; k++) Lval[k] = 0;
  for (int i = 0; i < C.nc; i++) {
    const uint64_t v = C.val[i];
    for (int k = 0; k < W; k++) {
      uint32_t s = C.set[i * NW + k];
      while (s) {
        const int bit = __builtin_ctz(s);
        s &= s - 1;
        Lval[k * 32 + bit] = v;
      }
    }
  }
  for (int m = 0; m < P; m++) {
    const bool live = (alive[m >> 5] >> (m & 31)) & 1u;
    for (int k = 0; k < W; k++) Rmat[static_cast<size_t>(m) * W + k] = live ? src[m * NW + k] : 0u;
  }
}

// ---- phase 2, one sub-chunk: D(a) = max(L, R x D(b)).  `D` holds Lval on entry and the resolved
// values on return; Dnext = resolved values of the sub-chunk to the right (zeros beyond the text).
RJ_HD void cs_resolve(int P, int W, uint64_t* D, const uint32_t* Rmat, const uint64_t* Dnext) {
  for (int m = 0; m < P; m++) {
    const uint64_t v = Dnext[m];
    if (v == 0) continue;
    for (int k = 0; k < W; k++) {
      uint32_t s = Rmat[static_cast<size_t>(m) * W + k];
      while (s) {
        const int bit = __builtin_ctz(s);
        s &= s - 1;
        if (D[k * 32 + bit] < v) D[k * 32 + bit] = v;
      }
    }
  }
}

// Two-level resolve: the transfer of a GROUP of consecutive sub-chunks composed into one, so that the
// sequential pass runs over groups (cs_resolve per group), not over sub-chunks.  (Lc_in, Rc_in) =
// the transfer of the sub-chunks to the right inside the group (identity / zeros to begin with);
// (L, R) = the next sub-chunk to the left:   Lc_out = L (+) R x Lc_in,   Rc_out[m] = R x Rc_in[m].
RJ_HD void cs_compose(int P, int W, const uint64_t* L, const uint32_t* R, const uint64_t* Lc_in, const uint32_t* Rc_in,
                      uint64_t* Lc_out, uint32_t* Rc_out) {
  for (int k = 0; k < P; k++) Lc_out[k] = L[k];
  cs_resolve(P, W, Lc_out, R, Lc_in);
  for (int m = 0; m < P; m++) {
    for (int j = 0; j < W; j++) Rc_out[static_cast<size_t>(m) * W + j] = 0;
    for (int k0 = 0; k0 < W; k0++) {
      uint32_t s = Rc_in[static_cast<size_t>(m) * W + k0];
      while (s) {
        const int bit = __builtin_ctz(s);
        s &= s - 1;
        const uint32_t* row = R + static_cast<size_t>(k0 * 32 + bit) * W;
        for (int j = 0; j < W; j++) Rc_out[static_cast<size_t>(m) * W + j] |= row[j];
      }
    }
  }
}

// E(s) at boundary s from the state D(s): the largest value among the classes that hold a position
// able to START a match here (rev.last = forward first), else the empty match if allowed.
template <int NW>
RJ_HD uint64_t cs_longest_here(const DevProgram& R, const CsClasses<NW>& C, int ctx, uint64_t s) {
  const int W = R.n_words;
  const uint32_t* lr = R.last + static_cast<size_t>(ctx) * W;
  for (int i = 0; i < C.nc; i++) {
    uint32_t acc = 0;
    for (int k = 0; k < W; k++) acc |= C.set[i * NW + k] & lr[k];
    if (acc) return C.val[i];
  }
  return ((R.nullable >> (R.n_ctx > 1 ? ctx : 0)) & 1u) ? s : kCsNone;
}

// ---- phase 3: E(s) for the starts s of sub-chunk [a, b) that lie in [sb, se); Dnext = resolved D(b)
// (nullptr = nothing enters).  E is indexed from e_base.  The sub-chunk that holds b == n also owns
// the start position n (the empty match at the end of the text).
template <int NW>
RJ_HD void cs_emit(const DevProgram& R, const uint8_t* t, uint64_t n, uint64_t a, uint64_t b, uint64_t sb, uint64_t se,
                   const uint64_t* Dnext, CsClasses<NW>& C, uint64_t* E, uint64_t e_base) {
  const int W = R.n_words, P = R.n_pos;
  C.nc = 0;
  if (Dnext != nullptr) {
    // classes = the distinct values, largest first
    uint32_t left[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) left[k] = 0;
    for (int m = 0; m < P; m++)
      if (Dnext[m] != 0) left[m >> 5] |= 1u << (m & 31);
    for (;;) {
      uint64_t vmax = 0;
      for (int k = 0; k < W; k++) {
        uint32_t s = left[k];
        while (s) {
          const int bit = __builtin_ctz(s);
          s &= s - 1;
          const uint64_t v = Dnext[k * 32 + bit];
          vmax = v > vmax ? v : vmax;
        }
      }
      if (vmax == 0) break;
      for (int k = 0; k < W; k++) {
        uint32_t s = left[k], mine = 0;
        while (s) {
          const int bit = __builtin_ctz(s);
          s &= s - 1;
          if (Dnext[k * 32 + bit] == vmax) mine |= 1u << bit;
        }
        C.set[C.nc * NW + k] = mine;
        left[k] &= ~mine;
      }
      C.val[C.nc] = vmax;
      C.nc++;
    }
  }
  if (b == n && n >= sb && n < se) E[n - e_base] = ((R.nullable >> cs_context(R, t, n, n)) & 1u) ? n : kCsNone;
  for (uint64_t p = b; p-- > a;) {
    const int ctx = cs_context(R, t, n, p + 1);
    const uint32_t* clsrow = R.cls + static_cast<size_t>(t[p]) * W;
    cs_step_classes<NW>(R, C, ctx, clsrow, p + 1);
    if (p >= sb && p < se) E[p - e_base] = cs_longest_here<NW>(R, C, cs_context(R, t, n, p), p);
  }
}

// ---- phase 4: G[p] = the first chain position >= hi reached from a chain that stands at p
// (lo <= p < hi; hi = end of the sub-chunk's starts, clipped to se).
RJ_HD void cs_local_chain(const uint64_t* E, uint64_t* G, uint64_t e_base, uint64_t lo, uint64_t hi) {
  for (uint64_t p = hi; p-- > lo;) {
    const uint64_t e = E[p - e_base];
    uint64_t g;
    if (e != kCsNone) {
      const uint64_t t = e > p ? e : p + 1;
      g = t >= hi ? t : G[t - e_base];
    } else {
      g = p + 1 < hi ? G[p + 1 - e_base] : hi;
    }
    G[p - e_base] = g;
  }
}

// ---- phase 6 (one lane; the kernel has a wave-cooperative form): follow the chain that enters the
// sub-chunk's starts (they end at hi) at `cur`; the taken matches are compacted to the front of the
// sub-chunk's slab (index `slab` of E / G), begins into G, ends into E.  Returns their number.
RJ_HD uint32_t cs_take(uint64_t* E, uint64_t* G, uint64_t e_base, uint64_t slab, uint64_t hi, uint64_t cur) {
  uint32_t cnt = 0;
  while (cur < hi) {
    uint64_t s = cur;
    while (s < hi && E[s - e_base] == kCsNone) s++;
    if (s >= hi) break;
    const uint64_t e = E[s - e_base];
    G[slab + cnt] = s;  // slab + cnt <= s - e_base: behind the read position
    E[slab + cnt] = e;
    cnt++;
    cur = e > s ? e : s + 1;
  }
  return cnt;
}

}  // namespace rejit_amd
#endif
