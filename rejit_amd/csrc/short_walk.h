// rejit_amd/csrc/short_walk.h -- the automaton walk of SHORT bounded patterns (DevProgram::short_max <= 16) with the tables
// in LDS: what classify_shared_multi / classify_shared_general (plane_scan.hip) and the one-kernel count of general pattern
// sets (plane_count.hip) run per candidate.  Reference: the NFA inner loop, src/x64/codegen-x64.cc:535-640, restated as a
// position automaton (lowering.h).
#ifndef REJIT_AMD_SHORT_WALK_H_
#define REJIT_AMD_SHORT_WALK_H_

#include <hip/hip_runtime.h>

#include <type_traits>

#include "kernels.h"

namespace rejit_amd {

// Longest match from a start whose 16 text bytes are (lo, hi), `avail` of them inside the text; tables in LDS at
// t0:  first [W] last [W] linear [W] row_of [n_pos] rows [n_rows][W] cls [256][W]  (table_layout.h, one context).
// Same result as rj_lane_longest_short_at (device_program.h), written without divergent exits: the class rows of
// all bytes are fetched together (a row beyond the bytes a match may consume is zero, which kills the state),
// every step is straight-line code, and the walk through the rows of non-linear positions -- only those with a
// non-empty follow set count, ClassifyDesc::rowbits: none in an alternation of literals and classes -- sits
// behind a wave-uniform test.
template <int W, int MAXK>
__device__ __forceinline__ bool short_longest_lds(const uint32_t* t0, const ClassifyDesc& d, uint64_t lo, uint64_t hi, uint32_t avail,
                                                  uint32_t* length) {
  using State = typename std::conditional<W == 1, uint32_t, uint64_t>::type;
  const uint32_t NP = d.n_pos;
  const uint32_t steps = d.short_max < avail ? d.short_max : avail;  // bytes a match can consume here
  const uint32_t* cls = t0 + 3 * W + NP + d.n_rows * W;
  auto word = [&](const uint32_t* q) -> State {
    State v = q[0];
    if (W > 1) v |= static_cast<State>(static_cast<uint64_t>(q[W - 1]) << 32);
    return v;
  };
  State row[MAXK];
#pragma unroll
  for (int k = 0; k < MAXK; k++) {
    const uint32_t c = static_cast<uint32_t>(((k < 8 ? lo : hi) >> (8 * (k & 7))) & 0xFFu);
    const State v = word(cls + c * W);
    row[k] = static_cast<uint32_t>(k) < steps ? v : State{0};
  }
  const State first = word(t0), last = word(t0 + W), lin = word(t0 + 2 * W);
  State rowbits = d.rowbits[0];
  if (W > 1) rowbits |= static_cast<State>(static_cast<uint64_t>(d.rowbits[W - 1]) << 32);
  const int32_t* row_of = reinterpret_cast<const int32_t*>(t0 + 3 * W);
  const uint32_t* rows = t0 + 3 * W + NP;
  bool found = (d.nullable & 1u) != 0;
  uint32_t len = 0;
  State S = first & row[0];
#pragma unroll
  for (int k = 1; k <= MAXK; k++) {
    const bool acc = (S & last) != 0;  // S = the positions that consumed byte k - 1
    len = acc ? static_cast<uint32_t>(k) : len;
    found = found || acc;
    if (k < MAXK) {
      State T = (S & lin) << 1;
      State sp = S & rowbits;
      if (__ballot(sp != 0) != 0) {
        for (; sp; sp &= sp - 1) {
          const int bit = W == 1 ? __builtin_ctz(static_cast<uint32_t>(sp)) : __builtin_ctzll(static_cast<uint64_t>(sp));
          T |= word(rows + static_cast<uint32_t>(row_of[bit]) * W);
        }
      }
      S = T & row[k < MAXK ? k : 0];
    }
  }
  *length = len;
  return found;
}


}  // namespace rejit_amd
#endif
