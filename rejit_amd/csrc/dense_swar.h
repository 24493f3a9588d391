// rejit_amd/csrc/dense_swar.h -- the dense kernel's pre-steps with four starts per register.
//
// scan_dense_walk (kernels.hip) decides most starts of a 1-KiB chunk in registers before any walker sees
// them: a walk over random text is a byte or two long.  Round 1 did that with one 32-bit state per start
// and one LDS lookup per text byte (the class row), ~59 VALU instructions per text byte.  Here, for
// automata of <= 8 positions without assertions (DevProgram::swar):
//   * the class rows of four text bytes are ONE register, computed with byte-parallel range tests
//     (two adds and three bit operations per range and four bytes, no lookup);
//   * four consecutive starts share a register, one state byte each; at depth t they read the rows of
//     the bytes t further on, i.e. the row registers shifted by t bytes (v_alignbyte_b32);
//   * one step is  S' = ((S & step1) << 1 | (S & loop)) & rows  for four starts at once (no bit crosses
//     a byte: position 7 has no successor among 8 positions).
// The 16 per-start flags of a lane come out in "F layout": bit 8k + g = start 4g + k (group g = register
// g, k = byte in it) -- the four groups' byte flags OR-ed together with a shift each, no bit gathering.
// RJ_HD only: tests/support/carry_exec.cc compiles it with g++ and checks it against the scalar automaton.
#ifndef REJIT_AMD_DENSE_SWAR_H_
#define REJIT_AMD_DENSE_SWAR_H_

#include <stdint.h>

#include "device_program.h"

namespace rejit_amd {

// class rows of five text dwords (the lane's 16 bytes and the 4 after them): bit p of byte k = the byte
// may be consumed at position p
RJ_HD void rj_swar_rows5(const SwarPlan& pl, const uint32_t (&x)[5], uint32_t (&rows)[5]) {
  uint32_t x7[5], lowh[5], highh[5];
#pragma unroll
  for (int i = 0; i < 5; i++) {
    x7[i] = x[i] & 0x7f7f7f7fu;
    lowh[i] = ~x[i] & 0x80808080u;
    rows[i] = 0;
  }
  if (pl.n_ranges <= 4 && pl.n_low == pl.n_ranges) {
    // the usual pattern: a few ranges, all of them ASCII -- no choice between the halves per range and word, and
    // unrolled, so that the constants are loop-invariant scalars of the caller's chunk loop instead of three scalar
    // loads (and a wait) per range and chunk
#pragma unroll
    for (uint32_t r = 0; r < 4; r++) {
      if (r < pl.n_ranges) {
        const uint32_t lo = pl.add_lo[r], hi = pl.add_hi[r], sh = pl.shift[r];
#pragma unroll
        for (int i = 0; i < 5; i++) rows[i] |= ((x7[i] + lo) & ~(x7[i] + hi) & lowh[i]) >> sh;
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 5; i++) highh[i] = x[i] & 0x80808080u;
  if (pl.n_ranges <= 4) {
#pragma unroll
    for (uint32_t r = 0; r < 4; r++) {
      if (r < pl.n_ranges) {
        const uint32_t lo = pl.add_lo[r], hi = pl.add_hi[r], sh = pl.shift[r];
        if (r < pl.n_low) {
#pragma unroll
          for (int i = 0; i < 5; i++) rows[i] |= ((x7[i] + lo) & ~(x7[i] + hi) & lowh[i]) >> sh;
        } else {
#pragma unroll
          for (int i = 0; i < 5; i++) rows[i] |= ((x7[i] + lo) & ~(x7[i] + hi) & highh[i]) >> sh;
        }
      }
    }
    return;
  }
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
  for (uint32_t r = 0; r < pl.n_low; r++) {
    const uint32_t lo = pl.add_lo[r], hi = pl.add_hi[r], sh = pl.shift[r];
#pragma unroll
    for (int i = 0; i < 5; i++) rows[i] |= ((x7[i] + lo) & ~(x7[i] + hi) & lowh[i]) >> sh;
  }
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
  for (uint32_t r = pl.n_low; r < pl.n_ranges; r++) {
    const uint32_t lo = pl.add_lo[r], hi = pl.add_hi[r], sh = pl.shift[r];
#pragma unroll
    for (int i = 0; i < 5; i++) rows[i] |= ((x7[i] + lo) & ~(x7[i] + hi) & highh[i]) >> sh;
  }
}

// 0x80 in every byte of x that is not zero
RJ_HD uint32_t rj_swar_nz(uint32_t x) { return (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u; }

// bits 7, 15, 23, 31 -> bits 0..3
RJ_HD uint32_t rj_swar_movemask(uint32_t f) { return ((f >> 7) | (f >> 14) | (f >> 21) | (f >> 28)) & 0xFu; }

RJ_HD uint32_t rj_alignbyte(uint32_t hi, uint32_t lo, int bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbyte(hi, lo, static_cast<uint32_t>(bytes));
#else
  return static_cast<uint32_t>(((static_cast<uint64_t>(hi) << 32) | lo) >> (8 * bytes));
#endif
}

// F layout <-> one bit per start in text order
RJ_HD uint32_t rj_swar_f_from_starts(uint32_t m16) {
  uint32_t f = 0;
#pragma unroll
  for (int g = 0; g < 4; g++) f |= ((((m16 >> (4 * g)) & 0xFu) * 0x00204081u) & 0x01010101u) << g;
  return f;
}
RJ_HD uint32_t rj_swar_f_to_starts(uint32_t f) {
  uint32_t m = 0;
#pragma unroll
  for (int g = 0; g < 4; g++) {
    const uint32_t b = (f >> g) & 0x01010101u;
    m |= ((b | (b >> 7) | (b >> 14) | (b >> 21)) & 0xFu) << (4 * g);
  }
  return m;
}
// the flags of the starts one further on: start j -> j + 1 (the lane's last start drops out), bit 0 = carry_in
// (start 4g + k -> 4g + k + 1: the next byte of the same group, or byte 0 of the next group)
RJ_HD uint32_t rj_swar_f_next(uint32_t f, uint32_t carry_in) { return (f << 8) | ((f >> 23) & 0xEu) | carry_in; }
RJ_HD uint32_t rj_swar_f_last(uint32_t f) { return (f >> 27) & 1u; }  // the flag of start 15

RJ_HD uint32_t rj_rotr(uint32_t x, uint32_t r) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(x, x, r);
#else
  r &= 31u;
  return r ? (x >> r) | (x << (32u - r)) : x;
#endif
}

// The first D steps of the 16 starts whose bytes are rows[0..3] (rows[4]: the four bytes after them).
// All masks in F layout:
//   *walk     the start is still alive after D + 1 bytes, or passed through a position whose follow set
//             is not a shift / loop -- the walkers take it
//   *matched  some prefix of at most D bytes matched; H[g] byte k bit t - 1: start 4g + k matched t bytes.
//             A start that is not in *walk is decided: its longest match is the highest t.
//   *in_first (FIRST only) the start's byte can begin a match (DevProgram::loop_first: one first position)
// TWO_LAST: there are two accepting positions (else last_shift[1] == last_shift[0]); GEN: some position has a
// general follow row (pl.gen != 0) -- wave-uniform facts of the pattern, template parameters so that the steps of
// the common pattern (one accepting position, shifts and loops only) carry neither.
// The flag "accepting after t bytes" goes from bit 8k + last to bit 8k + t - 1 of h with ONE rotation (by
// last - t + 1 mod 32: the bit stays in its byte whichever way it moves) and one and-or.
template <int D, bool FIRST, bool TWO_LAST, bool GEN>
RJ_HD void rj_swar_presteps_as(const SwarPlan& pl, const uint32_t (&rows)[5], uint32_t* walk, uint32_t* matched, uint32_t (&H)[4],
                               uint32_t* in_first) {
  uint32_t w = 0, m = 0, inf = 0;
  uint32_t rot0[D], rot1[D];
#pragma unroll
  for (int t = 1; t <= D; t++) {
    rot0[t - 1] = (pl.last_shift[0] + 33u - static_cast<uint32_t>(t)) & 31u;
    rot1[t - 1] = (pl.last_shift[1] + 33u - static_cast<uint32_t>(t)) & 31u;
  }
#pragma unroll
  for (int g = 0; g < 4; g++) {
#if defined(__HIP_DEVICE_COMPILE__)
    // one group after the other: interleaving the four independent chains buys nothing (the kernel is
    // issue-bound, other waves fill the gaps) and costs the registers that decide the occupancy
    __builtin_amdgcn_sched_barrier(0);
#endif
    uint32_t S = pl.first & rows[g];
    if (FIRST) inf |= ((S >> pl.first_shift) & 0x01010101u) << g;
    uint32_t gen = 0, h = 0;
#pragma unroll
    for (int t = 1; t <= D; t++) {
      h |= rj_rotr(S, rot0[t - 1]) & (0x01010101u << (t - 1));
      if (TWO_LAST) h |= rj_rotr(S, rot1[t - 1]) & (0x01010101u << (t - 1));
      if (GEN) gen |= S & pl.gen;
      const uint32_t rt = t < 4 ? rj_alignbyte(rows[g + 1], rows[g], t) : rows[g + 1];
      S = (((S & pl.step1) << 1) | (S & pl.loopm)) & rt;
    }
    w |= (rj_swar_nz(GEN ? (S | gen) : S) >> 7) << g;
    m |= (((h + 0x7f7f7f7fu) & 0x80808080u) >> 7) << g;  // (h < 0x10 in every byte)
    H[g] = h;
  }
  *walk = w;
  *matched = m;
  *in_first = inf;
}

template <int D, bool FIRST>
RJ_HD void rj_swar_presteps(const SwarPlan& pl, const uint32_t (&rows)[5], uint32_t* walk, uint32_t* matched, uint32_t (&H)[4],
                            uint32_t* in_first) {
  if (pl.last_shift[0] == pl.last_shift[1] && pl.gen == 0) rj_swar_presteps_as<D, FIRST, false, false>(pl, rows, walk, matched, H, in_first);
  else rj_swar_presteps_as<D, FIRST, true, true>(pl, rows, walk, matched, H, in_first);
}

}  // namespace rejit_amd
#endif
