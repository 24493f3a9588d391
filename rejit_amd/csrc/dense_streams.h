// rejit_amd/csrc/dense_streams.h -- dense mode as BIT STREAMS: the class tests and the first automaton steps of 32
// start positions per register operation (round 4; kernel: dense_streams.hip).
//
// scan_dense_walk (kernels.hip) with the lane-packed pre-steps of dense_swar.h keeps one state BYTE per start, four
// starts per register: 23.5 VALU lane-operations per text byte on `[a-f]+[0-9]`, issue-bound at 0.21 of the HBM rate.
// Here everything is "vertical", as in the plane scan of the nine regexdna patterns:
//   * a lane owns 32 text bytes; a byte range [lo, hi] becomes ONE 32-bit stream T_r (bit j = byte j lies in the
//     range): three byte-parallel operations per dword and one v_dot4_u32_u8 per dword that packs the four 0x80 flags
//     (weights 1, 2, 4, 8 / 16 .. 128, accumulating) -- ~36 operations per range and 32 bytes;
//   * position k of the automaton (<= 8 positions: a chain with self loops, the shapes SwarPlan takes) reads the
//     stream S_k = OR of its ranges' T_r;
//   * the automaton state is POSITION-major: A_k holds, for each of the lane's 32 START positions, "the thread that
//     began there sits at position k".  Step t consumes the byte t places after every start at once:
//         A'_k = ((A_{k-1} & step_{k-1}) | (A_k & loop_k)) & W_k(t),   W_k(t) = the stream S_k shifted by t
//     -- 4 operations per position and step for 32 starts, where the start-major form needs ~10 per FOUR starts;
//   * the shift never needs the NEXT lane's bytes: a lane's 32 starts are the positions kStreamShift (16) bytes
//     BEFORE its own bytes (bit j = start at - 16 + j), so W_k(t) = alignbit(S_k, S_k of the lane below, 16 + t) for
//     all t <= 16; the lane below's word comes by DPP, lane 0's from the previous iteration (a scalar);
//   * the longest accepted length of every start is kept bit-sliced (4 registers: length - 1), the set of starts
//     whose thread is still alive after `depth` bytes goes to a scalar walk (rare: the kernel counts them).
// The functions below are host/device: tests/support/carry_exec.cc runs them lane by lane on the CPU against the
// scalar automaton and, as a whole MatchAll, against the oracle.
//
// Which patterns (make_stream_plan, table_layout.h): <= 8 positions in one state word, no assertions, not nullable,
// every follow set a shift and/or a self loop, <= 8 byte ranges in all, not at risk of the reference's ring artefact
// -- and NO TWO CANDIDATES CAN OVERLAP (decided on the automaton: no thread that has consumed a byte can consume a
// byte at which a candidate may begin).  Then the candidates (every start with a match; for `X+ rest` the first byte
// of every run of X, DevProgram::loop_first) ARE the result of the reference's left-most-longest selection
// (src/codegen.cc:36-86, src/x64/codegen-x64.cc:401-466), in order, and the kernel writes them once, at their
// final place.  Round 5: candidates that CAN overlap are taken too when no match is longer than 16 bytes
// (`[0-9][0-9][0-9]`, StreamPlan::select): the selection is then made in the kernel, rj_stream_select below.
#ifndef REJIT_AMD_DENSE_STREAMS_H_
#define REJIT_AMD_DENSE_STREAMS_H_

#include <stdint.h>

#include "device_program.h"

namespace rejit_amd {

constexpr int kStreamMaxRanges = 8;
constexpr int kStreamMaxPos = 8;
constexpr uint32_t kStreamShift = 16;  // a lane's starts lie this many bytes before its own 32 bytes; also the most steps

struct StreamPlan {
  uint32_t n_pos;       // 0: the pattern does not qualify
  uint32_t n_ranges;
  uint32_t depth;       // bytes consumed in registers: min(longest match, kStreamShift)
  uint32_t loop_first;  // candidates are the first bytes of the runs of position `first_pos`'s class
  uint32_t first_pos;
  // range r = [lo, hi] within one half of the byte values (dense_swar.h): b7 + add_lo carries into bit 7 iff b7 >= lo,
  // b7 + add_hi iff b7 > hi; replicated x4
  uint32_t add_lo[kStreamMaxRanges], add_hi[kStreamMaxRanges];
  uint32_t high_half;                    // bit r: the range lies in 0x80..0xff
  uint32_t range_pos[kStreamMaxRanges];  // bit k: position k consumes the bytes of range r
  uint32_t first, last, step, loop;      // bit k: may begin a match / accepts / passes to k + 1 / follows itself
  uint32_t select;      // candidates MAY overlap (`[0-9][0-9][0-9]`) and no match is longer than kStreamShift bytes: the kernel
                        // applies the reference's left-most-longest selection itself (rj_stream_select)
  uint32_t run_shape;   // 1: `X+`, 2: `X+ Y` with X and Y disjoint -- the run form of the steps (rj_stream_runs, round 6); 0: neither
};

RJ_HD uint32_t rj_udot4(uint32_t a, uint32_t b, uint32_t acc) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_udot4(a, b, acc, false);
#else
  for (int i = 0; i < 4; i++) acc += ((a >> (8 * i)) & 0xFFu) * ((b >> (8 * i)) & 0xFFu);
  return acc;
#endif
}

// the low 32 bits of (hi:lo) >> sh, sh in [0, 32]
RJ_HD uint32_t rj_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) {
#if defined(__HIP_DEVICE_COMPILE__)
  return sh >= 32u ? hi : __builtin_amdgcn_alignbit(hi, lo, sh);
#else
  return sh >= 32u ? hi : static_cast<uint32_t>(((static_cast<uint64_t>(hi) << 32) | lo) >> sh);
#endif
}

// x7 = the low seven bits of the lane's 32 bytes, half = 0x80 in every byte of the wanted half (low: ~x, high: x)
RJ_HD uint32_t rj_stream_range(const uint32_t (&x7)[8], const uint32_t (&half)[8], uint32_t add_lo, uint32_t add_hi) {
  uint32_t m[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const uint32_t f0 = (x7[2 * q] + add_lo) & ~(x7[2 * q] + add_hi) & half[2 * q];
    const uint32_t f1 = (x7[2 * q + 1] + add_lo) & ~(x7[2 * q + 1] + add_hi) & half[2 * q + 1];
    m[q] = rj_udot4(f0, 0x08040201u, rj_udot4(f1, 0x80402010u, 0u));  // (the flags weigh 128 each: the 8 mask bits << 7)
  }
  return (m[0] >> 7) | (m[1] << 1) | (m[2] << 9) | (m[3] << 17);
}

// the streams S_k of all positions over the lane's 32 bytes x[0..7]; `valid` = bit j: byte j lies inside the text.
// NR = the loop's compile-time bound: the plan's n_ranges <= NR (the constants of ranges beyond it are never touched, so
// they take no scalar registers in the kernel's chunk loop); HIGH = some range lies in 0x80..0xff (else the high-half
// flags are never formed -- the usual pattern is ASCII); rm[r][k] = 0 / ~0: position k reads range r.
template <int NP, int NR>
struct StreamRangeMasks {
  uint32_t m[NR][NP];
};
template <int NP, int NR>
RJ_HD StreamRangeMasks<NP, NR> rj_stream_range_masks(const StreamPlan& pl) {
  StreamRangeMasks<NP, NR> rm;
#pragma unroll
  for (int r = 0; r < NR; r++)
#pragma unroll
    for (int k = 0; k < NP; k++) rm.m[r][k] = static_cast<uint32_t>(r) < pl.n_ranges ? 0u - ((pl.range_pos[r] >> k) & 1u) : 0u;
  return rm;
}

template <int NP, int NR, bool HIGH>
RJ_HD void rj_stream_classes(const StreamPlan& pl, const StreamRangeMasks<NP, NR>& rm, const uint32_t (&x)[8], uint32_t valid, uint32_t (&S)[NP]) {
  uint32_t x7[8], lowh[8], highh[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    x7[i] = x[i] & 0x7f7f7f7fu;
    lowh[i] = ~x[i] & 0x80808080u;
    highh[i] = HIGH ? (x[i] & 0x80808080u) : 0u;
  }
#pragma unroll
  for (int k = 0; k < NP; k++) S[k] = 0;
#pragma unroll
  for (int r = 0; r < NR; r++) {
    // (ranges beyond the plan's n_ranges have zero masks: computed and dropped, at most NR / 2 - 1 of them)
    uint32_t T;
    if (HIGH && ((pl.high_half >> r) & 1u)) T = rj_stream_range(x7, highh, pl.add_lo[r], pl.add_hi[r]);
    else T = rj_stream_range(x7, lowh, pl.add_lo[r], pl.add_hi[r]);
#pragma unroll
    for (int k = 0; k < NP; k++) S[k] |= T & rm.m[r][k];
  }
#pragma unroll
  for (int k = 0; k < NP; k++) S[k] &= valid;
}

// The scalar walk of ONE start with nothing but the plan (no tables): the chain automaton's state in a register, the
// class of a byte from the plan's ranges.  For the rare start that outlives the register steps.  Returns the longest
// match's length (0: none); *overrun when the walk reached max_walk bytes.
RJ_HD uint32_t rj_stream_cls(const StreamPlan& pl, uint32_t b) {
  uint32_t c = 0;
  const uint32_t b7 = b & 0x7fu, high = b >> 7;
  for (uint32_t r = 0; r < pl.n_ranges; r++) {
    const uint32_t lo = 0x80u - (pl.add_lo[r] & 0xFFu), hi = 0x7fu - (pl.add_hi[r] & 0xFFu);
    if (((pl.high_half >> r) & 1u) == high && b7 >= lo && b7 <= hi) c |= pl.range_pos[r];
  }
  return c;
}
RJ_HD uint32_t rj_stream_walk(const StreamPlan& pl, const uint8_t* t, uint64_t n, uint64_t s, uint32_t max_walk, bool* overrun) {
  uint32_t S = pl.first & rj_stream_cls(pl, t[s]);
  uint32_t longest = 0, d = 1;
  while (S != 0) {
    if (S & pl.last) longest = d;
    if (s + d >= n) break;
    if (d >= max_walk) {
      *overrun = true;
      break;
    }
    const uint32_t T = ((S & pl.step) << 1) | (S & pl.loop);
    S = T & rj_stream_cls(pl, t[s + d]);
    d++;
  }
  return longest;
}

// 0 / ~0 per position: the plan's bit masks as operands (scalars of the kernel's chunk loop)
template <int NP>
struct StreamMasks {
  uint32_t first[NP], last[NP], step[NP], loop[NP];
};
template <int NP>
RJ_HD StreamMasks<NP> rj_stream_masks(const StreamPlan& pl) {
  StreamMasks<NP> m;
#pragma unroll
  for (int k = 0; k < NP; k++) {
    m.first[k] = 0u - ((pl.first >> k) & 1u);
    m.last[k] = 0u - ((pl.last >> k) & 1u);
    m.step[k] = 0u - ((pl.step >> k) & 1u);
    m.loop[k] = 0u - ((pl.loop >> k) & 1u);
  }
  return m;
}

// The first pl.depth steps of the lane's 32 starts (bit j = the start kStreamShift - j bytes BEFORE the lane's bytes).
//   S / Sb      the position streams of the lane's own 32 bytes / of the 32 bytes before them
//   start_mask  starts that count at all (own range, inside the text); the run-start rule is applied here
//   any(x)      is x != 0 in any lane of the wave (device: a ballot; host: the lane itself) -- the steps stop when
//               every thread of the wave has died
// Out: *matched  starts with a match of at most pl.depth bytes; len[b] = bit b of (its longest length - 1)
//      *alive    starts whose thread is still alive after kStreamShift (16) bytes: NOT decided (matched / len then hold the
//                longest match so far); empty when the longest possible match is pl.depth bytes
//      *cand     the candidate starts themselves (first byte fits; run starts only under loop_first)
template <int NP, typename Any>
RJ_HD void rj_stream_steps(const StreamPlan& pl, const StreamMasks<NP>& mk, const uint32_t (&S)[NP], const uint32_t (&Sb)[NP],
                           uint32_t start_mask, Any any, uint32_t* matched, uint32_t* alive, uint32_t (&len)[4], uint32_t* cand) {
  uint32_t A[NP], c0 = 0;
#pragma unroll
  for (int k = 0; k < NP; k++) {
    A[k] = rj_alignbit(S[k], Sb[k], 32u - kStreamShift) & mk.first[k];
    c0 |= A[k];
  }
  if (pl.loop_first) {
    // `X+ rest`: a start whose previous byte is in X too is never selected (DevProgram::loop_first; X = the class of the
    // one first position, picked by its mask -- no comparison of position numbers, each of which the compiler keeps as a
    // 64-bit lane mask in scalar registers)
    uint32_t prev = 0;
#pragma unroll
    for (int k = 0; k < NP; k++) prev |= rj_alignbit(S[k], Sb[k], 32u - kStreamShift - 1u) & mk.first[k];
    c0 &= ~prev;
  }
  c0 &= start_mask;
#pragma unroll
  for (int k = 0; k < NP; k++) A[k] &= c0;
  uint32_t m = 0, al = 0;
  len[0] = len[1] = len[2] = len[3] = 0;
#pragma unroll
  for (int t = 0; t < static_cast<int>(kStreamShift); t++) {
#if defined(__HIP_DEVICE_COMPILE__)
    // one step after the other: the shifted streams of all 16 steps are loop-invariant, and a scheduler that computes
    // them ahead of time pays for it in registers (occupancy), which buys nothing in an issue-bound kernel
    __builtin_amdgcn_sched_barrier(0);
#endif
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < NP; k++) acc |= A[k] & mk.last[k];
    m |= acc;
#pragma unroll
    for (int b = 0; b < 4; b++) len[b] = ((t >> b) & 1) ? (len[b] | acc) : (len[b] & ~acc);
    uint32_t N[NP], live = 0;
#pragma unroll
    for (int k = 0; k < NP; k++) {
      const uint32_t w = rj_alignbit(S[k], Sb[k], 32u - kStreamShift + static_cast<uint32_t>(t) + 1u);
      N[k] = (((k > 0 ? A[k > 0 ? k - 1 : 0] & mk.step[k > 0 ? k - 1 : 0] : 0u)) | (A[k] & mk.loop[k])) & w;
      live |= N[k];
    }
    // (no test for the plan's depth: a bounded pattern's threads are dead after its longest match, which is <= 16 bytes
    // whenever the plan's depth is below 16 -- sixteen uniform comparisons cost 32 scalar registers as lane masks)
    al = live;
    if (t + 1 == static_cast<int>(kStreamShift) || !any(live)) break;
#pragma unroll
    for (int k = 0; k < NP; k++) A[k] = N[k];
  }
  *matched = m;
  *alive = al;
  *cand = c0;
}

// `X+` and `X+ Y` (StreamPlan::run_shape; X, Y disjoint): the steps as ARITHMETIC on the lane's 64-byte window instead of one
// automaton step per consumed byte (round 6).  W = the X stream of the 32 bytes before the lane's own (low word) and of its own
// (high word); a lane's starts are the window bits 16 .. 47 (kStreamShift), so every start has at least 16 bytes of its run in
// the window.  The candidates are the run starts (bit set, the bit below clear); adding them to W carries each through its
// run and leaves a bit at the first byte behind it -- the run's END --, all runs at once, in one 64-bit addition: what sixteen
// steps of four operations per position did.  A match is a run whose end holds Y (`X+`: every run); its length is the
// distance of the two bits.  A run that reaches the window's top is not decided here (*alive: the scalar walk, as before).
//   *starts  the candidates (window bits 16 .. 47, under start_mask)      *ends  the END bits of the runs that match
RJ_HD void rj_stream_runs(uint32_t run_shape, uint32_t S0, uint32_t Sb0, uint32_t S1, uint32_t Sb1, uint32_t start_mask, uint64_t* starts, uint64_t* ends,
                          uint32_t* alive) {
  const uint64_t W = (static_cast<uint64_t>(S0) << 32) | Sb0;
  const uint64_t c = W & ~(W << 1) & (static_cast<uint64_t>(start_mask) << kStreamShift);
  const uint64_t T = W + c;               // (a carry out of bit 63 is the undecided run's)
  uint64_t end = T & ~W;
  if (run_shape == 2) end &= (static_cast<uint64_t>(S1) << 32) | Sb1;
  // the run that reaches bit 63: its start is the bit above the window's highest zero
  const uint64_t zeros = ~W;
  const uint32_t lead = zeros == 0 ? 64u : static_cast<uint32_t>(__builtin_clzll(zeros));
  const uint64_t top = (lead > 0 && lead < 64u) ? (1ull << (64u - lead)) : 0ull;   // (lead == 64: the run began before the window)
  *alive = static_cast<uint32_t>((c & top) >> kStreamShift);
  *starts = c;
  *ends = end;
}

// the next match of the lane, in text order: takes the lowest bit of *ends; j = its start (0 .. 31), the match's length
RJ_HD uint32_t rj_stream_run_next(uint32_t run_shape, uint64_t starts, uint64_t* ends, int* j) {
  const int e = __builtin_ctzll(*ends);
  *ends &= *ends - 1;
  const int s = 63 - __builtin_clzll(starts & ((1ull << e) - 1ull));   // (the run's own start: the highest candidate below its end)
  *j = s - static_cast<int>(kStreamShift);
  return static_cast<uint32_t>(e - s) + (run_shape == 2 ? 1u : 0u);
}

// longest length of start j from the bit-sliced registers
RJ_HD uint32_t rj_stream_len(const uint32_t (&len)[4], int j) {
  return 1u + (((len[0] >> j) & 1u) | (((len[1] >> j) & 1u) << 1) | (((len[2] >> j) & 1u) << 2) | (((len[3] >> j) & 1u) << 3));
}

// The reference's selection (left-most start, longest match from it, the next match begins at or behind its end:
// src/codegen.cc:36-86, src/x64/codegen-x64.cc:401-466, 494-500) among the 32 starts of one lane, for plans with
// `select`: T = the starts with a match, len = their longest lengths (bit-sliced, <= kStreamShift), d = how many of the
// lane's first starts lie inside a match selected in a lane below (0 .. kStreamShift - 1).  Returns the selected starts;
// *out = the same count for the NEXT lane.  A lane without any match resets the chain (a match is at most 16 bytes long,
// a lane 32 starts): the kernel resolves d lane by lane by speculation (d = 0, corrected from the lane below until
// nothing changes) and finds a tile's entry state in the 2 KiB before the tile (dense_streams.hip).
RJ_HD uint32_t rj_stream_select(uint32_t T, const uint32_t (&len)[4], uint32_t d, uint32_t* out) {
  uint32_t sel = 0, pos = d;
  uint32_t m = pos < 32u ? T & (~0u << pos) : 0u;
  while (m != 0) {
    const int j = __builtin_ctz(m);
    sel |= 1u << j;
    pos = static_cast<uint32_t>(j) + rj_stream_len(len, j);
    m = pos < 32u ? T & (~0u << pos) : 0u;
  }
  *out = pos > 32u ? pos - 32u : 0u;
  return sel;
}

}  // namespace rejit_amd
#endif
