// rejit_amd/csrc/plane_count.hip -- MatchAllCount for SEVERAL patterns in ONE kernel: the nine counts of regexdna
// (reference sample/regexdna.cc:51-67: one MatchAllCount per pattern, src/rejit.cc:203-208, each a run of
// FastForwardGen's multi-literal scan, src/x64/codegen-x64.cc:1102-1252, plus the NFA loop on its hits).
//
// Round 5.  The span pipeline answers the same question with three kernels -- plane_scan (the text, once),
// classify_shared_multi (an automaton walk per candidate), offsets_gather_check_multi (the (begin, end) lists laid
// out) -- and 44 % of its step was the two tails.  A caller that asks for COUNTS needs neither the lists nor, for
// the pattern sets this kernel takes (exact_count.h: fixed-length 8-byte patterns whose language lies within one byte
// of the scan's base windows), the automaton: a candidate's eight bytes are looked up in a table.  So here a wave
//   * streams its span of the text through the bit-plane test of plane_scan.hip in blocks of 2 KiB, of which a lane owns
//     TWO 16-byte pieces -- piece A at 16 * lane, piece B at 1024 + 16 * lane; bit 2k of a plane = byte k of A, bit
//     2k + 1 = byte k of B --: each of the wave's two loads covers one contiguous KiB, every cache line is asked for by
//     ONE instruction, and that is what lets the loads be non-temporal (stream_load.h: 6.9 instead of 6.0 TB/s for a
//     read-only stream; rounds 5-6 had a lane own 32 contiguous bytes, two loads per line, and the nt policy made that
//     form 19 % SLOWER).  The 8 positions that follow a piece are the first 8 of the same piece one lane up -- two DPP
//     moves, no load --, lane 63's piece A is followed by lane 0's piece B, and the codes of the NEXT block are computed
//     one step ahead, so that lane 63's piece B finds its neighbour (lane 0's piece A of the next block) in a register;
//   * keeps the candidates (one per 1.3 KiB on DNA) in an LDS ring of its own, as 32-bit offsets;
//   * classifies them 64 at a time -- whenever the ring holds that many, and at the end of the span --: two loads of
//     the candidate's text, exact_classify, a ballot per pattern;
//   * adds its counts to the workgroup's; the workgroup stores a row of its own (no atomics: see the end of the kernel),
//     and plane_count_finish -- one workgroup, queued behind the scan on a stream of the object's own, so that the next
//     scan kernel starts at once -- adds the rows up and hands counts, flags and the first / last match of every pattern
//     (what the carry exchange between shards needs, rj_multi_bounds) to pinned host memory.
// Nothing else is written: no candidate list in HBM, no (begin, end) arrays, no second or third launch.
//
// Exactness.  The table answers "does pattern p match these 8 bytes" exactly (exact_count.h).  The reference's count
// is that of the left-most-longest, non-overlapping selection (src/x64/codegen-x64.cc:401-466: a match that begins inside
// the match selected before it is dropped, :448-460); with all matches 8 bytes long it differs from the number of matching
// positions only when two matches of ONE pattern begin fewer than 8 bytes apart.  Candidates are classified in text
// order, so that needs two neighbours in the ring closer than 8 bytes with a pattern in common.  An isolated PAIR of
// such neighbours is resolved on the spot: the first stands (nothing overlaps it), the second is dropped for the patterns
// they share -- `agggtaaagggtaaa` counts once for `agggtaaa|tttaccct`, as in the reference.  Three or more candidates in a
// row, each closer than 8 bytes to the one before, would need the selection replayed along the chain: the run is flagged
// (kPcConflict) when a pattern occurs twice in such a chain (or the chain is longer than three), as is a pair that lies
// across two waves' spans (the wave looks at the 7 positions before its span itself), and the host repeats THAT run with
// the span pipeline.  A block that holds more candidates than the ring can take (kPcVoid) is handled the same way.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdlib>

#include "exact_count.h"
#include "kernels.h"
#include "short_walk.h"
#include "stream_load.h"

namespace rejit_amd {

namespace {

constexpr int kWave = 64;
constexpr uint64_t kBlock = 2048;     // bytes a wave takes per iteration: 64 lanes x 2 pieces of 16 B
constexpr uint32_t kPieceB = 1024;    // a lane's piece B begins this far behind its piece A
// Capture (ExactShape): a wave keeps the raw bytes of the last two blocks in LDS -- a block's 2 KiB in text order and, behind
// them, the first 8 bytes of the block that follows -- and a candidate takes its 8 bytes from there when it enters the ring.
constexpr uint32_t kStashWords = (2048 + 16) / 4;
#ifndef RJ_PC_RING
#define RJ_PC_RING 256
#endif
constexpr uint32_t kRing = RJ_PC_RING;       // candidate slots per wave; consumed 64 at a time, looked at every second block

__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x) & (kWave - 1); }

__device__ __forceinline__ uint32_t lanes_below(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}

// lane i <- lane i + 1 (wave_shl:1); lane 63 keeps `last`
__device__ __forceinline__ uint32_t from_lane_above(uint32_t v, uint32_t last) {
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(last), static_cast<int>(v), 0x130, 0xF, 0xF, false));
}
// lane i <- lane i - 1 (wave_shr:1); lane 0 keeps `first`
__device__ __forceinline__ uint32_t from_lane_below(uint32_t v, uint32_t first) {
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(first), static_cast<int>(v), 0x138, 0xF, 0xF, false));
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t x) {
  return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), CTRL, ROW_MASK, 0xF, true));
}
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x) {
  x += dpp_or_zero<0x111, 0xF>(x);
  x += dpp_or_zero<0x112, 0xF>(x);
  x += dpp_or_zero<0x114, 0xF>(x);
  x += dpp_or_zero<0x118, 0xF>(x);
  x += dpp_or_zero<0x142, 0xA>(x);
  x += dpp_or_zero<0x143, 0xC>(x);
  return x;
}

struct Consts {
  uint32_t cmask;   // 0x03030303 << code_shift
  uint32_t shift;
};

// the 2-bit codes of a dword's four bytes as one byte (times 2^shift)
__device__ __forceinline__ uint32_t codes4(uint32_t d, const Consts& k) { return __builtin_amdgcn_udot4(d & k.cmask, 0x40100401u, 0u, false); }

// the codes of 16 bytes: bits 2k, 2k + 1 = byte k
__device__ __forceinline__ uint32_t codes16(const uint4& v, const Consts& k) {
  const uint32_t s = k.shift;
  return (codes4(v.x, k) >> s) | (codes4(v.y, k) << (8 - s)) | (codes4(v.z, k) << (16 - s)) | (codes4(v.w, k) << (24 - s));
}

struct Raw {
  uint4 a, b;   // the lane's two pieces
};

// (a uniform base and a 32-bit lane offset: the load takes its address from a scalar pair + one register, no 64-bit
// pointer per lane is kept alive across the loop)
__device__ __forceinline__ void load_block(const uint8_t* block, uint32_t lane_off, Raw& r) {
  r.a = stream_load16(block + lane_off);
  r.b = stream_load16(block + lane_off + kPieceB);
}

__device__ __forceinline__ uint32_t guarded_dword(const uint8_t* text, uint64_t n, uint64_t at) {
  uint32_t v = 0;
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (at + k < n) v |= static_cast<uint32_t>(text[at + k]) << (8 * k);
  return v;
}

// Window positions of the lane's two pieces that lie within one code of a base: bit 2k = byte k of piece A, bit 2k + 1 =
// byte k of piece B.  ta / tb: the codes of the pieces; ha / hb: the codes of the 8 bytes behind each of them (bits 0..15).
template <int NB>
__device__ __forceinline__ uint32_t plane_test(uint32_t ta, uint32_t tb, uint32_t ha, uint32_t hb, const PlaneCountParams& a) {
  constexpr uint32_t kEven = 0x55555555u;
  const uint32_t L = (ta & kEven) | ((tb << 1) & ~kEven);
  const uint32_t H = ((ta >> 1) & kEven) | (tb & ~kEven);
  const uint32_t Ln = (ha & kEven) | ((hb << 1) & ~kEven);
  const uint32_t Hn = ((ha >> 1) & kEven) | (hb & ~kEven);
  // The four CODE planes (bit = the position's symbol has code c) of the 32 positions and of the 32 that follow.  "Window byte
  // i of base b fits at position p" is then ONE shifted plane -- alignbit(next[c], here[c], 2 i), c = the code of the base's
  // byte i -- where two bit planes compared against the base's masks took a shift each (shared by the bases), an XOR and a
  // three-input operation per base: 110 instead of 137 instructions per lane and block, in a kernel that runs at two thirds of
  // the VALU issue rate this device measures (profiles/r05_valu_rate.txt).  WHICH plane is a launch constant, and reading a
  // register picked by a run-time value is what the VGPR index mode is for (s_set_gpr_idx_*: the index in M0 is added to the
  // source registers of the VALU instructions that follow): the eight planes sit in FIXED registers v48..v51 (here) and
  // v52..v55 (next), `v_alignbit_b32 e, v52, v48, 2 i` under index c reads next[c], here[c].  Measured before this: a uniform
  // branch around four one-instruction arms per pick -- 0.101 ms per launch against 0.091 (taken branches cost the wave more
  // than the instructions saved); as selects the picks cost more than they save.
  // (M0: the index mode writes it.  It is a register the compiler reserves for itself -- an asm clobber of it is ignored, clang
  // says so -- and uses for LDS-DMA, s_movrel, GWS and messages only, none of which this file contains: tools/check_m0.py
  // looks at the generated code for any other reader of M0 in these kernels.)
  register uint32_t q0 asm("v48") = ~(L | H);
  register uint32_t q1 asm("v49") = L & ~H;
  register uint32_t q2 asm("v50") = ~L & H;
  register uint32_t q3 asm("v51") = L & H;
  register uint32_t n0 asm("v52") = ~(Ln | Hn);
  register uint32_t n1 asm("v53") = Ln & ~Hn;
  register uint32_t n2 asm("v54") = ~Ln & Hn;
  register uint32_t n3 asm("v55") = Ln & Hn;
  uint32_t codes = ~a.mask_bits;   // (mask bit set = that bit of the code is 0)
  asm volatile("" : "+s"(codes));
  auto code = [&](int b, int i) { return (codes >> (16 * b + 2 * i)) & 3u; };   // (wave-uniform: a scalar bit-field extract)
  uint32_t E[8][NB], Z[NB], O[NB];
#pragma unroll
  for (int half = 0; half < 2; half++) {
    // (the picks of four window bytes in one statement: nothing but the indexed reads runs under the index mode)
    if (NB == 1) {
      if (half == 0) {
      asm volatile(
          "s_set_gpr_idx_on %[c00], 0x3\n"
          "v_mov_b32 %[e00], v48\n"
          "s_set_gpr_idx_idx %[c10]\n"
          "v_alignbit_b32 %[e10], v52, v48, 2\n"
          "s_set_gpr_idx_idx %[c20]\n"
          "v_alignbit_b32 %[e20], v52, v48, 4\n"
          "s_set_gpr_idx_idx %[c30]\n"
          "v_alignbit_b32 %[e30], v52, v48, 6\n"
          "s_set_gpr_idx_off\n"
          : [e00] "=&v"(E[0][0]), [e10] "=&v"(E[1][0]), [e20] "=&v"(E[2][0]), [e30] "=&v"(E[3][0])
          : [c00] "s"(code(0, 0)), [c10] "s"(code(0, 1)), [c20] "s"(code(0, 2)), [c30] "s"(code(0, 3)), "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(n0), "v"(n1), "v"(n2), "v"(n3));
      } else {
      asm volatile(
          "s_set_gpr_idx_on %[c40], 0x3\n"
          "v_alignbit_b32 %[e40], v52, v48, 8\n"
          "s_set_gpr_idx_idx %[c50]\n"
          "v_alignbit_b32 %[e50], v52, v48, 10\n"
          "s_set_gpr_idx_idx %[c60]\n"
          "v_alignbit_b32 %[e60], v52, v48, 12\n"
          "s_set_gpr_idx_idx %[c70]\n"
          "v_alignbit_b32 %[e70], v52, v48, 14\n"
          "s_set_gpr_idx_off\n"
          : [e40] "=&v"(E[4][0]), [e50] "=&v"(E[5][0]), [e60] "=&v"(E[6][0]), [e70] "=&v"(E[7][0])
          : [c40] "s"(code(0, 4)), [c50] "s"(code(0, 5)), [c60] "s"(code(0, 6)), [c70] "s"(code(0, 7)), "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(n0), "v"(n1), "v"(n2), "v"(n3));
      }
    } else {
      if (half == 0) {
      asm volatile(
          "s_set_gpr_idx_on %[c00], 0x3\n"
          "v_mov_b32 %[e00], v48\n"
          "s_set_gpr_idx_idx %[c01]\n"
          "v_mov_b32 %[e01], v48\n"
          "s_set_gpr_idx_idx %[c10]\n"
          "v_alignbit_b32 %[e10], v52, v48, 2\n"
          "s_set_gpr_idx_idx %[c11]\n"
          "v_alignbit_b32 %[e11], v52, v48, 2\n"
          "s_set_gpr_idx_idx %[c20]\n"
          "v_alignbit_b32 %[e20], v52, v48, 4\n"
          "s_set_gpr_idx_idx %[c21]\n"
          "v_alignbit_b32 %[e21], v52, v48, 4\n"
          "s_set_gpr_idx_idx %[c30]\n"
          "v_alignbit_b32 %[e30], v52, v48, 6\n"
          "s_set_gpr_idx_idx %[c31]\n"
          "v_alignbit_b32 %[e31], v52, v48, 6\n"
          "s_set_gpr_idx_off\n"
          : [e00] "=&v"(E[0][0]), [e01] "=&v"(E[0][1]), [e10] "=&v"(E[1][0]), [e11] "=&v"(E[1][1]), [e20] "=&v"(E[2][0]), [e21] "=&v"(E[2][1]), [e30] "=&v"(E[3][0]), [e31] "=&v"(E[3][1])
          : [c00] "s"(code(0, 0)), [c01] "s"(code(1, 0)), [c10] "s"(code(0, 1)), [c11] "s"(code(1, 1)), [c20] "s"(code(0, 2)), [c21] "s"(code(1, 2)), [c30] "s"(code(0, 3)), [c31] "s"(code(1, 3)), "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(n0), "v"(n1), "v"(n2), "v"(n3));
      } else {
      asm volatile(
          "s_set_gpr_idx_on %[c40], 0x3\n"
          "v_alignbit_b32 %[e40], v52, v48, 8\n"
          "s_set_gpr_idx_idx %[c41]\n"
          "v_alignbit_b32 %[e41], v52, v48, 8\n"
          "s_set_gpr_idx_idx %[c50]\n"
          "v_alignbit_b32 %[e50], v52, v48, 10\n"
          "s_set_gpr_idx_idx %[c51]\n"
          "v_alignbit_b32 %[e51], v52, v48, 10\n"
          "s_set_gpr_idx_idx %[c60]\n"
          "v_alignbit_b32 %[e60], v52, v48, 12\n"
          "s_set_gpr_idx_idx %[c61]\n"
          "v_alignbit_b32 %[e61], v52, v48, 12\n"
          "s_set_gpr_idx_idx %[c70]\n"
          "v_alignbit_b32 %[e70], v52, v48, 14\n"
          "s_set_gpr_idx_idx %[c71]\n"
          "v_alignbit_b32 %[e71], v52, v48, 14\n"
          "s_set_gpr_idx_off\n"
          : [e40] "=&v"(E[4][0]), [e41] "=&v"(E[4][1]), [e50] "=&v"(E[5][0]), [e51] "=&v"(E[5][1]), [e60] "=&v"(E[6][0]), [e61] "=&v"(E[6][1]), [e70] "=&v"(E[7][0]), [e71] "=&v"(E[7][1])
          : [c40] "s"(code(0, 4)), [c41] "s"(code(1, 4)), [c50] "s"(code(0, 5)), [c51] "s"(code(1, 5)), [c60] "s"(code(0, 6)), [c61] "s"(code(1, 6)), [c70] "s"(code(0, 7)), [c71] "s"(code(1, 7)), "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(n0), "v"(n1), "v"(n2), "v"(n3));
      }
    }
#pragma unroll
    for (int i = 4 * half; i < 4 * half + 4; i++)
#pragma unroll
      for (int b = 0; b < NB; b++) {
        const uint32_t e = E[i][b];
        if (i == 0) {
          Z[b] = e;
        } else if (i == 1) {
          O[b] = Z[b] | e;
          Z[b] &= e;
        } else {
          O[b] = Z[b] | (O[b] & e);
          if (i < 7) Z[b] &= e;
        }
      }
  }
  return NB > 1 ? (O[0] | O[NB - 1]) : O[0];
}

// ---------------------------------------------------------------------------------------------------------------
// Round 6: the kernel is a template over the SHAPE of the pattern set.  A shape says how a block's 32 positions per lane
// are tested (the filter) and how a candidate is classified (the exact answer: which patterns match at it, how long):
//   ExactShape<NB>        8-byte patterns within one byte of <= 2 bases: bit-plane test through the VGPR index mode, table
//                         lookup per candidate (exact_count.h) -- regexdna's nine, the headline;
//   GeneralShape<W, K, T> any set the general one-pass plan takes (multi_pattern.hip: plan_plane_general -- <= 12 base
//                         windows of 4..8 compared bytes, exactly or within one code, windows at the patterns' own offsets,
//                         short bounded automata of any length <= 16): the bases one after the other against sixteen
//                         shifted planes, and per candidate every pattern's exact window test + its automaton from LDS
//                         (short_walk.h) -- alternations of literals of any length, k-mers of any k (the reference's
//                         FastForwardGen takes any alternation of literals: src/x64/codegen-x64.cc:1129-1252,
//                         src/codegen.cc:327-393).
// Everything else -- the streaming loop, the ring, the selection rule, the rows -- is shared.

struct Lens {
  uint64_t w[2];   // 5 bits per pattern (match length <= 16), twelve patterns per word
};
static_assert(kMaxFused <= 24, "two words of twelve lengths");
__device__ __forceinline__ uint32_t len_of(const Lens& l, uint32_t p) {   // (p wave-uniform)
  return static_cast<uint32_t>((p < 12 ? l.w[0] >> (5 * p) : l.w[1] >> (5 * (p - 12))) & 31u);
}

}  // namespace

// (the shapes are named types of rejit_amd: a profiler prints the kernels as plane_count<rejit_amd::ExactShape<2> > ...)
template <int NB>
struct ExactShape {
  using Args = PlaneCountParams;
  static constexpr uint32_t kFixedLen = 8;
  static constexpr bool kBlobInLds = false;
  static constexpr bool kList = false;
  static constexpr bool kCapture = true;   // a candidate's 8 bytes ride in the ring (taken from the wave's stash of the block)
  __device__ static __forceinline__ const PlaneCountParams& common(const Args& g) { return g; }
  __device__ static __forceinline__ uint32_t lmax(const Args&) { return 8u; }
  __device__ static __forceinline__ uint32_t offset_of(const Args&, const uint32_t*, uint32_t) { return 0u; }
  __device__ static __forceinline__ uint32_t test(uint32_t ta, uint32_t tb, uint32_t ha, uint32_t hb, const Args& g) { return plane_test<NB>(ta, tb, ha, hb, g); }
  // the patterns that match the 8 bytes at window position `pos` (the scan does not clip: windows before the range, or
  // with bytes beyond the end of the text, are dropped here)
  __device__ static __forceinline__ void classify(const Args& a, const uint32_t* table, uint64_t pos, bool have, uint32_t& mask, Lens&) {
    const bool ok = have && pos >= a.sb && pos < a.se && pos + 8 <= a.n;
    uint32_t lo = 0, hi = 0;
    if (ok) {
      __builtin_memcpy(&lo, a.text + pos, 4);
      __builtin_memcpy(&hi, a.text + pos + 4, 4);
    }
    mask = exact_classify<NB>(table, a.base_lo, a.base_hi, lo, hi);
    mask = ok ? mask : 0u;
  }
  // ... the same with the window's bytes at hand (lo = bytes 0..3, hi = bytes 4..7)
  __device__ static __forceinline__ void classify_bytes(const Args& a, const uint32_t* table, uint64_t pos, bool have, uint32_t lo, uint32_t hi, uint32_t& mask) {
    const bool ok = have && pos >= a.sb && pos < a.se && pos + 8 <= a.n;
    mask = exact_classify<NB>(table, a.base_lo, a.base_hi, lo, hi);
    mask = ok ? mask : 0u;
  }
};

// ListShape<NB>: ExactShape's test, the candidates not classified but written, in text order, to the wave's region of the
// span pipeline's shared candidate list (kernels.h: PlaneListParams) -- plane_scan<NB> in this kernel's layout.
template <int NB>
struct ListShape {
  using Args = PlaneListParams;
  static constexpr uint32_t kFixedLen = 8;
  static constexpr bool kBlobInLds = false;
  static constexpr bool kList = true;
  static constexpr bool kCapture = false;
  __device__ static __forceinline__ void classify_bytes(const Args&, const uint32_t*, uint64_t, bool, uint32_t, uint32_t, uint32_t& mask) { mask = 0; }
  __device__ static __forceinline__ const PlaneCountParams& common(const Args& g) { return g.c; }
  __device__ static __forceinline__ uint32_t lmax(const Args&) { return 8u; }
  __device__ static __forceinline__ uint32_t offset_of(const Args&, const uint32_t*, uint32_t) { return 0u; }
  __device__ static __forceinline__ uint32_t test(uint32_t ta, uint32_t tb, uint32_t ha, uint32_t hb, const Args& g) { return plane_test<NB>(ta, tb, ha, hb, g.c); }
  __device__ static __forceinline__ void classify(const Args&, const uint32_t*, uint64_t, bool, uint32_t& mask, Lens&) { mask = 0; }
};

namespace {

// The general test: bit 2k = byte k of the lane's piece A, bit 2k + 1 = byte k of its piece B (plane_test's layout).  Base after base in a
// loop that is NOT unrolled (<= 12 bases), and per base what plane_test does for its two: "window byte i of the base fits at
// position p" is ONE shifted code plane, picked by the VGPR index mode -- the four planes of the 32 positions sit in v48..v51,
// an all-ones plane in v52 (a compared byte beyond the set's n_cmp: always fits), the same for the 32 positions that follow
// in v56..v60, and `v_alignbit_b32 e, v56, v48, 2 i` under index idx[b][i] (0..3 the code, 4 don't care: a launch constant,
// eight dwords per base from the kernel arguments) reads next[idx], here[idx].  8 picks + 4 three-input ANDs per base; the
// first version of this function (two bit planes XORed with per-position masks, a scalar load + wait per mask pair) ran
// general_one_pass' four bases at 0.55 of peak and nine 12-mers at 0.32.
template <bool TOL>
__device__ __forceinline__ uint32_t general_test(uint32_t ta, uint32_t tb, uint32_t ha, uint32_t hb, const PlaneCountGParams& g) {
  constexpr uint32_t kEven = 0x55555555u;
  const uint32_t L = (ta & kEven) | ((tb << 1) & ~kEven);
  const uint32_t H = ((ta >> 1) & kEven) | (tb & ~kEven);
  const uint32_t Ln = (ha & kEven) | ((hb << 1) & ~kEven);
  const uint32_t Hn = ((ha >> 1) & kEven) | (hb & ~kEven);
  register uint32_t q0 asm("v48") = ~(L | H);
  register uint32_t q1 asm("v49") = L & ~H;
  register uint32_t q2 asm("v50") = ~L & H;
  register uint32_t q3 asm("v51") = L & H;
  register uint32_t q4 asm("v52") = ~0u;
  register uint32_t n0 asm("v56") = ~(Ln | Hn);
  register uint32_t n1 asm("v57") = Ln & ~Hn;
  register uint32_t n2 asm("v58") = ~Ln & Hn;
  register uint32_t n3 asm("v59") = Ln & Hn;
  register uint32_t n4 asm("v60") = ~0u;
  uint32_t c = 0;
#pragma clang loop unroll(disable)
  for (uint32_t b = 0; b < g.c.n_bases; b++) {
    const uint32_t* ix = g.idx[b];
    uint32_t e0, e1, e2, e3, e4, e5, e6, e7;
    asm volatile(
        "s_set_gpr_idx_on %[i0], 0x3\n"
        "v_mov_b32 %[e0], v48\n"
        "s_set_gpr_idx_idx %[i1]\n"
        "v_alignbit_b32 %[e1], v56, v48, 2\n"
        "s_set_gpr_idx_idx %[i2]\n"
        "v_alignbit_b32 %[e2], v56, v48, 4\n"
        "s_set_gpr_idx_idx %[i3]\n"
        "v_alignbit_b32 %[e3], v56, v48, 6\n"
        "s_set_gpr_idx_idx %[i4]\n"
        "v_alignbit_b32 %[e4], v56, v48, 8\n"
        "s_set_gpr_idx_idx %[i5]\n"
        "v_alignbit_b32 %[e5], v56, v48, 10\n"
        "s_set_gpr_idx_idx %[i6]\n"
        "v_alignbit_b32 %[e6], v56, v48, 12\n"
        "s_set_gpr_idx_idx %[i7]\n"
        "v_alignbit_b32 %[e7], v56, v48, 14\n"
        "s_set_gpr_idx_off\n"
        : [e0] "=&v"(e0), [e1] "=&v"(e1), [e2] "=&v"(e2), [e3] "=&v"(e3), [e4] "=&v"(e4), [e5] "=&v"(e5), [e6] "=&v"(e6), [e7] "=&v"(e7)
        : [i0] "s"(ix[0]), [i1] "s"(ix[1]), [i2] "s"(ix[2]), [i3] "s"(ix[3]), [i4] "s"(ix[4]), [i5] "s"(ix[5]), [i6] "s"(ix[6]), [i7] "s"(ix[7]), "v"(q0), "v"(q1), "v"(q2),
          "v"(q3), "v"(q4), "v"(n0), "v"(n1), "v"(n2), "v"(n3), "v"(n4));
    if (TOL) {
      // Z: no code differs so far; O: at most one does
      uint32_t Z = e0, O = e0 | e1;
      Z &= e1;
      O = Z | (O & e2);
      Z &= e2;
      O = Z | (O & e3);
      Z &= e3;
      O = Z | (O & e4);
      Z &= e4;
      O = Z | (O & e5);
      Z &= e5;
      O = Z | (O & e6);
      Z &= e6;
      O = Z | (O & e7);
      c |= O;
    } else {
      const uint32_t z0 = e0 & e1 & e2, z1 = e3 & e4 & e5;
      c |= z0 & z1 & e6 & e7;
    }
  }
  return c;
}

}  // namespace

template <int W, int MAXK, bool TOL>
struct GeneralShape {
  using Args = PlaneCountGParams;
  static constexpr uint32_t kFixedLen = 0;
  static constexpr bool kBlobInLds = true;
  static constexpr bool kList = false;
  static constexpr bool kCapture = false;
  __device__ static __forceinline__ void classify_bytes(const Args&, const uint32_t*, uint64_t, bool, uint32_t, uint32_t, uint32_t& mask) { mask = 0; }
  __device__ static __forceinline__ const PlaneCountParams& common(const Args& g) { return g.c; }
  __device__ static __forceinline__ uint32_t lmax(const Args& g) { return g.lmax; }
  __device__ static __forceinline__ uint32_t offset_of(const Args&, const uint32_t* blob, uint32_t p) {
    return reinterpret_cast<const ClassifyDesc*>(blob)[p].win_offset;
  }
  __device__ static __forceinline__ uint32_t test(uint32_t ta, uint32_t tb, uint32_t ha, uint32_t hb, const Args& g) { return general_test<TOL>(ta, tb, ha, hb, g); }
  // per pattern: the start s = w - its window offset inside the own range, the window inside the text, one of its windows
  // matches exactly -- then its automaton from s (classify_shared_general's steps 1 and 2, for 64 candidates at a time)
  __device__ static __forceinline__ void classify(const Args& g, const uint32_t* blob, uint64_t w, bool have, uint32_t& mask, Lens& lens) {
    const PlaneCountParams& a = g.c;
    const ClassifyDesc* desc = reinterpret_cast<const ClassifyDesc*>(blob);
    const uint32_t* tab = blob + g.desc_words;
    uint32_t lo = 0, hi = 0;
    if (have) {
      if (w + 8 <= a.n) {
        __builtin_memcpy(&lo, a.text + w, 4);
        __builtin_memcpy(&hi, a.text + w + 4, 4);
      } else {
        for (uint32_t q = 0; q < 8 && w + q < a.n; q++) {
          const uint32_t c = a.text[w + q];
          if (q < 4) lo |= c << (8 * q);
          else hi |= c << (8 * (q - 4));
        }
      }
    }
    uint32_t todo = 0;
    for (uint32_t p = 0; p < a.n_patterns; p++) {
      const ClassifyDesc& d = desc[p];
      const uint64_t s = w - d.win_offset;
      const bool ok = have && w >= d.win_offset && s >= a.sb && s < a.se && w + d.win_len <= a.n;
      bool win = false;
      for (uint32_t q = 0; q < d.n_windows; q++) win = win || ((((lo ^ d.v0[q]) & d.m0[q]) | ((hi ^ d.v1[q]) & d.m1[q])) == 0);
      todo |= (ok && win) ? 1u << p : 0u;
    }
    // in pass r every lane runs the r-th pattern whose window test ITS candidate passed (tables addressed per lane)
    mask = 0;
    lens.w[0] = lens.w[1] = 0;
    while (__ballot(todo != 0) != 0) {
      const bool act = todo != 0;
      const uint32_t p = act ? static_cast<uint32_t>(__builtin_ctz(todo)) : 0u;
      todo &= todo - 1;
      const ClassifyDesc& d = desc[p];
      const uint32_t* t0 = tab + d.tab;
      const uint64_t s = w - d.win_offset;
      uint64_t t_lo = 0, t_hi = 0;
      if (act) rj_load16(a.text, a.n, s, &t_lo, &t_hi);
      const uint32_t avail = act ? (a.n - s < 16 ? static_cast<uint32_t>(a.n - s) : 16u) : 0u;
      uint32_t len = 0;
      bool found;
      if (W == 1 || d.n_words <= 1) found = short_longest_lds<1, MAXK>(t0, d, t_lo, t_hi, avail, &len);
      else found = short_longest_lds<W, MAXK>(t0, d, t_lo, t_hi, avail, &len);
      if (found && act && len != 0) {
        mask |= 1u << p;
        if (p < 12) lens.w[0] |= static_cast<uint64_t>(len) << (5 * p);
        else lens.w[1] |= static_cast<uint64_t>(len) << (5 * (p - 12));
      }
    }
  }
};

// GeneralListShape<TOL>: GeneralShape's test, the candidates written to the span pipeline's shared regions (kernels.h:
// PlaneListGParams) for classify_shared_general -- what ListShape is to ExactShape.
template <bool TOL>
struct GeneralListShape {
  using Args = PlaneListGParams;
  static constexpr uint32_t kFixedLen = 0;
  static constexpr bool kBlobInLds = false;
  static constexpr bool kList = true;
  static constexpr bool kCapture = false;
  __device__ static __forceinline__ void classify_bytes(const Args&, const uint32_t*, uint64_t, bool, uint32_t, uint32_t, uint32_t& mask) { mask = 0; }
  __device__ static __forceinline__ const PlaneCountParams& common(const Args& g) { return g.g.c; }
  __device__ static __forceinline__ uint32_t lmax(const Args& g) { return g.g.lmax; }
  __device__ static __forceinline__ uint32_t offset_of(const Args&, const uint32_t*, uint32_t) { return 0u; }
  __device__ static __forceinline__ uint32_t test(uint32_t ta, uint32_t tb, uint32_t ha, uint32_t hb, const Args& g) { return general_test<TOL>(ta, tb, ha, hb, g.g); }
  __device__ static __forceinline__ void classify(const Args&, const uint32_t*, uint64_t, bool, uint32_t& mask, Lens&) { mask = 0; }
};

namespace {

// per wave: the ring, and what the classification carries from batch to batch
struct WaveState {
  uint32_t* ring;        // LDS, kRing slots: window positions as offsets from the span's first byte, in text order
                         // (capture: the window's bytes 0..3 in ring[kRing + slot], 4..7 in ring[2 kRing + slot])
  uint32_t* stash;       // LDS, capture: two slots of kStashWords
  uint32_t head, tail;   // wave-uniform, free-running
  uint32_t acc;          // lane p: matches of pattern p so far
  uint32_t flags;        // kPcConflict | kPcVoid (wave-uniform)
  uint32_t prev_rel, prev_valid;   // the last candidate classified so far
  // the wave's first / last match of pattern p (LDS): ends[4 p ..] = first offset, first length, last offset, last length
  // (offsets of the candidate, i.e. of the window; kNoEnd: none yet).  The last one is also the selection's state: the
  // match the next one of the pattern must not begin inside.
  uint32_t* ends;
  // ListShape: the wave's region in device memory (slot = the candidate's start as an absolute offset), its capacity
  uint64_t* list;
  uint32_t list_cap;
  uint64_t list_base;    // the span's first byte minus the windows' offset inside a match
};
constexpr uint32_t kNoEnd = 0xFFFFFFFFu;

// the candidates of one block, in text order, to the ring (LIST: to the wave's region in device memory; w.tail counts them)
// (CAPTURE: block_rel = the block's first byte as an offset from the span's, slot = the block's stash slot)
template <bool LIST, bool CAPTURE>
__device__ __forceinline__ void push_block(WaveState& w, uint32_t hm, uint32_t rel_lane, uint32_t block_rel = 0, uint32_t slot = 0) {
  const uint64_t any = __ballot(hm != 0);
  if (any == 0) return;  // wave-uniform
  auto put = [&](uint32_t idx, uint32_t rel) {
    if (LIST) {
      if (idx < w.list_cap) w.list[idx] = w.list_base + rel;
    } else {
      w.ring[idx & (kRing - 1)] = rel;
      if (CAPTURE) {
        // the window's 8 bytes: three aligned words of the stash, shifted (bytes 2048..2055 = the next block's first 8)
        const uint32_t o = rel - block_rel;
        const uint32_t* sw = w.stash + slot * kStashWords + (o >> 2);
        const uint32_t s0 = sw[0], s1 = sw[1], s2 = sw[2];
        w.ring[kRing + (idx & (kRing - 1))] = __builtin_amdgcn_alignbyte(s1, s0, o & 3u);
        w.ring[2 * kRing + (idx & (kRing - 1))] = __builtin_amdgcn_alignbyte(s2, s1, o & 3u);
      }
    }
  };
  // text order: the pieces A of lanes 0..63 (the block's first KiB), then the pieces B
  const uint32_t hm_a = hm & 0x55555555u, hm_b = hm & 0xAAAAAAAAu;
  const uint64_t several = __ballot((hm & (hm - 1)) != 0);
  if (several == 0) {
    // the usual case: no lane holds two candidates
    const uint64_t any_a = __ballot(hm_a != 0);
    if (hm != 0) {
      const uint32_t bit = static_cast<uint32_t>(__builtin_ctz(hm));
      const uint32_t at = hm_a != 0 ? lanes_below(any_a) : static_cast<uint32_t>(__popcll(any_a)) + lanes_below(any & ~any_a);
      put(w.tail + at, rel_lane + (bit >> 1) + ((bit & 1u) ? kPieceB : 0u));
    }
    w.tail += static_cast<uint32_t>(__popcll(any));
    return;
  }
  // (one prefix sum for both: piece A's count in the low half of the word, piece B's in the high half; <= 16 per lane and piece)
  const uint32_t c = __popc(hm_a) | (__popc(hm_b) << 16);
  const uint32_t inc = wave_inclusive_sum(c);
  const uint32_t tots = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(inc), kWave - 1));
  const uint32_t tot_a = tots & 0xFFFFu, tot = tot_a + (tots >> 16);
  if (LIST || tot <= kRing) {   // (ring: more -- the caller's occupancy test voids the run; nothing is written)
    const uint32_t before = inc - c;
    uint32_t idx = w.tail + (before & 0xFFFFu);
    for (uint32_t m = hm_a; m; m &= m - 1, idx++) put(idx, rel_lane + (static_cast<uint32_t>(__builtin_ctz(m)) >> 1));
    idx = w.tail + tot_a + (before >> 16);
    for (uint32_t m = hm_b; m; m &= m - 1, idx++) put(idx, rel_lane + kPieceB + (static_cast<uint32_t>(__builtin_ctz(m)) >> 1));
  }
  w.tail += tot;
}

// The first m (<= 64) candidates of the ring: classified, the selection rule applied, counted.
//
// The selection (reference src/x64/codegen-x64.cc:401-466: left-most-longest, non-overlapping; a match that begins inside
// the one selected before it is dropped, :448-460) is per pattern a walk along its matches in text order -- keep one when
// it begins at or behind the end of the last one KEPT.  A candidate whose distance to the candidate before it is at least
// the longest match (lmax) cannot begin inside anything: kept without asking.  The others (0.6 % of the candidates on DNA)
// are "suspects": for every pattern that matches at a suspect the wave walks that pattern's matches of the batch in order
// -- a scalar loop of two v_readlanes per match, starting from the pattern's last kept match (its ends[] entry, which is
// how the rule carries across batches) -- and drops what the walk drops.  Exact for pairs (`agggtaaagggtaaa` counts once)
// and for chains of any length (`agggtaaagggtaaagggtaaa`: the first and the third); round 5 voided the run on every pair.
template <class S>
__device__ __forceinline__ void classify_batch(WaveState& w, uint32_t m, const uint32_t* tab, const typename S::Args& g, uint64_t span_base) {
  const PlaneCountParams& a = S::common(g);
  const int lane = lane_id();
  const bool have = static_cast<uint32_t>(lane) < m;
  const uint32_t rel = w.ring[(w.head + static_cast<uint32_t>(lane)) & (kRing - 1)];
  uint32_t mask;
  Lens lens;
  if constexpr (S::kCapture) {
    const uint32_t lo = w.ring[kRing + ((w.head + static_cast<uint32_t>(lane)) & (kRing - 1))];
    const uint32_t hi = w.ring[2 * kRing + ((w.head + static_cast<uint32_t>(lane)) & (kRing - 1))];
    S::classify_bytes(g, tab, span_base + rel, have, lo, hi, mask);
  } else {
    S::classify(g, tab, span_base + rel, have, mask, lens);
  }
  const uint32_t before = from_lane_below(rel, w.prev_rel);
  const bool has_before = lane > 0 || w.prev_valid != 0;
  const bool close = have && has_before && (rel - before) < S::lmax(g);
  if (__ballot(close && mask != 0) != 0) {  // (wave-uniform)
    const uint32_t all_mask = mask;
    const uint32_t suspect = close ? all_mask : 0u;
    for (uint32_t p = 0; p < a.n_patterns; p++) {
      if (__ballot(((suspect >> p) & 1u) != 0) == 0) continue;  // (wave-uniform)
      const uint64_t all = __ballot(((all_mask >> p) & 1u) != 0);
      const uint32_t my_len = S::kFixedLen ? S::kFixedLen : len_of(lens, p);
      // the pattern's last kept match before this batch (every lane reads the same LDS words)
      const uint32_t last_rel = w.ends[4 * p + 2], last_len = w.ends[4 * p + 3];
      uint32_t kept_end = w.ends[4 * p] == kNoEnd ? 0u : static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(last_rel + last_len)));
      uint64_t kept = 0;
      for (uint64_t mm = all; mm != 0; mm &= mm - 1) {
        const int i = __builtin_ctzll(mm);
        const uint32_t wi = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(rel), i));
        const uint32_t li = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(my_len), i));
        if (wi >= kept_end) {
          kept |= 1ull << i;
          kept_end = wi + li;
        }
      }
      if ((((all & ~kept) >> lane) & 1ull) != 0) mask &= ~(1u << p);
    }
  }
  // counts: a ballot per pattern, lane p keeps pattern p's; the wave's first / last match of the pattern
  for (uint32_t p = 0; p < a.n_patterns; p++) {
    const uint64_t mine = __ballot(((mask >> p) & 1u) != 0);
    if (mine == 0) continue;  // (wave-uniform)
    if (lane == static_cast<int>(p)) w.acc += static_cast<uint32_t>(__popcll(mine));
    const int fl = __builtin_ctzll(mine), ll = 63 - __builtin_clzll(mine);
    const uint32_t my_len = S::kFixedLen ? S::kFixedLen : len_of(lens, p);
    const uint32_t first = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(rel), fl));
    const uint32_t last = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(rel), ll));
    const uint32_t first_len = S::kFixedLen ? S::kFixedLen : static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(my_len), fl));
    const uint32_t last_len = S::kFixedLen ? S::kFixedLen : static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(my_len), ll));
    if (lane == static_cast<int>(p)) {
      if (w.ends[4 * p] == kNoEnd) {
        w.ends[4 * p] = first;
        w.ends[4 * p + 1] = first_len;
      }
      w.ends[4 * p + 2] = last;
      w.ends[4 * p + 3] = last_len;
    }
  }
  w.prev_rel = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(rel), static_cast<int>(m - 1)));
  w.prev_valid = 1;
  w.head += m;
}

// The first candidates of a span may lie inside a match that begins in the span before it (another wave's): the wave
// classifies the lmax - 1 positions before its span itself.  Conservative: any pattern in common between those positions
// and the span's candidates in its first lmax - 1 bytes flags the run (kPcConflict: the host repeats it with the span
// pipeline).  (Those candidates are the ring's first entries -- text order -- and at most lmax - 1 <= 15: the low lanes of
// the span's first batch.)
template <class S>
__device__ __forceinline__ void check_span_start(WaveState& w, uint32_t m, const uint32_t* tab, const typename S::Args& g, uint64_t span_base) {
  const PlaneCountParams& a = S::common(g);
  const int lane = lane_id();
  const uint32_t reach = S::lmax(g) - 1u;   // (<= 15)
  const uint32_t rel = w.ring[(w.head + static_cast<uint32_t>(lane)) & (kRing - 1)];
  const bool near = static_cast<uint32_t>(lane) < m && static_cast<uint32_t>(lane) < reach && rel < reach;
  if (__ballot(near) == 0 || span_base == 0) return;  // (wave-uniform)
  uint32_t any[2] = {0, 0};  // patterns that match at one of the positions before the span / at a near candidate
#pragma unroll
  for (int round = 0; round < 2; round++) {
    const uint64_t pos = round == 0 ? span_base + static_cast<uint64_t>(lane) - reach : span_base + rel;
    const bool mine = round == 0 ? (static_cast<uint32_t>(lane) < reach && span_base + static_cast<uint64_t>(lane) >= reach) : near;
    uint32_t mask;
    Lens lens;
    S::classify(g, tab, pos, mine, mask, lens);
    for (uint32_t p = 0; p < a.n_patterns; p++)
      if (__ballot(((mask >> p) & 1u) != 0) != 0) any[round] |= 1u << p;
  }
  if ((any[0] & any[1]) != 0) w.flags |= kPcConflict;
}

// after every second block: the run is void when a block overfilled the ring; else classify while 64 are held
template <class S>
__device__ __forceinline__ void blocks_done(WaveState& w, const uint32_t* tab, const typename S::Args& g, uint64_t span_base, bool& first_batch) {
  const PlaneCountParams& a = S::common(g);
  if (w.tail - w.head > kRing) {  // more candidates than the ring takes: the run is void
    w.flags |= kPcVoid;
    w.head = w.tail;
    return;
  }
  while (w.tail - w.head >= a.batch_at) {
    const uint32_t held = w.tail - w.head;
    const uint32_t m = held < 64u ? held : 64u;
    if (first_batch) check_span_start<S>(w, m, tab, g, span_base);
    first_batch = false;
    classify_batch<S>(w, m, tab, g, span_base);
  }
}

}  // namespace

// (block indices are 32-bit inside the kernel: texts of up to 8 TiB)
template <class S>
__global__ __launch_bounds__(256) void plane_count(typename S::Args g) {
  // ExactShape: the table (exact_count.h); GeneralShape: the blob of descriptors + automaton tables (dynamic LDS)
  __shared__ __attribute__((aligned(16))) uint32_t table[kExactTabWords];
  extern __shared__ __attribute__((aligned(16))) uint32_t blob[];
  __shared__ uint32_t rings[4][kRing * (S::kCapture ? 3 : 1)];
  __shared__ __attribute__((aligned(16))) uint32_t stashes[4][S::kCapture ? 2 * kStashWords : 4];
  __shared__ uint32_t ends[4][4 * kExactMaxPatterns];
  __shared__ unsigned long long wave_bounds[4][kExactMaxPatterns][2];
  __shared__ uint32_t wave_counts[4][kExactMaxPatterns];
  __shared__ uint32_t wave_flags[4];
  const PlaneCountParams& a = S::common(g);
  const uint32_t* tab = S::kBlobInLds ? blob : table;
  const int lane = lane_id();
  const uint32_t wid = threadIdx.x >> 6;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>((static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6));
  Consts k;
  k.shift = a.code_shift;
  k.cmask = 0x03030303u << a.code_shift;
  // the blocks are dealt out evenly: wave w takes span_blocks of them, the first span_extra waves one more (a rounded-up
  // span for everybody left the last 8 % of the grid without work on the 500 MB text, and the others 13 blocks for 12)
  const uint32_t c0 = static_cast<uint32_t>(a.first_block) + wave * static_cast<uint32_t>(a.span_blocks) + (wave < a.span_extra ? wave : a.span_extra);
  const uint32_t c1 = c0 + static_cast<uint32_t>(a.span_blocks) + (wave < a.span_extra ? 1u : 0u);
  // blocks below fast_end: the block and the one behind it lie inside the text (unguarded loads)
  const uint32_t full = static_cast<uint32_t>(a.n / kBlock);
  uint32_t fast_end = full >= 1 ? full - 1 : 0;
  if (fast_end > c1) fast_end = c1;
  if (fast_end < c0) fast_end = c0;
  const uint64_t span_base = static_cast<uint64_t>(c0) * kBlock;
  WaveState w;
  w.ring = rings[wid];
  w.stash = stashes[wid];
  w.ends = ends[wid];
  w.head = w.tail = 0;
  w.acc = 0;
  w.flags = 0;
  w.prev_rel = w.prev_valid = 0;
  w.ends[lane] = kNoEnd;   // (4 x 32 entries: two per lane)
  w.ends[lane + 64] = kNoEnd;
  w.list = nullptr;
  w.list_cap = 0;
  w.list_base = 0;
  if constexpr (S::kList) {
    if (wave == 0 && lane < kCntSize)
      for (uint32_t p = 0; p < g.n_zero; p++) g.zero_counters[p][lane] = 0;
    w.list = g.hits + static_cast<uint64_t>(wave) * g.region_cap;
    w.list_cap = g.region_cap;
    w.list_base = span_base - g.offset;   // (a window before `offset` wraps to a start beyond every range: dropped by the classification)
  }
  bool first_batch = true;

  // A wave never loads a block that is not its own: prefetches beyond the span's last fast block are left out inside the
  // loop (wave-uniform branches; the buffers then keep codes nobody uses), and what lane 63 needs of the block BEHIND the
  // span -- the codes of its first 8 bytes -- comes from one 8-byte load, the same address in every lane.  History: clamped
  // to the text's last block only, the prefetches ran three blocks into the neighbour's span (FETCH_SIZE 1.22 x the text
  // on spans of 20 blocks); clamped to the span's own last block they were cache hits under the default load policy, but
  // a non-temporal line does not wait in the cache to be hit: with nt loads the clamped form cost the list kernel 9 %
  // (0.0867 -> 0.0790 ms per 500 MB without them, A/B in one gpurun call).
  const uint32_t last_own = fast_end > c0 ? fast_end - 1 : c0;
  auto blk = [&](uint32_t c) { return a.text + static_cast<uint64_t>(c < last_own ? c : last_own) * kBlock; };
  const uint32_t lane_rel = static_cast<uint32_t>(lane) * 16u;   // piece A; piece B kPieceB behind it
  Raw ra, rb;
  uint2 behind{0, 0};
  uint32_t c = c0;
  // capture: block ci's bytes to its stash slot (text order: piece A of lane L at 16 L, piece B at 1024 + 16 L) ...
  auto stash_put = [&](uint32_t ci, const Raw& r) __attribute__((always_inline)) {
    if constexpr (S::kCapture) {
      uint32_t* sl = w.stash + (ci & 1u) * kStashWords;
      *reinterpret_cast<uint4*>(sl + lane * 4) = r.a;
      *reinterpret_cast<uint4*>(sl + kPieceB / 4 + lane * 4) = r.b;
    }
  };
  // ... and behind them the first 8 bytes of the block that follows (lane 0's x, y)
  auto spill_put = [&](uint32_t ci, uint32_t x, uint32_t y) __attribute__((always_inline)) {
    if constexpr (S::kCapture) {
      if (lane == 0) *reinterpret_cast<uint2*>(w.stash + (ci & 1u) * kStashWords + kBlock / 4) = make_uint2(x, y);
    }
  };
  if (c < fast_end) {
    load_block(blk(c), lane_rel, ra);
    load_block(blk(c + 1), lane_rel, rb);
    behind = *reinterpret_cast<const uint2*>(a.text + static_cast<uint64_t>(fast_end) * kBlock);   // (block fast_end lies inside the text)
  }
  // the table / the blob (all waves), before the first wait for text
  if (S::kList) {
    // (nothing to stage: the candidates are classified by the next kernel)
  } else if (S::kBlobInLds) {
    const uint4* src = reinterpret_cast<const uint4*>(a.table);
    uint4* dst = reinterpret_cast<uint4*>(blob);
    for (uint32_t i = threadIdx.x; i < a.table_words / 4; i += blockDim.x) dst[i] = src[i];
  } else {
    for (uint32_t i = threadIdx.x; i < kExactTabWords; i += blockDim.x) table[i] = a.table[i];
  }
  __syncthreads();
  if (c < fast_end) {
    uint32_t xa = codes16(ra.a, k), xb = codes16(ra.b, k);   // block c
    stash_put(c, ra);
    if (S::kCapture) __builtin_amdgcn_sched_barrier(0);   // (the stash is written before the buffer is loaded again)
    load_block(blk(c + 2), lane_rel, ra);
    const uint32_t behind_codes = (codes4(behind.x, k) >> k.shift) | (codes4(behind.y, k) << (8 - k.shift));
    // two blocks per iteration: x = the codes of block c, rb = block c + 1, ra = block c + 2 (in flight)
    while (c + 1 < fast_end) {
      const uint32_t ya = codes16(rb.a, k), yb = codes16(rb.b, k);   // block c + 1
      // (the codes must exist BEFORE the buffer is loaded again: when the compiler sinks their computation towards its
      // use, the reload lands in other registers and is copied back at the loop's end -- behind a wait for it)
      asm volatile("" ::"v"(ya), "v"(yb));
      stash_put(c + 1, rb);
      spill_put(c, rb.a.x, rb.a.y);
      if (S::kCapture) __builtin_amdgcn_sched_barrier(0);
      if (c + 3 <= last_own) load_block(blk(c + 3), lane_rel, rb);
      {
        // (behind lane 63's piece A: lane 0's piece B; behind its piece B: lane 0's piece A of the next block)
        const uint32_t ha = from_lane_above(xa, static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(xb))));
        const uint32_t hb = from_lane_above(xb, static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(ya))));
        const uint32_t hm = S::test(xa, xb, ha, hb, g);
        push_block<S::kList, S::kCapture>(w, hm, (c - c0) * static_cast<uint32_t>(kBlock) + lane_rel, (c - c0) * static_cast<uint32_t>(kBlock), c & 1u);
      }
      __builtin_amdgcn_sched_barrier(0);
      xa = codes16(ra.a, k);   // block c + 2 (stale bytes when c + 2 == fast_end: not used then)
      xb = codes16(ra.b, k);
      asm volatile("" ::"v"(xa), "v"(xb));
      if (c + 2 < fast_end) {
        stash_put(c + 2, ra);
        spill_put(c + 1, ra.a.x, ra.a.y);
      } else {
        spill_put(c + 1, behind.x, behind.y);
      }
      if (S::kCapture) __builtin_amdgcn_sched_barrier(0);
      if (c + 4 <= last_own) load_block(blk(c + 4), lane_rel, ra);
      {
        const uint32_t next0 = c + 2 == fast_end ? behind_codes : static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(xa)));
        const uint32_t ha = from_lane_above(ya, static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(yb))));
        const uint32_t hb = from_lane_above(yb, next0);
        const uint32_t hm = S::test(ya, yb, ha, hb, g);
        push_block<S::kList, S::kCapture>(w, hm, (c + 1 - c0) * static_cast<uint32_t>(kBlock) + lane_rel, (c + 1 - c0) * static_cast<uint32_t>(kBlock), (c + 1) & 1u);
      }
      __builtin_amdgcn_sched_barrier(0);
      c += 2;
      if (!S::kList && w.tail - w.head >= a.batch_at) blocks_done<S>(w, tab, g, span_base, first_batch);
    }
    if (c + 1 == fast_end) {   // an odd block left: x holds its codes; behind it the span ends
      const uint32_t ha = from_lane_above(xa, static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(xb))));
      const uint32_t hb = from_lane_above(xb, static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(behind_codes))));
      const uint32_t hm = S::test(xa, xb, ha, hb, g);
      spill_put(c, behind.x, behind.y);
      push_block<S::kList, S::kCapture>(w, hm, (c - c0) * static_cast<uint32_t>(kBlock) + lane_rel, (c - c0) * static_cast<uint32_t>(kBlock), c & 1u);
      c++;
    }
    if (!S::kList) blocks_done<S>(w, tab, g, span_base, first_batch);
  }
  // the block(s) at the end of the text: guarded loads, the 8 bytes behind each of the lane's pieces read by the lane itself
  for (; c < c1; c++) {
    const uint64_t at = static_cast<uint64_t>(c) * kBlock + lane_rel, bt = at + kPieceB;
    uint4 va, vb;
    va.x = guarded_dword(a.text, a.n, at);
    va.y = guarded_dword(a.text, a.n, at + 4);
    va.z = guarded_dword(a.text, a.n, at + 8);
    va.w = guarded_dword(a.text, a.n, at + 12);
    vb.x = guarded_dword(a.text, a.n, bt);
    vb.y = guarded_dword(a.text, a.n, bt + 4);
    vb.z = guarded_dword(a.text, a.n, bt + 8);
    vb.w = guarded_dword(a.text, a.n, bt + 12);
    const uint32_t g0 = guarded_dword(a.text, a.n, at + 16), g1 = guarded_dword(a.text, a.n, at + 20);
    const uint32_t h0 = guarded_dword(a.text, a.n, bt + 16), h1 = guarded_dword(a.text, a.n, bt + 20);
    const uint32_t ha = (codes4(g0, k) >> k.shift) | (codes4(g1, k) << (8 - k.shift));
    const uint32_t hb = (codes4(h0, k) >> k.shift) | (codes4(h1, k) << (8 - k.shift));
    const uint32_t hm = S::test(codes16(va, k), codes16(vb, k), ha, hb, g);
    if (!S::kList && w.tail - w.head > kRing - 64u) blocks_done<S>(w, tab, g, span_base, first_batch);   // (room for this block)
    if constexpr (S::kCapture) {
      Raw gr;
      gr.a = va;
      gr.b = vb;
      stash_put(c, gr);
      const uint64_t nb = (static_cast<uint64_t>(c) + 1) * kBlock;
      spill_put(c, guarded_dword(a.text, a.n, nb), guarded_dword(a.text, a.n, nb + 4));
    }
    push_block<S::kList, S::kCapture>(w, hm, (c - c0) * static_cast<uint32_t>(kBlock) + lane_rel, (c - c0) * static_cast<uint32_t>(kBlock), c & 1u);
  }
  if constexpr (S::kList) {
    if (lane == 0) g.hit_counts[wave] = w.tail;   // (also beyond the region's capacity: the classification reports the overflow)
    return;
  }
  // what the ring still holds
  if (w.tail - w.head > kRing) {
    w.flags |= kPcVoid;
    w.head = w.tail;
  }
  while (w.tail != w.head) {
    const uint32_t held = w.tail - w.head;
    const uint32_t m = held < 64u ? held : 64u;
    if (first_batch) check_span_start<S>(w, m, tab, g, span_base);
    first_batch = false;
    classify_batch<S>(w, m, tab, g, span_base);
  }
  // wave -> workgroup (LDS) -> rows of the workgroup's own in device memory: plain stores, read by plane_count_finish
  // (the next kernel on the stream).  Measured alternatives, all inside this kernel: nine device-scope atomic adds and a
  // ticket per workgroup (kernel 0.094 -> 0.159 ms: the memory side performs same-address atomics one after the other);
  // rows + a ticket per group of 64 workgroups + a top ticket (0.183 ms: it is the atomic WITH RETURN that costs, ~18 us of
  // device time per thousand of them, whatever the address; an acq_rel ticket -- L2 write-back + invalidate per
  // workgroup -- 0.41 ms).
  // Every workgroup leaves its first / last match of every pattern beside its counts (round 6; round 5 kept them for the
  // 64 waves at either end of the grid only, and a caller who asked for the bounds of a pattern without a match there had
  // the span pipeline re-read a text that might be gone by then): plane_count_finish finds the first / last ROW with a
  // count from the rows it adds up anyway and reads two bounds per pattern.
  if (lane < kExactMaxPatterns) {
    const bool real = lane < static_cast<int>(a.n_patterns);
    wave_counts[wid][lane] = real ? w.acc : 0u;
    const uint32_t ef = w.ends[4 * lane], efl = w.ends[4 * lane + 1], el = w.ends[4 * lane + 2], ell = w.ends[4 * lane + 3];
    const uint64_t off = real ? S::offset_of(g, tab, static_cast<uint32_t>(lane)) : 0u;   // the match begins `off` bytes before its window
    wave_bounds[wid][lane][0] = ef != kNoEnd ? ((span_base + ef - off) | (static_cast<unsigned long long>(efl) << kPcLenShift)) : kPcNone;
    wave_bounds[wid][lane][1] = ef != kNoEnd ? ((span_base + el - off) | (static_cast<unsigned long long>(ell) << kPcLenShift)) : kPcNone;
  }
  if (lane == 0) wave_flags[wid] = w.flags;
  __syncthreads();
  if (wid != 0 || lane >= kExactMaxPatterns) return;
  uint32_t v = wave_counts[0][lane] + wave_counts[1][lane] + wave_counts[2][lane] + wave_counts[3][lane];
  if (lane == kExactMaxPatterns - 1) v = wave_flags[0] | wave_flags[1] | wave_flags[2] | wave_flags[3];   // (slot 31: the flags)
  a.wg_rows[static_cast<uint64_t>(blockIdx.x) * kExactMaxPatterns + lane] = v;
  if (lane < static_cast<int>(a.n_patterns) && v != 0) {   // (the waves of a workgroup hold consecutive spans)
    unsigned long long first = kPcNone, last = kPcNone;
#pragma unroll
    for (int q = 3; q >= 0; q--)
      if (wave_bounds[q][lane][0] != kPcNone) first = wave_bounds[q][lane][0];
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (wave_bounds[q][lane][1] != kPcNone) last = wave_bounds[q][lane][1];
    *reinterpret_cast<ulonglong2*>(a.wg_bounds + (static_cast<uint64_t>(blockIdx.x) * kExactMaxPatterns + lane) * 2) = ulonglong2{first, last};
  }
}

// The rows of plane_count added up: counts, flags and bounds to pinned host memory and to the device copy in `acc`
// (rj_multi_bounds_device).  kFinishGroups workgroups of the scan's own shape (4 waves, little LDS), so that they find a
// place between the workgroups of the NEXT scan, which is running when a caller keeps steps in flight: ONE workgroup
// of 1024 threads waited for 16 free wave slots on one CU -- until that scan had drained (93 us instead of 18).  As few
// DEPENDENT trips to memory as possible (a lone workgroup pays ~2 us per trip, more under a scan): thread t adds four
// patterns (t & 7) of its group's rows t >> 3, + 32, ... with up to 14 16-byte loads in flight and remembers the first /
// last row with a count; the group's sums go to acc, a ticket (eight arrivals: the cost of an atomic with return does not
// matter here) finds the last group, which adds them up and reads, in one more trip, the first match of every pattern's
// first row with a count and the last match of its last one.
// (<= 64 VGPRs -- __launch_bounds__(256, 8), eight loads in flight instead of fourteen, the reduce loops not unrolled: a workgroup of this kernel starts under a
// running scan as soon as ONE scan wave per SIMD has retired; with its 132 VGPRs of round 6's first version it waited for three)
constexpr uint32_t kFinishBatch = 8;
__global__ __launch_bounds__(256, 8) void plane_count_finish(PlaneCountParams a, uint32_t n_wg) {
  __shared__ unsigned long long part[32][kExactMaxPatterns];
  __shared__ uint32_t part_first[32][kExactMaxPatterns], part_last[32][kExactMaxPatterns];   // row + 1; 0: none
  __shared__ uint32_t is_last;
  const uint32_t p4 = threadIdx.x & 7u, q = threadIdx.x >> 3;
  const uint32_t per = (n_wg + gridDim.x - 1) / gridDim.x;
  const uint32_t r_lo = blockIdx.x * per, r_hi = r_lo + per < n_wg ? r_lo + per : n_wg;
  uint32_t sum[4] = {0, 0, 0, 0};   // (a thread adds <= n_wg / 256 rows of <= 163 840 matches each: 32 bits)
  uint32_t first_row[4] = {0, 0, 0, 0}, last_row[4] = {0, 0, 0, 0};
  const uint4* rows = reinterpret_cast<const uint4*>(a.wg_rows);
  for (uint32_t r = r_lo + q; r < r_hi; r += 32 * kFinishBatch) {
    uint4 v[kFinishBatch];
#pragma unroll
    for (uint32_t i = 0; i < kFinishBatch; i++) {
      const uint32_t rr = r + 32 * i;
      v[i] = rows[static_cast<uint64_t>(rr < r_hi ? rr : r) * 8 + p4];   // (clamped, not skipped: no branch between the loads)
      if (rr >= r_hi) v[i] = uint4{0, 0, 0, 0};
    }
#pragma unroll
    for (uint32_t i = 0; i < kFinishBatch; i++) {
      const uint32_t rr1 = r + 32 * i + 1;
      const uint32_t c[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
      sum[0] += v[i].x;
      sum[1] += v[i].y;
      sum[2] += v[i].z;
      sum[3] = p4 == 7 ? (sum[3] | v[i].w) : (sum[3] + v[i].w);   // (slot 31: the flags)
#pragma unroll
      for (int k = 0; k < 4; k++) {
        first_row[k] = (c[k] != 0 && first_row[k] == 0) ? rr1 : first_row[k];   // (rows in rising order)
        last_row[k] = c[k] != 0 ? rr1 : last_row[k];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    part[q][4 * p4 + i] = sum[i];
    part_first[q][4 * p4 + i] = first_row[i];
    part_last[q][4 * p4 + i] = last_row[i];
  }
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u, wave_id = threadIdx.x >> 6;
  if (wave_id == 0) {
    if (lane < kExactMaxPatterns) {
      unsigned long long t = 0;
      uint32_t f = 0, l = 0;
#pragma unroll 4
      for (uint32_t i = 0; i < 32; i++) {
        t = lane == kExactMaxPatterns - 1 ? (t | part[i][lane]) : (t + part[i][lane]);
        const uint32_t pf = part_first[i][lane], pl = part_last[i][lane];
        f = pf != 0 && (f == 0 || pf < f) ? pf : f;
        l = pl > l ? pl : l;
      }
      unsigned long long* g = a.acc + kPcGroupRows + blockIdx.x * 96;
      __hip_atomic_store(&g[lane], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&g[32 + lane], static_cast<unsigned long long>(f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&g[64 + lane], static_cast<unsigned long long>(l), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (eight arrivals per launch: a release / acquire ticket costs nothing measurable here -- it is what the memory model asks for)
    if (lane == 0) {
      const unsigned long long t = __hip_atomic_fetch_add(&a.acc[kPcTicket], 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      is_last = t + 1 == gridDim.x ? 1u : 0u;
      if (t + 1 == gridDim.x) __hip_atomic_store(&a.acc[kPcTicket], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (for the next run: exactly gridDim.x arrivals per launch)
    }
  }
  __syncthreads();
  if (!is_last || threadIdx.x >= kExactMaxPatterns) return;
  // the last group
  const uint32_t p = threadIdx.x;
  unsigned long long total = 0;
  uint32_t f = 0, l = 0;
#pragma unroll 2
  for (uint32_t g = 0; g < gridDim.x; g++) {
    const unsigned long long* row = a.acc + kPcGroupRows + g * 96;
    const unsigned long long v = __hip_atomic_load(&row[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t pf = static_cast<uint32_t>(__hip_atomic_load(&row[32 + p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const uint32_t pl = static_cast<uint32_t>(__hip_atomic_load(&row[64 + p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    total = p == kExactMaxPatterns - 1 ? (total | v) : (total + v);
    f = pf != 0 && (f == 0 || pf < f) ? pf : f;
    l = pl > l ? pl : l;
  }
  if (p == kExactMaxPatterns - 1) a.host_out[kPcHostFlags] = total;
  if (p >= a.n_patterns) return;
  unsigned long long first = kPcNone, last = kPcNone;
  if (total != 0 && f != 0) {   // (the scan kernel's plain stores: complete before this kernel began)
    first = a.wg_bounds[(static_cast<uint64_t>(f - 1) * kExactMaxPatterns + p) * 2];
    last = a.wg_bounds[(static_cast<uint64_t>(l - 1) * kExactMaxPatterns + p) * 2 + 1];
  }
  a.host_out[kPcHostCount + p] = total;
  a.host_out[kPcHostBounds + 2 * p] = first;
  a.host_out[kPcHostBounds + 2 * p + 1] = last;
  a.acc[kPcBounds + 2 * p] = first;       // the device copy
  a.acc[kPcBounds + 2 * p + 1] = last;
  a.acc[kPcTotals + p] = total;
}

__global__ void bounds_rows_counts_kernel(BoundsParams a, const unsigned long long* acc, int64_t offset, int first_round, int64_t* rows) {
  const int p = threadIdx.x;
  if (p >= a.n_lists) return;
  uint64_t n = a.count[p];
  int64_t fb = -1, fe = -1, lb = -1, le = -1;
  if (a.spans[p]) {
    const uint64_t* r = a.spans[p];
    if (n) {
      fb = static_cast<int64_t>(r[0]) + offset;
      fe = static_cast<int64_t>(r[1]) + offset;
      lb = static_cast<int64_t>(r[2 * (n - 1)]) + offset;
      le = static_cast<int64_t>(r[2 * (n - 1) + 1]) + offset;
    }
  } else {
    n = acc[kPcTotals + p];
    if (n) {
      const unsigned long long f = acc[kPcBounds + 2 * p], l = acc[kPcBounds + 2 * p + 1];
      const unsigned long long at = (1ull << kPcLenShift) - 1;
      fb = static_cast<int64_t>(f & at) + offset;
      fe = fb + static_cast<int64_t>(f >> kPcLenShift);
      lb = static_cast<int64_t>(l & at) + offset;
      le = lb + static_cast<int64_t>(l >> kPcLenShift);
    }
  }
  rows[8 * p + 0] = static_cast<int64_t>(n);
  if (first_round) {
    rows[8 * p + 1] = fb;
    rows[8 * p + 2] = fe;
    rows[8 * p + 5] = rows[8 * p + 6] = rows[8 * p + 7] = 0;
  }
  rows[8 * p + 3] = lb;
  rows[8 * p + 4] = le;
}

void launch_bounds_rows_counts(const BoundsParams& a, const unsigned long long* acc, int64_t offset, int first_round, int64_t* d_rows, hipStream_t st) {
  hipLaunchKernelGGL(bounds_rows_counts_kernel, dim3(1), dim3(64), 0, st, a, acc, offset, first_round, d_rows);
}

void launch_plane_count(const PlaneCountParams& a, int grid, hipEvent_t t0, hipEvent_t t1, hipStream_t st) {
  static const size_t pad = getenv("RJ_COUNT_LDS_PAD") ? static_cast<size_t>(atoi(getenv("RJ_COUNT_LDS_PAD"))) : 0;   // measurement: fewer workgroups per CU
  if (a.n_bases <= 1) hipExtLaunchKernelGGL((plane_count<ExactShape<1>>), dim3(grid), dim3(256), pad, st, t0, t1, 0, a);
  else hipExtLaunchKernelGGL((plane_count<ExactShape<2>>), dim3(grid), dim3(256), pad, st, t0, t1, 0, a);
}

void launch_plane_list(const PlaneListParams& a, int grid, hipEvent_t t0, hipEvent_t t1, hipStream_t st) {
  if (a.c.n_bases <= 1) hipExtLaunchKernelGGL((plane_count<ListShape<1>>), dim3(grid), dim3(256), 0, st, t0, t1, 0, a);
  else hipExtLaunchKernelGGL((plane_count<ListShape<2>>), dim3(grid), dim3(256), 0, st, t0, t1, 0, a);
}

void launch_plane_list_general(const PlaneListGParams& a, int grid, hipEvent_t t0, hipEvent_t t1, hipStream_t st) {
  if (a.g.tolerance) hipExtLaunchKernelGGL((plane_count<GeneralListShape<true>>), dim3(grid), dim3(256), 0, st, t0, t1, 0, a);
  else hipExtLaunchKernelGGL((plane_count<GeneralListShape<false>>), dim3(grid), dim3(256), 0, st, t0, t1, 0, a);
}

// max_words / max_short: the largest n_words / short_max among the patterns (the instantiation)
void launch_plane_count_general(const PlaneCountGParams& g, int max_words, uint32_t max_short, int grid, hipEvent_t t0, hipEvent_t t1, hipStream_t st) {
  const size_t lds = static_cast<size_t>(g.c.table_words) * sizeof(uint32_t);
  const dim3 gr(grid), b(256);
  if (g.tolerance) {
    if (max_words <= 1 && max_short <= 8) hipExtLaunchKernelGGL((plane_count<GeneralShape<1, 8, true>>), gr, b, lds, st, t0, t1, 0, g);
    else if (max_words <= 1) hipExtLaunchKernelGGL((plane_count<GeneralShape<1, 16, true>>), gr, b, lds, st, t0, t1, 0, g);
    else hipExtLaunchKernelGGL((plane_count<GeneralShape<2, 16, true>>), gr, b, lds, st, t0, t1, 0, g);
  } else {
    if (max_words <= 1 && max_short <= 8) hipExtLaunchKernelGGL((plane_count<GeneralShape<1, 8, false>>), gr, b, lds, st, t0, t1, 0, g);
    else if (max_words <= 1) hipExtLaunchKernelGGL((plane_count<GeneralShape<1, 16, false>>), gr, b, lds, st, t0, t1, 0, g);
    else hipExtLaunchKernelGGL((plane_count<GeneralShape<2, 16, false>>), gr, b, lds, st, t0, t1, 0, g);
  }
}

void launch_plane_count_finish(const PlaneCountParams& a, int grid, hipStream_t st) {
  hipLaunchKernelGGL(plane_count_finish, dim3(kPcFinishGroups), dim3(256), 0, st, a, static_cast<uint32_t>(grid));
}

}  // namespace rejit_amd
