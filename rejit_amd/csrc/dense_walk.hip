// rejit_amd/csrc/dense_walk.hip -- dense mode with a lane-sized automaton in ONE kernel (split out of kernels.hip in round 6):
// scan_dense_walk<NW, CTX, PD> (the no-fast-forward seeding + NFA loop of the reference, src/x64/codegen-x64.cc:535-677).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "behind_walk.h"
#include "dense_swar.h"
#include "device_program.h"
#include "kernel_util.h"
#include "kernels.h"

namespace rejit_amd {

// ---------------------------------------------------------------------------------------
// Dense mode, lane-sized automaton, everything in one kernel: find the candidate starts of a
// 1-KiB chunk, walk them, keep only the starts at which something matched.
//
// The list-based dense pipeline (scan_dense -> region_offsets -> verify -> mark/scan/compact)
// moves ~100 bytes of list traffic per candidate START; `[A-Z][a-z]+ [A-Z][a-z]+` over random
// ASCII has a start at 35% of the bytes, so 1 GB of text cost 35 GB of traffic (13.6 ms) for
// zero matches.  Here a start that does not match leaves no trace in HBM:
//   1. each lane tests its 16 bytes (first-byte bitmap / nullable contexts) -> 16-bit mask;
//      a wave scan ranks the chunk's candidates and their in-chunk offsets go to an LDS list;
//   2. persistent walkers (see verify_walkers) run the automaton from every listed start and
//      put the match length back into the candidate's LDS slot;
//   3. the slots are compacted in order and the survivors appended to the wave's region as
//      (begin -> region, end -> region_ends).
// Downstream is the windows pipeline's offsets_gather_check.  Tables and the next text byte as
// in verify_walkers.
// NW = 32-bit words of automaton state per lane (1, 2 or 4); CTX = the pattern has ^ / $.
// Positions inside the kernel are 32-bit offsets from the chunk base.
constexpr int kHalo = 64;                    // bytes after the chunk kept in LDS for the walkers
constexpr int kTextWindow = kChunk + kHalo;  // per wave

// PD = depth of the lane-packed pre-steps (dense_swar.h), 0: the pattern does not qualify.
// (amdgpu_waves_per_eu(6, 8): 80 instead of 90 VGPRs, six waves per SIMD instead of five -- measured 6 % on
// `[a-f]+[0-9]`, 10 % on `[@#]`; seven waves need scratch and are slower again.)
template <int NW, bool CTX, int PD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void scan_dense_walk(ScanParams a, DevProgram P, uint64_t* region_ends,
                                                       unsigned long long* counters) {
  extern __shared__ uint32_t tab[];
  __shared__ uint32_t fb[8];  // first-byte bitmap (indexed by data: LDS, not registers)
  if (threadIdx.x < 8) fb[threadIdx.x] = P.first_bytes[threadIdx.x];
  const int W = P.n_words, C = P.n_ctx, NP = P.n_pos > 0 ? P.n_pos : 1;
  const int o_last = C * W, o_lin = 2 * C * W, o_rowof = o_lin + W, o_rows = o_rowof + NP, o_cls = o_rows + C * P.n_rows * W;
  for (uint32_t i = threadIdx.x; i < P.table_words; i += blockDim.x) tab[i] = P.first[i];
  __syncthreads();
  const int lane = lane_id();
  // candidate slots of this wave: bits 0..9 offset inside the chunk, bits 10.. match length + 1
  uint32_t* slot = tab + ((P.table_words + 3u) & ~3u) + (threadIdx.x >> 6) * kChunk;
  // the chunk's text for the walkers (+ kHalo bytes after it): a global byte load per step made
  // every step wait a full memory latency -- the one-ahead prefetch cannot be waited for
  // separately (vmcnt counts in order) -- and the kernel was bound by exactly that
  uint8_t* txt = reinterpret_cast<uint8_t*>(tab + ((P.table_words + 3u) & ~3u) + 4 * kChunk) + (threadIdx.x >> 6) * kTextWindow;
  const uint64_t wave = scalar_wave_index();
  uint64_t* region = a.hits + wave * a.region_cap;
  uint64_t* ends = region_ends + wave * a.region_cap;
  uint32_t count = 0;  // survivors of this wave so far (wave-uniform)
  const uint64_t first_chunk = a.sb / kChunk;
  const uint64_t end_chunk = (a.se + kChunk - 1) / kChunk;  // se <= n + 1
  const WaveSpan span = wave_span(a, wave, first_chunk, end_chunk);
  // per-word constants: linear / loop / skip masks; first and last rows of context 0 (all there is
  // without assertions)
  uint32_t step1[NW], loopm[NW], skipm[NW], first0[NW], last0[NW];
#pragma unroll
  for (int q = 0; q < NW; q++) {
    const bool in = q < W;
    loopm[q] = in ? P.loop_mask[q] : 0u;
    skipm[q] = in ? P.skip_mask[q] : 0u;
    step1[q] = (in ? tab[o_lin + q] : 0u) | loopm[q] | skipm[q];  // positions that pass to i + 1
    first0[q] = in ? tab[q] : 0u;
    last0[q] = in ? tab[o_last + q] : 0u;
  }
  // contexts in which a non-empty match can start at all (bit c: first[c] is not empty)
  uint32_t first_ctx = 0;
  for (int c = 0; c < C; c++)
    for (int q = 0; q < W; q++)
      if (tab[c * W + q] != 0) first_ctx |= 1u << c;
  if (C == 1) first_ctx = 0xFu;

  // The lane's 16 bytes of the NEXT chunk and the run's "void" flag are loaded one iteration ahead: a load
  // at the top of the iteration that needs it put a full memory latency (two, with the flag) on every
  // wave's critical path.
  uint4 pre_v = make_uint4(0, 0, 0, 0);
  unsigned long long pre_flag = 0;
  bool pre_valid = false;  // (wave-uniform)
  // packed pre-steps of `X+...`: is the last byte of the span's chunk tail_it - 1 in X (both wave-uniform)
  uint32_t tail_in_x = 0;
  uint32_t tail_it = ~0u;
  // (a relaxed atomic load at device scope: fresh data, but -- unlike a volatile access -- nothing to wait for
  // until the value is used, an iteration later)
  auto load_flag = [&]() { return __hip_atomic_load(counters + kCntOverrun, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  // The loop's wave-uniform tests as 32-bit chunk counts relative to the span: a 64-bit ordering of two uniform
  // values has no scalar instruction -- the compiler copies both to vector registers and compares there, three VALU
  // instructions apiece in a kernel that is bound by exactly those.
  const uint32_t n_iter = static_cast<uint32_t>(span.c1 - span.c0);
  auto rel = [&](uint64_t chunk) -> uint32_t {
    return chunk <= span.c0 ? 0u : (chunk - span.c0 >= n_iter ? n_iter : static_cast<uint32_t>(chunk - span.c0));
  };
  const uint64_t own_lim = a.se < a.n ? a.se : a.n;
  const uint32_t full_until = rel(a.n / kChunk);                       // [0, full_until): base + kChunk <= n
  const uint32_t plus4_until = rel(a.n >= 4 ? (a.n - 4) / kChunk : 0);  // base + kChunk + 4 <= n
  const uint32_t clip_below = rel((a.sb + kChunk - 1) / kChunk);       // base < sb
  const uint32_t clip_from = rel(own_lim / kChunk);                    // base + kChunk > min(se, n)
  for (uint32_t it = 0; it < n_iter; it++) {
    const uint64_t c = span.c0 + it;
    const uint64_t base = c * kChunk;
    const uint64_t at = base + static_cast<uint64_t>(lane) * 16;
    const uint8_t* tbase = a.text + base;
    const bool full = it < full_until;
    uint4 v = pre_v;
    unsigned long long stop = pre_flag;
    if (!pre_valid) {
      stop = load_flag();
      if (full) v = *reinterpret_cast<const uint4*>(a.text + at);
    }
    pre_valid = it + 1 < full_until;  // (the next chunk belongs to the span and lies inside the text)
    if (pre_valid) {
      pre_v = *reinterpret_cast<const uint4*>(a.text + at + kChunk);
      pre_flag = load_flag();
    }
    // some walk of this run has hit P.max_walk: the run is void (the engine repeats it on the carry
    // scan), no point in finishing it
    if (stop != 0) break;
    // text length as seen from the chunk (a walk is cut at 2^20 bytes, so clamping is exact)
    const uint32_t n_rel = a.n - base < 0x7FFFFFFFull ? static_cast<uint32_t>(a.n - base) : 0x7FFFFFFFu;
    // ---- 1. candidate mask of the lane's 16 positions
    uint32_t d[6];
    if (full) {
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    } else {
      load_guarded(a.text, a.n, at, d);
    }
    uint32_t cand = 0;
    uint32_t fin = 0, flen = 0;  // starts already decided by the pre-steps, 2 bits of length each
    uint32_t Hs[4] = {0, 0, 0, 0};  // packed pre-steps: per start the lengths that matched (dense_swar.h)
    const bool packed = PD > 0 && it < plus4_until;  // (wave-uniform)
    if (packed) {
      // Pre-steps, four starts per register (dense_swar.h): class rows by byte-parallel range tests, the
      // first `depth` automaton steps of all 16 starts, no lookups and no divergence.  A start that is
      // dead after depth + 1 bytes is decided here; the others go to the walkers.
      uint32_t x[5] = {d[0], d[1], d[2], d[3], wave_from_lane_above(d[0])};
      if (lane == kWave - 1) x[4] = *reinterpret_cast<const uint32_t*>(a.text + base + kChunk);  // (in the text: see `packed`)
      uint32_t rows[5], walk, matched, in_x;
      rj_swar_rows5(P.swar, x, rows);
      // (masks in F layout from here on: bit 8k + g = the lane's start 4g + k, dense_swar.h)
      if (P.loop_first) rj_swar_presteps<(PD > 0 ? PD : 1), true>(P.swar, rows, &walk, &matched, Hs, &in_x);
      else rj_swar_presteps<(PD > 0 ? PD : 1), false>(P.swar, rows, &walk, &matched, Hs, &in_x);
      fin = matched & ~walk;
      cand = walk | fin;
      const uint64_t lim = own_lim;
      if (it < clip_below || it >= clip_from) {  // the chunks at the ends of the own range (a scalar test)
        const uint32_t hi = lim > at ? (lim - at < 16 ? static_cast<uint32_t>(lim - at) : 16u) : 0u;
        const uint32_t lo = a.sb > at ? (a.sb - at < 16 ? static_cast<uint32_t>(a.sb - at) : 16u) : 0u;
        const uint32_t range = rj_swar_f_from_starts(((1u << hi) - 1u) & ~((1u << lo) - 1u));
        fin &= range;
        cand &= range;
      }
      if (P.loop_first) {
        // `X+...`: a start whose previous byte is in X too is never selected (see DevProgram::loop_first).  The byte
        // before the chunk: the last lane's flag of the chunk before, when this wave has just been through it --
        // else one byte from the text
        uint32_t prev_in = wave_from_lane_below(rj_swar_f_last(in_x));
        uint32_t before = tail_in_x;
        if (tail_it != it) before = base > 0 ? ((fb[a.text[base - 1] >> 5] >> (a.text[base - 1] & 31)) & 1u) : 0u;
        if (lane == 0) prev_in = before;
        cand &= ~rj_swar_f_next(in_x, prev_in);
        tail_in_x = wave_last_lane(rj_swar_f_last(in_x));
        tail_it = it + 1;
      }
    } else if (PD == 0 && NW == 1 && !CTX && P.nullable == 0 && it < plus4_until) {
      // Pre-steps: the first kPre automaton steps of ALL 16 starts of the lane, in registers, with
      // no divergence.  Most starts die within a few bytes (a walk on random text is ~1.5 steps
      // long), and those never reach the walkers: a start that is dead after kPre + 1 bytes is
      // decided here (its longest match, if any, has length <= kPre).  Positions with a general
      // follow row are not stepped here: a state that holds one keeps the start for the walkers.
      constexpr int kPre = 2;
      uint32_t r[16 + kPre];  // class rows of the lane's bytes and of the kPre bytes after them
      const uint32_t nx = wave_from_lane_above(d[0]);
#pragma unroll
      for (int k = 0; k < 16 + kPre; k++) {
        const uint32_t byte = ((k < 16 ? d[k >> 2] : nx) >> (8 * (k & 3))) & 0xFFu;
        r[k] = tab[o_cls + byte];
      }
      // (all flags as 0/1 integers: comparisons would go through the scalar unit)
      auto nz = [](uint32_t x) -> uint32_t { return x < 1u ? x : 1u; };
      uint32_t walk = 0;
#pragma unroll
      for (int j = 0; j < 16; j++) {
        uint32_t S = first0[0] & r[j];
        uint32_t f = 0, gen = 0;
#pragma unroll
        for (int t = 1; t <= kPre; t++) {
          const uint32_t hit = nz(S & last0[0]);
          f = hit * t > f ? hit * t : f;
          gen |= S & ~step1[0];
          S = ((((S & step1[0]) << 1) | ((S & skipm[0]) << 2) | (S & loopm[0]))) & r[j + t];
        }
        const uint32_t alive = nz(S | gen);  // alive or decided, a start has a first byte
        walk |= alive << j;
        fin |= (nz(f) & (alive ^ 1u)) << j;
        flen |= f << (2 * j);
      }
      if (lane == kWave - 1) {  // no neighbour: the rows past the lane's bytes are not valid
        constexpr uint32_t tail = ((1u << kPre) - 1u) << (16 - kPre);
        uint32_t starts = 0;
#pragma unroll
        for (int j = 16 - kPre; j < 16; j++) starts |= static_cast<uint32_t>((first0[0] & r[j]) != 0) << j;
        walk = (walk & ~tail) | starts;
        fin &= ~tail;
      }
      const uint64_t lim = a.se < a.n ? a.se : a.n;
      const uint32_t hi = lim > at ? (lim - at < 16 ? static_cast<uint32_t>(lim - at) : 16u) : 0u;
      const uint32_t lo = a.sb > at ? (a.sb - at < 16 ? static_cast<uint32_t>(a.sb - at) : 16u) : 0u;
      const uint32_t range = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
      fin &= range;
      cand = (walk & range) | fin;
      if (P.loop_first) {
        // `X+...`: a start whose previous byte is in X too is never selected (see DevProgram::loop_first)
        uint32_t in_x = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) in_x |= static_cast<uint32_t>((first0[0] & r[j]) != 0) << j;
        uint32_t prev_in = wave_from_lane_below(in_x >> 15);
        if (lane == 0) prev_in = base > 0 ? ((fb[a.text[base - 1] >> 5] >> (a.text[base - 1] & 31)) & 1u) : 0u;
        cand &= ~((in_x << 1) | prev_in);
      }
    } else {
      // General form.  A position starts a candidate when its byte can begin a match in the
      // position's context, or (nullable patterns: x*, ^, $, ...) when the empty string matches
      // there.  Contexts (bit0: start of line, bit1: end of line) of all 16 positions come from one
      // line-break bitmask of the lane's bytes: sol = that mask shifted by one with the neighbour's
      // last byte shifted in, eol = the mask itself plus the end of the text.  Patterns that begin
      // with an assertion (`^[a-z]+:`) have an EMPTY first set outside their context: without the
      // context filter every [a-z] byte of the text was a candidate for the walkers.
      uint32_t lb = 0, first16 = 0;
      if (CTX) {
        // line breaks of the 16 bytes, four bytes per operation (a byte is \n or \r when one of the two
        // differences is zero)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint32_t both = rj_swar_nz(d[q] ^ 0x0a0a0a0au) & rj_swar_nz(d[q] ^ 0x0d0d0d0du);
          lb |= rj_swar_movemask(both ^ 0x80808080u) << (4 * q);
        }
      }
      if (P.n_pos != 0) {  // (only assertions: no byte begins a match)
#pragma unroll
        for (int j = 0; j < 16; j++) {
          const uint32_t cur = (d[j >> 2] >> (8 * (j & 3))) & 0xFFu;
          first16 |= ((fb[cur >> 5] >> (cur & 31)) & 1u) << j;
        }
      }
      // positions with a byte (s < n) / positions at all (s <= n)
      const uint32_t lt_n = a.n > at ? (a.n - at < 16 ? (1u << (a.n - at)) - 1u : 0xFFFFu) : 0u;
      const uint32_t le_n = a.n >= at ? (a.n - at < 15 ? (2u << (a.n - at)) - 1u : 0xFFFFu) : 0u;
      uint32_t null16 = P.nullable ? 0xFFFFu : 0u, allowed16 = 0xFFFFu;
      if (CTX) {
        lb &= lt_n;
        uint32_t prev_lb = wave_from_lane_below(lb >> 15);  // the neighbour's last byte
        if (lane == 0) prev_lb = base > 0 ? static_cast<uint32_t>(rj_line_break(a.text[base - 1])) : 1u;  // text start
        const uint32_t sol = ((lb << 1) | prev_lb) & 0xFFFFu;
        uint32_t eol = lb;
        if (a.n >= at && a.n - at < 16) eol |= 1u << (a.n - at);  // the end of the text
        const uint32_t in_ctx[4] = {~sol & ~eol, sol & ~eol, ~sol & eol, sol & eol};
        null16 = 0;
        allowed16 = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
          if ((P.nullable >> c) & 1u) null16 |= in_ctx[c];
          if ((first_ctx >> c) & 1u) allowed16 |= in_ctx[c];
        }
      }
      const uint32_t hi = a.se > at ? (a.se - at < 16 ? static_cast<uint32_t>(a.se - at) : 16u) : 0u;
      const uint32_t lo = a.sb > at ? (a.sb - at < 16 ? static_cast<uint32_t>(a.sb - at) : 16u) : 0u;
      const uint32_t range = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
      cand = ((first16 & allowed16 & lt_n) | (null16 & le_n)) & 0xFFFFu & range;
      if (P.loop_first) {  // (no assertions, not nullable: first16 is membership in X)
        const uint32_t in_x = first16 & lt_n;
        uint32_t prev_in = wave_from_lane_below(in_x >> 15);
        if (lane == 0) prev_in = base > 0 ? ((fb[a.text[base - 1] >> 5] >> (a.text[base - 1] & 31)) & 1u) : 0u;
        cand &= ~((in_x << 1) | prev_in);
      }
      if (P.n_pos == 0) {
        // only assertions (^, $, ^$): every candidate IS a match, the empty one -- no walk at all
        // (the line table of a grep-like caller is a MatchAll of "^", sample/jrep.cc:294)
        fin = cand;
        flen = 0;
      }
    }
    if (__ballot(cand != 0) == 0) continue;
    if ((packed || P.n_pos == 0) && __ballot((cand & ~fin) != 0) == 0) {
      // Every candidate of the chunk is decided already (the common chunk of `[a-f]+[0-9]`, every chunk of
      // `^`): no walkers, so no text window, no slots -- the lanes put their matches straight into the region
      const uint32_t mine = __popc(cand);
      const uint32_t inc = wave_inclusive_sum(mine);
      uint32_t pos = count + inc - mine;
      if (packed) {
#pragma unroll
        for (int g = 0; g < 4; g++) {  // (text order: group by group, byte by byte)
          uint32_t m = (cand >> g) & 0x01010101u;
          while (m) {
            const int b = __ffs(static_cast<int>(m)) - 1;  // 8k
            m &= m - 1;
            const uint32_t hb = (Hs[g] >> b) & 0xFu;
            if (pos < a.region_cap) {
              const uint64_t s = at + static_cast<uint64_t>(4 * g + (b >> 3));
              region[pos] = s;
              ends[pos] = s + (32u - static_cast<uint32_t>(__clz(static_cast<int>(hb))));
            }
            pos++;
          }
        }
      } else {
        uint32_t m = cand;
        while (m) {  // (only assertions: the empty match)
          const int j = __ffs(static_cast<int>(m)) - 1;
          m &= m - 1;
          if (pos < a.region_cap) {
            region[pos] = at + static_cast<uint64_t>(j);
            ends[pos] = at + static_cast<uint64_t>(j);
          }
          pos++;
        }
      }
      count += wave_last_lane(inc);
      continue;
    }
    *reinterpret_cast<uint4*>(txt + lane * 16) = make_uint4(d[0], d[1], d[2], d[3]);
    if (lane < kHalo / 4) {
      const uint64_t hp = base + kChunk + 4 * lane;
      uint32_t hv = 0;
      if (hp + 4 <= a.n) {
        hv = *reinterpret_cast<const uint32_t*>(a.text + hp);
      } else {
        for (int q = 0; q < 4; q++)
          if (hp + q < a.n) hv |= static_cast<uint32_t>(a.text[hp + q]) << (8 * q);
      }
      *reinterpret_cast<uint32_t*>(txt + kChunk + 4 * lane) = hv;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    // byte at chunk offset x < n_rel
    auto tb = [&](uint32_t x) -> uint32_t { return x < kChunk + kHalo ? txt[x] : tbase[x]; };
    const uint32_t mine = __popc(cand);
    const uint32_t inc = wave_inclusive_sum(mine);
    const uint32_t total = wave_last_lane(inc);
    if (packed) {
      uint32_t idx = inc - mine;
#pragma unroll
      for (int g = 0; g < 4; g++) {  // (text order: group by group, byte by byte; Hs[] stays in registers)
        uint32_t m = (cand >> g) & 0x01010101u;
        while (m) {
          const int b = __ffs(static_cast<int>(m)) - 1;  // 8k
          m &= m - 1;
          const int j = 4 * g + (b >> 3);
          // decided starts carry their length already (bits 10..: length + 1): the longest prefix that matched
          const uint32_t hb = (Hs[g] >> b) & 0xFu;
          const uint32_t len1 = (fin >> (b + g)) & 1u ? (32u - static_cast<uint32_t>(__clz(static_cast<int>(hb)))) + 1u : 0u;
          slot[idx++] = static_cast<uint32_t>(lane * 16 + j) | (len1 << 10);
        }
      }
    } else {
      uint32_t idx = inc - mine, m = cand;
      while (m) {
        const int j = __ffs(static_cast<int>(m)) - 1;
        m &= m - 1;
        // decided starts carry their length already (bits 10..: length + 1), the others go to the walkers
        const uint32_t len1 = (fin >> j) & 1u ? ((flen >> (2 * j)) & 3u) + 1u : 0u;
        slot[idx++] = static_cast<uint32_t>(lane * 16 + j) | (len1 << 10);
      }
    }
    // ---- 2. persistent walkers over slot[0 .. total)
    {
      uint32_t cursor = 0;
      bool active = false, found = false;
      uint32_t my_k = 0, prevb = 0, curb = 0;
      uint32_t s = 0, p = 0, e = 0, S[NW];  // offsets from the chunk base
#pragma unroll
      for (int q = 0; q < NW; q++) S[q] = 0;
      for (;;) {
        const uint64_t idle = __ballot(!active);
        if (idle != 0 && cursor < total) {
          const uint32_t k = cursor + __popcll(idle & ((1ull << lane) - 1ull));
          const uint32_t entry = !active && k < total ? slot[k] : 1u << 10;
          if ((entry >> 10) == 0) {  // (an entry decided by the pre-steps is left as it is)
            my_k = k;
            s = entry;
            active = true;
            found = false;
            e = 0;
            curb = s < n_rel ? tb(s) : '\n';
            int ctx = 0;
            if (CTX) {
              prevb = (base + s) > 0 ? *(tbase + s - 1) : '\n';  // s <= n here
              if (base + s == 0 || rj_line_break(prevb)) ctx |= 1;
              if (s == n_rel || rj_line_break(curb)) ctx |= 2;
            }
            if ((P.nullable >> ctx) & 1u) {
              found = true;
              e = s;
            }
            const bool can_start = s < n_rel && P.n_pos != 0;
            const int crow = o_cls + static_cast<int>(curb) * W;
#pragma unroll
            for (int q = 0; q < NW; q++) {
              const uint32_t f = CTX ? (q < W ? tab[ctx * W + q] : 0u) : first0[q];
              S[q] = can_start && q < W ? (f & tab[crow + q]) : 0u;
            }
            p = s + 1;
            prevb = curb;
            curb = p < n_rel ? tb(p) : '\n';
          }
          cursor += __popcll(idle);
          if (cursor > total) cursor = total;
        }
        if (__ballot(active) == 0) {
          if (cursor >= total) break;
          continue;  // a whole round of entries was already decided by the pre-steps: hand out the next
        }
        if (active) {
          uint32_t alive = 0;
#pragma unroll
          for (int q = 0; q < NW; q++) alive |= S[q];
          bool done = alive == 0;
          if (!done) {
            int ctx = 0;
            if (CTX) {
              if (rj_line_break(prevb)) ctx |= 1;  // p >= 1 here
              if (p == n_rel || rj_line_break(curb)) ctx |= 2;
            }
            uint32_t acc = 0;
#pragma unroll
            for (int q = 0; q < NW; q++) acc |= S[q] & (CTX ? (q < W ? tab[o_last + ctx * W + q] : 0u) : last0[q]);
            if (acc) {
              found = true;
              e = p;
            }
            if (p == n_rel) {
              done = true;
            } else if (p - s >= P.max_walk) {
              counters[kCntOverrun] = 1;
              done = true;
            } else {
              const uint32_t nextb = p + 1 < n_rel ? tb(p + 1) : '\n';
              uint32_t T[NW];
              uint32_t c1 = 0, c2 = 0;
#pragma unroll
              for (int q = 0; q < NW; q++) {
                const uint32_t x = S[q] & step1[q], y = S[q] & skipm[q];
                T[q] = (x << 1) | c1 | (y << 2) | c2 | (S[q] & loopm[q]);
                c1 = x >> 31;
                c2 = y >> 30;
              }
#pragma unroll
              for (int q = 0; q < NW; q++) {
                uint32_t sp = S[q] & ~step1[q];  // positions with a general follow set: OR their rows in
                while (sp) {
                  const int b = __ffs(static_cast<int>(sp)) - 1;
                  sp &= sp - 1;
                  const int row = o_rows + (ctx * P.n_rows + static_cast<int>(tab[o_rowof + q * 32 + b])) * W;
#pragma unroll
                  for (int j = 0; j < NW; j++)
                    if (j < W) T[j] |= tab[row + j];
                }
              }
              const int crow = o_cls + static_cast<int>(curb) * W;
#pragma unroll
              for (int q = 0; q < NW; q++) S[q] = q < W ? (T[q] & tab[crow + q]) : 0u;
              p++;
              prevb = curb;
              curb = nextb;
            }
          }
          if (done) {
            // length + 1 in bits 10..31 (a walk is cut at kMaxSimSteps = 2^20 bytes), 0 = no match
            slot[my_k] = s | (found ? (e - s + 1) << 10 : 0u);
            active = false;
          }
        }
      }
    }
    // ---- 3. ordered compaction into the region
    for (uint32_t kb = 0; kb < total; kb += kWave) {
      const uint32_t k = kb + lane;
      const uint32_t v = k < total ? slot[k] : 0u;
      const bool keep = (v >> 10) != 0;
      const uint64_t kept = __ballot(keep);
      const uint32_t pos = count + __popcll(kept & ((1ull << lane) - 1ull));
      if (keep && pos < a.region_cap) {
        const uint64_t s = base + (v & 1023u);
        region[pos] = s;
        ends[pos] = s + (v >> 10) - 1;
      }
      count += __popcll(kept);
    }
  }
  if (lane == 0) {
    if (count > a.region_cap) {  // the host grows the regions and runs again
      counters[kCntOverflow] = 1;
      atomicMax(&counters[kCntMaxRegion], static_cast<unsigned long long>(count));
    }
    a.hit_counts[wave] = count < a.region_cap ? count : a.region_cap;
  }
}


bool dense_walk_fits(const DevProgram& P) { return P.n_words <= 4 && P.table_words <= 8192; }

void launch_scan_dense_walk(const ScanParams& a, const DevProgram& P, int grid, uint64_t* region_ends,
                            unsigned long long* counters, hipEvent_t t0, hipEvent_t t1, hipStream_t st) {
  const size_t lds = (((static_cast<size_t>(P.table_words) + 3) & ~size_t{3}) + 4 * kChunk) * sizeof(uint32_t) + 4 * kTextWindow;
  const dim3 g(grid), b(256);
  const bool ctx = P.n_ctx > 1;
  const int pd = (P.n_words <= 1 && !ctx && P.swar.n_ranges != 0) ? static_cast<int>(P.swar.depth) : 0;
  if (pd == 1) {
    hipExtLaunchKernelGGL((scan_dense_walk<1, false, 1>), g, b, lds, st, t0, t1, 0, a, P, region_ends, counters);
  } else if (pd == 2) {
    hipExtLaunchKernelGGL((scan_dense_walk<1, false, 2>), g, b, lds, st, t0, t1, 0, a, P, region_ends, counters);
  } else if (pd == 4) {
    hipExtLaunchKernelGGL((scan_dense_walk<1, false, 4>), g, b, lds, st, t0, t1, 0, a, P, region_ends, counters);
  } else if (P.n_words <= 1) {
    if (ctx) hipExtLaunchKernelGGL((scan_dense_walk<1, true, 0>), g, b, lds, st, t0, t1, 0, a, P, region_ends, counters);
    else hipExtLaunchKernelGGL((scan_dense_walk<1, false, 0>), g, b, lds, st, t0, t1, 0, a, P, region_ends, counters);
  } else if (P.n_words == 2) {
    if (ctx) hipExtLaunchKernelGGL((scan_dense_walk<2, true, 0>), g, b, lds, st, t0, t1, 0, a, P, region_ends, counters);
    else hipExtLaunchKernelGGL((scan_dense_walk<2, false, 0>), g, b, lds, st, t0, t1, 0, a, P, region_ends, counters);
  } else {
    if (ctx) hipExtLaunchKernelGGL((scan_dense_walk<4, true, 0>), g, b, lds, st, t0, t1, 0, a, P, region_ends, counters);
    else hipExtLaunchKernelGGL((scan_dense_walk<4, false, 0>), g, b, lds, st, t0, t1, 0, a, P, region_ends, counters);
  }
}
}  // namespace rejit_amd
