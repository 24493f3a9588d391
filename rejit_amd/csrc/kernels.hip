// rejit_amd/csrc/kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the hot path.
//
// What the reference JITs per pattern (src/x64/codegen-x64.cc) is here a fixed family of
// kernels that interpret the lowered program (device_program.h):
//
//   scan_windows<K>   the "fast-forward" scan (FastForwardGen::VisitSingleMultipleChar /
//                     ::Generate multi-literal branch, codegen-x64.cc:1102-1403):
//                     streams the text from HBM with one coalesced 16-byte load per lane
//                     (1 KiB per wave instruction), builds the 16 unaligned 4-byte windows of
//                     the lane with v_alignbyte_b32 and compares them with the <=8 window
//                     constants held in SGPRs; the per-lane results live as 64-bit lane masks
//                     in SGPRs (v_cmp + s_or), so the streaming loop has no divergent code.
//                     Only when the wave-wide mask is non-zero are the hit offsets appended to
//                     the hit list (one wave-aggregated atomic per 64 hits).
//   scan_dense        the no-fast-forward case (GenerateMatchDirection seeding every
//                     position, codegen-x64.cc:544-554): every byte is tested against the
//                     256-bit first-byte set / the nullable contexts; survivors go to the
//                     hit list.
//   verify_lane<NQ>   the NFA inner loop (GenerateMatchDirection + GenerateTransitions,
//                     codegen-x64.cc:535-677): one candidate start per lane, automaton state
//                     in NQ 64-bit registers, longest match from that start.
//   verify_wave       same for automata of more than 128 positions: one candidate per wave,
//                     the state vector spread over the 64 lanes.
//   match_full        kMatchFull: one wave walks the whole text (codegen-x64.cc:162-164).
//   finalize_small    MatchAllAppendFilter + the non-overlap rule (src/codegen.cc:36-86,
//                     codegen-x64.cc:448-460) for <= kFinalizeCap candidates: LDS bitonic
//                     sort by begin, duplicate removal, left-most-longest selection.
//   select_sorted     the same selection over an already sorted candidate list of any size.
//
// No MFMA: there is no contraction anywhere on this path (integer compares on a byte stream);
// the roofline is HBM read bandwidth.
#include <cstring>

#include <hip/hip_runtime.h>

#include "device_program.h"
#include "kernels.h"

namespace rejit_amd {

namespace {

constexpr int kWave = 64;
constexpr int kChunk = 1024;  // bytes per wave iteration: 64 lanes x 16 B

__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x) & (kWave - 1); }

// Append `mine` (valid when pred) to list[] with one atomic per wave; returns nothing.
__device__ __forceinline__ void wave_append(bool pred, uint64_t mine, uint64_t* list, uint64_t cap,
                                            unsigned long long* counter) {
  const unsigned long long m = __ballot(pred);
  if (m == 0) return;
  const int lane = lane_id();
  const int leader = __ffsll(static_cast<long long>(m)) - 1;
  unsigned long long base = 0;
  if (lane == leader) base = atomicAdd(counter, static_cast<unsigned long long>(__popcll(m)));
  base = __shfl(base, leader);
  if (pred) {
    const unsigned long long idx = base + __popcll(m & ((1ull << lane) - 1ull));
    if (idx < cap) list[idx] = mine;
  }
}

__device__ __forceinline__ void wave_append_pair(bool pred, uint64_t b, uint64_t e, uint64_t* pairs,
                                                 uint64_t cap, unsigned long long* counter) {
  const unsigned long long m = __ballot(pred);
  if (m == 0) return;
  const int lane = lane_id();
  const int leader = __ffsll(static_cast<long long>(m)) - 1;
  unsigned long long base = 0;
  if (lane == leader) base = atomicAdd(counter, static_cast<unsigned long long>(__popcll(m)));
  base = __shfl(base, leader);
  if (pred) {
    const unsigned long long idx = base + __popcll(m & ((1ull << lane) - 1ull));
    if (idx < cap) {
      pairs[2 * idx] = b;
      pairs[2 * idx + 1] = e;
    }
  }
}

// 16 B of the lane + the 4 B that follow, guarded against the end of the text (tail chunk).
__device__ __forceinline__ void load_guarded(const uint8_t* text, uint64_t n, uint64_t at, uint32_t d[5]) {
#pragma unroll
  for (int q = 0; q < 5; q++) {
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint64_t p = at + 4 * q + k;
      if (p < n) v |= static_cast<uint32_t>(text[p]) << (8 * k);
    }
    d[q] = v;
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------
// Fast-forward window scan.
//
// Window position w is a hit iff (load32(text + w) & mask) == value[k] for some k < K; the
// candidate start is s = w - offset and must lie in [sb, se).  Scanned w range: [wlo, whi).
template <int K, bool MASKED>
__global__ __launch_bounds__(256) void scan_windows(ScanParams a, WindowSet ws) {
  const int lane = lane_id();
  const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_waves = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 6;
  const uint64_t first_chunk = a.wlo / kChunk;
  const uint64_t end_chunk = (a.whi + kChunk - 1) / kChunk;

  for (uint64_t c = first_chunk + wave; c < end_chunk; c += n_waves) {
    const uint64_t base = c * kChunk;
    const uint64_t at = base + static_cast<uint64_t>(lane) * 16;
    uint32_t d[5];
    if (base + kChunk + 4 <= a.n) {
      const uint4 v = *reinterpret_cast<const uint4*>(a.text + at);
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      d[4] = *reinterpret_cast<const uint32_t*>(a.text + at + 16);  // neighbour's first dword (L1 hit)
    } else {
      load_guarded(a.text, a.n, at, d);
    }
    // the 16 unaligned 4-byte windows of this lane
    uint32_t x[16];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      x[4 * q] = d[q];
      x[4 * q + 1] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], 1);
      x[4 * q + 2] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], 2);
      x[4 * q + 3] = __builtin_amdgcn_alignbyte(d[q + 1], d[q], 3);
    }
    bool any = false;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint32_t v = MASKED ? (x[j] & ws.mask) : x[j];
#pragma unroll
      for (int k = 0; k < K; k++) any |= (v == ws.value[k]);
    }
    if (__ballot(any) == 0) continue;  // wave-uniform: the common case leaves here

    // rare path: per-lane 16-bit hit mask, then append the hit offsets in position order
    uint32_t hm = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint32_t v = MASKED ? (x[j] & ws.mask) : x[j];
      bool hit = false;
#pragma unroll
      for (int k = 0; k < K; k++) hit |= (v == ws.value[k]);
      const uint64_t w = at + j;
      hit = hit && w >= a.wlo && w < a.whi;
      hm |= static_cast<uint32_t>(hit) << j;
    }
#pragma unroll 1
    for (int j = 0; j < 16; j++) {
      wave_append((hm >> j) & 1u, at + j - ws.offset, a.hits, a.hits_cap, a.counters + kCntHits);
    }
  }
}

// ---------------------------------------------------------------------------------------
// Dense scan: every position s in [sb, se) that can start a match goes to the hit list.
__global__ __launch_bounds__(256) void scan_dense(ScanParams a, DevProgram P) {
  __shared__ uint32_t fb[8];
  if (threadIdx.x < 8) fb[threadIdx.x] = P.first_bytes[threadIdx.x];
  __syncthreads();
  const int lane = lane_id();
  const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_waves = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 6;
  const uint64_t first_chunk = a.sb / kChunk;
  const uint64_t end_chunk = (a.se + kChunk - 1) / kChunk;  // se <= n + 1
  const bool ctxed = P.n_ctx > 1;

  for (uint64_t c = first_chunk + wave; c < end_chunk; c += n_waves) {
    const uint64_t base = c * kChunk;
    const uint64_t at = base + static_cast<uint64_t>(lane) * 16;
    uint32_t d[5];
    if (base + kChunk + 4 <= a.n) {
      const uint4 v = *reinterpret_cast<const uint4*>(a.text + at);
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      d[4] = *reinterpret_cast<const uint32_t*>(a.text + at + 16);
    } else {
      load_guarded(a.text, a.n, at, d);
    }
    uint32_t prev = '\n';  // byte before the lane's first byte ('\n' stands for "start of text")
    if (at > 0 && at <= a.n) prev = a.text[at - 1];
    uint32_t cand = 0;  // bit j: position at + j is a candidate start
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint32_t cur = (d[j >> 2] >> (8 * (j & 3))) & 0xFFu;
      const uint64_t s = at + j;
      bool ok = false;
      if (s < a.n) ok = (fb[cur >> 5] >> (cur & 31)) & 1u;
      if (P.nullable && s <= a.n) {
        int ctx = 0;
        if (ctxed) {
          if (s == 0 || rj_line_break(prev)) ctx |= 1;
          if (s == a.n || rj_line_break(cur)) ctx |= 2;
        }
        ok = ok || ((P.nullable >> ctx) & 1u);
      }
      ok = ok && s >= a.sb && s < a.se;
      cand |= static_cast<uint32_t>(ok) << j;
      prev = cur;
    }
    if (__ballot(cand != 0) == 0) continue;
#pragma unroll 1
    for (int j = 0; j < 16; j++) {
      wave_append((cand >> j) & 1u, at + j, a.hits, a.hits_cap, a.counters + kCntHits);
    }
  }
}

// ---------------------------------------------------------------------------------------
// Verify: longest match from every hit; survivors become (begin,end) candidates.
template <int NQ>
__global__ __launch_bounds__(256) void verify_lane(VerifyParams a, DevProgram P) {
  unsigned long long n_hits = a.counters[kCntHits];
  if (n_hits > a.hits_cap) {
    if (blockIdx.x == 0 && threadIdx.x == 0) a.counters[kCntOverflow] = 1;
    n_hits = a.hits_cap;
  }
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  // all lanes of a wave iterate together (wave-aggregated append needs the full wave)
  const uint64_t first = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  for (uint64_t base = first - lane_id(); base < n_hits; base += stride) {
    const uint64_t i = base + lane_id();
    bool found = false;
    uint64_t s = 0, e = 0;
    if (i < n_hits) {
      s = a.hits[i];
      found = rj_lane_longest<NQ>(P, a.text, a.n, s, &e);
    }
    wave_append_pair(found, s, e, a.cands, a.cands_cap, a.counters + kCntCands);
  }
}

namespace {

// Automaton state spread over the wave: lane l holds 32-bit words l, l+64, ... (NR of them).
template <int NR>
__device__ bool wave_longest(const DevProgram& P, const uint8_t* t, uint64_t n, uint64_t s, uint64_t* end,
                             bool anchored_full) {
  const int lane = lane_id();
  const int W = P.n_words;
  const bool ctxed = P.n_ctx > 1;
  int ctx = ctxed ? rj_context(t, n, s) : 0;
  bool found = false;
  if (!anchored_full || s == n) {
    if ((P.nullable >> ctx) & 1u) {
      found = true;
      *end = s;
    }
  }
  if (s >= n || P.n_pos == 0) return found;
  uint32_t S[NR], lin[NR];
  {
    const uint32_t* fr = P.first + ctx * W;
    const uint32_t* cr = P.cls + static_cast<uint32_t>(t[s]) * W;
#pragma unroll
    for (int r = 0; r < NR; r++) {
      const int wi = r * kWave + lane;
      S[r] = wi < W ? (fr[wi] & cr[wi]) : 0u;
      lin[r] = wi < W ? P.linear[wi] : 0u;
    }
  }
  uint64_t p = s + 1;
  for (;;) {
    bool alive = false;
#pragma unroll
    for (int r = 0; r < NR; r++) alive |= S[r] != 0;
    if (__ballot(alive) == 0) break;
    ctx = ctxed ? rj_context(t, n, p) : 0;
    if (!anchored_full || p == n) {
      const uint32_t* lr = P.last + ctx * W;
      bool acc = false;
#pragma unroll
      for (int r = 0; r < NR; r++) {
        const int wi = r * kWave + lane;
        if (wi < W) acc |= (S[r] & lr[wi]) != 0;
      }
      if (__ballot(acc) != 0) {
        found = true;
        *end = p;
      }
    }
    if (p == n) break;
    uint32_t T[NR];
    uint32_t carry_in = 0;  // bit shifted out of the previous register's lane 63
#pragma unroll
    for (int r = 0; r < NR; r++) {
      const uint32_t x = S[r] & lin[r];
      const uint32_t hi = x >> 31;
      uint32_t up = __shfl_up(hi, 1);
      if (lane == 0) up = carry_in;
      T[r] = (x << 1) | up;
      carry_in = __shfl(hi, kWave - 1);
    }
#pragma unroll
    for (int r = 0; r < NR; r++) {
      uint32_t sp = S[r] & ~lin[r];
      for (;;) {
        const unsigned long long m = __ballot(sp != 0);
        if (m == 0) break;
        const int src = __ffsll(static_cast<long long>(m)) - 1;
        const uint32_t spv = __shfl(sp, src);
        const int b = __ffs(static_cast<int>(spv)) - 1;
        const int pos = (r * kWave + src) * 32 + b;
        const uint32_t* row = P.rows + (static_cast<size_t>(ctx) * P.n_rows + P.row_of[pos]) * W;
#pragma unroll
        for (int r2 = 0; r2 < NR; r2++) {
          const int wi = r2 * kWave + lane;
          if (wi < W) T[r2] |= row[wi];
        }
        if (lane == src) sp &= sp - 1;
      }
    }
    const uint32_t* cr = P.cls + static_cast<uint32_t>(t[p]) * W;
#pragma unroll
    for (int r = 0; r < NR; r++) {
      const int wi = r * kWave + lane;
      S[r] = wi < W ? (T[r] & cr[wi]) : 0u;
    }
    p++;
  }
  return found;
}

}  // namespace

template <int NR>
__global__ __launch_bounds__(256) void verify_wave(VerifyParams a, DevProgram P) {
  unsigned long long n_hits = a.counters[kCntHits];
  if (n_hits > a.hits_cap) {
    if (blockIdx.x == 0 && threadIdx.x == 0) a.counters[kCntOverflow] = 1;
    n_hits = a.hits_cap;
  }
  const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_waves = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 6;
  for (uint64_t i = wave; i < n_hits; i += n_waves) {
    const uint64_t s = a.hits[i];
    uint64_t e = 0;
    const bool found = wave_longest<NR>(P, a.text, a.n, s, &e, false);
    if (found && lane_id() == 0) {
      const unsigned long long idx = atomicAdd(a.counters + kCntCands, 1ull);
      if (idx < a.cands_cap) {
        a.cands[2 * idx] = s;
        a.cands[2 * idx + 1] = e;
      }
    }
  }
}

// kMatchFull: result[0] = 1 iff the automaton started at 0 accepts exactly at n.
template <int NR>
__global__ __launch_bounds__(64) void match_full(const uint8_t* text, uint64_t n, DevProgram P, int* result) {
  uint64_t e = 0;
  const bool found = wave_longest<NR>(P, text, n, 0, &e, true);
  if (lane_id() == 0) result[0] = (found && e == n) ? 1 : 0;
}

// ---------------------------------------------------------------------------------------
// Finalize (small): sort by begin, drop duplicates, left-most-longest selection.
// One workgroup; cands are read from HBM once, everything else happens in LDS.
__global__ __launch_bounds__(1024) void finalize_small(FinalizeParams a) {
  __shared__ uint64_t key[kFinalizeCap];
  __shared__ uint64_t val[kFinalizeCap];
  __shared__ int all_disjoint;
  const unsigned long long n_raw = a.counters[kCntCands];
  if (n_raw > a.cands_cap || a.counters[kCntOverflow] != 0) {
    if (threadIdx.x == 0) {  // a list overflowed: the host grows it and retries
      a.counters[kCntOverflow] = 1;
      a.counters[kCntFinal] = ~0ull;
    }
    return;
  }
  if (n_raw > kFinalizeCap) {  // too many for LDS: the host takes the large path
    if (threadIdx.x == 0) a.counters[kCntFinal] = ~0ull;
    return;
  }
  const int n = static_cast<int>(n_raw);
  int m = 1;
  while (m < n) m <<= 1;
  for (int i = threadIdx.x; i < m; i += blockDim.x) {
    key[i] = i < n ? a.cands[2 * i] : ~0ull;
    val[i] = i < n ? a.cands[2 * i + 1] : ~0ull;
  }
  if (threadIdx.x == 0) all_disjoint = 1;
  __syncthreads();
  // bitonic sort on (key, val)
  for (int k = 2; k <= m; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < m; i += blockDim.x) {
        const int l = i ^ j;
        if (l > i) {
          const bool up = (i & k) == 0;
          const uint64_t ki = key[i], kl = key[l];
          if ((ki > kl) == up && ki != kl) {
            key[i] = kl; key[l] = ki;
            const uint64_t vi = val[i];
            val[i] = val[l]; val[l] = vi;
          }
        }
      }
      __syncthreads();
    }
  }
  // fast exit: pairwise disjoint, no duplicates, no empty matches -> selection is the identity
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const bool ok = val[i] > key[i] && (i == 0 || key[i] >= val[i - 1]);
    if (!ok) all_disjoint = 0;
  }
  __syncthreads();
  if (all_disjoint && (n == 0 || key[0] >= a.carry_cur)) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      if (static_cast<uint64_t>(i) < a.out_cap) {
        a.out[2 * i] = key[i];
        a.out[2 * i + 1] = val[i];
      }
    }
    if (threadIdx.x == 0) a.counters[kCntFinal] = static_cast<unsigned long long>(n);
    return;
  }
  // general case: sequential definition (clusters of overlapping candidates are tiny in practice)
  if (threadIdx.x == 0) {
    RjSelectState st;
    st.cur = a.carry_cur;
    st.prev_end = a.carry_prev_end;
    st.have_prev = a.have_prev != 0;
    unsigned long long out_n = 0;
    for (int i = 0; i < n; i++) {
      if (i > 0 && key[i] == key[i - 1]) continue;  // duplicate begin
      bool taken;
      if (rj_select_step(&st, key[i], val[i], &taken)) {
        if (out_n < a.out_cap) {
          a.out[2 * out_n] = key[i];
          a.out[2 * out_n + 1] = val[i];
        }
        out_n++;
      }
    }
    a.counters[kCntFinal] = out_n;
  }
}

// Selection over a sorted candidate list of any size (large path), one lane.
// TODO(round 2): cluster-parallel version (prefix-max of ends -> independent clusters).
__global__ void select_sorted(const uint64_t* keys, const uint64_t* vals, uint64_t n, FinalizeParams a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  RjSelectState st;
  st.cur = a.carry_cur;
  st.prev_end = a.carry_prev_end;
  st.have_prev = a.have_prev != 0;
  unsigned long long out_n = 0;
  for (uint64_t i = 0; i < n; i++) {
    if (i > 0 && keys[i] == keys[i - 1]) continue;
    bool taken;
    if (rj_select_step(&st, keys[i], vals[i], &taken)) {
      if (out_n < a.out_cap) {
        a.out[2 * out_n] = keys[i];
        a.out[2 * out_n + 1] = vals[i];
      }
      out_n++;
    }
  }
  a.counters[kCntFinal] = out_n;
}

// Parallel check used by the large path: is the sorted list already a valid result?
__global__ void check_disjoint(const uint64_t* keys, const uint64_t* vals, uint64_t n, int* flag) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool ok = vals[i] > keys[i] && (i == 0 || keys[i] >= vals[i - 1]);
  if (!ok) *flag = 0;
}

__global__ void interleave_pairs(const uint64_t* keys, const uint64_t* vals, uint64_t n, uint64_t* out, uint64_t cap) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n || i >= cap) return;
  out[2 * i] = keys[i];
  out[2 * i + 1] = vals[i];
}

__global__ void split_pairs(const uint64_t* pairs, uint64_t n, uint64_t* keys, uint64_t* vals) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = pairs[2 * i];
  vals[i] = pairs[2 * i + 1];
}

// ---------------------------------------------------------------------------------------
// Launchers (host side of the <<< >>> syntax lives here so engine.cc stays plain C++).
namespace {
int grid_for_scan(uint64_t chunks) {
  // memory-bound streaming: ~8 workgroups of 256 threads per CU, grid-stride the rest
  const uint64_t waves = chunks;
  uint64_t blocks = (waves + 3) / 4;
  if (blocks > 256u * 8u) blocks = 256u * 8u;
  if (blocks == 0) blocks = 1;
  return static_cast<int>(blocks);
}
}  // namespace

template <bool MASKED>
static void launch_windows_k(int k, const ScanParams& a, const WindowSet& ws, int grid, hipStream_t st) {
  switch (k) {
    case 1: hipLaunchKernelGGL((scan_windows<1, MASKED>), dim3(grid), dim3(256), 0, st, a, ws); break;
    case 2: hipLaunchKernelGGL((scan_windows<2, MASKED>), dim3(grid), dim3(256), 0, st, a, ws); break;
    case 3: hipLaunchKernelGGL((scan_windows<3, MASKED>), dim3(grid), dim3(256), 0, st, a, ws); break;
    case 4: hipLaunchKernelGGL((scan_windows<4, MASKED>), dim3(grid), dim3(256), 0, st, a, ws); break;
    case 5: hipLaunchKernelGGL((scan_windows<5, MASKED>), dim3(grid), dim3(256), 0, st, a, ws); break;
    case 6: hipLaunchKernelGGL((scan_windows<6, MASKED>), dim3(grid), dim3(256), 0, st, a, ws); break;
    case 7: hipLaunchKernelGGL((scan_windows<7, MASKED>), dim3(grid), dim3(256), 0, st, a, ws); break;
    default: hipLaunchKernelGGL((scan_windows<8, MASKED>), dim3(grid), dim3(256), 0, st, a, ws); break;
  }
}

void launch_scan_windows(const ScanParams& a, const WindowSet& ws, int n_windows, hipStream_t st) {
  if (a.whi <= a.wlo) return;
  const uint64_t chunks = (a.whi + kChunk - 1) / kChunk - a.wlo / kChunk;
  const int grid = grid_for_scan(chunks);
  if (ws.mask == 0xFFFFFFFFu) launch_windows_k<false>(n_windows, a, ws, grid, st);
  else launch_windows_k<true>(n_windows, a, ws, grid, st);
}

void launch_scan_dense(const ScanParams& a, const DevProgram& P, hipStream_t st) {
  if (a.se <= a.sb) return;
  const uint64_t chunks = (a.se + kChunk - 1) / kChunk - a.sb / kChunk;
  hipLaunchKernelGGL(scan_dense, dim3(grid_for_scan(chunks)), dim3(256), 0, st, a, P);
}

void launch_verify(const VerifyParams& a, const DevProgram& P, uint64_t expected_hits, hipStream_t st) {
  const int W = P.n_words;
  if (W <= 4) {
    uint64_t blocks = (expected_hits + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    if (W <= 2) hipLaunchKernelGGL((verify_lane<1>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, a, P);
    else hipLaunchKernelGGL((verify_lane<2>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, a, P);
    return;
  }
  uint64_t blocks = (expected_hits + 3) / 4;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  const unsigned g = static_cast<unsigned>(blocks);
  if (W <= 64) hipLaunchKernelGGL((verify_wave<1>), dim3(g), dim3(256), 0, st, a, P);
  else if (W <= 128) hipLaunchKernelGGL((verify_wave<2>), dim3(g), dim3(256), 0, st, a, P);
  else hipLaunchKernelGGL((verify_wave<4>), dim3(g), dim3(256), 0, st, a, P);
}

void launch_match_full(const uint8_t* text, uint64_t n, const DevProgram& P, int* result, hipStream_t st) {
  const int W = P.n_words;
  if (W <= 64) hipLaunchKernelGGL((match_full<1>), dim3(1), dim3(64), 0, st, text, n, P, result);
  else if (W <= 128) hipLaunchKernelGGL((match_full<2>), dim3(1), dim3(64), 0, st, text, n, P, result);
  else hipLaunchKernelGGL((match_full<4>), dim3(1), dim3(64), 0, st, text, n, P, result);
}

void launch_finalize_small(const FinalizeParams& a, hipStream_t st) {
  hipLaunchKernelGGL(finalize_small, dim3(1), dim3(1024), 0, st, a);
}

void launch_split_pairs(const uint64_t* pairs, uint64_t n, uint64_t* keys, uint64_t* vals, hipStream_t st) {
  if (n == 0) return;
  hipLaunchKernelGGL(split_pairs, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, pairs, n, keys, vals);
}

void launch_check_disjoint(const uint64_t* keys, const uint64_t* vals, uint64_t n, int* flag, hipStream_t st) {
  if (n == 0) return;
  hipLaunchKernelGGL(check_disjoint, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, keys, vals, n, flag);
}

void launch_interleave_pairs(const uint64_t* keys, const uint64_t* vals, uint64_t n, uint64_t* out, uint64_t cap,
                             hipStream_t st) {
  if (n == 0) return;
  hipLaunchKernelGGL(interleave_pairs, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, keys, vals, n, out, cap);
}

void launch_select_sorted(const uint64_t* keys, const uint64_t* vals, uint64_t n, const FinalizeParams& a,
                          hipStream_t st) {
  hipLaunchKernelGGL(select_sorted, dim3(1), dim3(64), 0, st, keys, vals, n, a);
}

}  // namespace rejit_amd
