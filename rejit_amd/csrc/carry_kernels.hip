// rejit_amd/csrc/carry_kernels.hip -- the linear-time carry scan on the GPU: thin kernels around the
// per-sub-chunk bodies of carry_scan.h (which the CPU tests run against the oracle).
//
// Replaces, for inputs on which the per-start verifier would be quadratic, the reference's one-byte-
// per-iteration NFA loop with merged threads (GenerateMatchDirection / SetState / ClearStates,
// src/x64/codegen-x64.cc:535-640, 951-987, 1075-1097).  One LANE owns one sub-chunk of the text and
// walks it backwards with the reverse automaton; the lanes of a wave hold their private state
// (classes, symbolic sources) interleaved in LDS (one state word) or in a global scratch slab (wider
// automata).  Bound: VALU / LDS issue (tens of instructions per text byte), not HBM -- this is the
// fallback that keeps unbounded repetitions over long runs linear, not the streaming path.
#include <hip/hip_runtime.h>

#include "carry_scan.h"
#include "kernels.h"

namespace rejit_amd {

namespace {

constexpr int kCsWave = 64;

// stage R's tables in LDS (when `lds_words` says they fit) and return a descriptor that points there
__device__ __forceinline__ DevProgram cs_stage(const DevProgram& R, uint32_t* tab, uint32_t lds_words) {
  DevProgram Q = R;
  if (lds_words >= R.table_words) {
    for (uint32_t i = threadIdx.x; i < R.table_words; i += blockDim.x) tab[i] = R.first[i];
    const int W = R.n_words, C = R.n_ctx, NP = R.n_pos > 0 ? R.n_pos : 1;
    Q.first = tab;
    Q.last = tab + C * W;
    Q.linear = tab + 2 * C * W;
    Q.row_of = reinterpret_cast<const int32_t*>(tab + 2 * C * W + W);
    Q.rows = tab + 2 * C * W + W + NP;
    Q.cls = tab + 2 * C * W + W + NP + C * R.n_rows * W;
  }
  __syncthreads();
  return Q;
}

// The lane-private arrays of one wave: cval [P] uint64, cset [P*NW], src [P*NW] uint32, element i of
// lane l at [i * 64 + l].  `slab` = this wave's slab (LDS or global), 8-byte aligned.
template <int NW>
__device__ __forceinline__ void cs_private(void* slab, int P, int lane, CsClasses<NW>* C, CsArr<uint32_t>* src) {
  const int np = P > 0 ? P : 1;
  uint64_t* v = static_cast<uint64_t*>(slab);
  uint32_t* s = reinterpret_cast<uint32_t*>(v + static_cast<size_t>(np) * kCsWave);
  C->val = CsArr<uint64_t>{v + lane, kCsWave};
  C->set = CsArr<uint32_t>{s + lane, kCsWave};
  C->nc = 0;
  *src = CsArr<uint32_t>{s + static_cast<size_t>(np) * NW * kCsWave + lane, kCsWave};
}

}  // namespace

__host__ __device__ inline size_t cs_private_bytes(int n_pos, int nw) {
  const size_t np = n_pos > 0 ? static_cast<size_t>(n_pos) : 1;
  return np * kCsWave * (8 + 2 * 4 * static_cast<size_t>(nw));
}

// phase 1.  Sub-chunk i of the run = bytes [a0 + i*sub, min(.. + sub, n)); vals / mats: [m][P] / [m][P*W].
template <int NW, bool LDS_STATE>
__global__ __launch_bounds__(64) void cs_summarize_kernel(DevProgram R, const uint8_t* text, uint64_t n, uint64_t a0, uint64_t sub,
                                                          uint64_t m, uint64_t* vals, uint32_t* mats, uint8_t* scratch,
                                                          uint32_t lds_table_words) {
  extern __shared__ uint64_t cs_lds[];
  const int lane = threadIdx.x & (kCsWave - 1);
  const int np = R.n_pos > 0 ? R.n_pos : 1;
  const size_t priv = cs_private_bytes(R.n_pos, NW);
  uint8_t* lds = reinterpret_cast<uint8_t*>(cs_lds);
  const DevProgram Q = cs_stage(R, reinterpret_cast<uint32_t*>(lds + (LDS_STATE ? priv : 0)), lds_table_words);
  CsClasses<NW> C;
  CsArr<uint32_t> src;
  cs_private<NW>(LDS_STATE ? static_cast<void*>(lds) : static_cast<void*>(scratch + static_cast<size_t>(blockIdx.x) * priv), R.n_pos, lane,
                 &C, &src);
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kCsWave + lane; i < m; i += static_cast<uint64_t>(gridDim.x) * kCsWave) {
    const uint64_t a = a0 + i * sub;
    uint64_t b = a + sub;
    if (b > n) b = n;
    cs_summarize<NW>(Q, text, n, a < n ? a : n, b, C, src, vals + i * np, mats + i * static_cast<uint64_t>(np) * R.n_words);
  }
}

// phase 2, two levels (a strictly sequential pass over 32 Ki sub-chunk summaries was 15 of the 17 ms of
// `[acgt]+` over 64 MiB): (a) every group of kCsGroup consecutive sub-chunks composes its transfers
// into one, a lane per group; (b) one wave resolves the groups right to left; (c) every group resolves
// its own sub-chunks from the state entering at its right edge, a lane per group again.
constexpr uint64_t kCsGroup = 64;

__global__ __launch_bounds__(64) void cs_group_compose_kernel(int P, int W, uint64_t m, const uint64_t* vals, const uint32_t* mats,
                                                              uint64_t* gvals, uint32_t* gmats, uint8_t* scratch) {
  const int np = P > 0 ? P : 1;
  const uint64_t ng = (m + kCsGroup - 1) / kCsGroup;
  const uint64_t g = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  // ping-pong buffers of this lane: 2 x (np values + np*W words)
  const size_t one = static_cast<size_t>(np) * 8 + static_cast<size_t>(np) * W * 4;
  uint8_t* mine = scratch + g * 2 * one;
  uint64_t* la = reinterpret_cast<uint64_t*>(mine);
  uint32_t* ra = reinterpret_cast<uint32_t*>(mine + static_cast<size_t>(np) * 8);
  uint64_t* lb = reinterpret_cast<uint64_t*>(mine + one);
  uint32_t* rb = reinterpret_cast<uint32_t*>(mine + one + static_cast<size_t>(np) * 8);
  for (int k = 0; k < P; k++) {
    la[k] = 0;
    for (int j = 0; j < W; j++) ra[static_cast<size_t>(k) * W + j] = 0;
    ra[static_cast<size_t>(k) * W + (k >> 5)] = 1u << (k & 31);  // identity
  }
  uint64_t hi = (g + 1) * kCsGroup;
  if (hi > m) hi = m;
  for (uint64_t i = hi; i-- > g * kCsGroup;) {
    cs_compose(P, W, vals + i * np, mats + i * static_cast<uint64_t>(np) * W, la, ra, lb, rb);
    uint64_t* tl = la; la = lb; lb = tl;
    uint32_t* tr = ra; ra = rb; rb = tr;
  }
  for (int k = 0; k < P; k++) {
    gvals[g * np + k] = la[k];
    for (int j = 0; j < W; j++) gmats[(g * np + k) * W + j] = ra[static_cast<size_t>(k) * W + j];
  }
}

// (b): right to left over the groups, one wave, lane = position (more than 64 positions: each lane
// takes several)
__global__ __launch_bounds__(64) void cs_resolve_kernel(int P, int W, uint64_t m, uint64_t* vals, const uint32_t* mats) {
  const int np = P > 0 ? P : 1;
  const int lane = threadIdx.x;
  __shared__ uint64_t dn[1024];
  for (int k = lane; k < 1024; k += kCsWave) dn[k] = 0;
  __syncthreads();
  for (uint64_t i = m; i-- > 0;) {
    uint64_t* D = vals + i * np;
    const uint32_t* Rm = mats + i * static_cast<uint64_t>(np) * W;
    for (int k = lane; k < P; k += kCsWave) {
      uint64_t d = D[k];
      for (int mm = 0; mm < P; mm++) {
        const uint64_t v = dn[mm];
        if (v > d && ((Rm[static_cast<size_t>(mm) * W + (k >> 5)] >> (k & 31)) & 1u)) d = v;
      }
      D[k] = d;
    }
    __syncthreads();
    for (int k = lane; k < P; k += kCsWave) dn[k] = D[k];
    __syncthreads();
  }
}

// (c): gvals[g + 1] = resolved state entering group g at its right edge (zeros beyond the last group)
__global__ __launch_bounds__(64) void cs_group_apply_kernel(int P, int W, uint64_t m, uint64_t* vals, const uint32_t* mats,
                                                            const uint64_t* gvals) {
  const int np = P > 0 ? P : 1;
  const uint64_t ng = (m + kCsGroup - 1) / kCsGroup;
  const uint64_t g = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  uint64_t hi = (g + 1) * kCsGroup;
  if (hi > m) hi = m;
  for (uint64_t i = hi; i-- > g * kCsGroup;)
    cs_resolve(P, W, vals + i * np, mats + i * static_cast<uint64_t>(np) * W, i + 1 == hi ? gvals + (g + 1) * np : vals + (i + 1) * np);
}

// phase 3: E(s) for the own sub-chunks (the first m_own of the run)
template <int NW, bool LDS_STATE>
__global__ __launch_bounds__(64) void cs_emit_kernel(DevProgram R, const uint8_t* text, uint64_t n, uint64_t a0, uint64_t sub,
                                                     uint64_t m, uint64_t m_own, uint64_t sb, uint64_t se, const uint64_t* vals,
                                                     uint64_t* E, uint8_t* scratch, uint32_t lds_table_words,
                                                     unsigned long long* longest) {
  extern __shared__ uint64_t cs_lds[];
  const int lane = threadIdx.x & (kCsWave - 1);
  const int np = R.n_pos > 0 ? R.n_pos : 1;
  const size_t priv = cs_private_bytes(R.n_pos, NW);
  uint8_t* lds = reinterpret_cast<uint8_t*>(cs_lds);
  const DevProgram Q = cs_stage(R, reinterpret_cast<uint32_t*>(lds + (LDS_STATE ? priv : 0)), lds_table_words);
  CsClasses<NW> C;
  CsArr<uint32_t> src;
  cs_private<NW>(LDS_STATE ? static_cast<void*>(lds) : static_cast<void*>(scratch + static_cast<size_t>(blockIdx.x) * priv), R.n_pos, lane,
                 &C, &src);
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kCsWave + lane; i < m_own; i += static_cast<uint64_t>(gridDim.x) * kCsWave) {
    const uint64_t a = a0 + i * sub;
    uint64_t b = a + sub;
    if (b > n) b = n;
    cs_emit<NW>(Q, text, n, a < n ? a : n, b, sb, se, i + 1 < m ? vals + (i + 1) * np : nullptr, C, E, a0);
    // the longest candidate seen (the host drops its "this text needs the carry scan" hint when every
    // candidate would have fitted the parallel verifier's walk): E(s) - s is largest at a start
    // that is the first of its sub-chunk to reach that end -- a scan of the lane's own slice
    uint64_t best = 0;
    const uint64_t lo = a > sb ? a : sb;
    uint64_t hi = a + sub < se ? a + sub : se;
    for (uint64_t p = lo; p < hi; p++) {
      const uint64_t e = E[p - a0];
      if (e != kCsNone && e - p > best) best = e - p;
    }
    if (best > 0) atomicMax(longest, static_cast<unsigned long long>(best));
  }
}

// phase 4
__global__ __launch_bounds__(64) void cs_local_chain_kernel(const uint64_t* E, uint64_t* G, uint64_t a0, uint64_t sub, uint64_t m_own,
                                                            uint64_t sb, uint64_t se) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= m_own) return;
  uint64_t lo = a0 + i * sub, hi = lo + sub;
  if (lo < sb) lo = sb;
  if (hi > se) hi = se;
  cs_local_chain(E, G, a0, lo, hi);
}

// phase 5: the chain hops from sub-chunk to sub-chunk (one thread; entry[] preset to kCsNone)
__global__ void cs_global_chain_kernel(const uint64_t* G, uint64_t a0, uint64_t sub, uint64_t cur, uint64_t se, uint64_t* entry) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  while (cur < se) {
    entry[(cur - a0) / sub] = cur;
    cur = G[cur - a0];
  }
}

// phase 6: one wave per entered sub-chunk follows the chain inside it; 64 positions are searched
// for the next start with a match at a time.  Taken matches go to the front of the sub-chunk's slab.
__global__ __launch_bounds__(256) void cs_take_kernel(uint64_t* E, uint64_t* G, uint64_t a0, uint64_t sub, uint64_t m_own, uint64_t se,
                                                      const uint64_t* entry, uint32_t* counts, unsigned long long* total) {
  const int lane = threadIdx.x & (kCsWave - 1);
  const uint64_t n_waves = (static_cast<uint64_t>(gridDim.x) * blockDim.x) >> 6;
  for (uint64_t i = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6; i < m_own; i += n_waves) {
    uint64_t cur = entry[i];
    uint32_t cnt = 0;
    if (cur != kCsNone) {
      uint64_t hi = a0 + (i + 1) * sub;
      if (hi > se) hi = se;
      const uint64_t slab = i * sub;
      while (cur < hi) {
        const uint64_t s = cur + lane;
        const uint64_t e = s < hi ? E[s - a0] : kCsNone;
        const uint64_t have = __ballot(e != kCsNone);
        if (have == 0) {
          cur += kCsWave;
          continue;
        }
        const int l = __ffsll(static_cast<long long>(have)) - 1;
        const uint64_t s0 = cur + l;
        const uint64_t e0 = __shfl(e, l);
        if (lane == 0) {  // slab + cnt <= s0 - a0: every lane has read its entry already
          G[slab + cnt] = s0;
          E[slab + cnt] = e0;
        }
        cnt++;
        cur = e0 > s0 ? e0 : s0 + 1;
      }
    }
    if (lane == 0) {
      counts[i] = cnt;
      if (cnt) atomicAdd(total, static_cast<unsigned long long>(cnt));
    }
  }
}

// ------------------------------------------------------------------------------------- launchers
// (round 4: up to 1024 positions -- 256 before, which left cyclic automata wider than that with RJ_TOO_LARGE at match time)
int cs_state_words(const DevProgram& R) {
  return R.n_words <= 1 ? 1 : R.n_words <= 2 ? 2 : R.n_words <= 4 ? 4 : R.n_words <= 8 ? 8 : R.n_words <= 16 ? 16 : R.n_words <= 32 ? 32 : R.n_words <= 64 ? 64 : R.n_words <= 128 ? 128 : R.n_words <= 256 ? 256 : 0;
}

namespace {
struct CsLaunch {
  unsigned grid;
  bool lds_state;
  size_t lds_bytes;
  uint32_t lds_table_words;
  size_t scratch_bytes;
};

CsLaunch cs_plan(const DevProgram& R, uint64_t lanes) {
  CsLaunch L{};
  const int nw = cs_state_words(R);
  const size_t priv = cs_private_bytes(R.n_pos, nw);
  uint64_t waves = (lanes + kCsWave - 1) / kCsWave;
  if (waves < 1) waves = 1;
  if (waves > 8192) waves = 8192;  // grid-stride beyond (bounds the scratch slab)
  L.grid = static_cast<unsigned>(waves);
  L.lds_state = nw == 1 && priv <= 40 * 1024;
  const size_t table_bytes = static_cast<size_t>(R.table_words) * 4;
  const size_t room = 64 * 1024 - (L.lds_state ? priv : 0);
  L.lds_table_words = table_bytes <= room ? R.table_words : 0;
  L.lds_bytes = (L.lds_state ? priv : 0) + static_cast<size_t>(L.lds_table_words) * 4;
  L.scratch_bytes = L.lds_state ? 0 : priv * waves;
  return L;
}
}  // namespace

size_t cs_scratch_bytes(const DevProgram& R, uint64_t lanes) { return cs_plan(R, lanes).scratch_bytes; }

void launch_cs_summarize(const DevProgram& R, const uint8_t* text, uint64_t n, uint64_t a0, uint64_t sub, uint64_t m, uint64_t* vals,
                         uint32_t* mats, uint8_t* scratch, hipStream_t st) {
  const CsLaunch L = cs_plan(R, m);
#define RJ_CS_SUM(NW, LDS) \
  hipLaunchKernelGGL((cs_summarize_kernel<NW, LDS>), dim3(L.grid), dim3(64), L.lds_bytes, st, R, text, n, a0, sub, m, vals, mats, scratch, L.lds_table_words)
  switch (cs_state_words(R)) {
    case 1: if (L.lds_state) RJ_CS_SUM(1, true); else RJ_CS_SUM(1, false); break;
    case 2: RJ_CS_SUM(2, false); break;
    case 4: RJ_CS_SUM(4, false); break;
    case 8: RJ_CS_SUM(8, false); break;
    case 16: RJ_CS_SUM(16, false); break;
    case 32: RJ_CS_SUM(32, false); break;
    case 64: RJ_CS_SUM(64, false); break;
    case 128: RJ_CS_SUM(128, false); break;
    default: RJ_CS_SUM(256, false); break;
  }
#undef RJ_CS_SUM
}

size_t cs_resolve_scratch_bytes(const DevProgram& R, uint64_t m) {
  const size_t np = R.n_pos > 0 ? static_cast<size_t>(R.n_pos) : 1, W = static_cast<size_t>(R.n_words);
  const size_t ng = (m + kCsGroup - 1) / kCsGroup;
  const size_t one = np * 8 + np * W * 4;
  // group values [ng + 1][np], group matrices [ng][np * W], ping-pong buffers
  return (ng + 1) * np * 8 + ng * np * W * 4 + ng * 2 * one + 64;
}

void launch_cs_resolve(const DevProgram& R, uint64_t m, uint64_t* vals, const uint32_t* mats, uint8_t* scratch, hipStream_t st) {
  const size_t np = R.n_pos > 0 ? static_cast<size_t>(R.n_pos) : 1, W = static_cast<size_t>(R.n_words);
  const uint64_t ng = (m + kCsGroup - 1) / kCsGroup;
  uint64_t* gvals = reinterpret_cast<uint64_t*>(scratch);
  uint32_t* gmats = reinterpret_cast<uint32_t*>(scratch + (ng + 1) * np * 8);
  uint8_t* ping = scratch + (((ng + 1) * np * 8 + ng * np * W * 4 + 15) & ~static_cast<size_t>(15));
  (void)hipMemsetAsync(gvals + ng * np, 0, np * 8, st);  // nothing enters the last group
  const unsigned blocks = static_cast<unsigned>((ng + 63) / 64);
  hipLaunchKernelGGL(cs_group_compose_kernel, dim3(blocks), dim3(64), 0, st, R.n_pos, R.n_words, m, vals, mats, gvals, gmats, ping);
  hipLaunchKernelGGL(cs_resolve_kernel, dim3(1), dim3(64), 0, st, R.n_pos, R.n_words, ng, gvals, gmats);
  hipLaunchKernelGGL(cs_group_apply_kernel, dim3(blocks), dim3(64), 0, st, R.n_pos, R.n_words, m, vals, mats, gvals);
}

void launch_cs_emit(const DevProgram& R, const uint8_t* text, uint64_t n, uint64_t a0, uint64_t sub, uint64_t m, uint64_t m_own,
                    uint64_t sb, uint64_t se, const uint64_t* vals, uint64_t* E, uint8_t* scratch, unsigned long long* longest,
                    hipStream_t st) {
  const CsLaunch L = cs_plan(R, m);  // the same plan as the summaries (one scratch slab serves both)
  unsigned grid = static_cast<unsigned>(std::min<uint64_t>(L.grid, (m_own + kCsWave - 1) / kCsWave));
  if (grid < 1) grid = 1;
#define RJ_CS_EMIT(NW, LDS) \
  hipLaunchKernelGGL((cs_emit_kernel<NW, LDS>), dim3(grid), dim3(64), L.lds_bytes, st, R, text, n, a0, sub, m, m_own, sb, se, vals, E, scratch, L.lds_table_words, longest)
  switch (cs_state_words(R)) {
    case 1: if (L.lds_state) RJ_CS_EMIT(1, true); else RJ_CS_EMIT(1, false); break;
    case 2: RJ_CS_EMIT(2, false); break;
    case 4: RJ_CS_EMIT(4, false); break;
    case 8: RJ_CS_EMIT(8, false); break;
    case 16: RJ_CS_EMIT(16, false); break;
    case 32: RJ_CS_EMIT(32, false); break;
    case 64: RJ_CS_EMIT(64, false); break;
    case 128: RJ_CS_EMIT(128, false); break;
    default: RJ_CS_EMIT(256, false); break;
  }
#undef RJ_CS_EMIT
}

void launch_cs_chain(uint64_t* E, uint64_t* G, uint64_t a0, uint64_t sub, uint64_t m_own, uint64_t sb, uint64_t se, uint64_t cur,
                     uint64_t* entry, uint32_t* counts, unsigned long long* total, hipStream_t st) {
  hipLaunchKernelGGL(cs_local_chain_kernel, dim3(static_cast<unsigned>((m_own + 63) / 64)), dim3(64), 0, st, E, G, a0, sub, m_own, sb, se);
  hipLaunchKernelGGL(cs_global_chain_kernel, dim3(1), dim3(64), 0, st, G, a0, sub, cur, se, entry);
  uint64_t blocks = (m_own + 3) / 4;
  blocks = blocks < 1 ? 1 : blocks > 16384 ? 16384 : blocks;
  hipLaunchKernelGGL(cs_take_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, E, G, a0, sub, m_own, se, entry, counts, total);
}

}  // namespace rejit_amd
