// rejit_amd/csrc/select_kernels.hip -- the left-most-longest selection when candidates overlap (split out of kernels.hip in round 6):
// finalize_small (one workgroup, LDS) and the blocked chain selection chain_* for lists of any size (reference: MatchAllAppendFilter +
// CheckMatch, src/codegen.cc:36-86, src/x64/codegen-x64.cc:401-466), the zero-length rule, detect_adjacent.
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
// Finalize (small): sort by begin, drop duplicates, left-most-longest selection.
// One workgroup; cands are read from HBM once, everything else happens in LDS.
__global__ __launch_bounds__(1024) void finalize_small(FinalizeParams a) {
  __shared__ uint64_t key[kFinalizeCap];
  __shared__ uint64_t val[kFinalizeCap];
  __shared__ int all_disjoint;
  __shared__ int n_valid;
  const unsigned long long n_raw = a.counters[kCntHits] * a.expand;  // candidate slots
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
  const int n_slots = static_cast<int>(n_raw);
  int m = 1;
  while (m < n_slots) m <<= 1;
  if (threadIdx.x == 0) {
    all_disjoint = 1;
    n_valid = 0;
  }
  __syncthreads();
  int mine = 0;
  for (int i = threadIdx.x; i < m; i += blockDim.x) {
    // starts without a match sort to the end (the input is already ordered by begin; the sort is
    // kept because it also serves callers that pass unordered candidates, and costs ~3 us)
    const bool ok = i < n_slots && a.cand_end[i] != kNoMatch;
    key[i] = ok ? a.cand_begin[i] : ~0ull;
    val[i] = ok ? a.cand_end[i] : ~0ull;
    mine += ok;
  }
  if (mine) atomicAdd(&n_valid, mine);
  __syncthreads();
  const int n = n_valid;
  if (threadIdx.x == 0) a.counters[kCntCands] = static_cast<unsigned long long>(n);
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
  if (a.detect_adjacent) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const uint64_t e = val[i];
      if (e <= key[i]) continue;
      int lo = 0, hi = n;  // first index with key >= e
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (key[mid] < e) lo = mid + 1; else hi = mid;
      }
      if (lo < n && key[lo] == e) a.counters[kCntAdjacent] = 1;
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
      if (a.detect_conflict && key[i] < st.cur && val[i] > st.cur) a.counters[kCntConflict] = 1;
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


// Selection over a sorted candidate list of any size (large path).
//
// The greedy rule (reference: MatchAllAppendFilter + CheckMatch, src/codegen.cc:36-86,
// codegen-x64.cc:401-466) is a CHAIN over the candidates: after taking i the next one taken is
//     nxt[i] = the first j > i with begin[j] >= max(end[i], begin[i] + 1).
// With M[i] = max end of the candidates before i (exclusive prefix max, computed by the caller)
// candidate i is a HEAD iff begin[i] >= max(M[i], carry_cur) and begin[i] > begin[i-1]: nothing
// before it can overlap it, so every chain passes through it.  Round 1 let the thread of a head walk
// its whole cluster, which is sequential in the cluster's size -- `[ab]{40}c*` over 4 MiB of a/b is
// ONE cluster of 4 M overlapping candidates: 1.05 s in that kernel.  Now the list is cut into blocks:
//   chain_next    nxt[] by (galloping) binary search, one thread per candidate
//   chain_local   per block, right to left: G[i] = where a chain that stands at i leaves the block
//   chain_hop     from every block that holds a head (and from the chain's first candidate) hop block
//                 to block through G until a block with a head of its own: the entry points
//   chain_mark    per block: follow nxt[] from the entry point (or the first head) to the block's end
// Sequential depth: block + (largest cluster / block) + block instead of the largest cluster.
constexpr uint64_t kChainNone = ~0ull;

// the block size balances the three sequential stretches (block + cluster / block + block): about
// sqrt(n), so that a few thousand candidates are not walked by a handful of lanes for half a millisecond
static uint64_t chain_block(uint64_t n) {
  uint64_t b = 32;
  while (b < 1024 && b * b < n) b <<= 1;
  return b;
}

__global__ void chain_next(const uint64_t* keys, const uint64_t* vals, uint64_t n, uint64_t carry_cur, uint64_t* nxt, uint64_t* i0_out) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i == 0) {  // the chain's first candidate: the first index with begin >= carry_cur
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
      const uint64_t mid = (lo + hi) >> 1;
      if (keys[mid] < carry_cur) lo = mid + 1; else hi = mid;
    }
    *i0_out = lo;
  }
  if (i >= n) return;
  const uint64_t b = keys[i], e = vals[i];
  const uint64_t cur = e > b ? e : b + 1;
  uint64_t lo = i + 1, hi = n;  // first index in (i, n] with key >= cur
  // the next candidate usually is close: gallop before bisecting
  uint64_t step = 1;
  while (lo + step < n && keys[lo + step] < cur) {
    lo += step + 1;
    step <<= 1;
  }
  if (lo + step < hi) hi = lo + step;
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (keys[mid] < cur) lo = mid + 1; else hi = mid;
  }
  nxt[i] = lo;
}

// One WAVE per block of B <= 1024 candidates: the block's `nxt` is staged in LDS (coalesced), lane 0 resolves
// G back to front there (a dependent step per candidate costs an LDS access, not a round trip to L2 as when one
// lane walked a block in global memory -- that was 125 us of the complex-regex tail), the heads are found by
// all lanes, G goes back coalesced.
__global__ __launch_bounds__(64) void chain_local(const uint64_t* keys, const uint64_t* pmax, const uint64_t* nxt, uint64_t n,
                                                  uint64_t carry_cur, uint64_t B, const uint64_t* i0_ptr, uint64_t* G,
                                                  uint64_t* first_head, uint64_t* entry) {
  __shared__ uint64_t s_nxt[1024];
  __shared__ uint64_t s_g[1024];
  const uint64_t blk = blockIdx.x;
  const uint64_t lo = blk * B;
  if (lo >= n) return;
  const uint64_t hi = lo + B < n ? lo + B : n;
  const uint32_t len = static_cast<uint32_t>(hi - lo), lane = threadIdx.x;
  uint64_t head = kChainNone;
  for (uint32_t k = lane; k < len; k += 64) {
    const uint64_t i = lo + k;
    s_nxt[k] = nxt[i];
    const uint64_t floor_i = pmax[i] > carry_cur ? pmax[i] : carry_cur;
    if (head == kChainNone && keys[i] >= floor_i && (i == 0 || keys[i] > keys[i - 1])) head = i;  // (the lane's first: k ascends)
  }
  __syncthreads();
  if (lane == 0)
    for (uint32_t k = len; k-- > 0;) {
      const uint64_t t = s_nxt[k];
      s_g[k] = t >= hi ? t : s_g[t - lo];
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {  // the block's first head = the minimum over the lanes
    const uint64_t other = __shfl_xor(head, o);
    head = other < head ? other : head;
  }
  __syncthreads();
  for (uint32_t k = lane; k < len; k += 64) G[lo + k] = s_g[k];
  if (lane == 0) {
    first_head[blk] = head;
    const uint64_t i0 = *i0_ptr;
    entry[blk] = (i0 < n && i0 / B == blk) ? i0 : kChainNone;
  }
}

__global__ __launch_bounds__(64) void chain_hop(const uint64_t* G, const uint64_t* first_head, const uint64_t* i0_ptr, uint64_t n,
                                                uint64_t B, uint64_t* entry) {
  const uint64_t blk = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (blk * B >= n) return;
  const uint64_t i0 = *i0_ptr;
  uint64_t start = first_head[blk];
  if (i0 < n && i0 / B == blk) start = i0;  // (a head of this block, if any, lies at or after i0)
  else if (start == kChainNone || start < i0) return;
  uint64_t idx = G[start];
  while (idx < n) {
    const uint64_t b2 = idx / B;
    entry[b2] = idx;
    if (first_head[b2] != kChainNone) break;  // that block's own thread goes on from its head
    idx = G[idx];
  }
}

// (a wave per block as well: the chain through the block is followed in LDS)
__global__ __launch_bounds__(64) void chain_mark(const uint64_t* nxt, const uint64_t* first_head, const uint64_t* entry,
                                                 const uint64_t* i0_ptr, uint64_t n, uint64_t B, uint8_t* taken) {
  __shared__ uint64_t s_nxt[1024];
  __shared__ uint8_t s_taken[1024];
  const uint64_t blk = blockIdx.x;
  const uint64_t lo = blk * B;
  if (lo >= n) return;
  const uint64_t hi = lo + B < n ? lo + B : n;
  uint64_t i = entry[blk];
  if (i == kChainNone) {
    i = first_head[blk];
    if (i == kChainNone || i < *i0_ptr) return;  // no chain comes through this block (wave-uniform)
  }
  const uint32_t len = static_cast<uint32_t>(hi - lo), lane = threadIdx.x;
  for (uint32_t k = lane; k < len; k += 64) {
    s_nxt[k] = nxt[lo + k];
    s_taken[k] = 0;
  }
  __syncthreads();
  if (lane == 0)
    while (i < hi) {
      s_taken[i - lo] = 1;
      i = s_nxt[i - lo];
    }
  __syncthreads();
  for (uint32_t k = lane; k < len; k += 64)
    if (s_taken[k]) taken[lo + k] = 1;
}

// idx[i] = i + 1 if candidate i was taken else 0 (input of the "last taken before i" max-scan)
__global__ void taken_index(const uint8_t* taken, uint64_t n, uint64_t* idx) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) idx[i] = taken[i] ? i + 1 : 0;
}

// zero-length rule (reference src/codegen.cc:65-73): a taken empty match that begins where the
// previously taken match ended is not reported.  keep[i] in {0,1} as uint64 for the sum-scan.
__global__ void apply_zero_length_rule(const uint64_t* keys, const uint64_t* vals, const uint8_t* taken,
                                       const uint64_t* last_taken, uint64_t n, uint64_t carry_prev_end,
                                       int have_prev, uint64_t* keep, unsigned long long* conflict) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool k = taken[i] != 0;
  if (conflict != nullptr && !k && last_taken[i] > 0) {
    // behind mode: a candidate hidden by the match taken before it must not reach beyond that match
    const uint64_t lb = keys[last_taken[i] - 1], le = vals[last_taken[i] - 1];
    const uint64_t cur = le > lb ? le : lb + 1;
    if (keys[i] != lb && keys[i] < cur && vals[i] > cur) *conflict = 1;
  }
  if (k && keys[i] == vals[i]) {
    const uint64_t lt = last_taken[i];  // 1-based index of the last taken candidate before i
    if (lt > 0) {
      if (vals[lt - 1] == keys[i]) k = false;
    } else if (have_prev && carry_prev_end == keys[i]) {
      k = false;
    }
  }
  keep[i] = k ? 1 : 0;
}

__global__ void compact_kept(const uint64_t* keys, const uint64_t* vals, const uint64_t* keep, const uint64_t* pos,
                             uint64_t n, uint64_t* out, uint64_t out_cap, unsigned long long* counters) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (keep[i]) {
    const uint64_t o = pos[i];
    if (o < out_cap) {
      out[2 * o] = keys[i];
      out[2 * o + 1] = vals[i];
    }
  }
  if (i == n - 1) counters[kCntFinal] = pos[i] + keep[i];
}

__global__ void detect_adjacent(const uint64_t* keys, const uint64_t* vals, unsigned long long* counters) {
  const uint64_t n = counters[kCntCands];
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint64_t e = vals[i];
    if (e <= keys[i]) continue;
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
      const uint64_t mid = (lo + hi) >> 1;
      if (keys[mid] < e) lo = mid + 1; else hi = mid;
    }
    if (lo < n && keys[lo] == e) counters[kCntAdjacent] = 1;
  }
}


// ---------------------------------------------------------------------------------------
// Launchers
void launch_finalize_small(const FinalizeParams& a, hipStream_t st) {
  hipLaunchKernelGGL(finalize_small, dim3(1), dim3(1024), 0, st, a);
}

void launch_detect_adjacent(const uint64_t* keys, const uint64_t* vals, uint64_t n_upper, unsigned long long* counters,
                            hipStream_t st) {
  uint64_t blocks = (n_upper + 255) / 256;
  blocks = blocks < 1 ? 1 : blocks > 4096 ? 4096 : blocks;
  hipLaunchKernelGGL(detect_adjacent, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, keys, vals, counters);
}

static unsigned blocks_for(uint64_t n) { return static_cast<unsigned>((n + 255) / 256); }

size_t chain_select_scratch_bytes(uint64_t n) { return ((n + 31) / 32 * 2 + 2) * sizeof(uint64_t); }

void launch_chain_select(const uint64_t* keys, const uint64_t* vals, const uint64_t* pmax, uint64_t n, uint64_t carry_cur,
                         uint8_t* taken, uint64_t* nxt, uint64_t* G, uint64_t* blocks_scratch, hipStream_t st) {
  const uint64_t B = chain_block(n);
  const uint64_t nb = (n + B - 1) / B;
  uint64_t* first_head = blocks_scratch;
  uint64_t* entry = blocks_scratch + nb;
  uint64_t* i0 = blocks_scratch + 2 * nb;
  (void)hipMemsetAsync(taken, 0, n, st);
  hipLaunchKernelGGL(chain_next, dim3(blocks_for(n)), dim3(256), 0, st, keys, vals, n, carry_cur, nxt, i0);
  const unsigned lb = static_cast<unsigned>((nb + 63) / 64);
  const unsigned wb = static_cast<unsigned>(nb);  // a wave per block of candidates
  hipLaunchKernelGGL(chain_local, dim3(wb), dim3(64), 0, st, keys, pmax, nxt, n, carry_cur, B, i0, G, first_head, entry);
  hipLaunchKernelGGL(chain_hop, dim3(lb), dim3(64), 0, st, G, first_head, i0, n, B, entry);
  hipLaunchKernelGGL(chain_mark, dim3(wb), dim3(64), 0, st, nxt, first_head, entry, i0, n, B, taken);
}

void launch_taken_index(const uint8_t* taken, uint64_t n, uint64_t* idx, hipStream_t st) {
  hipLaunchKernelGGL(taken_index, dim3(blocks_for(n)), dim3(256), 0, st, taken, n, idx);
}

void launch_zero_length_rule(const uint64_t* keys, const uint64_t* vals, const uint8_t* taken,
                             const uint64_t* last_taken, uint64_t n, uint64_t carry_prev_end, int have_prev,
                             uint64_t* keep, unsigned long long* conflict, hipStream_t st) {
  hipLaunchKernelGGL(apply_zero_length_rule, dim3(blocks_for(n)), dim3(256), 0, st, keys, vals, taken, last_taken, n,
                     carry_prev_end, have_prev, keep, conflict);
}

void launch_compact_kept(const uint64_t* keys, const uint64_t* vals, const uint64_t* keep, const uint64_t* pos,
                         uint64_t n, uint64_t* out, uint64_t out_cap, unsigned long long* counters, hipStream_t st) {
  hipLaunchKernelGGL(compact_kept, dim3(blocks_for(n)), dim3(256), 0, st, keys, vals, keep, pos, n, out, out_cap,
                     counters);
}
}  // namespace rejit_amd
