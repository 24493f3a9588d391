// rejit_amd/csrc/exact_replay.hip -- the reference's answer, ring artefact included, on texts of any
// size (exact_replay.h has the argument): synchronisation points from the all-starts position automaton,
// then the reference's own loop per segment, one lane per segment.
//
//   xr_lookup      the two ends of the ownership range: first proven synchronisation point >= sb, >= se
//   xr_chunk_sync  per 1-KiB chunk the first proven synchronisation point (lane per chunk, early exit)
//   xr_segments    per chunk that has one: where its segment ends; the last point and the longest segment
//   xr_replay      lane per segment: rj_replay_segment, ring in LDS (lane-interleaved) when it fits, pairs
//                  into the segment's own stretch of a scratch array (a segment has at most one match
//                  per byte)
//   xr_offsets     exclusive scan of the per-chunk counts (one workgroup), running total across batches
//   xr_compact     scratch -> result pairs
//
// The text is taken in batches of 64 MiB so that the scratch is bounded (1 GiB); a batch ends at its last
// synchronisation point, where the next one begins.  A stretch of more than kMaxSegment bytes without a
// proven synchronisation point is not replayed (one lane: 2-6 us per byte, tools/replay_probe.py): run_exact reports "not
// done" and the caller keeps the result of the parallel pipeline (documented semantics).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include "engine_internal.h"
#include "exact_replay.h"

namespace rejit_amd {
namespace {

constexpr uint64_t kChunk = 1024;
constexpr uint64_t kBatchChunks = 65536;        // 64 MiB of text per batch
constexpr uint64_t kMaxSegment = 16ull << 20;   // longest stretch one lane is asked to replay
constexpr int kReplayLanes = 64;                // lanes per workgroup of xr_replay
constexpr int kLdsRingSlots = 96;               // times x states up to this: ring in LDS (64 lanes x 96 x 8 B = 48 KiB)

// state words shared by the kernels of one run
enum { kXrY0 = 0, kXrY1, kXrLastSync, kXrMaxGap, kXrTotal, kXrStateSize };

template <int NQ>
__global__ void xr_lookup(DevProgram P, const uint8_t* t, uint64_t n, uint64_t sb, uint64_t se, unsigned long long* state) {
  const int i = threadIdx.x;
  if (blockIdx.x != 0 || i > 1) return;
  const uint64_t x = i == 0 ? sb : se;
  state[i == 0 ? kXrY0 : kXrY1] = x > n ? n + 1 : rj_first_sync<NQ>(P, t, n, x);
}

template <int NQ>
__global__ void xr_chunk_sync(DevProgram P, const uint8_t* t, uint64_t n, uint64_t ys, uint64_t y1, uint64_t n_chunks,
                              uint64_t* sync) {
  const uint64_t c = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (c >= n_chunks) return;
  const uint64_t c0 = ys + c * kChunk;
  sync[c] = rj_chunk_first_sync<NQ>(P, t, n, c0, std::min(c0 + kChunk, y1), c == 0);
}

// seg_end[c]: end of the segment that begins at sync[c] -- the next synchronisation point of the batch, y1
// in the last batch (`final`), kNoSync for the batch's last point otherwise (the next batch starts there)
__global__ void xr_segments(const uint64_t* sync, uint64_t n_chunks, uint64_t y1, int final, uint64_t* seg_end,
                            unsigned long long* state) {
  const uint64_t c = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (c >= n_chunks) return;
  const uint64_t a = sync[c];
  uint64_t b = kNoSync;
  if (a != kNoSync) {
    for (uint64_t k = c + 1; k < n_chunks && b == kNoSync; k++) b = sync[k];
    if (b == kNoSync) {
      atomicMax(&state[kXrLastSync], static_cast<unsigned long long>(a));
      if (final) b = y1;
    }
    if (b != kNoSync) atomicMax(&state[kXrMaxGap], static_cast<unsigned long long>(b - a));
  }
  seg_end[c] = b;
}

__global__ void xr_begin_batch(unsigned long long* state, uint64_t ys) {
  state[kXrLastSync] = ys;
  state[kXrMaxGap] = 0;
}

struct LdsRing {
  int64_t* base;  // + lane; slot stride = kReplayLanes
  __device__ int64_t& operator()(int slot) const { return base[slot * kReplayLanes]; }
};
struct GlobalRing {
  int64_t* base;  // + lane; slot stride = total lanes
  uint64_t stride;
  __device__ int64_t& operator()(int slot) const { return base[static_cast<uint64_t>(slot) * stride]; }
};

template <bool LDS>
__global__ void __launch_bounds__(kReplayLanes)
xr_replay(DevGraph G, const uint8_t* t, uint64_t n, uint64_t ys, const uint64_t* sync, const uint64_t* seg_end, uint64_t n_chunks,
          int64_t* ring_mem, uint64_t* scratch, uint32_t* counts) {
  extern __shared__ int64_t lds_ring[];
  const uint64_t lanes = static_cast<uint64_t>(gridDim.x) * kReplayLanes;
  const uint64_t lane = static_cast<uint64_t>(blockIdx.x) * kReplayLanes + threadIdx.x;
  for (uint64_t c = lane; c < n_chunks; c += lanes) {
    const uint64_t a = sync[c], b = seg_end[c];
    uint32_t m = 0;
    if (a != kNoSync && b != kNoSync) {
      uint64_t* out = scratch + 2 * (a - ys);
      if (LDS) m = static_cast<uint32_t>(rj_replay_segment(G, t, n, a, b, LdsRing{lds_ring + threadIdx.x}, out));
      else m = static_cast<uint32_t>(rj_replay_segment(G, t, n, a, b, GlobalRing{ring_mem + lane, lanes}, out));
    }
    counts[c] = m;
  }
}

// one workgroup: offs[c] = total so far + exclusive prefix of counts; total += sum
__global__ void __launch_bounds__(1024) xr_offsets(const uint32_t* counts, uint64_t n_chunks, uint64_t* offs, unsigned long long* state) {
  __shared__ uint64_t part[1024];
  const uint64_t per = (n_chunks + 1023) / 1024;
  const uint64_t lo = std::min(n_chunks, threadIdx.x * per), hi = std::min(n_chunks, lo + per);
  uint64_t sum = 0;
  for (uint64_t c = lo; c < hi; c++) sum += counts[c];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const uint64_t v = threadIdx.x >= static_cast<unsigned>(d) ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  const uint64_t base = state[kXrTotal];
  uint64_t run = base + part[threadIdx.x] - sum;
  for (uint64_t c = lo; c < hi; c++) {
    offs[c] = run;
    run += counts[c];
  }
  __syncthreads();
  if (threadIdx.x == 1023) state[kXrTotal] = base + part[1023];
}

// a wave per chunk: the segment's pairs to their place in the result
__global__ void xr_compact(const uint64_t* sync, const uint32_t* counts, const uint64_t* offs, uint64_t n_chunks, uint64_t ys,
                           const uint64_t* scratch, uint64_t* out, uint64_t cap) {
  const uint64_t c = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const uint32_t l = threadIdx.x & 63;
  if (c >= n_chunks) return;
  const uint32_t m = counts[c];
  if (m == 0) return;
  const uint64_t* from = scratch + 2 * (sync[c] - ys);
  const uint64_t to = offs[c];
  for (uint32_t j = l; j < 2 * m; j += 64)
    if (to + (j >> 1) < cap) out[2 * to + j] = from[j];
}

template <int NQ>
int run_exact_nq(rj_scan* s, const uint8_t* d_text, uint64_t n, uint64_t sb, uint64_t se, hipStream_t st) {
  const rj_program* rp = s->prog;
  const DevProgram& P = rp->dev;
  const DevGraph& G = rp->graph;
  RJ_HIP(s->xr_state.reserve(kXrStateSize * sizeof(unsigned long long)));
  unsigned long long* state = s->xr_state.as<unsigned long long>();
  unsigned long long h[kXrStateSize];
  uint64_t y0 = 0, y1 = n + 1;
  if (!(sb == 0 && se == n + 1)) {
    hipLaunchKernelGGL(xr_lookup<NQ>, dim3(1), dim3(64), 0, st, P, d_text, n, sb, se, state);
    RJ_HIP(hipMemcpyAsync(h, state, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    RJ_HIP(hipStreamSynchronize(st));
    y0 = h[kXrY0];
    y1 = h[kXrY1];
  }
  const int slots = G.n_states * G.times;
  const bool lds = slots <= kLdsRingSlots;
  // (the pairs go to a buffer of their own: when a later batch turns out not to be replayable the caller
  // keeps the result it has)
  for (int attempt = 0; attempt < 2; attempt++) {
    if (y0 >= y1) {  // nothing begins in this range
      s->result_count = 0;
      s->result = s->out.as<uint64_t>();
      return 1;
    }
    RJ_HIP(hipMemsetAsync(state + kXrTotal, 0, sizeof(unsigned long long), st));
    if (s->xr_out_cap < (y1 - y0) / 8 + 1024) {  // (a first guess; the count decides below)
      s->xr_out_cap = (y1 - y0) / 8 + 1024;
      RJ_HIP(s->xr_out.reserve(s->xr_out_cap * 2 * sizeof(uint64_t)));
    }
    const uint64_t all_chunks = (y1 - y0 + kChunk - 1) / kChunk;
    const uint64_t cap_chunks = std::min(all_chunks, kBatchChunks);
    RJ_HIP(s->xr_sync.reserve(cap_chunks * sizeof(uint64_t)));
    RJ_HIP(s->xr_seg_end.reserve(cap_chunks * sizeof(uint64_t)));
    RJ_HIP(s->xr_offs.reserve(cap_chunks * sizeof(uint64_t)));
    RJ_HIP(s->xr_counts.reserve(cap_chunks * sizeof(uint32_t)));
    RJ_HIP(s->xr_scratch.reserve((cap_chunks * kChunk + 1) * 2 * sizeof(uint64_t)));
    const int replay_blocks = static_cast<int>(std::min<uint64_t>((cap_chunks + kReplayLanes - 1) / kReplayLanes, 2048));
    if (!lds) RJ_HIP(s->ring.reserve(static_cast<size_t>(replay_blocks) * kReplayLanes * slots * sizeof(int64_t)));
    for (uint64_t ys = y0; ys < y1;) {
      const uint64_t nb = std::min((y1 - ys + kChunk - 1) / kChunk, kBatchChunks);
      const int final = ys + nb * kChunk >= y1 ? 1 : 0;
      hipLaunchKernelGGL(xr_begin_batch, dim3(1), dim3(1), 0, st, state, ys);
      const int grid = static_cast<int>((nb + 255) / 256);
      hipLaunchKernelGGL(xr_chunk_sync<NQ>, dim3(grid), dim3(256), 0, st, P, d_text, n, ys, y1, nb, s->xr_sync.as<uint64_t>());
      hipLaunchKernelGGL(xr_segments, dim3(grid), dim3(256), 0, st, s->xr_sync.as<uint64_t>(), nb, y1, final,
                         s->xr_seg_end.as<uint64_t>(), state);
      RJ_HIP(hipMemcpyAsync(h, state, sizeof(h), hipMemcpyDeviceToHost, st));
      RJ_HIP(hipStreamSynchronize(st));
      const uint64_t last = h[kXrLastSync];
      // the batch's last point opens a segment that the next batch replays; when it is the batch's first
      // point as well, more than a batch of text has no synchronisation point
      if (h[kXrMaxGap] > kMaxSegment || (!final && last == ys)) return 0;
      const int rb = static_cast<int>(std::min<uint64_t>((nb + kReplayLanes - 1) / kReplayLanes, static_cast<uint64_t>(replay_blocks)));
      if (lds)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(xr_replay<true>), dim3(rb), dim3(kReplayLanes),
                           static_cast<size_t>(slots) * kReplayLanes * sizeof(int64_t), st, G, d_text, n, ys, s->xr_sync.as<uint64_t>(),
                           s->xr_seg_end.as<uint64_t>(), nb, nullptr, s->xr_scratch.as<uint64_t>(), s->xr_counts.as<uint32_t>());
      else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(xr_replay<false>), dim3(rb), dim3(kReplayLanes), 0, st, G, d_text, n, ys,
                           s->xr_sync.as<uint64_t>(), s->xr_seg_end.as<uint64_t>(), nb, s->ring.as<int64_t>(),
                           s->xr_scratch.as<uint64_t>(), s->xr_counts.as<uint32_t>());
      hipLaunchKernelGGL(xr_offsets, dim3(1), dim3(1024), 0, st, s->xr_counts.as<uint32_t>(), nb, s->xr_offs.as<uint64_t>(), state);
      hipLaunchKernelGGL(xr_compact, dim3(static_cast<int>((nb * 64 + 255) / 256)), dim3(256), 0, st, s->xr_sync.as<uint64_t>(),
                         s->xr_counts.as<uint32_t>(), s->xr_offs.as<uint64_t>(), nb, ys, s->xr_scratch.as<uint64_t>(),
                         s->xr_out.as<uint64_t>(), s->xr_out_cap);
      ys = final ? y1 : last;
    }
    RJ_HIP(hipMemcpyAsync(h, state, sizeof(h), hipMemcpyDeviceToHost, st));
    RJ_HIP(hipStreamSynchronize(st));
    RJ_HIP(hipGetLastError());
    const uint64_t total = h[kXrTotal];
    if (total <= s->xr_out_cap) {
      s->result_count = total;
      s->result = s->xr_out.as<uint64_t>();
      return 1;
    }
    // the result did not fit: make room and replay once more
    s->xr_out_cap = total + 1;
    RJ_HIP(s->xr_out.reserve(s->xr_out_cap * 2 * sizeof(uint64_t)));
  }
  return rj_fail(RJ_DEVICE_ERROR, "exact replay: result kept overflowing");
}

}  // namespace

bool exact_replay_fits(const rj_program* rp) { return rp->host->q8_risk && rp->dev.n_words <= 8 && rp->graph.n_states > 0; }

int run_exact(rj_scan* s, const uint8_t* d_text, uint64_t n, uint64_t sb, uint64_t se, hipStream_t st) {
  const rj_program* rp = s->prog;
  if (!exact_replay_fits(rp)) return 0;
  if (se > n + 1) se = n + 1;
  if (rp->dev.n_words <= 2) return run_exact_nq<1>(s, d_text, n, sb, se, st);
  if (rp->dev.n_words <= 4) return run_exact_nq<2>(s, d_text, n, sb, se, st);
  return run_exact_nq<4>(s, d_text, n, sb, se, st);
}

}  // namespace rejit_amd
