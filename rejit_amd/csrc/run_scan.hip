// rejit_amd/csrc/run_scan.hip -- the kernels of run_scan.h: MatchAll of `X+` / `A L*` / `A L* B` / `X+ B` patterns in two
// passes over the text and one small scan over tile summaries, whatever the length of the runs (reference: the NFA loop's
// one long-lived thread, src/x64/codegen-x64.cc:535-581, its last accepting position :426-461, the restart behind the
// match :487-503).
//
//   run_summary  a wave per tile of 8 KiB (four iterations of 2 KiB: a lane owns 32 bytes, the next iteration's in flight):
//                the class streams of A, B and the breaks (dense_streams.h: rj_stream_range), then the tile's events in
//                text order with the wave as ONE sequential machine -- its state is the open segment's (s1, q), kept in
//                scalar registers; an iteration without a break costs two ballots, a break one trip round a loop.  Leaves
//                the tile's summary: its first break, the first A / last B before it, the matches closed by its other
//                breaks (they depend on nothing outside the tile), the segment open at its end.
//   run_resolve  ONE workgroup: the summaries composed (tiles without a break hand the pending thread on, tiles with one
//                replace it: associative), every tile's incoming state and the number of its first output pair.
//   run_emit     run_summary's walk again with the incoming state known: the pairs at their final place.
// Cost: the streams (~36 VALU per range and 32 bytes) twice; FETCH_SIZE 2 x the text.  Texts with a break every few bytes
// take one loop trip per break -- slower than dense_streams, which is tried first for the patterns it takes.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "kernels.h"
#include "run_scan.h"

namespace rejit_amd {

namespace {

constexpr int kWave = 64;
constexpr uint32_t kIterBytes = 2048;
constexpr unsigned long long kNone = ~0ull, kBlocked = ~0ull - 1;

__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x) & (kWave - 1); }

// the three streams of the lane's 32 bytes at `at`: A, B, breaks (a byte outside L; the text's end is one)
__device__ __forceinline__ void run_streams_of(const RunParams& a, uint64_t at, const uint4& v0, const uint4& v1, bool loaded, uint32_t* SA, uint32_t* SB,
                                               uint32_t* BR) {
  uint32_t x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
  uint32_t valid = ~0u;
  if (!loaded) {   // (an iteration that touches the end of the text: byte by byte)
    valid = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) x[q] = 0;
#pragma unroll 1
    for (int j = 0; j < 32; j++)
      if (at + j < a.n) {
        x[j >> 2] |= static_cast<uint32_t>(a.text[at + j]) << (8 * (j & 3));
        valid |= 1u << j;
      }
  }
  uint32_t x7[8], lowh[8], highh[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    x7[i] = x[i] & 0x7f7f7f7fu;
    lowh[i] = ~x[i] & 0x80808080u;
    highh[i] = x[i] & 0x80808080u;
  }
  const RunPlan& pl = a.plan;
  uint32_t sa = 0, sl = 0, sb = 0;
#pragma unroll
  for (int r = 0; r < kRunMaxRanges; r++) {
    if (static_cast<uint32_t>(r) >= pl.n_ranges) break;   // (wave-uniform)
    uint32_t T;
    if ((pl.high_half >> r) & 1u) T = rj_stream_range(x7, highh, pl.add_lo[r], pl.add_hi[r]);
    else T = rj_stream_range(x7, lowh, pl.add_lo[r], pl.add_hi[r]);
    sa |= ((pl.a_ranges >> r) & 1u) ? T : 0u;
    sl |= ((pl.l_ranges >> r) & 1u) ? T : 0u;
    sb |= ((pl.b_ranges >> r) & 1u) ? T : 0u;
  }
  sa = (pl.a_neg ? ~sa : sa) & valid;
  sl = (pl.l_neg ? ~sl : sl) & valid;
  sb = pl.has_b ? ((pl.b_neg ? ~sb : sb) & valid) : 0u;
  uint32_t br = ~sl & valid;
  if (a.n >= at && a.n - at < 32) br |= 1u << static_cast<uint32_t>(a.n - at);   // the end of the text closes the last segment
  *SA = sa;
  *SB = sb;
  *BR = br;
}

// bits of the lane's word whose index lane * 32 + bit lies in [lo, hi)
__device__ __forceinline__ uint32_t clip(uint32_t m, int lane, uint32_t lo, uint32_t hi) {
  const uint32_t base = static_cast<uint32_t>(lane) * 32u;
  const uint32_t l = lo > base ? lo - base : 0u, h = hi > base ? hi - base : 0u;
  const uint32_t below_h = h >= 32u ? ~0u : ((1u << h) - 1u), below_l = l >= 32u ? ~0u : ((1u << l) - 1u);
  return m & below_h & ~below_l;
}
// index of the first / last set bit over the wave (kIterBytes: none)
__device__ __forceinline__ uint32_t first_set(uint32_t m) {
  const uint64_t b = __ballot(m != 0);
  if (b == 0) return kIterBytes;
  const int l = __builtin_ctzll(b);
  const uint32_t w = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(m), l));
  return static_cast<uint32_t>(l) * 32u + static_cast<uint32_t>(__builtin_ctz(w));
}
__device__ __forceinline__ uint32_t last_set(uint32_t m) {
  const uint64_t b = __ballot(m != 0);
  if (b == 0) return kIterBytes;
  const int l = 63 - __builtin_clzll(b);
  const uint32_t w = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(m), l));
  return static_cast<uint32_t>(l) * 32u + 31u - static_cast<uint32_t>(__builtin_clz(w));
}

struct Open {
  unsigned long long s, q;   // the open segment's first A (kNone / kBlocked) and last B behind it (kNone)
};

__device__ __forceinline__ bool real(unsigned long long s) { return s < kBlocked; }

// The events of one iteration in text order.  on_close(state, r): the break at absolute position r closes the segment.
// any_b (may be null): the last B position met before the first close of this call chain (the caller resets it).
template <class Close>
__device__ __forceinline__ void run_iteration(const RunParams& a, uint64_t it_base, uint32_t SA, uint32_t SB, uint32_t BR, Open& st, unsigned long long* any_b,
                                              bool* track_any, Close on_close) {
  const int lane = lane_id();
  const uint32_t a_lo = a.min_start > it_base ? (a.min_start - it_base < kIterBytes ? static_cast<uint32_t>(a.min_start - it_base) : kIterBytes) : 0u;
  uint32_t lo = 0;
  auto part = [&](uint32_t hi_a, uint32_t hi_b) {   // A in [lo, hi_a), B in (s1, hi_b)
    if (any_b && *track_any && a.plan.has_b) {
      const uint32_t p = last_set(clip(SB, lane, lo, hi_b));
      if (p != kIterBytes) *any_b = it_base + p;
    }
    if (st.s == kNone) {
      const uint32_t p = first_set(clip(SA, lane, lo > a_lo ? lo : a_lo, hi_a));
      if (p != kIterBytes) st.s = it_base + p;
    }
    if (a.plan.has_b && real(st.s)) {
      const uint32_t from = st.s >= it_base ? static_cast<uint32_t>(st.s - it_base) + 1u : 0u;
      const uint32_t p = last_set(clip(SB, lane, from > lo ? from : lo, hi_b));
      if (p != kIterBytes) st.q = it_base + p;
    }
  };
  uint32_t brw = BR;
  uint64_t brm = __ballot(brw != 0);
  while (brm != 0) {
    const int l = __builtin_ctzll(brm);
    const uint32_t w = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(brw), l));
    const uint32_t r = static_cast<uint32_t>(l) * 32u + static_cast<uint32_t>(__builtin_ctz(w));
    part(r, r + 1);          // a start lies before the break, B may be the break itself
    on_close(st, it_base + r);
    st.s = kNone;
    st.q = kNone;
    lo = r;                  // (the next segment's starts begin AT the break)
    if (lane == l) brw &= brw - 1;
    if ((w & (w - 1)) == 0) brm &= brm - 1;
  }
  part(kIterBytes, kIterBytes);
}

}  // namespace

__global__ __launch_bounds__(256) void run_summary(RunParams a) {
  const int lane = lane_id();
  const uint64_t tile = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  if (tile >= a.n_tiles) return;
  const uint64_t base = (a.first_tile + tile) * kRunTile;
  RunSummary sum;
  sum.r1 = kNone;
  sum.a1 = sum.b1 = sum.b1a = kNone;
  sum.open_s = sum.open_q = kNone;
  sum.cnt = 0;
  Open st{kNone, kNone};
  unsigned long long any_b = kNone;
  bool before_first = true;
  unsigned long long cnt = 0;
  constexpr int kIters = static_cast<int>(kRunTile / kIterBytes);
  uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
  bool loaded = false;
  auto fetch = [&](int it) {
    const uint64_t at = base + static_cast<uint64_t>(it) * kIterBytes + static_cast<uint64_t>(lane) * 32;
    loaded = base + static_cast<uint64_t>(it + 1) * kIterBytes <= a.n;   // (wave-uniform)
    if (loaded) {
      v0 = *reinterpret_cast<const uint4*>(a.text + at);
      v1 = *reinterpret_cast<const uint4*>(a.text + at + 16);
    }
  };
  fetch(0);
#pragma unroll 1
  for (int it = 0; it < kIters; it++) {
    const uint64_t it_base = base + static_cast<uint64_t>(it) * kIterBytes;
    if (it_base > a.n) break;   // (nothing here, not even the text's end)
    uint32_t SA, SB, BR;
    run_streams_of(a, it_base + static_cast<uint64_t>(lane) * 32, v0, v1, loaded, &SA, &SB, &BR);
    if (it + 1 < kIters) fetch(it + 1);
    run_iteration(a, it_base, SA, SB, BR, st, &any_b, &before_first, [&](const Open& o, uint64_t r) {
      if (before_first) {   // the tile's first break: what it closes depends on the tiles before
        sum.r1 = r;
        sum.a1 = o.s;
        sum.b1a = o.q;
        sum.b1 = any_b;
        before_first = false;
      } else if (real(o.s) && (!a.plan.has_b || o.q != kNone) && o.s >= a.sb && o.s < a.se) {
        cnt++;
      }
    });
  }
  if (before_first) {   // no break in the tile
    sum.a1 = st.s;
    sum.b1a = st.q;
    sum.b1 = any_b;
  } else {
    sum.open_s = st.s;
    sum.open_q = st.q;
  }
  sum.cnt = cnt;
  if (lane == 0) a.summaries[tile] = sum;
}

namespace {

// a tile (or a run of tiles) as a function on the open segment's state
struct Elem {
  unsigned long long hb;   // holds a break
  unsigned long long a1, b1, b1a, open_s, open_q;
};
__device__ __forceinline__ Elem elem_of(const RunSummary& s) { return Elem{s.r1 != kNone ? 1ull : 0ull, s.a1, s.b1, s.b1a, s.open_s, s.open_q}; }
// the state behind a stretch without a break
__device__ __forceinline__ Open pass_on(const Elem& g, Open in) {
  Open o;
  if (in.s == kNone) {
    o.s = g.a1;
    o.q = g.a1 != kNone ? g.b1a : kNone;
  } else {
    o.s = in.s;
    o.q = (real(in.s) && g.b1 != kNone) ? g.b1 : in.q;
  }
  return o;
}
// f, then g
__device__ __forceinline__ Elem compose(const Elem& f, const Elem& g) {
  Elem e;
  if (!f.hb) {   // the stretch before the composite's first break: f's whole, then g's beginning
    e.a1 = f.a1 != kNone ? f.a1 : g.a1;
    e.b1 = g.b1 != kNone ? g.b1 : f.b1;
    e.b1a = f.a1 != kNone ? (g.b1 != kNone ? g.b1 : f.b1a) : g.b1a;
  } else {
    e.a1 = f.a1;
    e.b1 = f.b1;
    e.b1a = f.b1a;
  }
  e.hb = (f.hb | g.hb) ? 1ull : 0ull;
  if (g.hb) {
    e.open_s = g.open_s;
    e.open_q = g.open_q;
  } else if (f.hb) {
    const Open o = pass_on(g, Open{f.open_s, f.open_q});
    e.open_s = o.s;
    e.open_q = o.q;
  } else {
    e.open_s = e.open_q = kNone;
  }
  return e;
}
// the state behind an element; *emits: its first break closes a match that counts
__device__ __forceinline__ Open apply(const RunParams& a, const Elem& g, Open in, bool* emits) {
  *emits = false;
  if (!g.hb) return pass_on(g, in);
  const Open at_break = pass_on(g, in);
  *emits = real(at_break.s) && (!a.plan.has_b || at_break.q != kNone) && at_break.s >= a.sb && at_break.s < a.se;
  return Open{g.open_s, g.open_q};
}

}  // namespace

// ONE workgroup of 1024 threads: thread t owns the tiles [t C, (t + 1) C).  Its tiles composed into one element; an inclusive
// scan of the 1024 elements in LDS (ten rounds: the composition is associative; the first version let thread 0 walk them one after
// the other -- 197 us of the 280 us a 64 MiB text took); the element BEFORE a thread's chunk applied to the run's initial state is
// the chunk's incoming state; the chunk walked again with it: every tile's incoming state and count; the counts' prefix sums the
// same way.
__global__ __launch_bounds__(1024) void run_resolve(RunParams a) {
  __shared__ Elem chunk[2][1024];
  __shared__ unsigned long long sums[2][1024];
  const uint32_t t = threadIdx.x;
  const uint64_t C = (a.n_tiles + 1023) / 1024;
  const uint64_t lo = static_cast<uint64_t>(t) * C, hi = lo + C < a.n_tiles ? lo + C : a.n_tiles;
  const Elem identity{0, kNone, kNone, kNone, kNone, kNone};   // (a stretch without a break, an A or a B hands every state on)
  // (the loops over a thread's tiles load four summaries at a time: one thread's loads, issued one after the other and each
  // waited for, were most of this kernel -- 58 us for the 8192 tiles of a 64 MiB text, three passes of eight dependent trips)
  constexpr int kBatch = 4;
  Elem e = identity;
  for (uint64_t i = lo; i < hi; i += kBatch) {
    RunSummary sm[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; k++) sm[k] = a.summaries[i + k < hi ? i + k : i];
#pragma unroll
    for (int k = 0; k < kBatch; k++)
      if (i + k < hi) e = (i + k == lo) ? elem_of(sm[k]) : compose(e, elem_of(sm[k]));
  }
  int cur = 0;
  chunk[0][t] = e;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    Elem v = chunk[cur][t];
    if (t >= d) v = compose(chunk[cur][t - d], v);
    chunk[cur ^ 1][t] = v;
    cur ^= 1;
    __syncthreads();
  }
  const Open initial{a.blocked_in ? kBlocked : kNone, kNone};
  bool emits;
  Open st = t == 0 ? initial : apply(a, chunk[cur][t - 1], initial, &emits);
  unsigned long long total = 0;
  for (uint64_t i = lo; i < hi; i += kBatch) {
    RunSummary sm[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; k++) sm[k] = a.summaries[i + k < hi ? i + k : i];
#pragma unroll
    for (int k = 0; k < kBatch; k++) {
      if (i + k >= hi) break;
      const Open next = apply(a, elem_of(sm[k]), st, &emits);
      RunTileIn ti;
      ti.s = st.s;
      ti.q = st.q;
      ti.off = sm[k].cnt + (emits ? 1ull : 0ull);   // (the tile's count for now: turned into its offset below)
      ti.pad = 0;
      a.tile_in[i + k] = ti;
      total += ti.off;
      st = next;
    }
  }
  int sc = 0;
  sums[0][t] = total;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    unsigned long long v = sums[sc][t];
    if (t >= d) v += sums[sc][t - d];
    sums[sc ^ 1][t] = v;
    sc ^= 1;
    __syncthreads();
  }
  if (t == 1023) {
    const unsigned long long run = sums[sc][1023];
    a.counters[kCntFinal] = run;
    a.counters[kCntCands] = run;
    a.counters[kCntHits] = run;
    if (a.host_counters) a.host_counters[kCntFinal] = run;
  }
  unsigned long long off = sums[sc][t] - total;
  for (uint64_t i = lo; i < hi; i += kBatch) {
    unsigned long long c[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; k++) c[k] = a.tile_in[i + k < hi ? i + k : i].off;
#pragma unroll
    for (int k = 0; k < kBatch; k++) {
      if (i + k >= hi) break;
      a.tile_in[i + k].off = off;
      off += c[k];
    }
  }
}

__global__ __launch_bounds__(256) void run_emit(RunParams a) {
  const int lane = lane_id();
  const uint64_t tile = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  if (tile >= a.n_tiles) return;
  const uint64_t base = (a.first_tile + tile) * kRunTile;
  const RunTileIn in = a.tile_in[tile];
  Open st{in.s, in.q};
  unsigned long long pos = in.off;
  bool no_track = false;
  constexpr int kIters = static_cast<int>(kRunTile / kIterBytes);
  uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
  bool loaded = false;
  auto fetch = [&](int it) {
    const uint64_t at = base + static_cast<uint64_t>(it) * kIterBytes + static_cast<uint64_t>(lane) * 32;
    loaded = base + static_cast<uint64_t>(it + 1) * kIterBytes <= a.n;
    if (loaded) {
      v0 = *reinterpret_cast<const uint4*>(a.text + at);
      v1 = *reinterpret_cast<const uint4*>(a.text + at + 16);
    }
  };
  fetch(0);
#pragma unroll 1
  for (int it = 0; it < kIters; it++) {
    const uint64_t it_base = base + static_cast<uint64_t>(it) * kIterBytes;
    if (it_base > a.n) break;
    uint32_t SA, SB, BR;
    run_streams_of(a, it_base + static_cast<uint64_t>(lane) * 32, v0, v1, loaded, &SA, &SB, &BR);
    if (it + 1 < kIters) fetch(it + 1);
    run_iteration(a, it_base, SA, SB, BR, st, nullptr, &no_track, [&](const Open& o, uint64_t r) {
      if (real(o.s) && (!a.plan.has_b || o.q != kNone) && o.s >= a.sb && o.s < a.se) {
        if (lane == 0 && pos < a.out_cap) *reinterpret_cast<ulonglong2*>(a.out + 2 * pos) = make_ulonglong2(o.s, a.plan.has_b ? o.q + 1 : r);
        pos++;
      }
    });
  }
}

uint64_t run_tiles(uint64_t sb, uint64_t n, uint64_t* first_tile) {
  *first_tile = sb / kRunTile;
  return n / kRunTile - *first_tile + 1;   // (the tile that holds position n -- the text's end -- is the last)
}

void launch_run_summary(const RunParams& a, hipEvent_t t0, hipEvent_t t1, hipStream_t st) {
  const unsigned grid = static_cast<unsigned>((a.n_tiles + 3) / 4);
  hipExtLaunchKernelGGL(run_summary, dim3(grid), dim3(256), 0, st, t0, t1, 0, a);
}
void launch_run_resolve(const RunParams& a, hipStream_t st) { hipLaunchKernelGGL(run_resolve, dim3(1), dim3(1024), 0, st, a); }
void launch_run_emit(const RunParams& a, hipEvent_t t1, hipStream_t st) {
  const unsigned grid = static_cast<unsigned>((a.n_tiles + 3) / 4);
  hipExtLaunchKernelGGL(run_emit, dim3(grid), dim3(256), 0, st, nullptr, t1, 0, a);
}

}  // namespace rejit_amd
