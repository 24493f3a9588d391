// rejit_amd/csrc/engine.hip -- host orchestration of the device pipeline and the C ABI
// (include/rejit_hip.h).  Replaces the reference's API shim src/rejit.cc:150-267 (lazy
// Compile + indirect call into JIT code) and its result sink src/codegen.cc:36-86.
//
//   rj_compile : parse -> lower -> upload the automaton tables to HBM once
//   rj_scan_run: memset counters -> scan kernel(s) -> verify kernel -> finalize kernel
//                -> ONE host synchronisation to read {hits, candidates, final count};
//                lists that overflowed are grown and the run repeated; more than
//                kFinalizeCap candidates take the rocPRIM radix-sort path.
//
// There is no CPU matching code in this library: without a working HIP device every
// entry point fails with RJ_DEVICE_ERROR.
#include <cstring>  // must precede rocprim (its texture iterator uses ::memset)

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rejit_hip.h"
#include "device_program.h"
#include "kernels.h"
#include "lowering.h"

using namespace rejit_amd;

namespace {

thread_local std::string g_error;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
  return code;
}

#define RJ_HIP(call)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail(RJ_DEVICE_ERROR, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// The reference's generated code never touches errno, and its callers rely on that
// (sample/jrep.cc:281-285 tests `if (errno)` right after mmap).  The HIP runtime does set it
// (probing files, ioctls), so every entry point restores the caller's value.
struct ErrnoGuard {
  int saved;
  ErrnoGuard() : saved(errno) {}
  ~ErrnoGuard() { errno = saved; }
};

// The HIP runtime's own load-time initialisers (they run before ours: dependency order) can
// leave errno set before main() starts -- e.g. ENOENT from probing amdgpu.ids.  A process that
// links the reference's librejit.a starts with errno == 0, so restore that.
__attribute__((constructor)) void rj_library_loaded() { errno = 0; }

struct DeviceBuffer {
  void* p = nullptr;
  size_t bytes = 0;
  ~DeviceBuffer() {
    if (p) (void)hipFree(p);
  }
  // grow-only; contents are NOT preserved
  hipError_t reserve(size_t want) {
    if (want <= bytes) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) bytes = want;
    return e;
  }
  template <class T>
  T* as() const { return static_cast<T*>(p); }
};

}  // namespace

struct rj_program {
  std::unique_ptr<Program> host;
  DevProgram dev{};
  DeviceBuffer tables;
  DevGraph graph{};        // uploaded only for patterns with q8_risk
  DeviceBuffer graph_blob;
  int device = 0;
  int window_alphabet = 0;  // distinct byte values among the fixed window bytes
  std::string pattern;
};

struct rj_scan {
  const rj_program* prog = nullptr;
  DeviceBuffer counters, hits, cand_begin, cand_end, out, keys_out, vals_out, sort_tmp, flag;
  DeviceBuffer scan_a, scan_b, taken;  // large-path selection scratch
  DeviceBuffer ring;                   // exact sequential kernel
  uint64_t hits_cap = 0, cands_cap = 0, out_cap = 0;
  unsigned long long* host_counters = nullptr;  // pinned
  int* host_flag = nullptr;                     // pinned
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  rj_stats stats{};
  const uint64_t* result = nullptr;  // device pointer to the final pairs
  uint64_t result_count = 0;
  // host-text path
  DeviceBuffer text;
  hipStream_t own_stream = nullptr;
};

namespace {

int upload_program(rj_program* rp) {
  const Program& P = *rp->host;
  const int W = P.n_words;
  const int C = P.has_assertions ? kNumCtx : 1;
  const int R = std::max(P.n_rows, 1);
  const int Pn = std::max(P.n_pos, 1);
  // layout in 32-bit words
  size_t off_first = 0;
  size_t off_last = off_first + static_cast<size_t>(C) * W;
  size_t off_linear = off_last + static_cast<size_t>(C) * W;
  size_t off_rowof = off_linear + W;
  size_t off_rows = off_rowof + Pn;
  size_t off_cls = off_rows + static_cast<size_t>(C) * R * W;
  size_t total = off_cls + static_cast<size_t>(256) * W;
  std::vector<uint32_t> blob(total, 0u);
  for (int c = 0; c < C; c++) {
    std::copy(P.first[c].begin(), P.first[c].end(), blob.begin() + off_first + static_cast<size_t>(c) * W);
    std::copy(P.last[c].begin(), P.last[c].end(), blob.begin() + off_last + static_cast<size_t>(c) * W);
    for (int r = 0; r < P.n_rows; r++)
      std::copy(P.rows[c].begin() + static_cast<long>(r) * W, P.rows[c].begin() + static_cast<long>(r + 1) * W,
                blob.begin() + off_rows + (static_cast<size_t>(c) * R + r) * W);
  }
  std::copy(P.linear.begin(), P.linear.end(), blob.begin() + off_linear);
  for (int i = 0; i < P.n_pos; i++) blob[off_rowof + i] = static_cast<uint32_t>(P.row_of[static_cast<size_t>(i)]);
  std::copy(P.cls.begin(), P.cls.end(), blob.begin() + off_cls);

  RJ_HIP(hipGetDevice(&rp->device));
  RJ_HIP(rp->tables.reserve(total * sizeof(uint32_t)));
  RJ_HIP(hipMemcpy(rp->tables.p, blob.data(), total * sizeof(uint32_t), hipMemcpyHostToDevice));
  const uint32_t* base = rp->tables.as<uint32_t>();
  DevProgram& D = rp->dev;
  D.n_pos = P.n_pos;
  D.n_words = W;
  D.n_ctx = C;
  D.n_rows = R;
  D.nullable = 0;
  for (int c = 0; c < kNumCtx; c++)
    if (P.nullable[C == 1 ? 0 : c]) D.nullable |= 1u << c;
  D.mode = P.mode == ScanMode::Windows ? 1 : 0;
  D.n_windows = static_cast<int>(P.windows.size());
  D.win_offset = P.windows.empty() ? 0 : P.windows[0].offset;
  D.win_len = P.windows.empty() ? 0 : P.windows[0].len;
  for (int k = 0; k < kDevMaxWindows; k++) {
    // unused slots repeat the last window, so kernels instantiated for a larger K stay exact
    const FFWindow w = P.windows.empty() ? FFWindow{} : P.windows[std::min<size_t>(static_cast<size_t>(k), P.windows.size() - 1)];
    D.win_value0[k] = w.value0;
    D.win_mask0[k] = w.mask0;
    D.win_value1[k] = w.value1;
    D.win_mask1[k] = w.mask1;
  }
  {
    ByteSet seen;
    for (const FFWindow& w : P.windows)
      for (uint32_t k = 0; k < w.len; k++) {
        const uint32_t m = k < 4 ? (w.mask0 >> (8 * k)) : (w.mask1 >> (8 * (k - 4)));
        const uint32_t v = k < 4 ? (w.value0 >> (8 * k)) : (w.value1 >> (8 * (k - 4)));
        if (m & 0xFFu) seen.add(static_cast<uint8_t>(v & 0xFFu));
      }
    rp->window_alphabet = seen.count();
  }
  for (int k = 0; k < 8; k++) D.first_bytes[k] = P.first_bytes.w[k];
  D.min_len = P.min_len;
  D.first = base + off_first;
  D.last = base + off_last;
  D.linear = base + off_linear;
  D.row_of = reinterpret_cast<const int32_t*>(base + off_rowof);
  D.rows = base + off_rows;
  D.cls = base + off_cls;
  if (P.q8_risk) {
    // graph for the exact sequential kernel: int32 arrays, then class bitmaps, then literal bytes
    const Graph& g = P.graph;
    const size_t nb = g.byte_edges.size(), nc = g.control_edges.size();
    std::vector<int32_t> ints;
    std::vector<uint32_t> classes;
    std::string lits;
    std::vector<int32_t> be_src, be_dst, be_len, be_off, ce_src, ce_dst, ce_kind;
    size_t longest = 1;
    for (const ByteEdge& e : g.byte_edges) {
      be_src.push_back(e.src);
      be_dst.push_back(e.dst);
      if (!e.bytes.empty()) {
        be_len.push_back(static_cast<int32_t>(e.bytes.size()));
        be_off.push_back(static_cast<int32_t>(lits.size()));
        lits += e.bytes;
        longest = std::max(longest, e.bytes.size());
      } else {
        be_len.push_back(0);
        be_off.push_back(static_cast<int32_t>(classes.size() / 8));
        for (int k = 0; k < 8; k++) classes.push_back(e.cls.w[k]);
      }
    }
    for (const ControlEdge& c : g.control_edges) {
      ce_src.push_back(c.src);
      ce_dst.push_back(c.dst);
      ce_kind.push_back(c.kind == ControlKind::Epsilon ? 0 : c.kind == ControlKind::StartOfLine ? 1 : 2);
    }
    const size_t words = 4 * nb + 3 * nc + classes.size();
    const size_t bytes = words * 4 + lits.size() + 16;
    std::vector<uint8_t> gb(bytes, 0);
    uint32_t* w = reinterpret_cast<uint32_t*>(gb.data());
    size_t o = 0;
    auto put = [&](const std::vector<int32_t>& v) {
      size_t at = o;
      if (!v.empty()) memcpy(w + o, v.data(), v.size() * 4);
      o += v.size();
      return at;
    };
    const size_t o_src = put(be_src), o_dst = put(be_dst), o_len = put(be_len), o_off = put(be_off);
    const size_t o_cs = put(ce_src), o_cd = put(ce_dst), o_ck = put(ce_kind);
    const size_t o_cls = o;
    if (!classes.empty()) memcpy(w + o, classes.data(), classes.size() * 4);
    o += classes.size();
    if (!lits.empty()) memcpy(gb.data() + o * 4, lits.data(), lits.size());
    RJ_HIP(rp->graph_blob.reserve(bytes));
    RJ_HIP(hipMemcpy(rp->graph_blob.p, gb.data(), bytes, hipMemcpyHostToDevice));
    const uint32_t* gbase = rp->graph_blob.as<uint32_t>();
    DevGraph& G = rp->graph;
    G.n_states = g.n_states;
    G.entry = g.entry;
    G.exit = g.exit;
    G.n_byte_edges = static_cast<int32_t>(nb);
    G.n_control_edges = static_cast<int32_t>(nc);
    G.times = 1 + static_cast<int32_t>(std::min<size_t>(longest, 64));
    G.be_src = reinterpret_cast<const int32_t*>(gbase + o_src);
    G.be_dst = reinterpret_cast<const int32_t*>(gbase + o_dst);
    G.be_len = reinterpret_cast<const int32_t*>(gbase + o_len);
    G.be_off = reinterpret_cast<const int32_t*>(gbase + o_off);
    G.ce_src = reinterpret_cast<const int32_t*>(gbase + o_cs);
    G.ce_dst = reinterpret_cast<const int32_t*>(gbase + o_cd);
    G.ce_kind = reinterpret_cast<const int32_t*>(gbase + o_ck);
    G.cls = gbase + o_cls;
    G.lit = reinterpret_cast<const uint8_t*>(gbase + o);
  }
  return RJ_OK;
}

int ensure_lists(rj_scan* s, uint64_t hits_cap, uint64_t cands_cap) {
  if (hits_cap > s->hits_cap) {
    RJ_HIP(s->hits.reserve(hits_cap * sizeof(uint64_t)));
    s->hits_cap = hits_cap;
  }
  if (cands_cap > s->cands_cap) {
    RJ_HIP(s->cand_begin.reserve(cands_cap * sizeof(uint64_t)));
    RJ_HIP(s->cand_end.reserve(cands_cap * sizeof(uint64_t)));
    RJ_HIP(s->out.reserve(cands_cap * 2 * sizeof(uint64_t)));
    s->cands_cap = cands_cap;
    s->out_cap = cands_cap;
  }
  return RJ_OK;
}

// Large path: more candidates than finalize_small sorts in LDS.
int finalize_large(rj_scan* s, uint64_t n_cands, uint64_t text_len, const FinalizeParams& fp, hipStream_t st) {
  s->stats.large_path = 1;
  RJ_HIP(s->keys_out.reserve(n_cands * sizeof(uint64_t)));
  RJ_HIP(s->vals_out.reserve(n_cands * sizeof(uint64_t)));
  uint64_t* kin = s->cand_begin.as<uint64_t>();
  uint64_t* vin = s->cand_end.as<uint64_t>();
  // begins are < 2^bits: sort only the bits that can differ
  unsigned bits = 1;
  while (bits < 64 && (text_len >> bits) != 0) bits++;
  size_t tmp_bytes = 0;
  RJ_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, kin, s->keys_out.as<uint64_t>(), vin, s->vals_out.as<uint64_t>(),
                                   n_cands, 0, bits, st));
  RJ_HIP(s->sort_tmp.reserve(std::max<size_t>(tmp_bytes, 16)));
  RJ_HIP(rocprim::radix_sort_pairs(s->sort_tmp.p, tmp_bytes, kin, s->keys_out.as<uint64_t>(), vin,
                                   s->vals_out.as<uint64_t>(), n_cands, 0, bits, st));
  if (fp.detect_adjacent)
    launch_detect_adjacent(s->keys_out.as<uint64_t>(), s->vals_out.as<uint64_t>(), n_cands,
                           s->counters.as<unsigned long long>(), st);
  // common case: the sorted candidates already are the result
  *s->host_flag = 1;
  RJ_HIP(hipMemcpyAsync(s->flag.p, s->host_flag, sizeof(int), hipMemcpyHostToDevice, st));
  launch_check_and_interleave(s->keys_out.as<uint64_t>(), s->vals_out.as<uint64_t>(), n_cands, fp.carry_cur,
                              s->out.as<uint64_t>(), s->out_cap, s->flag.as<int>(), st);
  RJ_HIP(hipMemcpyAsync(s->host_flag, s->flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
  RJ_HIP(hipStreamSynchronize(st));
  RJ_HIP(hipGetLastError());
  if (fp.detect_adjacent)
    RJ_HIP(hipMemcpy(s->host_counters + kCntAdjacent, s->counters.as<unsigned long long>() + kCntAdjacent,
                     sizeof(unsigned long long), hipMemcpyDeviceToHost));
  if (*s->host_flag == 1) {
    s->result_count = n_cands;
    return RJ_OK;
  }
  // general case: cluster-parallel selection
  //   pmax  = exclusive prefix max of the ends           (scan_a)
  //   taken = per-cluster sequential walk                 (taken)
  //   last  = exclusive prefix max of (taken ? i+1 : 0)   (scan_b; cand_begin reused as scratch)
  //   keep  = taken minus the zero-length rule            (scan_a)
  //   pos   = exclusive prefix sum of keep                (scan_b)
  RJ_HIP(s->scan_a.reserve(n_cands * sizeof(uint64_t)));
  RJ_HIP(s->scan_b.reserve(n_cands * sizeof(uint64_t)));
  RJ_HIP(s->taken.reserve(n_cands));
  uint64_t* keys = s->keys_out.as<uint64_t>();
  uint64_t* vals = s->vals_out.as<uint64_t>();
  uint64_t* sa = s->scan_a.as<uint64_t>();
  uint64_t* sb = s->scan_b.as<uint64_t>();
  uint64_t* scratch = s->cand_begin.as<uint64_t>();  // free again after the sort
  auto scan_max = [&](uint64_t* in, uint64_t* out) -> hipError_t {
    size_t bytes = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, bytes, in, out, uint64_t{0}, n_cands, rocprim::maximum<uint64_t>(), st);
    if (e != hipSuccess) return e;
    e = s->sort_tmp.reserve(std::max<size_t>(bytes, 16));
    if (e != hipSuccess) return e;
    return rocprim::exclusive_scan(s->sort_tmp.p, bytes, in, out, uint64_t{0}, n_cands, rocprim::maximum<uint64_t>(), st);
  };
  auto scan_sum = [&](uint64_t* in, uint64_t* out) -> hipError_t {
    size_t bytes = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, bytes, in, out, uint64_t{0}, n_cands, rocprim::plus<uint64_t>(), st);
    if (e != hipSuccess) return e;
    e = s->sort_tmp.reserve(std::max<size_t>(bytes, 16));
    if (e != hipSuccess) return e;
    return rocprim::exclusive_scan(s->sort_tmp.p, bytes, in, out, uint64_t{0}, n_cands, rocprim::plus<uint64_t>(), st);
  };
  RJ_HIP(scan_max(vals, sa));
  launch_select_walk(keys, vals, sa, n_cands, fp.carry_cur, s->taken.as<uint8_t>(), st);
  launch_taken_index(s->taken.as<uint8_t>(), n_cands, scratch, st);
  RJ_HIP(scan_max(scratch, sb));
  launch_zero_length_rule(keys, vals, s->taken.as<uint8_t>(), sb, n_cands, fp.carry_prev_end, fp.have_prev, sa, st);
  RJ_HIP(scan_sum(sa, sb));
  launch_compact_kept(keys, vals, sa, sb, n_cands, s->out.as<uint64_t>(), s->out_cap, s->counters.as<unsigned long long>(), st);
  RJ_HIP(hipMemcpyAsync(s->host_counters, s->counters.p, kCntSize * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  RJ_HIP(hipStreamSynchronize(st));
  RJ_HIP(hipGetLastError());
  s->result_count = s->host_counters[kCntFinal];
  return RJ_OK;
}

constexpr uint64_t kExactLimit = 1u << 20;  // bytes the one-lane exact kernel is allowed to walk

int run_pipeline(rj_scan* s, const uint8_t* d_text, uint64_t n, uint64_t sb, uint64_t se, uint64_t carry_cur,
                 uint64_t carry_prev_end, int have_prev, hipStream_t st) {
  const rj_program* rp = s->prog;
  const DevProgram& D = rp->dev;
  if (se > n + 1) se = n + 1;
  s->stats = rj_stats{};
  s->result = nullptr;
  s->result_count = 0;
  if (sb >= se) return RJ_OK;
  if ((reinterpret_cast<uintptr_t>(d_text) & 15u) != 0) return fail(RJ_BAD_ARGUMENT, "device text must be 16-byte aligned");

  // Dense mode produces a hit for a sizeable fraction of the bytes: walk the starts in
  // segments so the hit list stays bounded.  Windows mode: one segment.
  const bool windows = D.mode == 1;
  const uint64_t seg = windows ? (se - sb) : std::min<uint64_t>(se - sb, 1ull << 27);
  // a workgroup appends to segment (blockIdx % kHitSegs): with few workgroups one segment may
  // receive everything, so small runs give every segment room for the whole range
  const uint64_t hits_limit = static_cast<uint64_t>(kHitSegs) * (seg + 64);
  uint64_t want_hits = windows ? std::max<uint64_t>(1u << 16, (se - sb) / 512)
                               : (seg <= (1u << 20) ? hits_limit : seg + seg / 8 + 64 * kHitSegs);
  uint64_t want_cands = std::max<uint64_t>(1u << 16, (se - sb) / 512);

  for (int attempt = 0; attempt < 8; attempt++) {
    int rc = ensure_lists(s, std::max(want_hits, s->hits_cap), std::max(want_cands, s->cands_cap));
    if (rc != RJ_OK) return rc;
    RJ_HIP(hipMemsetAsync(s->counters.p, 0, kCntSize * sizeof(unsigned long long), st));
    RJ_HIP(hipEventRecord(s->ev[0], st));
    float scan_ms_total = 0.f;
    (void)scan_ms_total;
    for (uint64_t lo = sb; lo < se; lo += seg) {
      const uint64_t hi = std::min(se, lo + seg);
      if (lo != sb)
        RJ_HIP(hipMemsetAsync(s->counters.as<unsigned long long>() + kCntHits, 0, kHitSegs * sizeof(unsigned long long), st));
      ScanParams sp{};
      sp.text = d_text;
      sp.n = n;
      sp.sb = lo;
      sp.se = hi;
      sp.hits = s->hits.as<uint64_t>();
      sp.hits_cap = s->hits_cap;
      sp.counters = s->counters.as<unsigned long long>();
      if (lo == sb) RJ_HIP(hipEventRecord(s->ev[1], st));
      if (windows) {
        WindowSet ws{};
        bool masked = false;
        for (int k = 0; k < kDevMaxWindows; k++) {
          ws.value0[k] = D.win_value0[k];
          ws.mask0[k] = D.win_mask0[k];
          ws.value1[k] = D.win_value1[k];
          ws.mask1[k] = D.win_mask1[k];
          masked |= D.win_mask0[k] != 0xFFFFFFFFu || (D.win_len > 4 && D.win_mask1[k] != 0xFFFFFFFFu);
        }
        ws.masked = masked;
        ws.two_level = rp->window_alphabet > 4;
        ws.len = D.win_len;
        ws.offset = D.win_offset;
        sp.wlo = lo + D.win_offset;
        // a window must fit into the text: w + win_len <= n
        const uint64_t last_w = n >= D.win_len ? n - D.win_len + 1 : 0;
        sp.whi = std::min(hi + D.win_offset, last_w);
        launch_scan_windows(sp, ws, D.n_windows, st);
      } else {
        launch_scan_dense(sp, D, st);
      }
      if (hi == se) RJ_HIP(hipEventRecord(s->ev[2], st));
      VerifyParams vp{};
      vp.text = d_text;
      vp.n = n;
      vp.hits = s->hits.as<uint64_t>();
      vp.hits_cap = s->hits_cap;
      vp.cand_begin = s->cand_begin.as<uint64_t>();
      vp.cand_end = s->cand_end.as<uint64_t>();
      vp.cands_cap = s->cands_cap;
      vp.counters = s->counters.as<unsigned long long>();
      launch_verify(vp, D, windows ? 65536 : (hi - lo) / 8 + 1, st);
    }
    FinalizeParams fp{};
    fp.cand_begin = s->cand_begin.as<uint64_t>();
    fp.cand_end = s->cand_end.as<uint64_t>();
    fp.cands_cap = s->cands_cap;
    fp.hits_cap = s->hits_cap;
    fp.out = s->out.as<uint64_t>();
    fp.out_cap = s->out_cap;
    fp.counters = s->counters.as<unsigned long long>();
    fp.carry_cur = carry_cur;
    fp.carry_prev_end = carry_prev_end;
    fp.have_prev = have_prev;
    // bit-exactness with the reference's ring artefact (Q8) can only be at stake when the
    // pattern is at risk AND a candidate begins exactly where another one ends
    const bool whole_text = sb == 0 && se == n + 1 && carry_cur == 0 && !have_prev;
    fp.detect_adjacent = rp->host->q8_risk && whole_text;
    launch_finalize_small(fp, st);
    RJ_HIP(hipEventRecord(s->ev[3], st));
    RJ_HIP(hipMemcpyAsync(s->host_counters, s->counters.p, kCntSize * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    RJ_HIP(hipStreamSynchronize(st));
    RJ_HIP(hipGetLastError());
    unsigned long long n_hits = 0, max_seg = 0;
    for (int k = 0; k < kHitSegs; k++) {
      n_hits += s->host_counters[kCntHits + k];
      max_seg = std::max(max_seg, s->host_counters[kCntHits + k]);
    }
    const unsigned long long n_cands = s->host_counters[kCntCands];
    s->stats.n_hits = n_hits;
    s->stats.n_candidates = n_cands;
    (void)hipEventElapsedTime(&s->stats.scan_ms, s->ev[1], s->ev[2]);
    (void)hipEventElapsedTime(&s->stats.total_ms, s->ev[0], s->ev[3]);
    if (s->host_counters[kCntOverflow] != 0 || n_cands > s->cands_cap || max_seg > s->hits_cap / kHitSegs) {
      // grow whichever list overflowed and run again
      s->stats.retries++;
      // segments fill unevenly: size for the fullest one
      want_hits = std::max<uint64_t>(s->hits_cap, std::min<uint64_t>(std::max<uint64_t>(max_seg * kHitSegs * 2, s->hits_cap * 4),
                                                                     hits_limit));
      want_cands = std::max<uint64_t>(s->cands_cap, std::min<uint64_t>(std::max<uint64_t>(n_cands * 2, s->cands_cap * 4), (se - sb) + 64));
      if (want_hits == s->hits_cap && want_cands == s->cands_cap) return fail(RJ_DEVICE_ERROR, "device lists cannot grow further");
      continue;
    }
    if (s->host_counters[kCntFinal] == ~0ull) {
      rc = finalize_large(s, n_cands, n + 1, fp, st);
      if (rc != RJ_OK) return rc;
    } else {
      s->result_count = s->host_counters[kCntFinal];
    }
    if (fp.detect_adjacent && s->host_counters[kCntAdjacent] != 0 && n <= kExactLimit) {
      // run the reference's own sequential algorithm on one lane and take ITS answer
      RJ_HIP(s->ring.reserve(static_cast<size_t>(rp->graph.times) * rp->graph.n_states * sizeof(int64_t)));
      rc = ensure_lists(s, s->hits_cap, std::max<uint64_t>(s->cands_cap, n + 2));
      if (rc != RJ_OK) return rc;
      launch_exact_sequential(d_text, n, rp->graph, s->ring.as<int64_t>(), s->out.as<uint64_t>(), s->out_cap,
                              s->counters.as<unsigned long long>(), st);
      RJ_HIP(hipMemcpyAsync(s->host_counters, s->counters.p, kCntSize * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
      RJ_HIP(hipStreamSynchronize(st));
      RJ_HIP(hipGetLastError());
      s->result_count = s->host_counters[kCntFinal];
      s->stats.exact_path = 1;
    }
    s->stats.n_matches = s->result_count;
    s->result = s->out.as<uint64_t>();
    return RJ_OK;
  }
  return fail(RJ_DEVICE_ERROR, "device lists kept overflowing");
}

int scan_init(rj_scan* s) {
  RJ_HIP(s->counters.reserve(kCntSize * sizeof(unsigned long long)));
  RJ_HIP(s->flag.reserve(16));
  RJ_HIP(hipHostMalloc(reinterpret_cast<void**>(&s->host_counters), kCntSize * sizeof(unsigned long long)));
  RJ_HIP(hipHostMalloc(reinterpret_cast<void**>(&s->host_flag), 16));
  for (auto& e : s->ev) RJ_HIP(hipEventCreate(&e));
  return RJ_OK;
}

// one scratch per (thread, program) for the host-text entry points
struct HostScans {
  std::vector<std::pair<const rj_program*, rj_scan*>> v;
  ~HostScans() {
    for (auto& p : v) rj_scan_destroy(p.second);
  }
};
thread_local HostScans g_host_scans;

int host_scan_for(const rj_program* prog, rj_scan** out) {
  for (auto& p : g_host_scans.v)
    if (p.first == prog) {
      *out = p.second;
      return RJ_OK;
    }
  rj_scan* s = nullptr;
  int rc = rj_scan_create(prog, &s);
  if (rc != RJ_OK) return rc;
  if (hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking) != hipSuccess) {
    rj_scan_destroy(s);
    return fail(RJ_DEVICE_ERROR, "hipStreamCreate failed");
  }
  if (g_host_scans.v.size() >= 16) {  // bound the cache
    rj_scan_destroy(g_host_scans.v.front().second);
    g_host_scans.v.erase(g_host_scans.v.begin());
  }
  g_host_scans.v.emplace_back(prog, s);
  *out = s;
  return RJ_OK;
}

int stage_text(rj_scan* s, const char* text, size_t n, const uint8_t** d_text) {
  RJ_HIP(s->text.reserve(((n + 64 + 4095) / 4096) * 4096));
  if (n) RJ_HIP(hipMemcpyAsync(s->text.p, text, n, hipMemcpyHostToDevice, s->own_stream));
  *d_text = s->text.as<uint8_t>();
  return RJ_OK;
}

}  // namespace

extern "C" {

const char* rj_last_error(void) { return g_error.c_str(); }

int rj_device_count(void) {
  ErrnoGuard errno_guard;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int rj_compile(const char* regexp, rj_program** out) {
  ErrnoGuard errno_guard;
  if (!regexp || !out) return fail(RJ_BAD_ARGUMENT, "null argument");
  *out = nullptr;
  g_error.clear();
  LowerResult lr = lower(regexp);
  if (lr.status != 0) {
    g_error = lr.message;
    return lr.status == kParseError ? RJ_PARSER_ERROR : RJ_TOO_LARGE;
  }
  auto rp = std::make_unique<rj_program>();
  rp->host = std::move(lr.program);
  rp->pattern = regexp;
  int rc = upload_program(rp.get());
  if (rc != RJ_OK) return rc;
  *out = rp.release();
  return RJ_OK;
}

void rj_program_free(rj_program* prog) {
  ErrnoGuard errno_guard;
  if (!prog) return;
  // drop cached host scans of this thread that refer to the program
  auto& v = g_host_scans.v;
  for (size_t i = 0; i < v.size();) {
    if (v[i].first == prog) {
      rj_scan_destroy(v[i].second);
      v.erase(v.begin() + static_cast<long>(i));
    } else {
      i++;
    }
  }
  delete prog;
}

int rj_program_info(const rj_program* prog, rj_info* info) {
  if (!prog || !info) return fail(RJ_BAD_ARGUMENT, "null argument");
  const Program& P = *prog->host;
  info->n_positions = P.n_pos;
  info->n_words = P.n_words;
  info->has_assertions = P.has_assertions;
  info->scan_mode = P.mode == ScanMode::Windows;
  info->n_windows = static_cast<int32_t>(P.windows.size());
  info->window_offset = prog->dev.win_offset;
  info->window_len = prog->dev.win_len;
  info->min_len = P.min_len;
  info->max_len = P.max_len;
  return RJ_OK;
}

int rj_scan_create(const rj_program* prog, rj_scan** out) {
  ErrnoGuard errno_guard;
  if (!prog || !out) return fail(RJ_BAD_ARGUMENT, "null argument");
  auto s = std::make_unique<rj_scan>();
  s->prog = prog;
  int rc = scan_init(s.get());
  if (rc != RJ_OK) return rc;
  *out = s.release();
  return RJ_OK;
}

void rj_scan_destroy(rj_scan* s) {
  ErrnoGuard errno_guard;
  if (!s) return;
  if (s->host_counters) (void)hipHostFree(s->host_counters);
  if (s->host_flag) (void)hipHostFree(s->host_flag);
  for (auto& e : s->ev)
    if (e) (void)hipEventDestroy(e);
  if (s->own_stream) (void)hipStreamDestroy(s->own_stream);
  delete s;
}

int64_t rj_scan_run(rj_scan* s, const void* d_text, uint64_t n, uint64_t own_begin, uint64_t own_end,
                    uint64_t carry_cur, uint64_t carry_prev_end, int have_prev, void* hip_stream) {
  ErrnoGuard errno_guard;
  if (!s) return fail(RJ_BAD_ARGUMENT, "null scan");
  int rc = run_pipeline(s, static_cast<const uint8_t*>(d_text), n, own_begin, own_end, carry_cur, carry_prev_end,
                        have_prev, static_cast<hipStream_t>(hip_stream));
  if (rc != RJ_OK) return rc;
  return static_cast<int64_t>(s->result_count);
}

const uint64_t* rj_scan_device_spans(const rj_scan* s) { return s ? s->result : nullptr; }

int64_t rj_scan_copy_spans(const rj_scan* s, uint64_t* host_spans, uint64_t cap) {
  ErrnoGuard errno_guard;
  if (!s) return fail(RJ_BAD_ARGUMENT, "null scan");
  const uint64_t k = std::min<uint64_t>(cap, s->result_count);
  if (k) {
    hipError_t e = hipMemcpy(host_spans, s->result, k * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(RJ_DEVICE_ERROR, "hipMemcpy failed: %s", hipGetErrorString(e));
  }
  return static_cast<int64_t>(s->result_count);
}

int rj_scan_stats(const rj_scan* s, rj_stats* stats) {
  if (!s || !stats) return fail(RJ_BAD_ARGUMENT, "null argument");
  *stats = s->stats;
  return RJ_OK;
}

int rj_scan_match_full(rj_scan* s, const void* d_text, uint64_t n, void* hip_stream) {
  ErrnoGuard errno_guard;
  if (!s) return fail(RJ_BAD_ARGUMENT, "null scan");
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  launch_match_full(static_cast<const uint8_t*>(d_text), n, s->prog->dev, s->flag.as<int>(), st);
  RJ_HIP(hipMemcpyAsync(s->host_flag, s->flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
  RJ_HIP(hipStreamSynchronize(st));
  RJ_HIP(hipGetLastError());
  return *s->host_flag ? 1 : 0;
}

int64_t rj_match_all(const rj_program* prog, const char* text, size_t n, uint64_t** spans) {
  ErrnoGuard errno_guard;
  if (spans) *spans = nullptr;
  if (!prog || (!text && n)) return fail(RJ_BAD_ARGUMENT, "null argument");
  rj_scan* s = nullptr;
  int rc = host_scan_for(prog, &s);
  if (rc != RJ_OK) return rc;
  const uint8_t* d_text = nullptr;
  rc = stage_text(s, text, n, &d_text);
  if (rc != RJ_OK) return rc;
  rc = run_pipeline(s, d_text, n, 0, n + 1, 0, 0, 0, s->own_stream);
  if (rc != RJ_OK) return rc;
  if (spans && s->result_count) {
    uint64_t* h = static_cast<uint64_t*>(malloc(s->result_count * 2 * sizeof(uint64_t)));
    if (!h) return fail(RJ_DEVICE_ERROR, "out of host memory");
    hipError_t e = hipMemcpy(h, s->result, s->result_count * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
      free(h);
      return fail(RJ_DEVICE_ERROR, "hipMemcpy failed: %s", hipGetErrorString(e));
    }
    *spans = h;
  }
  return static_cast<int64_t>(s->result_count);
}

void rj_free_spans(uint64_t* spans) { free(spans); }

int rj_match_first(const rj_program* prog, const char* text, size_t n, uint64_t* begin, uint64_t* end) {
  ErrnoGuard errno_guard;
  // kMatchFirst == first element of kMatchAll (left-most longest); see DESIGN.md
  uint64_t* spans = nullptr;
  int64_t c = rj_match_all(prog, text, n, &spans);
  if (c < 0) return static_cast<int>(c);
  if (c > 0) {
    if (begin) *begin = spans[0];
    if (end) *end = spans[1];
  }
  rj_free_spans(spans);
  return c > 0 ? 1 : 0;
}

int rj_match_anywhere(const rj_program* prog, const char* text, size_t n) {
  ErrnoGuard errno_guard;
  int64_t c = rj_match_all(prog, text, n, nullptr);
  if (c < 0) return static_cast<int>(c);
  return c > 0 ? 1 : 0;
}

int rj_match_full(const rj_program* prog, const char* text, size_t n) {
  ErrnoGuard errno_guard;
  if (!prog || (!text && n)) return fail(RJ_BAD_ARGUMENT, "null argument");
  rj_scan* s = nullptr;
  int rc = host_scan_for(prog, &s);
  if (rc != RJ_OK) return rc;
  const uint8_t* d_text = nullptr;
  rc = stage_text(s, text, n, &d_text);
  if (rc != RJ_OK) return rc;
  return rj_scan_match_full(s, d_text, n, s->own_stream);
}

}  // extern "C"
