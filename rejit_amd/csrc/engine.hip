// rejit_amd/csrc/engine.hip -- host orchestration of the device pipeline and the C ABI
// (include/rejit_hip.h).  Replaces the reference's API shim src/rejit.cc:150-267 (lazy
// Compile + indirect call into JIT code) and its result sink src/codegen.cc:36-86.
//
//   rj_compile : parse -> lower -> upload the automaton tables to HBM once
//   rj_scan_run: scan kernel -> verify inside the hit regions -> offsets + gather + check
//                -> ONE host synchronisation (the last kernel writes the counters into pinned
//                host memory); regions that overflowed are grown and the run repeated; candidates
//                that overlap go through the selection kernels.  Dense patterns: one kernel
//                (scan_dense_walk) + the gather.  Automata of more than 128 positions: the list
//                pipeline (region_offsets, verify_wave, finalize_small / mark + compact).
//                Texts of a few KiB: one workgroup, one launch (match_small).  A walk that outlives
//                max_walk: the run is repeated on the linear-time carry scan (linear.hip).  Patterns at risk
//                of the reference's ring artefact whose candidates touch, and ranges of such patterns: the
//                reference's own loop replayed between synchronisation points (exact_replay.hip).
// rj_multi_* (several patterns over one text) live in multi_pattern.hip, the host-text entry points
// (rj_match_all, rj_match_all_batch, rj_match_first / anywhere / full, rj_replace_all) in host_api.hip, the
// split of one call over all visible devices in multi_device.hip.
//
// There is no CPU matching code in this library: without a working HIP device every
// entry point fails with RJ_DEVICE_ERROR.
#include <cstring>  // must precede rocprim (its texture iterator uses ::memset)

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "engine_internal.h"
#include "table_layout.h"

using namespace rejit_amd;

namespace rejit_amd {

thread_local std::string g_error;

int rj_fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
  return code;
}

}  // namespace rejit_amd

#define fail ::rejit_amd::rj_fail

namespace {

// The HIP runtime's own load-time initialisers (they run before ours: dependency order) can
// leave errno set before main() starts -- e.g. ENOENT from probing amdgpu.ids.  A process that
// links the reference's librejit.a starts with errno == 0, so restore that.
__attribute__((constructor)) void rj_library_loaded() { errno = 0; }

}  // namespace

namespace {

int upload_program(rj_program* rp) {
  const Program& P = *rp->host;
  const int W = P.n_words;
  const int C = P.has_assertions ? kNumCtx : 1;
  const TableBlob blob = make_table_blob(P, P.n_pos, P.n_words, P.has_assertions);
  const size_t total = blob.words.size();
  const size_t off_first = blob.off_first, off_last = blob.off_last, off_linear = blob.off_linear, off_rowof = blob.off_rowof,
               off_rows = blob.off_rows, off_cls = blob.off_cls;
  const int R = blob.R;

  RJ_HIP(hipGetDevice(&rp->device));
  RJ_HIP(rp->tables.reserve(total * sizeof(uint32_t)));
  RJ_HIP(hipMemcpy(rp->tables.p, blob.words.data(), total * sizeof(uint32_t), hipMemcpyHostToDevice));
  {
    // the reverse automaton for the linear-time carry scan (linear.hip) and backward passes
    const TableBlob rb = make_table_blob(P.rev, P.n_pos, P.n_words, P.has_assertions);
    RJ_HIP(rp->rev_tables.reserve(rb.words.size() * sizeof(uint32_t)));
    RJ_HIP(hipMemcpy(rp->rev_tables.p, rb.words.data(), rb.words.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    rp->rev = DevProgram{};
    point_tables(&rp->rev, rp->rev_tables.as<uint32_t>(), rb, P.n_pos);
    rp->rev.nullable = nullable_bits(P);
  }
  rp->walk = WalkDesc{};
  if (W <= 4 && (P.floating || P.behind) && getenv("RJ_NO_LDS_WALK") == nullptr) {  // (env: measurement override)
    const int nq = W <= 2 ? 1 : 2;
    std::vector<uint64_t> fw = make_walk_blob(blob, P.n_pos, nq);
    if (fw.size() & 1) fw.push_back(0);  // (copied to LDS 16 bytes at a time)
    const size_t words = fw.size();
    if (P.behind) {
      std::vector<uint64_t> rv = make_walk_blob(make_table_blob(P.rev, P.n_pos, P.n_words, P.has_assertions), P.n_pos, nq);
      rv.resize(words, 0);
      fw.insert(fw.end(), rv.begin(), rv.end());
    }
    RJ_HIP(rp->walk_tables.reserve(fw.size() * sizeof(uint64_t)));
    RJ_HIP(hipMemcpy(rp->walk_tables.p, fw.data(), fw.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    rp->walk.blob = rp->walk_tables.as<uint64_t>();
    rp->walk.rev_blob = P.behind ? rp->walk_tables.as<uint64_t>() + words : nullptr;
    rp->walk.words = static_cast<uint32_t>(words);
    rp->walk.nq = nq;
    rp->walk.n_ctx = C;
    rp->walk.n_pos = P.n_pos;
  }
  const uint32_t* base = rp->tables.as<uint32_t>();
  DevProgram& D = rp->dev;
  D.n_pos = P.n_pos;
  D.table_words = static_cast<uint32_t>(total);
  loop_skip_masks(P, D.loop_mask, D.skip_mask);
  D.swar = getenv("RJ_NO_SWAR") == nullptr ? make_swar_plan(P) : SwarPlan{};  // (env: measurement override)
  D.short_max = getenv("RJ_NO_SHORT") == nullptr ? short_match_bound(P) : 0u;   // (env: measurement override)
  D.n_words = W;
  D.n_ctx = C;
  D.n_rows = R;
  D.nullable = 0;
  for (int c = 0; c < kNumCtx; c++)
    if (P.nullable[C == 1 ? 0 : c]) D.nullable |= 1u << c;
  fill_windows(&D, P);
  {
    ByteSet seen;
    for (const FFWindow& w : P.windows)
      for (uint32_t k = 0; k < w.len; k++) {
        const uint32_t m = k < 4 ? (w.mask0 >> (8 * k)) : (w.mask1 >> (8 * (k - 4)));
        const uint32_t v = k < 4 ? (w.value0 >> (8 * k)) : (w.value1 >> (8 * (k - 4)));
        if (m & 0xFFu) seen.add(static_cast<uint8_t>(v & 0xFFu));
      }
    rp->window_alphabet = seen.count();
    // nibble filter: the alphabet's low nibbles must be distinct (else the pattern's own letters
    // alias each other) and every mask byte must be all-or-nothing
    bool ok = true;
    uint32_t nib_seen = 0;
    for (int b = 0; b < 256; b++)
      if (seen.has(static_cast<uint8_t>(b))) {
        ok = ok && !((nib_seen >> (b & 15)) & 1u);
        nib_seen |= 1u << (b & 15);
      }
    for (const FFWindow& w : P.windows)
      for (uint32_t k = 0; k < 8; k++) {
        const uint32_t m = (k < 4 ? (w.mask0 >> (8 * k)) : (w.mask1 >> (8 * (k - 4)))) & 0xFFu;
        ok = ok && (m == 0 || m == 0xFFu);
      }
    rp->window_nibbles = ok && getenv("RJ_NO_NIBBLE") == nullptr;
  }
  {
    // Batches of texts are concatenated with a separator that no pattern position can consume, so
    // that every walk dies at a text's end exactly as it would at the end of the text.  A line
    // break gives the `^` / `$` contexts of a text begin / end for free; without assertions any
    // dead byte will do.  Patterns at risk of Q8 are matched text by text (the exact kernel has
    // whole-text state).
    auto dead = [&](int b) {
      for (int k = 0; k < W; k++)
        if (P.cls[static_cast<size_t>(b) * W + k] != 0) return false;
      return true;
    };
    rp->batch_separator = -1;
    if (!P.q8_risk) {
      if (dead('\n')) rp->batch_separator = '\n';
      else if (dead('\r')) rp->batch_separator = '\r';
      else if (!P.has_assertions)
        for (int b = 0; b < 256 && rp->batch_separator < 0; b++)
          if (dead(b)) rp->batch_separator = b;
    }
  }
  // Windows with few fixed bytes hit several percent of the positions of ordinary text (estimate:
  // every fixed byte passes 1/64 of the positions): that is dense work, and the fused dense
  // kernel (candidates walked in place, no hit lists) serves it better than hits + verify.
  if (D.mode == 1 && !P.floating && dense_walk_fits(D) && getenv("RJ_NO_DENSE_WALK") == nullptr) {
    double density = 0.0;
    for (const FFWindow& w : P.windows) {
      int fixed = 0;
      for (uint32_t k = 0; k < w.len; k++)
        fixed += ((k < 4 ? (w.mask0 >> (8 * k)) : (w.mask1 >> (8 * (k - 4)))) & 0xFFu) != 0;
      double p = 1.0;
      for (int k = 0; k < fixed; k++) p /= 64.0;
      density += p;
    }
    if (density > 0.03) D.mode = 0;
  }
  // How far one start may be walked before the run is handed to the linear-time carry scan
  // (linear.hip).  Dense mode walks a start at a sizeable share of the bytes, so a long-lived
  // candidate means many of them: cut early.  Window hits are rare: a long line is cheaper to walk.
  D.max_walk = static_cast<uint32_t>(kMaxSimSteps);
  // (automata of more than 1024 positions: the carry scan takes them since round 5, but its cost grows with the square of the
  // width -- a start is walked for up to 2^20 bytes before a run goes there)
  if (linear_path_cheap(rp)) D.max_walk = (D.mode == 0 || D.behind) ? 4096u : 65536u;  // (behind: three walks per hit, a step ~1 us)
  if (const char* mw = getenv("RJ_MAX_WALK"))  // test / measurement override
    if (atoi(mw) > 0) D.max_walk = static_cast<uint32_t>(std::min<long>(atol(mw), static_cast<long>(kMaxSimSteps)));
  // (patterns at risk of the ring artefact keep every start as a candidate: the test "a candidate begins where
  // another one ends" that sends a text to the exact replay is made on the candidates)
  D.loop_first = run_start_rule(P) && !P.q8_risk && getenv("RJ_NO_LOOP_FIRST") == nullptr ? 1u : 0u;  // (env: measurement override)
  rp->stream = getenv("RJ_NO_STREAMS") == nullptr ? make_stream_plan(P, D.loop_first != 0, P.q8_risk) : StreamPlan{};  // (env: measurement override)
  rp->run = getenv("RJ_NO_RUNS") == nullptr ? make_run_plan(P) : RunPlan{};   // (env: measurement override)
  D.float_range = P.floating ? P.float_max - P.float_min + 1 : 1;
  D.float_max = P.floating ? P.float_max : 0;
  for (int k = 0; k < 8; k++) D.first_bytes[k] = P.first_bytes.w[k];
  D.min_len = P.min_len;
  D.first = base + off_first;
  D.last = base + off_last;
  D.linear = base + off_linear;
  D.row_of = reinterpret_cast<const int32_t*>(base + off_rowof);
  D.rows = base + off_rows;
  D.cls = base + off_cls;
  rp->walk.nullable = D.nullable;
  rp->walk.max_walk = D.max_walk;
  if (P.q8_risk) {
    // graph for the exact replay kernels (table_layout.h: int32 arrays, class bitmaps, literal bytes)
    const GraphBlob gb = make_graph_blob(P.graph);
    RJ_HIP(rp->graph_blob.reserve(gb.bytes.size()));
    RJ_HIP(hipMemcpy(rp->graph_blob.p, gb.bytes.data(), gb.bytes.size(), hipMemcpyHostToDevice));
    point_graph(&rp->graph, rp->graph_blob.as<uint8_t>(), gb);
  }
  return RJ_OK;
}

}  // namespace

namespace rejit_amd {
// Grow-only device lists.  hits: n_regions x region_cap;  candidates: one slot per hit.
int ensure_lists(rj_scan* s, uint32_t n_regions, uint32_t region_cap, uint64_t cands_cap) {
  const uint64_t hit_slots = static_cast<uint64_t>(n_regions) * region_cap;
  RJ_HIP(s->hits.reserve(hit_slots * sizeof(uint64_t)));
  RJ_HIP(s->hit_counts.reserve(static_cast<size_t>(n_regions) * sizeof(uint32_t)));
  RJ_HIP(s->hit_offsets.reserve((static_cast<size_t>(n_regions) + 1) * sizeof(uint64_t)));
  if (cands_cap > s->cands_cap) {
    RJ_HIP(s->cand_begin.reserve(cands_cap * sizeof(uint64_t)));
    RJ_HIP(s->cand_end.reserve(cands_cap * sizeof(uint64_t)));
    RJ_HIP(s->out.reserve(cands_cap * 2 * sizeof(uint64_t)));
    s->cands_cap = cands_cap;
    s->out_cap = cands_cap;
  }
  return RJ_OK;
}
}  // namespace rejit_amd

namespace {

// Large path: more hit slots than finalize_small handles in LDS.  The slots are already in
// text order, so no sort: drop the kNoMatch slots, then check / select.
hipError_t prefix_scan(rj_scan* s, uint64_t* in, uint64_t* out, uint64_t count, bool is_max, hipStream_t st) {
  size_t bytes = 0;
  hipError_t e = is_max ? rocprim::exclusive_scan(nullptr, bytes, in, out, uint64_t{0}, count, rocprim::maximum<uint64_t>(), st)
                        : rocprim::exclusive_scan(nullptr, bytes, in, out, uint64_t{0}, count, rocprim::plus<uint64_t>(), st);
  if (e != hipSuccess) return e;
  e = s->sort_tmp.reserve(std::max<size_t>(bytes, 16));
  if (e != hipSuccess) return e;
  return is_max ? rocprim::exclusive_scan(s->sort_tmp.p, bytes, in, out, uint64_t{0}, count, rocprim::maximum<uint64_t>(), st)
                : rocprim::exclusive_scan(s->sort_tmp.p, bytes, in, out, uint64_t{0}, count, rocprim::plus<uint64_t>(), st);
}

int check_and_select(rj_scan* s, uint64_t n_upper, const FinalizeParams& fp, hipStream_t st);

int finalize_large(rj_scan* s, uint64_t n_slots, bool unsorted, uint64_t text_len, const FinalizeParams& fp, hipStream_t st) {
  s->stats.large_path = 1;
  RJ_HIP(s->keys_out.reserve(n_slots * sizeof(uint64_t)));
  RJ_HIP(s->vals_out.reserve(n_slots * sizeof(uint64_t)));
  RJ_HIP(s->scan_a.reserve(n_slots * sizeof(uint64_t)));
  RJ_HIP(s->scan_b.reserve(n_slots * sizeof(uint64_t)));
  uint64_t* keys = s->keys_out.as<uint64_t>();
  uint64_t* vals = s->vals_out.as<uint64_t>();
  uint64_t* sa = s->scan_a.as<uint64_t>();
  uint64_t* sb = s->scan_b.as<uint64_t>();
  auto scan = [&](uint64_t* in, uint64_t* out, uint64_t count, bool is_max) -> hipError_t {
    return prefix_scan(s, in, out, count, is_max, st);
  };
  // 1. ordered compaction of the verified candidates
  launch_mark_valid(s->cand_end.as<uint64_t>(), n_slots, sa, st);
  RJ_HIP(scan(sa, sb, n_slots, false));
  launch_compact_valid(s->cand_begin.as<uint64_t>(), s->cand_end.as<uint64_t>(), sa, sb, n_slots, keys, vals,
                       s->counters.as<unsigned long long>(), st);
  if (unsorted) {
    // floating windows: the starts derived from neighbouring hits interleave (and may repeat), so
    // this mode -- rare hits by construction -- pays for a sort of the compacted candidates
    RJ_HIP(hipMemcpyAsync(s->host_counters, s->counters.p, kCntSize * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    RJ_HIP(hipStreamSynchronize(st));
    const uint64_t nc = s->host_counters[kCntCands];
    if (nc > 1) {
      unsigned bits = 1;
      while (bits < 64 && (text_len >> bits) != 0) bits++;
      uint64_t* k2 = s->cand_begin.as<uint64_t>();  // free again: reuse as sort output
      uint64_t* v2 = s->cand_end.as<uint64_t>();
      size_t tmp_bytes = 0;
      RJ_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, k2, vals, v2, nc, 0, bits, st));
      RJ_HIP(s->sort_tmp.reserve(std::max<size_t>(tmp_bytes, 16)));
      RJ_HIP(rocprim::radix_sort_pairs(s->sort_tmp.p, tmp_bytes, keys, k2, vals, v2, nc, 0, bits, st));
      RJ_HIP(hipMemcpyAsync(keys, k2, nc * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
      RJ_HIP(hipMemcpyAsync(vals, v2, nc * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
    }
  }
  return check_and_select(s, n_slots, fp, st);
}

// The ordered candidates are in keys_out / vals_out, their count in counters[kCntCands] (device).
// Common case, without a host round trip before it: the candidates already are the result.
int check_and_select(rj_scan* s, uint64_t n_slots, const FinalizeParams& fp, hipStream_t st) {
  uint64_t* keys = s->keys_out.as<uint64_t>();
  uint64_t* vals = s->vals_out.as<uint64_t>();
  if (fp.detect_adjacent) launch_detect_adjacent(keys, vals, n_slots, s->counters.as<unsigned long long>(), st);
  launch_check_and_interleave(keys, vals, s->counters.as<unsigned long long>() + kCntCands, n_slots, fp.carry_cur,
                              s->out.as<uint64_t>(), s->out_cap, s->counters.as<unsigned long long>() + kCntUnordered, st);
  RJ_HIP(hipMemcpyAsync(s->host_counters, s->counters.p, kCntSize * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  RJ_HIP(hipStreamSynchronize(st));
  RJ_HIP(hipGetLastError());
  return resolve_selection(s, fp, st);
}

}  // namespace

namespace rejit_amd {
// host_counters hold the counters of a finished check: take the candidates as they are, or run
// the cluster-parallel selection over keys_out / vals_out.
int resolve_selection(rj_scan* s, const FinalizeParams& fp, hipStream_t st) {
  uint64_t* keys = s->keys_out.as<uint64_t>();
  uint64_t* vals = s->vals_out.as<uint64_t>();
  if (s->host_counters[kCntOverflow] != 0) return RJ_OK;  // the caller grows the regions and runs again
  const uint64_t n_cands = s->host_counters[kCntCands];
  s->stats.n_candidates = n_cands;
  if (n_cands == 0) {
    s->result_count = 0;
    return RJ_OK;
  }
  if (s->host_counters[kCntUnordered] == 0) {
    s->result_count = n_cands;
    return RJ_OK;
  }
  if (n_cands <= kFinalizeCap) {
    // few candidates: one workgroup sorts (a no-op here), deduplicates and selects in LDS
    FinalizeParams f = fp;
    f.cand_begin = keys;
    f.cand_end = vals;
    f.cands_cap = n_cands;
    f.out = s->out.as<uint64_t>();
    f.out_cap = s->out_cap;
    f.counters = s->counters.as<unsigned long long>();
    f.detect_adjacent = 0;
    f.detect_conflict = fp.detect_conflict;
    f.expand = 1;
    unsigned long long* scratch = reinterpret_cast<unsigned long long*>(s->host_flag);  // pinned, 16 bytes
    *scratch = n_cands;  // finalize_small takes its slot count from counters[kCntHits]
    RJ_HIP(hipMemcpyAsync(s->counters.as<unsigned long long>() + kCntHits, scratch, sizeof(unsigned long long),
                          hipMemcpyHostToDevice, st));
    launch_finalize_small(f, st);
    RJ_HIP(hipMemcpyAsync(s->host_counters, s->counters.p, kCntSize * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    RJ_HIP(hipStreamSynchronize(st));
    RJ_HIP(hipGetLastError());
    s->result_count = s->host_counters[kCntFinal];
    return RJ_OK;
  }
  // general case: cluster-parallel selection
  RJ_HIP(s->scan_a.reserve(n_cands * sizeof(uint64_t)));
  RJ_HIP(s->scan_b.reserve(n_cands * sizeof(uint64_t)));
  RJ_HIP(s->cand_begin.reserve(n_cands * sizeof(uint64_t)));
  uint64_t* sa = s->scan_a.as<uint64_t>();
  uint64_t* sb = s->scan_b.as<uint64_t>();
  auto scan = [&](uint64_t* in, uint64_t* out, uint64_t count, bool is_max) -> hipError_t {
    return prefix_scan(s, in, out, count, is_max, st);
  };
  //   pmax  = exclusive prefix max of the ends            (sa)
  //   taken = per-cluster sequential walk                  (taken)
  //   last  = exclusive prefix max of (taken ? i+1 : 0)    (sb; cand_begin reused as scratch)
  //   keep  = taken minus the zero-length rule             (sa)
  //   pos   = exclusive prefix sum of keep                 (sb)
  RJ_HIP(s->taken.reserve(n_cands));
  uint64_t* scratch = s->cand_begin.as<uint64_t>();
  RJ_HIP(scan(vals, sa, n_cands, true));
  RJ_HIP(s->chain_blocks.reserve(chain_select_scratch_bytes(n_cands)));
  launch_chain_select(keys, vals, sa, n_cands, fp.carry_cur, s->taken.as<uint8_t>(), sb, scratch, s->chain_blocks.as<uint64_t>(), st);
  launch_taken_index(s->taken.as<uint8_t>(), n_cands, scratch, st);
  RJ_HIP(scan(scratch, sb, n_cands, true));
  launch_zero_length_rule(keys, vals, s->taken.as<uint8_t>(), sb, n_cands, fp.carry_prev_end, fp.have_prev, sa,
                          fp.detect_conflict ? s->counters.as<unsigned long long>() + kCntConflict : nullptr, st);
  RJ_HIP(scan(sa, sb, n_cands, false));
  launch_compact_kept(keys, vals, sa, sb, n_cands, s->out.as<uint64_t>(), s->out_cap, s->counters.as<unsigned long long>(), st);
  RJ_HIP(hipMemcpyAsync(s->host_counters, s->counters.p, kCntSize * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  RJ_HIP(hipStreamSynchronize(st));
  RJ_HIP(hipGetLastError());
  s->result_count = s->host_counters[kCntFinal];
  return RJ_OK;
}
}  // namespace rejit_amd

namespace rejit_amd {

// The window constants a scan kernel takes (exact or nibble form, see WindowSet).
WindowSet make_window_set(const rj_program* rp) {
  const DevProgram& D = rp->dev;
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
  ws.nibble = D.win_len > 4 && !ws.two_level && rp->window_nibbles;
  if (ws.nibble) {
    bool nib_masked = false;
    for (int k = 0; k < kDevMaxWindows; k++) {
      uint32_t v = 0, m = 0;
      for (int i = 0; i < 8; i++) {
        const uint32_t vb = (i < 4 ? (D.win_value0[k] >> (8 * i)) : (D.win_value1[k] >> (8 * (i - 4)))) & 0xFFu;
        const uint32_t mb = (i < 4 ? (D.win_mask0[k] >> (8 * i)) : (D.win_mask1[k] >> (8 * (i - 4)))) & 0xFFu;
        const int at = 8 * (i & 3) + 4 * (i >> 2);
        if (mb) {
          v |= (vb & 15u) << at;
          m |= 15u << at;
        }
      }
      ws.value0[k] = v;
      ws.mask0[k] = m;
      nib_masked |= m != 0xFFFFFFFFu;
    }
    ws.masked = nib_masked;
  }
  ws.len = D.win_len;
  ws.offset = D.win_offset;
  return ws;
}

// Assertion-only patterns (`^`, `$`: the line table of a grep-like caller) from the beginning of a selection: every
// position in a matching context is a match, so the result is written once, in place (emit_scan.hip).
// 1 = done, 0 = not applicable / the prefix scan timed out (the caller takes the dense kernel), < 0 = error.
static int run_assertions(rj_scan* s, const uint8_t* d_text, uint64_t n, uint64_t sb, uint64_t se, hipStream_t st) {
  const DevProgram& D = s->prog->dev;
  static const bool off = getenv("RJ_NO_EMIT") != nullptr;  // measurement override
  if (off || D.n_pos != 0 || D.nullable == 0 || se <= sb) return 0;
  uint64_t cap = std::max<uint64_t>(s->hits_hint + s->hits_hint / 8 + 1024, (se - sb) / 48 + 1024);
  RJ_HIP(s->scan_a.reserve(emit_scratch_bytes(sb, se)));
  for (int attempt = 0; attempt < 3; attempt++) {
    if (cap > s->out_cap) {
      RJ_HIP(s->out.reserve(cap * 2 * sizeof(uint64_t)));
      s->out_cap = cap;
    }
    RJ_HIP(hipMemsetAsync(s->counters.p, 0, kCntSize * sizeof(unsigned long long), st));
    s->host_counters[kCntOverrun] = 0;
    s->host_counters[kCntFinal] = 0;
    launch_emit_assertions(d_text, n, sb, se, D.nullable, s->scan_a.as<unsigned long long>(), s->out.as<uint64_t>(), s->out_cap,
                           s->counters.as<unsigned long long>(), s->host_counters, s->t0(), s->ev[2], st);
    RJ_HIP(hipStreamSynchronize(st));
    RJ_HIP(hipGetLastError());
    float ms = 0.f;
    if (s->timing) (void)hipEventElapsedTime(&ms, s->ev[1], s->ev[2]);
    s->stats.scan_ms += ms;
    if (s->host_counters[kCntOverrun] != 0) return 0;
    const uint64_t cnt = s->host_counters[kCntFinal];
    if (cnt > s->out_cap) {  // more matches than room: once more with room for all of them
      s->stats.retries++;
      cap = cnt + cnt / 16 + 1024;
      continue;
    }
    s->result_count = cnt;
    s->result = s->out.as<uint64_t>();
    s->hits_hint = cnt;
    s->stats.n_hits += cnt;
    s->stats.n_candidates += cnt;
    return 1;
  }
  return 0;
}

// Dense patterns whose candidates cannot overlap (make_stream_plan) from the beginning of a selection: one pass, the pairs
// written once, in place (dense_streams.hip).  1 = done, 0 = not applicable / void run (the caller takes scan_dense_walk,
// and from there the carry scan), < 0 = error.
static int run_streams(rj_scan* s, const uint8_t* d_text, uint64_t n, uint64_t sb, uint64_t se, hipStream_t st) {
  const rj_program* rp = s->prog;
  if (rp->stream.n_pos == 0 || s->streams_off || se <= sb) return 0;
  StreamParams a{};
  a.text = d_text;
  a.n = n;
  a.sb = sb;
  a.se = se;
  a.n_tiles = stream_tiles(sb, se, n, &a.first_tile);
  if (a.n_tiles == 0) {
    s->result_count = 0;
    s->result = s->out.as<uint64_t>();
    return 1;
  }
  a.plan = rp->stream;
  // (a staged pair keeps its length in 17 bits, dense_streams.hip: a longer walk must void the run, not wrap into the begin)
  a.max_walk = std::min<uint32_t>(rp->dev.max_walk, (1u << 17) - 1u);
  uint64_t cap = std::max<uint64_t>(s->hits_hint + s->hits_hint / 8 + 1024, (se - sb) / 64 + 1024);
  RJ_HIP(s->scan_a.reserve(stream_scratch_bytes(a.n_tiles)));
  for (int attempt = 0; attempt < 3; attempt++) {
    if (cap > s->out_cap) {
      RJ_HIP(s->out.reserve(cap * 2 * sizeof(uint64_t)));
      s->out_cap = cap;
    }
    RJ_HIP(hipMemsetAsync(s->counters.p, 0, kCntSize * sizeof(unsigned long long), st));
    s->host_counters[kCntOverrun] = 0;
    s->host_counters[kCntFinal] = 0;
    a.out = s->out.as<uint64_t>();
    a.out_cap = s->out_cap;
    a.counters = s->counters.as<unsigned long long>();
    a.host_counters = s->host_counters;
    launch_dense_streams(a, s->scan_a.as<unsigned long long>(), s->t0(), s->ev[2], st);
    unsigned long long slow = 0;
    RJ_HIP(hipMemcpyAsync(&s->host_counters[kCntSlowStarts], s->counters.as<unsigned long long>() + kCntSlowStarts, sizeof(unsigned long long),
                          hipMemcpyDeviceToHost, st));
    RJ_HIP(hipStreamSynchronize(st));
    RJ_HIP(hipGetLastError());
    slow = s->host_counters[kCntSlowStarts];
    float ms = 0.f;
    if (s->timing) (void)hipEventElapsedTime(&ms, s->ev[1], s->ev[2]);
    s->stats.scan_ms += ms;
    if (s->host_counters[kCntOverrun] != 0) {
      s->streams_off = true;  // (a long-lived thread or a time-out: this text is not for the register steps)
      s->stats.retries++;
      return 0;
    }
    // a text on which many starts outlive the register steps (long runs of the loop's class): the scalar walks are
    // chains of loads from device memory; scan_dense_walk's LDS walkers do that better -- from the next call on
    // ... and a run shape (`[A-Z][a-z]+` over words longer than the steps' 16 bytes) has the run kernels behind it, which take the
    // text at 1.0-1.5 ms per GiB whatever the runs' length.  A scalar walk holds its whole wave up: 37 600 of them per GiB (one per
    // 28 KiB) made this kernel 5.2 ms where the run kernels take 1.05 (tools/probes/zoo.py) -- ~100 ns apiece: they pay from a walk
    // per ~256 KiB on
    if (slow > (se - sb) / (rp->run.ok ? 262144 : 512)) s->streams_off = true;
    const uint64_t cnt = s->host_counters[kCntFinal];
    if (cnt > s->out_cap) {  // more matches than room: once more with room for all of them
      s->stats.retries++;
      cap = cnt + cnt / 16 + 1024;
      continue;
    }
    s->result_count = cnt;
    s->result = s->out.as<uint64_t>();
    s->hits_hint = cnt;
    s->stats.n_hits += cnt;
    s->stats.n_candidates += cnt;
    s->stats.stream_path = 1;
    s->stats.slow_starts = static_cast<int32_t>(std::min<unsigned long long>(slow, 0x7fffffffull));
    return 1;
  }
  return 0;
}

// Patterns with one long-lived thread in one loop position (run_scan.h: `X+`, `A L*`, `A L* B`, `X+ B`): two passes over
// the text from the tile of sb to its END (the segment that holds the range's last start closes at the first break behind
// it, wherever that is) and a scan over tile summaries; linear in the text whatever the length of the runs.  1 = done
// (s->out, s->result_count), 0 = not this path, < 0 = error.
static int run_runs(rj_scan* s, const uint8_t* d_text, uint64_t n, uint64_t sb, uint64_t se, uint64_t carry_cur, uint64_t carry_prev_end, int have_prev,
                    hipStream_t st) {
  const rj_program* rp = s->prog;
  if (!rp->run.ok || se <= sb) return 0;
  RunParams a{};
  a.text = d_text;
  a.n = n;
  // (`A L+` plans: the kernels' starts are the marks one byte behind a match's begin -- the three positions move with them)
  const uint64_t lag = rp->run.lag;
  a.sb = sb + lag;
  a.se = std::min<uint64_t>(se, n) + lag;   // (a match consumes at least one byte: none begins at n)
  a.min_start = std::max<uint64_t>(sb, have_prev ? carry_cur : 0) + lag;
  a.plan = rp->run;
  if (have_prev && rp->run.has_b && carry_prev_end >= 1 && carry_prev_end <= n) {
    // the carried-in match ended behind its B at carry_prev_end - 1: when that byte is no break, the segment goes on and has
    // had its match (run_scan.h)
    uint8_t last = 0;
    RJ_HIP(hipMemcpyAsync(&last, d_text + carry_prev_end - 1, 1, hipMemcpyDeviceToHost, st));
    RJ_HIP(hipStreamSynchronize(st));
    const Program& P = *rp->host;
    const int l_pos = P.n_pos == 3 ? 1 : 0;
    a.blocked_in = ((P.cls[last] >> l_pos) & 1u) ? 1u : 0u;
  }
  a.tile_bytes = run_tile_bytes(n - std::min<uint64_t>(a.min_start, n));
  a.n_tiles = run_tiles(a.min_start, n, a.tile_bytes, &a.first_tile);
  RJ_HIP(s->run_summaries.reserve(sizeof(RunSummary) * run_resolve_slots(a.n_tiles)));
  RJ_HIP(s->run_tile_in.reserve(sizeof(RunTileIn) * run_resolve_slots(a.n_tiles)));
  a.summaries = s->run_summaries.as<RunSummary>();
  a.tile_in = s->run_tile_in.as<RunTileIn>();
  a.counters = s->counters.as<unsigned long long>();
  a.host_counters = s->host_counters;
  a.out = s->out.as<uint64_t>();
  a.out_cap = s->out_cap;
  RJ_HIP(hipMemsetAsync(s->counters.p, 0, kCntSize * sizeof(unsigned long long), st));
  s->host_counters[kCntFinal] = 0;
  launch_run_summary(a, s->t0(), nullptr, st);
  launch_run_resolve(a, st);
  RJ_HIP(hipStreamSynchronize(st));   // (the number of pairs sizes the output)
  RJ_HIP(hipGetLastError());
  const uint64_t cnt = s->host_counters[kCntFinal];
  // (`^` in front of the shape is a mask on the kernels' start stream, `$` behind it a test of the closing break: run_scan.hip)
  if (s->count_only_run && sb == 0 && se > n) {   // MatchAllCount of the whole text: one pass over it
    s->result_count = cnt;
    s->result = nullptr;
    s->hits_hint = cnt;
    s->stats.n_hits += cnt;
    s->stats.n_candidates += cnt;
    s->stats.run_path = 1;
    return 1;
  }
  if (cnt > s->out_cap) {
    const uint64_t cap = cnt + cnt / 16 + 1024;
    RJ_HIP(s->out.reserve(cap * 2 * sizeof(uint64_t)));
    s->out_cap = cap;
  }
  a.out = s->out.as<uint64_t>();
  a.out_cap = s->out_cap;
  if (cnt) launch_run_emit(a, s->ev[2], st);
  else RJ_HIP(hipEventRecord(s->ev[2], st));
  RJ_HIP(hipStreamSynchronize(st));
  RJ_HIP(hipGetLastError());
  float ms = 0.f;
  if (s->timing) (void)hipEventElapsedTime(&ms, s->ev[1], s->ev[2]);
  s->stats.scan_ms += ms;   // (first pass's start to the second pass's end, the scan between them included)
  s->result_count = cnt;
  s->result = s->out.as<uint64_t>();
  s->hits_hint = cnt;
  s->stats.n_hits += cnt;
  s->stats.n_candidates += cnt;
  s->stats.run_path = 1;
  return 1;
}

// The PAIR shape (run_scan.h: `"[^"]*"`, `'[^'\n]*'`): the same three steps with the pair kernels -- whole texts without a
// carried-in state only (what crosses a cut is the parity of the Q bytes since the last reset, which a shard does not know).
// 1 = done, 0 = not this path, < 0 = error.
static int run_pairs(rj_scan* s, const uint8_t* d_text, uint64_t n, uint64_t sb, uint64_t se, hipStream_t st) {
  const rj_program* rp = s->prog;
  if (!rp->run.pair || sb != 0 || se < n || n == 0) return 0;
  RunParams a{};
  a.text = d_text;
  a.n = n;
  a.sb = 0;
  a.se = n;
  a.plan = rp->run;
  a.tile_bytes = run_tile_bytes(n);
  a.n_tiles = run_tiles(0, n, a.tile_bytes, &a.first_tile);
  RJ_HIP(s->run_summaries.reserve(sizeof(RunSummary) * run_resolve_slots(a.n_tiles)));
  RJ_HIP(s->run_tile_in.reserve(sizeof(RunTileIn) * run_resolve_slots(a.n_tiles)));
  a.summaries = s->run_summaries.as<RunSummary>();
  a.tile_in = s->run_tile_in.as<RunTileIn>();
  a.counters = s->counters.as<unsigned long long>();
  a.host_counters = s->host_counters;
  a.out = s->out.as<uint64_t>();
  a.out_cap = s->out_cap;
  RJ_HIP(hipMemsetAsync(s->counters.p, 0, kCntSize * sizeof(unsigned long long), st));
  s->host_counters[kCntFinal] = 0;
  launch_pair_summary(a, s->t0(), nullptr, st);
  launch_pair_resolve(a, st);
  RJ_HIP(hipStreamSynchronize(st));   // (the number of pairs sizes the output)
  RJ_HIP(hipGetLastError());
  const uint64_t cnt = s->host_counters[kCntFinal];
  if (!s->count_only_run) {
    if (cnt > s->out_cap) {
      const uint64_t cap = cnt + cnt / 16 + 1024;
      RJ_HIP(s->out.reserve(cap * 2 * sizeof(uint64_t)));
      s->out_cap = cap;
    }
    a.out = s->out.as<uint64_t>();
    a.out_cap = s->out_cap;
    if (cnt) launch_pair_emit(a, s->ev[2], st);
    else RJ_HIP(hipEventRecord(s->ev[2], st));
    RJ_HIP(hipStreamSynchronize(st));
    RJ_HIP(hipGetLastError());
    float ms = 0.f;
    if (s->timing) (void)hipEventElapsedTime(&ms, s->ev[1], s->ev[2]);
    s->stats.scan_ms += ms;
  }
  s->result_count = cnt;
  s->result = s->count_only_run ? nullptr : s->out.as<uint64_t>();
  s->hits_hint = cnt;
  s->stats.n_hits += cnt;
  s->stats.n_candidates += cnt;
  s->stats.run_path = 2;
  return 1;
}

constexpr uint64_t kExactLimit = 1u << 20;  // bytes the one-lane exact kernel is allowed to walk
constexpr uint64_t kDenseSegment = 1ull << 27;  // dense mode: starts per pipeline run (bounds the lists)

// One full pipeline over the starts [sb, se): scan -> region offsets -> verify -> finalize.
// Results: s->out (device, ordered pairs), s->result_count.
static int run_range(rj_scan* s, const uint8_t* d_text, uint64_t n, uint64_t sb, uint64_t se, uint64_t carry_cur,
                     uint64_t carry_prev_end, int have_prev, hipStream_t st, bool force_dense = false) {
  const rj_program* rp = s->prog;
  // Windows behind an unbounded prefix give ONE candidate per hit (the left-most start): enough for a
  // run from the beginning of the text, not for an own range that begins inside it (a start clipped by
  // the range or by a carried-in match would be missing) -- such runs, and repeats after a conflict,
  // take the dense path, which considers every start.
  const bool as_dense = force_dense || (rp->dev.behind && (sb != 0 || carry_cur != 0 || s->behind_conflicts));
  DevProgram dense_copy;
  if (as_dense && rp->dev.mode == 1) {
    dense_copy = rp->dev;
    dense_copy.mode = 0;
    dense_copy.behind = 0;
    if (linear_path_cheap(rp)) dense_copy.max_walk = std::min<uint32_t>(dense_copy.max_walk, 4096u);
  }
  const DevProgram& D = (as_dense && rp->dev.mode == 1) ? dense_copy : rp->dev;
  const bool windows = D.mode == 1;
  const bool behind = windows && D.behind != 0;
  s->result_count = 0;
  s->result = nullptr;
  // the previous text needed the linear-time path: go there directly (run_linear clears the hint
  // when the text turns out not to need it)
  // one long-lived thread in one loop position (run_scan.h): the text's bit streams answer, whatever the runs' length --
  // after dense_streams (faster where the runs are short) has given such a text up, or at once for the patterns that
  // kernel does not take (`a.*b`) and for runs under a carry
  static const bool runs_first = getenv("RJ_RUNS_FIRST") != nullptr;   // measurement override
  const bool fresh = carry_cur == 0 && !have_prev;
  // (a pattern with a fast-forward window -- `a.*b`, `<[^>]*>`: the window is their first byte -- goes there only once a walk
  // has outlived max_walk on this scan's text: linear_hint.  NOTE: the hint is cleared by nobody on this path; a scan object
  // whose texts stop having long runs keeps the run kernels, which are never wrong and never quadratic.)
  static const bool no_pairs = getenv("RJ_NO_PAIRS") != nullptr;   // measurement override
  if (rp->run.pair && fresh && !no_pairs) {   // (`"[^"]*"`: every Q byte is a window hit and a walk; the pair kernels are two streaming passes)
    int rc = run_pairs(s, d_text, n, sb, se, st);
    if (rc != 0) return rc < 0 ? rc : RJ_OK;
  }
  // A WINDOWS-mode run shape -- its window is the ONE byte of A: `a.*b`, `#.*`, `<[^>]*>`, `\([^)]*\)`, ` +` -- over a range that reaches the
  // text's end: the run kernels first.  On everyday text such a byte is everywhere, every hit is a walk, and the window path ran 1 GiB
  // of log-like text at 32-155 GB/s (`a.*b` 33.8 ms, `<[^>]*>` 22.7, `#.*` 6.9, ` +` 8.0) where the run kernels take 0.55-1.5 ms
  // (tools/probes/zoo.py).  Where the byte is RARE the window scan is ~1.6 x faster (one pass at the literal scan's rate): a run
  // that found few matches sends the next one there, and a window scan that meets dense hits sends the scan object back for good.
  static const bool no_window_runs = getenv("RJ_NO_WINDOW_RUNS") != nullptr;   // measurement override
  const bool window_runs = windows && se >= n && n - std::min(sb, n) >= (256u << 10) && (!s->runs_sparse || s->window_dense) && !no_window_runs;
  // (`^[A-Z][a-z]+`: a dense-mode shape behind `^` -- its candidates are the line starts, few and cheap on the general path (1.2 ms per
  // GiB of log-like text), where the run kernels would write every capitalised word and filter 98 % of them away again (1.8 ms))
  // (... and where the shape is `X+` / `X+ Y` -- ` +`, `=+`, `-+>` -- dense_streams decides runs of up to 47 bytes in registers in ONE pass:
  // ` +` over the same text 0.7 ms against the run kernels' 1.5; texts of longer runs send it back through streams_off, as in dense mode)
  if (window_runs && fresh && rp->stream.n_pos != 0 && rp->stream.run_shape != 0 && !rp->stream.select && !s->streams_off && !s->linear_hint && !runs_first) {
    int rc = run_streams(s, d_text, n, sb, se, st);
    if (rc != 0) return rc < 0 ? rc : RJ_OK;
  }
  static const bool bol_dense_general = getenv("RJ_BOL_DENSE_GENERAL") != nullptr;   // measurement override: the policy before the in-kernel `^`
  const bool bol_dense = rp->run.bol != 0 && !windows && !runs_first && bol_dense_general;
  if (rp->run.ok && (runs_first || s->linear_hint || window_runs || (!windows && !bol_dense && (!fresh || rp->stream.n_pos == 0 || s->streams_off)))) {
    int rc = run_runs(s, d_text, n, sb, se, carry_cur, carry_prev_end, have_prev, st);
    if (rc != 0) {
      // (fewer than a match per 64 KiB: at ~11 ns per hit the window path wins below a hit per ~32 KiB)
      if (rc == 1 && windows && !s->linear_hint) s->runs_sparse = s->result_count * 65536 < n - std::min(sb, n);
      return rc < 0 ? rc : RJ_OK;
    }
  }
  if (s->linear_hint && linear_path_fits(rp)) return run_linear(s, d_text, n, sb, se, carry_cur, carry_prev_end, have_prev, st);
  if (!windows && fresh) {
    int rc = run_assertions(s, d_text, n, sb, se, st);
    if (rc != 0) return rc < 0 ? rc : RJ_OK;
    rc = run_streams(s, d_text, n, sb, se, st);
    if (rc != 0) return rc < 0 ? rc : RJ_OK;
    if (rp->run.ok && s->streams_off) {   // (dense_streams has just given this text up: long runs)
      rc = run_runs(s, d_text, n, sb, se, carry_cur, carry_prev_end, have_prev, st);
      if (rc != 0) return rc < 0 ? rc : RJ_OK;
    }
  }

  // what the scan kernel walks, in 1-KiB chunks
  ScanParams sp{};
  sp.text = d_text;
  sp.n = n;
  sp.sb = sb;
  sp.se = se;
  uint64_t first_chunk, end_chunk;
  const uint32_t expand = windows ? D.float_range : 1;
  if (windows) {
    // fixed windows: w = s + offset.  floating: w in [s + float_min, s + float_max]
    const uint64_t float_min = D.float_max + 1 - D.float_range;
    sp.wlo = sb + (expand > 1 ? float_min : D.win_offset);
    const uint64_t last_w = n >= D.win_len ? n - D.win_len + 1 : 0;  // a window must fit: w + len <= n
    sp.whi = behind ? last_w : std::min(se + (expand > 1 ? D.float_max : D.win_offset), last_w);
    if (sp.whi < sp.wlo) sp.whi = sp.wlo;
    first_chunk = sp.wlo / 1024;
    end_chunk = (sp.whi + 1023) / 1024;
  } else {
    first_chunk = sb / 1024;
    end_chunk = (se + 1023) / 1024;
  }
  const uint64_t chunks = end_chunk > first_chunk ? end_chunk - first_chunk : 0;
  const ScanGeometry geo = scan_geometry(std::max<uint64_t>(chunks, 1));
  sp.span_chunks = geo.span_chunks;
  // dense: any position may be a hit; windows: start small, grow on overflow
  const uint64_t region_full = geo.span_chunks * 1024;
  // dense mode with a lane-sized automaton: one kernel finds, walks and compacts the candidates,
  // so its regions hold verified matches only (few) and are sized like the windows regions
  static const bool no_dense_walk = getenv("RJ_NO_DENSE_WALK") != nullptr;  // measurement override
  const bool dense_walk = !windows && dense_walk_fits(D) && !no_dense_walk;
  uint64_t region_cap = windows      ? std::min<uint64_t>(std::max<uint64_t>(s->region_cap_hint, 64), region_full)
                        : dense_walk ? std::min<uint64_t>(std::max<uint64_t>(s->region_cap_hint, 256), region_full)
                                     : region_full;

  // fixed windows + an automaton that fits a lane: candidates are verified and compacted inside
  // their hit regions (no global compaction, no sort)
  static const bool no_float_regions = getenv("RJ_NO_FLOAT_REGIONS") != nullptr;  // measurement override
  const bool floating_regions = windows && expand > 1 && D.n_words <= 4 && !no_float_regions;
  const bool in_regions = (windows && D.n_words <= 4 && (expand == 1 || floating_regions)) || dense_walk;  // (behind: n_words <= 4 by plan)

  for (int attempt = 0; attempt < 6; attempt++) {
    const uint64_t slots = static_cast<uint64_t>(geo.n_regions) * region_cap;
    // candidate slots: one per (hit, possible start); floating windows start with room for a few
    // thousand hits and grow when a run needs more
    uint64_t cand_slots = (expand == 1 || floating_regions) ? slots : std::max<uint64_t>(s->hits_hint * expand * 2, 1u << 16);
    int rc = ensure_lists(s, geo.n_regions, static_cast<uint32_t>(region_cap), std::max<uint64_t>(cand_slots, 1u << 12));
    if (rc != RJ_OK) return rc;
    if (in_regions) RJ_HIP(s->valid_counts.reserve(static_cast<size_t>(geo.n_regions) * sizeof(uint32_t)));
    sp.hits = s->hits.as<uint64_t>();
    sp.region_cap = static_cast<uint32_t>(region_cap);
    sp.hit_counts = s->hit_counts.as<uint32_t>();
    if (in_regions && windows) sp.zero_counters = s->counters.as<unsigned long long>();  // the scan kernel clears them
    else RJ_HIP(hipMemsetAsync(s->counters.p, 0, kCntSize * sizeof(unsigned long long), st));
    if (windows) {
      const WindowSet ws = make_window_set(rp);
      launch_scan_windows(sp, ws, D.n_windows, geo.grid, s->t0(), s->ev[2], st);
    } else if (dense_walk) {
      launch_scan_dense_walk(sp, D, geo.grid, s->cand_end.as<uint64_t>(), s->counters.as<unsigned long long>(), s->t0(),
                             s->ev[2], st);
    } else {
      launch_scan_dense(sp, D, geo.grid, s->t0(), s->ev[2], st);
    }
    FinalizeParams fp{};
    fp.cand_begin = s->cand_begin.as<uint64_t>();
    fp.cand_end = s->cand_end.as<uint64_t>();
    fp.cands_cap = s->cands_cap;
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
    fp.detect_conflict = behind;
    fp.expand = expand;
    VerifyParams vp{};
    if (!in_regions)
      launch_region_offsets(s->hit_counts.as<uint32_t>(), geo.n_regions, static_cast<uint32_t>(region_cap),
                            s->hit_offsets.as<uint64_t>(), s->counters.as<unsigned long long>(), st);
    vp.text = d_text;
    vp.n = n;
    vp.hits = s->hits.as<uint64_t>();
    vp.offsets = s->hit_offsets.as<uint64_t>();
    vp.n_regions = geo.n_regions;
    vp.region_cap = static_cast<uint32_t>(region_cap);
    vp.cand_begin = s->cand_begin.as<uint64_t>();
    vp.cand_end = s->cand_end.as<uint64_t>();
    vp.counters = s->counters.as<unsigned long long>();
    vp.sb = sb;
    vp.se = se;
    vp.expand = expand;
    vp.float_max = D.float_max;
    if (expand > 1 && !floating_regions) {
      // the slot count depends on the hit count, which only the device knows yet: verify must not
      // write past the candidate arrays, so floating runs read the count first (hits are rare)
      RJ_HIP(hipMemcpyAsync(s->host_counters, s->counters.p, kCntSize * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
      RJ_HIP(hipStreamSynchronize(st));
      if (s->host_counters[kCntOverflow] == 0) {
        const uint64_t need = s->host_counters[kCntHits] * expand;
        rc = ensure_lists(s, geo.n_regions, static_cast<uint32_t>(region_cap), std::max<uint64_t>(need, s->cands_cap));
        if (rc != RJ_OK) return rc;
        vp.cand_begin = s->cand_begin.as<uint64_t>();
        vp.cand_end = s->cand_end.as<uint64_t>();
      }
    }
    if (in_regions) {
      // verify + compact inside the regions, lay the survivors out, check / select: one sync
      // (verifying at the tail of the scan kernel instead was measured: 5 us slower per pass)
      const uint64_t* begins = s->hits.as<uint64_t>();
      // floating windows: the wave that verifies a region also applies the selection rule to the region's candidates,
      // which then usually are the result already (verify_lds.hip); not for patterns at risk of the ring artefact
      // (the adjacency test below wants every candidate)
      bool local_select = floating_regions && !rp->host->q8_risk && !s->no_local_select && getenv("RJ_NO_LOCAL_SELECT") == nullptr;
      if (floating_regions) {
        if (!launch_verify_floating_lds(vp, D, rp->walk, s->hit_counts.as<uint32_t>(), s->valid_counts.as<uint32_t>(),
                                        s->cand_begin.as<uint64_t>(), s->cand_end.as<uint64_t>(), local_select, st)) {
          local_select = false;
          launch_verify_floating_in_regions(vp, D, s->hit_counts.as<uint32_t>(), s->valid_counts.as<uint32_t>(),
                                            s->cand_begin.as<uint64_t>(), s->cand_end.as<uint64_t>(), st);
        }
        begins = s->cand_begin.as<uint64_t>();
      } else if (behind) {
        if (!launch_verify_behind_lds(vp, D, rp->walk, s->hit_counts.as<uint32_t>(), s->valid_counts.as<uint32_t>(),
                                      s->cand_end.as<uint64_t>(), st))
          launch_verify_behind_in_regions(vp, D, rp->rev, s->hit_counts.as<uint32_t>(), s->valid_counts.as<uint32_t>(),
                                          s->cand_end.as<uint64_t>(), st);
      } else if (!dense_walk) {
        launch_verify_in_regions(vp, D, s->hit_counts.as<uint32_t>(), s->valid_counts.as<uint32_t>(), s->cand_end.as<uint64_t>(), st);
      }
      const uint32_t* survivors = dense_walk ? s->hit_counts.as<uint32_t>() : s->valid_counts.as<uint32_t>();
      // (letting the last workgroup publish the counters to pinned host memory instead of the copy
      // below was measured: slower, its agent-scope fence writes L2 back)
      s->host_counters[kCntUnordered] = 0;  // the kernel below writes the pinned block itself
      s->host_counters[kCntAdjacent] = 0;
      s->host_counters[kCntConflict] = 0;
      // many candidates per region expected (the previous run had them, or the pattern is dense: its FIRST run
      // over a text has no history, and the one-launch form took 20 ms for the 8.4 M matches of `x*` over 16 MiB
      // where the second launch costs microseconds when there is little to copy): lay out first, then copy with a
      // wave per region
      uint64_t *off_scratch = nullptr, *prev_scratch = nullptr;
      if (dense_walk || s->hits_hint > static_cast<uint64_t>(geo.n_regions) * 16) {
        RJ_HIP(s->hit_offsets.reserve((static_cast<size_t>(geo.n_regions) + 1) * sizeof(uint64_t)));
        RJ_HIP(s->scan_a.reserve(static_cast<size_t>(geo.n_regions) * sizeof(uint64_t)));
        off_scratch = s->hit_offsets.as<uint64_t>();
        prev_scratch = s->scan_a.as<uint64_t>();
      }
      launch_offsets_gather_check(survivors, begins, s->cand_end.as<uint64_t>(), geo.n_regions,
                                  static_cast<uint32_t>(region_cap), fp.carry_cur, s->out.as<uint64_t>(), s->out_cap,
                                  s->counters.as<unsigned long long>(), s->host_counters, off_scratch, prev_scratch, st,
                                  fp.carry_prev_end, fp.have_prev);
      RJ_HIP(hipStreamSynchronize(st));
      RJ_HIP(hipGetLastError());
      if (local_select && s->host_counters[kCntUnordered] != 0 && s->host_counters[kCntOverflow] == 0 && s->host_counters[kCntOverrun] == 0) {
        // a match reaches from one region into the candidates of the next: the general selection needs every
        // candidate -- verify once more without the in-region rule (the hit lists are untouched: floating
        // candidates go to their own arrays), and remember it for this scan's next runs
        s->no_local_select = true;
        s->stats.retries++;
        launch_verify_floating_lds(vp, D, rp->walk, s->hit_counts.as<uint32_t>(), s->valid_counts.as<uint32_t>(),
                                   s->cand_begin.as<uint64_t>(), s->cand_end.as<uint64_t>(), false, st);
        s->host_counters[kCntUnordered] = 0;
        launch_offsets_gather_check(survivors, begins, s->cand_end.as<uint64_t>(), geo.n_regions,
                                    static_cast<uint32_t>(region_cap), fp.carry_cur, s->out.as<uint64_t>(), s->out_cap,
                                    s->counters.as<unsigned long long>(), s->host_counters, off_scratch, prev_scratch, st,
                                    fp.carry_prev_end, fp.have_prev);
        RJ_HIP(hipStreamSynchronize(st));
        RJ_HIP(hipGetLastError());
      }
      if (s->host_counters[kCntUnordered] != 0 && s->host_counters[kCntOverflow] == 0) {
        // not the result yet: the selection kernels work on begin[] / end[]
        const uint64_t nc = s->host_counters[kCntCands];
        RJ_HIP(s->keys_out.reserve(nc * sizeof(uint64_t)));
        RJ_HIP(s->vals_out.reserve(nc * sizeof(uint64_t)));
        launch_split_pairs(s->out.as<uint64_t>(), s->counters.as<unsigned long long>() + kCntCands, nc,
                           s->keys_out.as<uint64_t>(), s->vals_out.as<uint64_t>(), st);
        if (behind && nc > kFinalizeCap) {
          // one candidate per hit, in hit order: their begins are not sorted (a later hit may have an
          // earlier left-most start).  Few candidates are sorted by finalize_small; many here.
          unsigned bits = 1;
          while (bits < 64 && ((n + 1) >> bits) != 0) bits++;
          RJ_HIP(s->cand_begin.reserve(nc * sizeof(uint64_t)));
          uint64_t* k2 = s->cand_begin.as<uint64_t>();
          uint64_t* v2 = s->cand_end.as<uint64_t>();   // (region ends: consumed by the gather already)
          RJ_HIP(s->cand_end.reserve(nc * sizeof(uint64_t)));
          v2 = s->cand_end.as<uint64_t>();
          size_t tmp_bytes = 0;
          RJ_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, s->keys_out.as<uint64_t>(), k2, s->vals_out.as<uint64_t>(), v2, nc, 0, bits, st));
          RJ_HIP(s->sort_tmp.reserve(std::max<size_t>(tmp_bytes, 16)));
          RJ_HIP(rocprim::radix_sort_pairs(s->sort_tmp.p, tmp_bytes, s->keys_out.as<uint64_t>(), k2, s->vals_out.as<uint64_t>(), v2, nc, 0, bits, st));
          RJ_HIP(hipMemcpyAsync(s->keys_out.p, k2, nc * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
          RJ_HIP(hipMemcpyAsync(s->vals_out.p, v2, nc * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
        }
        if (fp.detect_adjacent) {
          // (overlapping candidates: adjacency is no longer a neighbour property)
          launch_detect_adjacent(s->keys_out.as<uint64_t>(), s->vals_out.as<uint64_t>(), nc, s->counters.as<unsigned long long>(), st);
          RJ_HIP(hipMemcpyAsync(&s->host_counters[kCntAdjacent], s->counters.as<unsigned long long>() + kCntAdjacent,
                                sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
          RJ_HIP(hipStreamSynchronize(st));
        }
      }
      rc = resolve_selection(s, fp, st);
      if (rc != RJ_OK) return rc;
      if (behind && s->host_counters[kCntOverflow] == 0 && (s->host_counters[kCntConflict] != 0 || s->host_counters[kCntOverrun] != 0)) {
        // a hidden candidate reaches beyond the match that hides it, or a walk from a hit ran into the
        // limit: every start has to be considered -- the dense path (and behind it the carry scan)
        s->behind_conflicts = true;
        s->stats.retries++;
        return run_range(s, d_text, n, sb, se, carry_cur, carry_prev_end, have_prev, st, true);
      }
    } else {
      launch_verify(vp, D, windows ? std::max<uint64_t>(s->hits_hint, 1u << 14) : (se - sb) / 8 + 1, st);
      launch_finalize_small(fp, st);
      RJ_HIP(hipMemcpyAsync(s->host_counters, s->counters.p, kCntSize * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
      RJ_HIP(hipStreamSynchronize(st));
      RJ_HIP(hipGetLastError());
    }
    const unsigned long long n_hits = s->host_counters[kCntHits];
    float ms = 0.f;
    if (s->timing) (void)hipEventElapsedTime(&ms, s->ev[1], s->ev[2]);
    s->stats.scan_ms += ms;
    s->stats.n_hits += n_hits;
    if (s->host_counters[kCntOverflow] != 0 && s->host_counters[kCntOverrun] == 0) {
      // a region overflowed: size every region for the fullest one seen (x2) and run again
      s->stats.retries++;
      const uint64_t want = std::min<uint64_t>(std::max<uint64_t>(s->host_counters[kCntMaxRegion] * 2, region_cap * 4), region_full);
      if (want <= region_cap) return fail(RJ_DEVICE_ERROR, "hit regions cannot grow further");
      region_cap = want;
      s->region_cap_hint = static_cast<uint32_t>(std::min<uint64_t>(want, 1u << 20));
      s->stats.n_hits -= n_hits;
      continue;
    }
    s->hits_hint = n_hits;
    if (windows && rp->run.ok && n_hits * 32768 > se - sb) s->window_dense = true;   // (a hit every 32 KiB or denser: the run kernels from now on)
    if (s->host_counters[kCntOverrun] != 0) {
      // some start was still alive after max_walk bytes (an unbounded repetition over a long run):
      // walking every start on its own is quadratic there.  The carry scan is linear in the text
      // whatever the pattern, like the reference's loop (codegen-x64.cc:535-640).
      s->linear_hint = true;
      s->stats.retries++;
      if (rp->run.ok) {   // one long-lived thread in one loop position: the run kernels (two passes over the text)
        rc = run_runs(s, d_text, n, sb, se, carry_cur, carry_prev_end, have_prev, st);
        if (rc != 0) return rc < 0 ? rc : RJ_OK;
      }
      return run_linear(s, d_text, n, sb, se, carry_cur, carry_prev_end, have_prev, st);
    }
    if (in_regions) {
      // check_and_select produced the result
    } else if (s->host_counters[kCntFinal] == ~0ull) {
      rc = finalize_large(s, n_hits * expand, expand > 1, n + 1, fp, st);
      if (rc != RJ_OK) return rc;
    } else {
      s->result_count = s->host_counters[kCntFinal];
      s->stats.n_candidates += s->host_counters[kCntCands];
    }
    // (run_pipeline then takes the reference's own answer from the exact replay)
    if (fp.detect_adjacent && s->host_counters[kCntAdjacent] != 0) s->want_exact = true;
    return RJ_OK;
  }
  return fail(RJ_DEVICE_ERROR, "hit regions kept overflowing");
}

// automata too wide for the exact replay's synchronisation scan: the reference's loop on ONE lane over the
// whole text, up to kExactLimit bytes
static int run_exact_one_lane(rj_scan* s, const uint8_t* d_text, uint64_t n, hipStream_t st) {
  const rj_program* rp = s->prog;
  RJ_HIP(s->ring.reserve(static_cast<size_t>(rp->graph.times) * rp->graph.n_states * sizeof(int64_t)));
  int rc = ensure_lists(s, 1, 1, std::max<uint64_t>(s->cands_cap, n + 2));
  if (rc != RJ_OK) return rc;
  launch_exact_sequential(d_text, n, rp->graph, s->ring.as<int64_t>(), s->out.as<uint64_t>(), s->out_cap,
                          s->counters.as<unsigned long long>(), st);
  RJ_HIP(hipMemcpyAsync(s->host_counters, s->counters.p, kCntSize * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  RJ_HIP(hipStreamSynchronize(st));
  RJ_HIP(hipGetLastError());
  s->result_count = s->host_counters[kCntFinal];
  s->result = s->out.as<uint64_t>();
  return RJ_OK;
}

int run_pipeline(rj_scan* s, const uint8_t* d_text, uint64_t n, uint64_t sb, uint64_t se, uint64_t carry_cur,
                 uint64_t carry_prev_end, int have_prev, hipStream_t st) {
  const rj_program* rp = s->prog;
  if (se > n + 1) se = n + 1;
  s->stats = rj_stats{};
  s->result = nullptr;
  s->result_count = 0;
  s->want_exact = false;
  if (sb >= se) return RJ_OK;
  if ((reinterpret_cast<uintptr_t>(d_text) & 15u) != 0) return fail(RJ_BAD_ARGUMENT, "device text must be 16-byte aligned");
  const auto wall0 = std::chrono::steady_clock::now();
  const bool whole_text = sb == 0 && se == n + 1 && carry_cur == 0 && !have_prev;
  if (rp->host->q8_risk && !whole_text && exact_replay_fits(rp)) {
    // A pattern at risk of the reference's ring artefact (Q8) over a RANGE of the text: the artefact's
    // state crosses any cut that is not a synchronisation point of the reference's loop, so the range owns
    // whole segments between such points -- [first point >= sb, first point >= se) -- and replays the
    // reference's loop over them (exact_replay.hip).  Neighbouring ranges agree on the points, their
    // results concatenate to the reference's answer, and no carry is needed (nothing crosses a point).
    int rc = run_exact(s, d_text, n, sb, se, st);
    if (rc < 0) return rc;
    if (rc == 1) {
      s->stats.exact_path = s->xr_parts != 0 ? 2 : 1;
      s->stats.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - wall0).count();
      s->stats.n_matches = s->result_count;
      return RJ_OK;
    }
    // (a stretch too long to replay: the documented semantics, as for any other pattern)
  }
  static const bool no_small = getenv("RJ_NO_SMALL") != nullptr;  // measurement override
  if (n <= kSmallMaxText && rp->dev.n_words <= 4 && rp->dev.table_words <= kSmallMaxTableWords && !no_small &&
      small_lds_bytes(rp->dev, static_cast<uint32_t>(n)) <= small_lds_limit()) {
    // one launch, one synchronise: the whole MatchAll in one workgroup (kernels.hip: match_small)
    if (s->small_out == nullptr) {
      // (pinned, mapped, coherent host memory: the kernel writes the result where the host reads it)
      RJ_HIP(hipHostMalloc(reinterpret_cast<void**>(&s->small_out), static_cast<size_t>(kSmallMaxCands) * 2 * sizeof(uint64_t),
                           hipHostMallocCoherent | hipHostMallocMapped));
      RJ_HIP(hipHostMalloc(reinterpret_cast<void**>(&s->small_hdr), 2 * sizeof(unsigned long long), hipHostMallocCoherent | hipHostMallocMapped));
    }
    SmallParams sp{};
    sp.text = d_text;
    sp.n = static_cast<uint32_t>(n);
    sp.sb = static_cast<uint32_t>(sb);
    sp.se = static_cast<uint32_t>(se);
    sp.carry_cur = carry_cur;
    sp.carry_prev_end = carry_prev_end;
    sp.have_prev = have_prev;
    sp.q8_risk = rp->host->q8_risk ? 1 : 0;
    sp.out = s->small_out;
    sp.out_cap = kSmallMaxCands;
    sp.hdr = s->small_hdr;
    // (Polling the pinned header instead of waiting for the stream was tried: the call's latency did not
    // move -- it is launch + kernel, not the wake-up -- and results occasionally arrived stale.)
    s->small_hdr[1] = 1;
    launch_match_small(sp, rp->dev, st);
    RJ_HIP(hipStreamSynchronize(st));
    RJ_HIP(hipGetLastError());
    if (s->small_hdr[1] == 0) {
      s->result_count = s->small_hdr[0];
      s->result = s->small_out;   // pinned host memory: readable from the device and from the host
      s->stats.n_candidates = s->result_count;
      s->stats.n_matches = s->result_count;
      s->stats.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - wall0).count();
      return RJ_OK;
    }
    // too many candidates, a long walk, or a Q8-sensitive adjacency: the general pipeline
  }
  const bool windows = rp->dev.mode == 1;
  // (the pair kernels take whole texts only: a dense-mode pair pattern -- `["'][^"']*["']` -- must not be cut into segments)
  const bool single_run = windows || dense_walk_fits(rp->dev) || se - sb <= kDenseSegment || (s->linear_hint && linear_path_fits(rp)) ||
                          (rp->run.pair != 0 && whole_text);
  if (single_run) {
    int rc = run_range(s, d_text, n, sb, se, carry_cur, carry_prev_end, have_prev, st);
    if (rc != RJ_OK) return rc;
    // (the carry scan may have accumulated segments elsewhere; a count-only run of the run kernels leaves no list)
    if (s->result == nullptr && !(s->count_only_run && s->stats.run_path)) s->result = s->out.as<uint64_t>();
  } else {
    // Dense mode over a long range: a hit slot for a sizeable fraction of the bytes would not
    // fit, so the starts are walked in segments; the selection state is carried from one
    // segment to the next exactly as between the shards of a multi-GPU run.
    uint64_t total = 0;
    for (uint64_t lo = sb, hi; lo < se; lo = hi) {
      // (once a segment needed the carry scan the rest of the range goes there in one piece: its
      // summaries cover the text to its end, so many small runs would repeat that work)
      hi = (s->linear_hint && linear_path_fits(rp)) ? se : std::min(se, lo + kDenseSegment);
      int rc = run_range(s, d_text, n, lo, hi, carry_cur, carry_prev_end, have_prev, st);
      if (rc != RJ_OK) return rc;
      if (s->result_count) {
        const uint64_t* from = s->result ? s->result : s->out.as<uint64_t>();
        RJ_HIP(s->acc_out.grow_keep((total + s->result_count) * 2 * sizeof(uint64_t), total * 2 * sizeof(uint64_t)));
        RJ_HIP(hipMemcpyAsync(s->acc_out.as<uint64_t>() + 2 * total, from, s->result_count * 2 * sizeof(uint64_t),
                              hipMemcpyDeviceToDevice, st));
        uint64_t last[2];
        RJ_HIP(hipMemcpyAsync(last, from + 2 * (s->result_count - 1), sizeof(last), hipMemcpyDeviceToHost, st));
        RJ_HIP(hipStreamSynchronize(st));
        carry_cur = last[1] > last[0] ? last[1] : last[0] + 1;
        carry_prev_end = last[1];
        have_prev = 1;
        total += s->result_count;
      }
    }
    s->result_count = total;
    s->result = s->acc_out.as<uint64_t>();
  }
  if (rp->host->q8_risk && whole_text && (s->want_exact || s->stats.linear_path || !single_run)) {
    // the result may differ from the reference's by the ring artefact (a candidate begins exactly where
    // another ends; the carry scan and the segmented dense runs do not look for that): take the
    // reference's own answer
    int rc = 0;
    if (exact_replay_fits(rp)) rc = run_exact(s, d_text, n, 0, n + 1, st);
    else if (n <= kExactLimit && rp->graph.n_states > 0) rc = run_exact_one_lane(s, d_text, n, st) == RJ_OK ? 1 : RJ_DEVICE_ERROR;
    if (rc < 0) return rc;
    if (rc == 1) s->stats.exact_path = s->xr_parts != 0 ? 2 : 1;
  }
  // (every run_range ends with a stream synchronise, so the host clock covers the whole pipeline
  // and no event commands are needed on the stream)
  s->stats.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  s->stats.n_matches = s->result_count;
  return RJ_OK;
}

static int scan_init(rj_scan* s) {
  RJ_HIP(s->counters.reserve(kCounterBlockWords * sizeof(unsigned long long)));
  RJ_HIP(hipMemset(s->counters.p, 0, kCounterBlockWords * sizeof(unsigned long long)));
  RJ_HIP(s->flag.reserve(16));
  RJ_HIP(hipHostMalloc(reinterpret_cast<void**>(&s->host_counters), kCntSize * sizeof(unsigned long long)));
  RJ_HIP(hipHostMalloc(reinterpret_cast<void**>(&s->host_flag), 16));
  for (auto& e : s->ev) RJ_HIP(hipEventCreate(&e));
  return RJ_OK;
}

}  // namespace rejit_amd

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
  static std::atomic<uint64_t> next_id{1};
  rp->id = next_id.fetch_add(1);
  int rc = upload_program(rp.get());
  if (rc != RJ_OK) return rc;
  *out = rp.release();
  return RJ_OK;
}

void rj_program_free(rj_program* prog) {
  ErrnoGuard errno_guard;
  if (!prog) return;
  forget_host_scans(prog->id);  // (host_api.hip: this thread's cached scans of the program)
  forget_combiner(prog->id);
  for (rj_program* r : prog->replicas)
    if (r != nullptr) rj_program_free(r);
  delete prog;
}

int rj_program_info(const rj_program* prog, rj_info* info) {
  if (!prog || !info) return fail(RJ_BAD_ARGUMENT, "null argument");
  const Program& P = *prog->host;
  info->n_positions = P.n_pos;
  info->n_words = P.n_words;
  info->has_assertions = P.has_assertions;
  info->scan_mode = prog->dev.mode;  // the mode the kernels run (the engine may prefer dense over weak windows)
  info->n_windows = static_cast<int32_t>(P.windows.size());
  info->window_offset = prog->dev.win_offset;
  info->window_len = prog->dev.win_len;
  info->min_len = P.min_len;
  info->max_len = P.max_len;
  info->ring_artefact_risk = P.q8_risk ? 1 : 0;
  info->reserved = 0;
  return RJ_OK;
}

namespace {
// rj_set_default_timing: whether NEW rj_scan objects take the scan kernel's start event (rj_stats.scan_ms).  Measured on the
// regexdna step: the start event of hipExtLaunchKernelGGL costs ~6.5 us between two kernels of a stream, the end event nothing.
std::atomic<int>* default_timing() {   // (a pointer: this sits inside the extern "C" block, a reference to a class type draws a warning)
  static std::atomic<int> v{getenv("RJ_KERNEL_TIMING") && atoi(getenv("RJ_KERNEL_TIMING")) != 0 ? 1 : 0};
  return &v;
}
}  // namespace

int rj_set_default_timing(int on) { return default_timing()->exchange(on != 0 ? 1 : 0); }

int rj_scan_set_timing(rj_scan* s, int on) {
  if (!s) return fail(RJ_BAD_ARGUMENT, "null scan");
  s->timing = on != 0;
  return RJ_OK;
}

int rj_scan_create(const rj_program* prog, rj_scan** out) {
  ErrnoGuard errno_guard;
  if (!prog || !out) return fail(RJ_BAD_ARGUMENT, "null argument");
  auto s = std::make_unique<rj_scan>();
  s->prog = prog;
  s->timing = default_timing()->load() != 0;
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
  if (s->pinned) (void)hipHostFree(s->pinned);
  if (s->small_out) (void)hipHostFree(s->small_out);
  if (s->small_hdr) (void)hipHostFree(s->small_hdr);
  if (s->small_text) (void)hipHostFree(s->small_text);
  if (s->gx_host) (void)hipHostFree(s->gx_host);
  for (auto& e : s->ev)
    if (e) (void)hipEventDestroy(e);
  if (s->own_stream) (void)hipStreamDestroy(s->own_stream);
  if (s->tail_stream) (void)hipStreamDestroy(s->tail_stream);
  if (s->counter) rj_multi_destroy(s->counter);
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

// Start / finish: the scan of the next pattern does not have to wait for the (latency-bound)
// verify + gather kernels of this one.  rj_scan_start enqueues the scan on the caller's stream and
// the tail on the scan object's own stream (ordered after the scan by an event) and returns;
// rj_scan_finish waits for the tail.  Whole text, no carry.  Anything but the common outcome (a
// region overflowed, candidates overlap, a pattern the in-region pipeline does not take) is handled
// by running the ordinary synchronous pipeline in rj_scan_finish.
static int scan_start(rj_scan* s, const void* d_text, uint64_t n, void* hip_stream) {
  if (!s || (!d_text && n)) return fail(RJ_BAD_ARGUMENT, "null argument");
  if (s->pending) return fail(RJ_BAD_ARGUMENT, "rj_scan_start: the previous start has not been finished");
  if ((reinterpret_cast<uintptr_t>(d_text) & 15u) != 0) return fail(RJ_BAD_ARGUMENT, "device text must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  s->pending = true;
  s->pending_launched = false;
  s->pending_text = static_cast<const uint8_t*>(d_text);
  s->pending_n = n;
  s->pending_stream = st;
  const rj_program* rp = s->prog;
  const DevProgram& D = rp->dev;
  if (!(D.mode == 1 && D.float_range == 1 && D.n_words <= 4 && !rp->host->q8_risk && !D.behind && n >= 16)) return RJ_OK;
  // (default priority: with a high-priority tail stream the runtime preempts the running scan's
  // waves for every tail kernel -- measured 1.13 -> 1.68 ms per step.  At equal priority the tails
  // mostly run when the queued scans have drained; what is saved is the host round trip per call.)
  if (!s->tail_stream) RJ_HIP(hipStreamCreateWithFlags(&s->tail_stream, hipStreamNonBlocking));
  ScanParams sp{};
  sp.text = s->pending_text;
  sp.n = n;
  sp.sb = 0;
  sp.se = n + 1;
  sp.wlo = D.win_offset;
  const uint64_t last_w = n >= D.win_len ? n - D.win_len + 1 : 0;
  sp.whi = std::min<uint64_t>(n + 1 + D.win_offset, last_w);
  if (sp.whi < sp.wlo) sp.whi = sp.wlo;
  const uint64_t first_chunk = sp.wlo / 1024, end_chunk = (sp.whi + 1023) / 1024;
  const ScanGeometry geo = scan_geometry(std::max<uint64_t>(end_chunk > first_chunk ? end_chunk - first_chunk : 0, 1));
  sp.span_chunks = geo.span_chunks;
  const uint64_t region_cap = std::min<uint64_t>(std::max<uint64_t>(s->region_cap_hint, 64), geo.span_chunks * 1024);
  int rc = ensure_lists(s, geo.n_regions, static_cast<uint32_t>(region_cap), static_cast<uint64_t>(geo.n_regions) * region_cap);
  if (rc != RJ_OK) return rc;
  RJ_HIP(s->valid_counts.reserve(static_cast<size_t>(geo.n_regions) * sizeof(uint32_t)));
  sp.hits = s->hits.as<uint64_t>();
  sp.region_cap = static_cast<uint32_t>(region_cap);
  sp.hit_counts = s->hit_counts.as<uint32_t>();
  sp.zero_counters = s->counters.as<unsigned long long>();
  s->stats = rj_stats{};
  s->result = nullptr;
  s->result_count = 0;
  launch_scan_windows(sp, make_window_set(rp), D.n_windows, geo.grid, s->t0(), s->ev[2], st);
  RJ_HIP(hipEventRecord(s->ev[0], st));
  RJ_HIP(hipStreamWaitEvent(s->tail_stream, s->ev[0], 0));
  VerifyParams vp{};
  vp.text = s->pending_text;
  vp.n = n;
  vp.hits = s->hits.as<uint64_t>();
  vp.n_regions = geo.n_regions;
  vp.region_cap = static_cast<uint32_t>(region_cap);
  vp.counters = s->counters.as<unsigned long long>();
  vp.sb = 0;
  vp.se = n + 1;
  vp.expand = 1;
  vp.float_max = D.float_max;
  launch_verify_in_regions(vp, D, s->hit_counts.as<uint32_t>(), s->valid_counts.as<uint32_t>(), s->cand_end.as<uint64_t>(),
                           s->tail_stream);
  s->host_counters[kCntUnordered] = 0;
  s->host_counters[kCntAdjacent] = 0;
  launch_offsets_gather_check(s->valid_counts.as<uint32_t>(), s->hits.as<uint64_t>(), s->cand_end.as<uint64_t>(), geo.n_regions,
                              static_cast<uint32_t>(region_cap), 0, s->out.as<uint64_t>(), s->out_cap,
                              s->counters.as<unsigned long long>(), s->host_counters, nullptr, nullptr, s->tail_stream);
  s->pending_launched = true;
  return RJ_OK;
}

int rj_scan_start(rj_scan* s, const void* d_text, uint64_t n, void* hip_stream) {
  ErrnoGuard errno_guard;
  const bool was_pending = s && s->pending;
  const int rc = scan_start(s, d_text, n, hip_stream);
  if (rc != RJ_OK && s && !was_pending) s->pending = false;  // a failed start leaves nothing to finish
  return rc;
}

int64_t rj_scan_finish(rj_scan* s) {
  ErrnoGuard errno_guard;
  if (!s) return fail(RJ_BAD_ARGUMENT, "null scan");
  if (!s->pending) return fail(RJ_BAD_ARGUMENT, "rj_scan_finish without rj_scan_start");
  s->pending = false;
  if (s->pending_launched) {
    RJ_HIP(hipStreamSynchronize(s->tail_stream));
    RJ_HIP(hipGetLastError());
    const unsigned long long* hc = s->host_counters;
    if (hc[kCntOverflow] == 0 && hc[kCntOverrun] == 0 && hc[kCntUnordered] == 0) {
      float ms = 0.f;
      if (s->timing) (void)hipEventElapsedTime(&ms, s->ev[1], s->ev[2]);
      s->stats.scan_ms = ms;
      s->stats.n_hits = hc[kCntHits];
      s->stats.n_candidates = hc[kCntCands];
      s->hits_hint = hc[kCntHits];
      s->result_count = hc[kCntCands];
      s->result = s->out.as<uint64_t>();
      s->stats.n_matches = s->result_count;
      return static_cast<int64_t>(s->result_count);
    }
    if (hc[kCntOverflow] != 0)  // size the regions for the fullest one before running again
      s->region_cap_hint = static_cast<uint32_t>(std::min<uint64_t>(std::max<uint64_t>(hc[kCntMaxRegion] * 2, 256), 1u << 20));
  }
  int rc = run_pipeline(s, s->pending_text, s->pending_n, 0, s->pending_n + 1, 0, 0, 0, s->pending_stream);
  if (rc != RJ_OK) return rc;
  return static_cast<int64_t>(s->result_count);
}

int64_t rj_scan_copy_spans(const rj_scan* s, uint64_t* host_spans, uint64_t cap) {
  ErrnoGuard errno_guard;
  if (!s) return fail(RJ_BAD_ARGUMENT, "null scan");
  const uint64_t k = std::min<uint64_t>(cap, s->result_count);
  if (k && !s->result) return fail(RJ_BAD_ARGUMENT, "the last run was counts-only: there is no span list (rj_multi_set_counts_only / rj_scan_count)");
  if (k) {
    // (hipMemcpyDefault: the destination may be host or device memory.  The run has been synchronised
    // before it returned; the copy goes over a non-blocking stream of the calling thread -- hipMemcpy on the
    // null stream would wait for every other thread's blocking streams)
    static thread_local hipStream_t copy_stream = nullptr;
    static thread_local int copy_device = -1;
    int dev = 0;
    RJ_HIP(hipGetDevice(&dev));
    if (copy_stream == nullptr || copy_device != dev) {
      RJ_HIP(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));  // (one per thread and device change; never destroyed)
      copy_device = dev;
    }
    RJ_HIP(hipMemcpyAsync(host_spans, s->result, k * 2 * sizeof(uint64_t), hipMemcpyDefault, copy_stream));
    RJ_HIP(hipStreamSynchronize(copy_stream));
  }
  return static_cast<int64_t>(s->result_count);
}

int64_t rj_scan_copy_gathered_spans(const rj_scan* s, uint64_t* host_spans, uint64_t cap) {
  ErrnoGuard errno_guard;
  if (!s) return fail(RJ_BAD_ARGUMENT, "null scan");
  if (!s->gathered) return 0;   // (not the root, or no gather yet)
  const uint64_t k = std::min<uint64_t>(cap, s->gathered_count);
  if (k) {
    if (!host_spans) return fail(RJ_BAD_ARGUMENT, "null argument");
    RJ_HIP(hipMemcpy(host_spans, s->gathered, k * 2 * sizeof(uint64_t), hipMemcpyDefault));
  }
  return static_cast<int64_t>(s->gathered_count);
}

int64_t rj_scan_replace(rj_scan* s, const void* d_text, uint64_t n, const char* with, uint64_t with_len, void* d_out,
                        uint64_t out_cap, void* hip_stream) {
  ErrnoGuard errno_guard;
  if (!s || (!with && with_len)) return fail(RJ_BAD_ARGUMENT, "null argument");
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const uint64_t m = s->result_count;
  const uint64_t* spans = s->result;
  if (m && !spans) return fail(RJ_BAD_ARGUMENT, "the last run was counts-only: there is no span list to replace (rj_multi_set_counts_only / rj_scan_count)");
  RJ_HIP(s->with_buf.reserve(std::max<uint64_t>(with_len, 16)));
  if (with_len) RJ_HIP(hipMemcpyAsync(s->with_buf.p, with, with_len, hipMemcpyHostToDevice, st));
  RJ_HIP(s->scan_a.reserve(std::max<uint64_t>(m, 1) * sizeof(uint64_t)));
  RJ_HIP(s->scan_b.reserve(std::max<uint64_t>(m, 1) * sizeof(uint64_t)));
  RJ_HIP(s->long_gaps.reserve((m + 1) * 3 * sizeof(uint64_t)));
  RJ_HIP(hipMemsetAsync(s->counters.p, 0, kCntSize * sizeof(unsigned long long), st));
  if (m) {
    launch_match_lengths(spans, m, s->scan_a.as<uint64_t>(), st);
    size_t bytes = 0;
    RJ_HIP(rocprim::exclusive_scan(nullptr, bytes, s->scan_a.as<uint64_t>(), s->scan_b.as<uint64_t>(), uint64_t{0}, m,
                                   rocprim::plus<uint64_t>(), st));
    RJ_HIP(s->sort_tmp.reserve(std::max<size_t>(bytes, 16)));
    RJ_HIP(rocprim::exclusive_scan(s->sort_tmp.p, bytes, s->scan_a.as<uint64_t>(), s->scan_b.as<uint64_t>(), uint64_t{0}, m,
                                   rocprim::plus<uint64_t>(), st));
  }
  launch_replace_gather(static_cast<const uint8_t*>(d_text), n, spans, s->scan_b.as<uint64_t>(), m, s->with_buf.as<uint8_t>(),
                        with_len, static_cast<uint8_t*>(d_out), out_cap, s->long_gaps.as<uint64_t>(),
                        s->counters.as<unsigned long long>(), st);
  RJ_HIP(hipMemcpyAsync(s->host_counters, s->counters.p, kCntSize * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  RJ_HIP(hipStreamSynchronize(st));
  RJ_HIP(hipGetLastError());
  const uint64_t new_len = s->host_counters[kCntFinal];
  if (new_len > out_cap) return fail(RJ_BAD_ARGUMENT, "replace output needs %llu bytes", static_cast<unsigned long long>(new_len));
  return static_cast<int64_t>(new_len);
}

void rj_free_text(char* text) { free(text); }

int rj_scan_stats(const rj_scan* s, rj_stats* stats) {
  if (!s || !stats) return fail(RJ_BAD_ARGUMENT, "null argument");
  *stats = s->stats;
  return RJ_OK;
}

int rj_scan_stats_sized(const rj_scan* s, void* stats, size_t struct_size) {
  if (!s || !stats) return fail(RJ_BAD_ARGUMENT, "null argument");
  memcpy(stats, &s->stats, std::min(struct_size, sizeof(rj_stats)));   // (a caller built against an older header gets its own fields)
  return static_cast<int>(sizeof(rj_stats));
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

}  // extern "C"

