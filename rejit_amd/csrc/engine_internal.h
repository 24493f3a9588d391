// rejit_amd/csrc/engine_internal.h -- what the host-side translation units of the engine share:
// error plumbing, grow-only device buffers, and the objects behind the opaque C-ABI handles
// (include/rejit_hip.h).  engine.hip holds the device pipeline and the device-text C ABI (rj_compile,
// rj_scan_*), multi_pattern.hip rj_multi_*, host_api.hip the host-text entry points, linear.hip the linear-time
// carry scan, exact_replay.hip the reference-exact replay, multi_device.hip the split of one call over all
// visible GPUs.
#ifndef REJIT_AMD_ENGINE_INTERNAL_H_
#define REJIT_AMD_ENGINE_INTERNAL_H_

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cerrno>
#include <memory>
#include <string>
#include <vector>

#include "../../include/rejit_hip.h"
#include "device_program.h"
#include "kernels.h"
#include "lowering.h"

namespace rejit_amd {

// sets the calling thread's rj_last_error() text and returns `code`
int rj_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define RJ_HIP(call)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return ::rejit_amd::rj_fail(RJ_DEVICE_ERROR, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// The reference's generated code never touches errno, and its callers rely on that
// (sample/jrep.cc:281-285 tests `if (errno)` right after mmap).  The HIP runtime does set it
// (probing files, ioctls), so every entry point restores the caller's value.
struct ErrnoGuard {
  int saved;
  ErrnoGuard() : saved(errno) {}
  ~ErrnoGuard() { errno = saved; }
};

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
  // grow, preserving the first `keep` bytes
  hipError_t grow_keep(size_t want, size_t keep) {
    if (want <= bytes) return hipSuccess;
    void* q = nullptr;
    size_t cap = std::max(want, bytes * 2);
    hipError_t e = hipMalloc(&q, cap);
    if (e != hipSuccess) return e;
    if (p && keep) e = hipMemcpy(q, p, keep, hipMemcpyDeviceToDevice);
    if (p) (void)hipFree(p);
    p = q;
    bytes = cap;
    return e;
  }
  template <class T>
  T* as() const { return static_cast<T*>(p); }
};

}  // namespace rejit_amd

struct rj_program {
  std::unique_ptr<rejit_amd::Program> host;
  rejit_amd::DevProgram dev{};
  rejit_amd::DeviceBuffer tables;
  rejit_amd::DevProgram rev{};   // the reverse automaton (table fields only), see lowering.h Program::rev
  rejit_amd::DeviceBuffer rev_tables;
  // the same tables padded for the LDS walkers (lds_walk.h / verify_lds.hip): automata of <= 128 positions whose
  // scan plan has floating windows or windows behind an unbounded prefix; walk.blob == nullptr otherwise
  rejit_amd::WalkDesc walk{};
  rejit_amd::DeviceBuffer walk_tables;
  rejit_amd::DevGraph graph{};        // uploaded only for patterns with q8_risk
  rejit_amd::DeviceBuffer graph_blob;
  int device = 0;
  std::vector<rj_program*> replicas;  // multi_device.hip: the same pattern compiled on the other devices (index = device), lazily
  uint64_t id = 0;          // unique per compile: keys per-thread caches (a freed program's address may be reused)
  int window_alphabet = 0;  // distinct byte values among the fixed window bytes
  bool window_nibbles = false;  // those values differ in their low nibble (nibble filter usable)
  int batch_separator = -1;  // byte that ends a text inside a concatenated batch, -1: none exists
  rejit_amd::StreamPlan stream{};  // dense mode as bit streams (dense_streams.h): n_pos == 0 when the pattern does not qualify
  rejit_amd::RunPlan run{};        // one long-lived thread in one loop position (run_scan.h): ok == 0 when the pattern does not qualify
  std::string pattern;
};

struct rj_scan {
  const rj_program* prog = nullptr;
  rejit_amd::DeviceBuffer counters, hits, hit_counts, valid_counts, hit_offsets, cand_begin, cand_end, out, acc_out, keys_out, vals_out, sort_tmp, flag;
  rejit_amd::DeviceBuffer scan_a, scan_b, taken, chain_blocks;  // large-path selection scratch
  rejit_amd::DeviceBuffer ring;                   // exact sequential kernel / exact replay with a ring too big for LDS
  rejit_amd::DeviceBuffer xr_state, xr_sync, xr_seg_end, xr_offs, xr_counts, xr_scratch, xr_out;  // exact replay (exact_replay.hip)
  uint64_t xr_out_cap = 0;
  rejit_amd::DeviceBuffer xr_snaps, xr_raw_n, xr_fix_ring;  // long segments taken in parts (speculate and verify)
  uint64_t xr_parts = 0, xr_rounds = 0;                     // of the last exact replay: parts of long segments, rounds beyond the first
  bool want_exact = false;         // the run just made may differ from the reference by the ring artefact (Q8)
  rejit_amd::DeviceBuffer with_buf, long_gaps, repl_out;  // replace_gather
  uint64_t repl_len = 0;           // rj_replace_all_begin: the new text waits in repl_out for rj_replace_all_fetch
  bool repl_valid = false;
  // carry scan (linear.hip): summaries (resolved in place), reachability matrices, E / G slabs,
  // entry points, per-sub-chunk counts, wide-automaton scratch
  rejit_amd::DeviceBuffer cs_vals, cs_mats, cs_e, cs_g, cs_entry, cs_counts, cs_scratch, cs_acc, cs_groups;
  bool linear_hint = false;        // the previous run needed the carry scan: go there directly
  rejit_amd::DeviceBuffer run_summaries, run_tile_in;  // run_scan.hip
  bool count_only_run = false;   // (scan_count: this run's pairs are not wanted -- the run kernels stop behind their resolve)
  // a WINDOWS-mode run shape (`a.*b`, `#.*`, `<[^>]*>`, ` +`: the window is one byte) takes the run kernels first; runs_sparse: they
  // found few matches on this scan's text, the next run tries the window scan (faster when its hits are rare); window_dense: that
  // scan met dense hits on this scan's text (every hit a walk: 20-40 x slower than the run kernels) -- the run kernels for good
  bool runs_sparse = false, window_dense = false;
  bool streams_off = false;        // dense_streams ran into a void run or too many scalar walks on this scan's text: scan_dense_walk
  bool behind_conflicts = false;   // behind mode gave a conflict / overrun on this scan's text: stay dense
  bool no_local_select = false;    // floating windows: the in-region selection left overlapping candidates on this text
  uint64_t cands_cap = 0, out_cap = 0;
  uint32_t region_cap_hint = 64;   // hit-region size that sufficed last time (windows mode)
  uint64_t hits_hint = 0;          // hits of the previous run (sizes the verify grid)
  unsigned long long* host_counters = nullptr;  // pinned
  int* host_flag = nullptr;                     // pinned
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  bool timing = false;             // rj_scan_set_timing: the scan kernel's START event (rj_stats.scan_ms); its end event is always recorded
  hipEvent_t t0() const { return timing ? ev[1] : nullptr; }
  rj_stats stats{};
  const uint64_t* result = nullptr;  // device pointer to the final pairs
  uint64_t result_count = 0;
  // rj_scan_start / rj_scan_finish
  hipStream_t tail_stream = nullptr;
  bool pending = false, pending_launched = false;
  const uint8_t* pending_text = nullptr;
  uint64_t pending_n = 0;
  hipStream_t pending_stream = nullptr;
  // small texts (match_small): pinned, device-visible buffers -- the result pairs + header the kernel writes,
  // and the staging copy of a small HOST text (the kernel reads it over PCIe: no copy command, no device buffer)
  uint64_t* small_out = nullptr;
  unsigned long long* small_hdr = nullptr;
  char* small_text = nullptr;
  // host-text path
  rejit_amd::DeviceBuffer text;
  char* pinned = nullptr;  // staging for rj_match_all_batch
  size_t pinned_cap = 0;
  hipStream_t own_stream = nullptr;
  // rj_scan_gather_spans: carry rows (this rank's 8 integers + every rank's), the pinned decision + rows copy, the
  // shard's pairs as global offsets, and (root) the whole list
  rejit_amd::DeviceBuffer gx_rows, gx_send, gx_out;
  int64_t* gx_host = nullptr;
  const uint64_t* gathered = nullptr;
  uint64_t gathered_count = 0;
  // rj_scan_count / rj_match_all(..., NULL): a private rj_multi of this one pattern on the counts path (multi_pattern.hip:
  // scan_count); state 0 not looked at yet, 1 the pattern has the shape, -1 it has not
  struct rj_multi* counter = nullptr;
  int counter_state = 0;
};

namespace rejit_amd {

// engine.hip: the device pipeline.  run_pipeline = MatchAll of the starts [sb, se) of a device-resident text
// (every path: small texts, windows / dense, carry scan, exact replay); results in s->result, s->result_count.
int ensure_lists(rj_scan* s, uint32_t n_regions, uint32_t region_cap, uint64_t cands_cap);
int resolve_selection(rj_scan* s, const FinalizeParams& fp, hipStream_t st);
WindowSet make_window_set(const rj_program* rp);
int run_pipeline(rj_scan* s, const uint8_t* d_text, uint64_t n, uint64_t sb, uint64_t se, uint64_t carry_cur, uint64_t carry_prev_end,
                 int have_prev, hipStream_t st);
// multi_pattern.hip: MatchAllCount of the scan's pattern over a device text -- the one-kernel count (plane_count.hip) when
// the pattern has the shape, else the pipeline; the count or rj_status
int64_t scan_count(rj_scan* s, const uint8_t* d_text, uint64_t n, hipStream_t st);
// host_api.hip
void forget_host_scans(uint64_t program_id);
void forget_combiner(uint64_t program_id);
extern thread_local std::string g_error;   // engine.hip: the calling thread's rj_last_error() text
// linear.hip: MatchAll of the starts [sb, se) in time linear in the text (carry_scan.h); results as
// after run_range (s->out, s->result_count).  RJ_TOO_LARGE when the automaton is wider than the
// carry kernels take.
bool linear_path_fits(const rj_program* rp);
bool linear_path_cheap(const rj_program* rp);  // <= 1024 positions: state in registers
int run_linear(rj_scan* s, const uint8_t* d_text, uint64_t n, uint64_t sb, uint64_t se, uint64_t carry_cur,
               uint64_t carry_prev_end, int have_prev, hipStream_t st);

// exact_replay.hip: the reference's answer, ring artefact included, for the starts the range [sb, se) owns
// (whole segments between synchronisation points: [first point >= sb, first point >= se)); the text buffer
// is taken to be the whole text.  1 = done (s->out, s->result_count), 0 = not done (pattern not at risk /
// too wide, or a stretch without synchronisation point too long to replay), < 0 = error.
bool exact_replay_fits(const rj_program* rp);
int run_exact(rj_scan* s, const uint8_t* d_text, uint64_t n, uint64_t sb, uint64_t se, hipStream_t st);

// multi_device.hip: one call over every visible device (false = not applicable, take the one-device path)
bool multi_device_match_all(const rj_program* prog, const char* text, size_t n, uint64_t** spans, int64_t* result);
bool multi_device_match_all_batch(const rj_program* prog, const char* const* texts, const size_t* sizes, size_t n_texts, uint64_t* counts,
                                  uint64_t** spans, int64_t* result);
// engine.hip: MatchAll of the starts [own_begin, own_end) of a host text on the program's device (spans relative
// to `text`), and the one-device batch
int64_t rj_match_range_host(const rj_program* prog, const char* text, size_t n, uint64_t own_begin, uint64_t own_end, uint64_t carry_cur,
                            uint64_t carry_prev_end, int have_prev, uint64_t** spans);
int64_t rj_match_all_batch_one_device(const rj_program* prog, const char* const* texts, const size_t* sizes, size_t n_texts,
                                      uint64_t* counts, uint64_t** spans);

// the calling thread runs on `device` until the guard goes (a program's tables live in ONE device's HBM)
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) (void)hipSetDevice(device);
    else prev = -1;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

}  // namespace rejit_amd
#endif
