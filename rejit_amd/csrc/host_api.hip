// rejit_amd/csrc/host_api.hip -- the entry points that take HOST text, i.e. what a rejit.h caller reaches
// (rejit_api.cc): rj_match_all / rj_match_all_batch / rj_match_first / rj_match_anywhere / rj_match_full /
// rj_replace_all.  They replace the four JIT function pointers of the reference (src/regexp.h:533-536) called
// from src/rejit.cc:150-227.  A scratch rj_scan per (thread, program), a non-blocking stream of its own, small
// texts read in place from pinned memory, large ones copied; then run_pipeline (engine.hip).

#include <cstring>
#include <emmintrin.h>

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "engine_internal.h"

using namespace rejit_amd;

#define fail ::rejit_amd::rj_fail

namespace {

// RJ_TRACE_HOST=1: the phases of the host-text batch calls on stderr (packing, upload, device pipeline, results)
bool trace_host() {
  static const bool on = getenv("RJ_TRACE_HOST") != nullptr;
  return on;
}
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// one scratch per (thread, program) for the host-text entry points
struct HostScans {
  std::vector<std::pair<uint64_t, rj_scan*>> v;  // keyed by rj_program::id, not by address
  ~HostScans() {
    for (auto& p : v) rj_scan_destroy(p.second);
  }
};
thread_local HostScans g_host_scans;

// Programs freed by ANY thread: a thread's cached scans of them are dropped the next time it looks for a scan (the
// persistent shard workers of multi_device.hip keep a scan -- with a device text buffer the size of their shard -- per
// program they served; rj_program_free only reaches the calling thread's cache directly).
std::mutex g_freed_mu;
std::vector<uint64_t> g_freed_ids;
std::atomic<uint64_t> g_freed_epoch{0};
thread_local uint64_t g_seen_epoch = 0;

void purge_freed_scans() {
  const uint64_t epoch = g_freed_epoch.load(std::memory_order_acquire);
  if (epoch == g_seen_epoch) return;
  std::vector<uint64_t> freed;
  {
    std::lock_guard<std::mutex> lk(g_freed_mu);
    freed = g_freed_ids;
  }
  g_seen_epoch = epoch;
  auto& v = g_host_scans.v;
  for (size_t i = 0; i < v.size();) {
    if (std::find(freed.begin(), freed.end(), v[i].first) != freed.end()) {
      rj_scan_destroy(v[i].second);
      v.erase(v.begin() + static_cast<long>(i));
    } else {
      i++;
    }
  }
}

int host_scan_for(const rj_program* prog, rj_scan** out) {
  purge_freed_scans();
  for (auto& p : g_host_scans.v)
    if (p.first == prog->id) {
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
  g_host_scans.v.emplace_back(prog->id, s);
  *out = s;
  return RJ_OK;
}

// the first `pairs` result pairs of the last run into host memory (the results of the small-text kernel
// already ARE in host memory)
hipError_t copy_result_pairs(rj_scan* s, uint64_t* dst, uint64_t first, uint64_t pairs, hipStream_t st) {
  if (pairs == 0) return hipSuccess;
  if (s->result == s->small_out && s->small_out != nullptr) {
    memcpy(dst, s->small_out + 2 * first, pairs * 2 * sizeof(uint64_t));
    return hipSuccess;
  }
  hipError_t e = hipMemcpyAsync(dst, s->result + 2 * first, pairs * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  return e;
}

// ---------------------------------------------------------------------------------------------------------------
// Large results on their way back to HOST memory (round 6).  A rejit.h caller's buffers are ordinary (pageable) memory
// -- sample/regexdna.cc:41-47: a std::string of 500 MB that eleven ReplaceAll calls rewrite.  Measured on the GPU box
// (tools/probes/repl_probe.py, 1 GB): the runtime uploads from pageable memory at the link's rate (56 GB/s), but a plain
// download into freshly allocated pages takes 76-110 ms per GB (one thread faults the pages in) and free() of such a
// buffer another 70-90 ms.  So (a) the download is the library's own: a ring of pinned slices per calling thread, the DMA
// of slice k + 1 running while a few worker threads copy slice k into the caller's memory side by side (the box has 256
// host threads); (b) rj_replace_all_begin / _fetch let the caller take the new text into a buffer it already owns
// (Regej::ReplaceAll: the string itself -- no allocation, no free, pages already there).
constexpr size_t kStageSlice = 16u << 20;      // bytes per pinned slice
constexpr int kStageSlots = 6;                 // slices in the ring (96 MiB of pinned memory per calling thread that moves large buffers)
constexpr size_t kStageMin = 32u << 20;        // buffers below this take the plain copy

// memcpy into a staging buffer nobody reads on this side: non-temporal stores (no read-for-ownership of the destination
// lines, nothing of the destination left in the caches); SSE2, the x86-64 baseline
inline void stream_copy(char* dst, const char* src, size_t n) {
  static const bool off = getenv("RJ_NO_STREAM_COPY") != nullptr;  // measurement override
  if (n < 4096 || off) {
    memcpy(dst, src, n);
    return;
  }
  const size_t head = (16 - (reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u;
  if (head) memcpy(dst, src, head);
  dst += head;
  src += head;
  n -= head;
  const size_t blocks = n / 64;
  for (size_t i = 0; i < blocks; i++) {
    const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src) + 0), b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src) + 1);
    const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src) + 2), d = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src) + 3);
    _mm_stream_si128(reinterpret_cast<__m128i*>(dst) + 0, a);
    _mm_stream_si128(reinterpret_cast<__m128i*>(dst) + 1, b);
    _mm_stream_si128(reinterpret_cast<__m128i*>(dst) + 2, c);
    _mm_stream_si128(reinterpret_cast<__m128i*>(dst) + 3, d);
    src += 64;
    dst += 64;
  }
  _mm_sfence();
  if (n & 63) memcpy(dst, src, n & 63);
}

// a few persistent workers: run(parts, fn) executes fn(0) .. fn(parts - 1) on them and on the caller
class CopyPool {
 public:
  static CopyPool& get() {
    static CopyPool* pool = new CopyPool;   // (never destroyed: worker threads at process exit)
    return *pool;
  }
  unsigned width() const { return n_workers_ + 1; }
  void run(unsigned parts, const std::function<void(unsigned)>& fn) {
    if (parts <= 1 || n_workers_ == 0) {
      for (unsigned i = 0; i < parts; i++) fn(i);
      return;
    }
    std::unique_lock<std::mutex> lk(mu_);   // (one parallel section at a time: concurrent callers take turns)
    busy_.wait(lk, [&] { return !running_; });
    running_ = true;
    fn_ = &fn;
    parts_ = parts;
    next_ = 1;           // part 0 is the caller's
    left_ = parts - 1;
    generation_++;
    lk.unlock();
    work_.notify_all();
    fn(0);
    lk.lock();
    // (the caller helps with what the workers have not picked up yet)
    while (next_ < parts_) {
      const unsigned part = next_++;
      lk.unlock();
      fn(part);
      lk.lock();
      --left_;
    }
    done_.wait(lk, [&] { return left_ == 0; });
    running_ = false;
    lk.unlock();
    busy_.notify_one();
  }
  void parallel_copy(char* dst, const char* src, size_t bytes) {
    const unsigned parts = static_cast<unsigned>(std::min<size_t>(width(), std::max<size_t>(bytes >> 20, 1)));
    const size_t per = ((bytes + parts - 1) / parts + 4095) & ~static_cast<size_t>(4095);
    run(parts, [&](unsigned part) {
      const size_t lo = std::min(bytes, per * part), hi = std::min(bytes, lo + per);
      if (hi > lo) memcpy(dst + lo, src + lo, hi - lo);
    });
  }

 private:
  CopyPool() {
    static const int env = getenv("RJ_COPY_THREADS") ? atoi(getenv("RJ_COPY_THREADS")) : 0;  // measurement override
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned n = env > 0 ? static_cast<unsigned>(env) : std::min(24u, std::max(3u, hw / 8));
    n_workers_ = n - 1;
    for (unsigned i = 0; i < n_workers_; i++) std::thread([this] { worker(); }).detach();
  }
  void worker() {
    uint64_t seen = 0;
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      work_.wait(lk, [&] { return generation_ != seen && next_ < parts_; });
      while (next_ < parts_) {
        const unsigned part = next_++;
        const std::function<void(unsigned)>* fn = fn_;
        lk.unlock();
        (*fn)(part);
        lk.lock();
        if (--left_ == 0) done_.notify_all();
      }
      seen = generation_;
    }
  }
  std::mutex mu_;
  std::condition_variable work_, done_, busy_;
  unsigned n_workers_ = 0;
  bool running_ = false;
  const std::function<void(unsigned)>* fn_ = nullptr;
  unsigned parts_ = 0, next_ = 0, left_ = 0;
  uint64_t generation_ = 0;
};

// the calling thread's ring of pinned slices (allocated at its first large copy, per device context the memory is
// visible to all devices: hipHostMalloc's default is portable enough for the one-process-many-devices paths here)
struct StageRing {
  char* base = nullptr;
  hipEvent_t ev[kStageSlots] = {};
  bool ok = false, failed = false;
  ~StageRing() {
    if (base) (void)hipHostFree(base);
    for (auto& e : ev)
      if (e) (void)hipEventDestroy(e);
  }
  bool ready() {
    if (ok) return true;
    if (failed) return false;   // (a failed attempt is not repeated: the plain copy serves)
    failed = true;
    if (hipHostMalloc(reinterpret_cast<void**>(&base), kStageSlice * kStageSlots, hipHostMallocPortable) != hipSuccess) {
      base = nullptr;
      (void)hipGetLastError();
      return false;
    }
    for (auto& e : ev)
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
    failed = false;
    ok = true;
    return true;
  }
};
thread_local StageRing g_stage;

// device -> host, complete on return
int staged_download(char* dst, const void* d_src, size_t n, hipStream_t st) {
  static const bool off = getenv("RJ_NO_STAGED_COPY") != nullptr;  // measurement override
  if (n < kStageMin || off || !g_stage.ready()) {
    if (n) RJ_HIP(hipMemcpyAsync(dst, d_src, n, hipMemcpyDeviceToHost, st));
    RJ_HIP(hipStreamSynchronize(st));
    return RJ_OK;
  }
  CopyPool& pool = CopyPool::get();
  const size_t slices = (n + kStageSlice - 1) / kStageSlice;
  // DMAs run kStageSlots - 1 slices ahead of the host copies that empty the ring
  auto queue = [&](size_t k) -> hipError_t {
    const size_t lo = k * kStageSlice, len = std::min(kStageSlice, n - lo);
    const int slot = static_cast<int>(k % kStageSlots);
    hipError_t e = hipMemcpyAsync(g_stage.base + static_cast<size_t>(slot) * kStageSlice, static_cast<const char*>(d_src) + lo, len, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipEventRecord(g_stage.ev[slot], st);
    return e;
  };
  size_t queued = 0;
  for (; queued < slices && queued + 1 < static_cast<size_t>(kStageSlots); queued++) RJ_HIP(queue(queued));
  for (size_t k = 0; k < slices; k++) {
    const size_t lo = k * kStageSlice, len = std::min(kStageSlice, n - lo);
    const int slot = static_cast<int>(k % kStageSlots);
    RJ_HIP(hipEventSynchronize(g_stage.ev[slot]));
    if (queued < slices) RJ_HIP(queue(queued++));   // (its slot is the one emptied LAST round: kStageSlots - 1 in flight)
    pool.parallel_copy(dst + lo, g_stage.base + static_cast<size_t>(slot) * kStageSlice, len);
  }
  return RJ_OK;
}

int stage_text(rj_scan* s, const char* text, size_t n, const uint8_t** d_text) {
  static const bool no_small = getenv("RJ_NO_SMALL") != nullptr;
  const DevProgram& D = s->prog->dev;
  if (n <= kSmallMaxText && D.n_words <= 4 && D.table_words <= kSmallMaxTableWords && !no_small &&
      small_lds_bytes(D, static_cast<uint32_t>(n)) <= small_lds_limit()) {
    // a small text stays in (pinned) host memory: match_small reads it over PCIe in one round trip, which
    // beats a copy command plus its completion wait; should the general pipeline have to take the run
    // after all, its kernels read the same memory (slower, rare)
    if (s->small_text == nullptr) RJ_HIP(hipHostMalloc(reinterpret_cast<void**>(&s->small_text), kSmallMaxText + 64));
    if (n) memcpy(s->small_text, text, n);
    *d_text = reinterpret_cast<const uint8_t*>(s->small_text);
    return RJ_OK;
  }
  RJ_HIP(s->text.reserve(((n + 64 + 4095) / 4096) * 4096));
  // (measured, round 6: the runtime's own pageable upload runs at the link's rate -- 56 GB/s for 2 GB; a pinned ring filled by
  // 24 threads was no faster.  The way BACK is where the library's own staging pays: staged_download)
  if (n) RJ_HIP(hipMemcpyAsync(s->text.p, text, n, hipMemcpyHostToDevice, s->own_stream));
  *d_text = s->text.as<uint8_t>();
  return RJ_OK;
}

}  // namespace

// rj_program_free (engine.hip): the calling thread's cached scans of a program go with it
void rejit_amd::forget_host_scans(uint64_t program_id) {
  {
    // (other threads drop theirs when they next look for a scan; the list only grows by 8 bytes per freed program
    // and is trimmed once it is long: ids are never reused, so forgetting old ones only delays a purge that no
    // cache of 16 entries can still need)
    std::lock_guard<std::mutex> lk(g_freed_mu);
    if (g_freed_ids.size() >= 4096) g_freed_ids.erase(g_freed_ids.begin(), g_freed_ids.begin() + 2048);
    g_freed_ids.push_back(program_id);
    g_freed_epoch.fetch_add(1, std::memory_order_release);
  }
  auto& v = g_host_scans.v;
  for (size_t i = 0; i < v.size();) {
    if (v[i].first == program_id) {
      rj_scan_destroy(v[i].second);
      v.erase(v.begin() + static_cast<long>(i));
    } else {
      i++;
    }
  }
}

extern "C" {

int64_t rj_replace_all_begin(const rj_program* prog, const char* text, size_t n, const char* with, size_t with_len, size_t* out_len) {
  ErrnoGuard errno_guard;
  if (!prog || (!text && n) || !out_len) return fail(RJ_BAD_ARGUMENT, "null argument");
  *out_len = 0;
  DeviceGuard on_device(prog->device);
  rj_scan* s = nullptr;
  int rc = host_scan_for(prog, &s);
  if (rc != RJ_OK) return rc;
  s->repl_len = 0;
  s->repl_valid = false;
  const uint8_t* d_text = nullptr;
  rc = stage_text(s, text, n, &d_text);
  if (rc != RJ_OK) return rc;
  rc = run_pipeline(s, d_text, n, 0, n + 1, 0, 0, 0, s->own_stream);
  if (rc != RJ_OK) return rc;
  const uint64_t m = s->result_count;
  // worst case: every match is empty and nothing is removed
  uint64_t cap = n + m * with_len + 64;
  RJ_HIP(s->repl_out.reserve(cap));
  int64_t new_len = rj_scan_replace(s, d_text, n, with, with_len, s->repl_out.p, cap, s->own_stream);
  if (new_len < 0) return new_len;
  s->repl_len = static_cast<uint64_t>(new_len);
  s->repl_valid = true;
  *out_len = static_cast<size_t>(new_len);
  return static_cast<int64_t>(m);
}

int rj_replace_all_fetch(const rj_program* prog, char* dst, size_t dst_cap) {
  ErrnoGuard errno_guard;
  if (!prog || (!dst && dst_cap)) return fail(RJ_BAD_ARGUMENT, "null argument");
  DeviceGuard on_device(prog->device);
  rj_scan* s = nullptr;
  for (auto& p : g_host_scans.v)
    if (p.first == prog->id) s = p.second;
  if (!s || !s->repl_valid) return fail(RJ_BAD_ARGUMENT, "rj_replace_all_fetch without rj_replace_all_begin on this thread");
  if (dst_cap < s->repl_len) return fail(RJ_BAD_ARGUMENT, "rj_replace_all_fetch: the new text has %llu bytes", static_cast<unsigned long long>(s->repl_len));
  s->repl_valid = false;
  return staged_download(dst, s->repl_out.p, static_cast<size_t>(s->repl_len), s->own_stream);
}

int64_t rj_replace_all(const rj_program* prog, const char* text, size_t n, const char* with, size_t with_len, char** out,
                       size_t* out_len) {
  ErrnoGuard errno_guard;
  if (!prog || (!text && n) || !out || !out_len) return fail(RJ_BAD_ARGUMENT, "null argument");
  *out = nullptr;
  *out_len = 0;
  size_t new_len = 0;
  const int64_t m = rj_replace_all_begin(prog, text, n, with, with_len, &new_len);
  if (m < 0) return m;
  char* h = static_cast<char*>(malloc(new_len + 1));
  if (!h) return fail(RJ_DEVICE_ERROR, "out of host memory");
  int rc = rj_replace_all_fetch(prog, h, new_len);
  if (rc != RJ_OK) {
    free(h);
    return rc;
  }
  h[new_len] = 0;
  *out = h;
  *out_len = new_len;
  return m;
}

}  // extern "C"

// MatchAll of the starts [own_begin, own_end) of a host text, on the program's device: H2D copy, device
// pipeline, D2H of the spans (relative to `text`).  rj_match_all is the whole-text case; multi_device.hip
// runs one of these per device.
int64_t rejit_amd::rj_match_range_host(const rj_program* prog, const char* text, size_t n, uint64_t own_begin, uint64_t own_end,
                                       uint64_t carry_cur, uint64_t carry_prev_end, int have_prev, uint64_t** spans) {
  if (spans) *spans = nullptr;
  DeviceGuard on_device(prog->device);
  rj_scan* s = nullptr;
  int rc = host_scan_for(prog, &s);
  if (rc != RJ_OK) return rc;
  const uint8_t* d_text = nullptr;
  rc = stage_text(s, text, n, &d_text);
  if (rc != RJ_OK) return rc;
  if (!spans && own_begin == 0 && own_end >= n + 1 && !have_prev && d_text != reinterpret_cast<const uint8_t*>(s->small_text)) {
    // the caller wants the NUMBER of matches (Regej::MatchAllCount, reference src/rejit.cc:203-208): the one-kernel count
    // for the patterns that have the shape (plane_count.hip), the pipeline for the others
    return scan_count(s, d_text, n, s->own_stream);
  }
  rc = run_pipeline(s, d_text, n, own_begin, own_end, carry_cur, carry_prev_end, have_prev, s->own_stream);
  if (rc != RJ_OK) return rc;
  if (spans && s->result_count) {
    uint64_t* h = static_cast<uint64_t*>(malloc(s->result_count * 2 * sizeof(uint64_t)));
    if (!h) return fail(RJ_DEVICE_ERROR, "out of host memory");
    // (on the scan's own stream: a blocking hipMemcpy goes through the NULL stream, which serialises
    // the streams of all the other threads that share the pattern)
    hipError_t e = copy_result_pairs(s, h, 0, s->result_count, s->own_stream);
    if (e != hipSuccess) {
      free(h);
      return fail(RJ_DEVICE_ERROR, "hipMemcpy failed: %s", hipGetErrorString(e));
    }
    *spans = h;
  }
  return static_cast<int64_t>(s->result_count);
}

namespace {

// The texts of a batch lie in the scan's device text buffer as laid out by `off` / `sizes` (text i at
// [off[i], off[i] + sizes[i]), separator bytes behind it up to off[i + 1]); the uploads are queued on the scan's
// stream.  One device pass, then one merge pass that hands every match to its text.
int64_t finish_packed(rj_scan* s, const uint64_t* off, const size_t* sizes, size_t n_texts, uint64_t total_bytes, uint64_t* counts,
                      uint64_t** spans) {
  const uint64_t n = total_bytes - 1;  // the last separator is the end of the buffer
  const double t0 = trace_host() ? now_ms() : 0;
  if (trace_host()) (void)hipStreamSynchronize(s->own_stream);
  const double t1 = trace_host() ? now_ms() : 0;
  int rc = run_pipeline(s, s->text.as<uint8_t>(), n, 0, n + 1, 0, 0, 0, s->own_stream);
  if (rc != RJ_OK) return rc;
  const double t2 = trace_host() ? now_ms() : 0;
  const uint64_t m = s->result_count;
  std::vector<uint64_t> pairs(2 * m);
  if (m) RJ_HIP(copy_result_pairs(s, pairs.data(), 0, m, s->own_stream));
  if (trace_host())
    fprintf(stderr, "rejit batch: %zu texts, %llu bytes: uploads still running %.3f ms, pipeline %.3f ms, %llu pairs back %.3f ms\n", n_texts,
            static_cast<unsigned long long>(total_bytes), t1 - t0, t2 - t1, static_cast<unsigned long long>(m), now_ms() - t2);
  // the matches are ordered by begin: one merge pass assigns them to their texts
  for (size_t i = 0; i < n_texts; i++) counts[i] = 0;
  size_t t = 0;
  uint64_t kept = 0;
  for (uint64_t k = 0; k < m; k++) {
    const uint64_t b = pairs[2 * k], e = pairs[2 * k + 1];
    while (t + 1 < n_texts && b >= off[t + 1]) t++;
    // an empty match between the separators of a gap -- before the first text (offsets[0] > 0), or behind a text's
    // own end -- belongs to no text
    if (b < off[t] || b > off[t] + sizes[t]) continue;
    if (e > off[t] + sizes[t]) return fail(RJ_DEVICE_ERROR, "internal: a match crosses a text boundary in a batch");
    counts[t]++;
    pairs[2 * kept] = b - off[t];
    pairs[2 * kept + 1] = e - off[t];
    kept++;
  }
  if (spans && kept) {
    uint64_t* h = static_cast<uint64_t*>(malloc(kept * 2 * sizeof(uint64_t)));
    if (!h) return fail(RJ_DEVICE_ERROR, "out of host memory");
    memcpy(h, pairs.data(), kept * 2 * sizeof(uint64_t));
    *spans = h;
  }
  return static_cast<int64_t>(kept);
}

// A batch that its owner laid out already (rj_match_all_packed, the call combiner): uploaded as it is.
// `s`: the scratch to use (null: the calling thread's own for this pattern).
int64_t run_packed(const rj_program* prog, const char* packed, const uint64_t* off, const size_t* sizes, size_t n_texts, uint64_t total_bytes,
                   uint64_t* counts, uint64_t** spans, rj_scan* s = nullptr) {
  if (spans) *spans = nullptr;
  DeviceGuard on_device(prog->device);
  if (!s) {
    int rc = host_scan_for(prog, &s);
    if (rc != RJ_OK) return rc;
  }
  struct DrainStream {
    hipStream_t st;
    ~DrainStream() { (void)hipStreamSynchronize(st); }
  } drain{s->own_stream};  // (the caller may refill `packed` as soon as this returns)
  RJ_HIP(s->text.reserve(((total_bytes + 64 + 4095) / 4096) * 4096));
  constexpr uint64_t kSlice = 64ull << 20;
  for (uint64_t lo = 0; lo < total_bytes; lo += kSlice) {
    const uint64_t len = std::min(kSlice, total_bytes - lo);
    RJ_HIP(hipMemcpyAsync(static_cast<char*>(s->text.p) + lo, packed + lo, len, hipMemcpyHostToDevice, s->own_stream));
  }
  return finish_packed(s, off, sizes, n_texts, total_bytes, counts, spans);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// Concurrent callers of one pattern are COMBINED.  The reference's jrep shares one compiled Regej between its
// worker threads and calls MatchAll once per file (sample/jrep.cc:288, workers :461-493); on the CPU that is N
// independent passes, here every call is a device round trip of >= 20 us whatever the size of the file, and
// concurrent round trips serialise in the runtime (jrep -j 8 was SLOWER than -j 0 in round 2).  So a caller with
// a small text reserves room in the pattern's open batch buffer (pinned host memory) and copies its text there
// itself -- the copies, and the page faults of an mmap'ed file, run in parallel on the callers' threads --;
// whoever finds no batch on the device becomes the leader, closes the buffer (new arrivals fill the second one),
// runs it as ONE packed batch -- one upload, one device pass -- and hands the results out; the others sleep on
// their own condition variable until their result is there or it is their turn to lead ("group commit").  A
// single caller leads a batch of one: the ordinary path plus an uncontended mutex.
namespace {

constexpr size_t kCombineBuffer = 8u << 20;    // bytes per batch buffer
constexpr size_t kCombineMaxText = 1u << 20;   // (a text this large gains nothing from sharing a pass)

struct PendingCall {
  const char* text;
  size_t n;
  bool want_spans;
  uint64_t off = 0;
  bool retry_alone = false;  // the combined pass failed: this call runs again on its own
  uint64_t* spans = nullptr;
  int64_t result = 0;
  std::string error;
  bool done = false;
  std::condition_variable cv;
};

struct Combiner {
  std::mutex mu;
  char* buf[2] = {nullptr, nullptr};   // pinned
  int open = 0;                        // the buffer arrivals fill
  uint64_t used = 0;                   // bytes reserved in it
  int copying[2] = {0, 0};             // callers still copying into a buffer
  std::vector<PendingCall*> calls;     // the open batch, in buffer order
  bool leader_active = false;
  rj_scan* scan = nullptr;             // the device scratch of the batches: ONE for all leaders (they lead one at a time;
                                       // a scratch per leading thread cost ~10 ms of allocations per thread: jrep -j 128)
  std::atomic<int> inside{0};          // calls of this pattern in flight (combined or not)
  std::condition_variable copied;      // a buffer's last copy has finished
  std::condition_variable room;        // the open buffer was swapped: there is room again
  ~Combiner() {
    if (buf[0]) (void)hipHostFree(buf[0]);   // (buf[1] is its second half)
    if (scan) rj_scan_destroy(scan);
  }
};

std::mutex g_combiners_mu;
std::unordered_map<uint64_t, std::shared_ptr<Combiner>> g_combiners;  // by rj_program::id

std::shared_ptr<Combiner> combiner_for(const rj_program* prog) {
  std::lock_guard<std::mutex> lk(g_combiners_mu);
  auto& c = g_combiners[prog->id];
  if (!c) c = std::make_shared<Combiner>();
  return c;
}

// the closed batch `batch` lies in `packed` (text, separator, text, separator, ...)
void run_combined(const rj_program* prog, rj_scan* scan, const char* packed, uint64_t used, std::vector<PendingCall*>& batch) {
  std::vector<uint64_t> off(batch.size() + 1);
  std::vector<size_t> sizes(batch.size());
  std::vector<uint64_t> counts(batch.size(), 0);
  for (size_t i = 0; i < batch.size(); i++) {
    off[i] = batch[i]->off;
    sizes[i] = batch[i]->n;
  }
  off[batch.size()] = used;
  uint64_t* all = nullptr;
  const int64_t total = run_packed(prog, packed, off.data(), sizes.data(), batch.size(), used, counts.data(), &all, scan);
  if (total < 0) {
    // The pass over the concatenation failed (one member's text drove a walk into its limit, a device list could not
    // grow, ...): the members are independent callers -- each of them repeats ITS call alone on the direct path (flagged
    // here, done by the caller's own thread in combined_match_all), so that one text's failure is that text's alone.
    for (PendingCall* p : batch) p->retry_alone = true;
    return;
  }
  uint64_t at = 0;
  for (size_t i = 0; i < batch.size(); i++) {
    PendingCall* p = batch[i];
    p->result = static_cast<int64_t>(counts[i]);
    if (p->want_spans && counts[i]) {
      p->spans = static_cast<uint64_t*>(malloc(counts[i] * 2 * sizeof(uint64_t)));
      if (!p->spans) {
        p->result = RJ_DEVICE_ERROR;
        p->error = "out of host memory";
      } else {
        memcpy(p->spans, all + 2 * at, counts[i] * 2 * sizeof(uint64_t));
      }
    }
    at += counts[i];
  }
  free(all);
}

// 0: take the direct path (this call is alone, or there is no pinned memory); 1: done, *result holds the answer
int combined_match_all(const rj_program* prog, const char* text, size_t n, uint64_t** spans, int64_t* result) {
  std::shared_ptr<Combiner> c = combiner_for(prog);
  struct Inside {
    std::atomic<int>& v;
    int seen;
    explicit Inside(std::atomic<int>& a) : v(a), seen(a.fetch_add(1) + 1) {}
    ~Inside() { v.fetch_sub(1); }
  } inside(c->inside);
  // a call that finds no other call of the pattern in flight takes the direct path -- a single-threaded caller
  // never comes further than this, and the batch buffers (pinned memory) are only ever allocated for patterns
  // that ARE called concurrently
  if (inside.seen < 2) {
    *result = rj_match_range_host(prog, text, n, 0, n + 1, 0, 0, 0, spans);
    return 1;
  }
  PendingCall me{text, n, spans != nullptr};
  std::unique_lock<std::mutex> lk(c->mu);
  if (!c->buf[0]) {
    DeviceGuard on_device(prog->device);
    if (hipHostMalloc(reinterpret_cast<void**>(&c->buf[0]), 2 * kCombineBuffer) != hipSuccess) {
      c->buf[0] = nullptr;
      return 0;
    }
    c->buf[1] = c->buf[0] + kCombineBuffer;
    if (rj_scan_create(prog, &c->scan) != RJ_OK || hipStreamCreateWithFlags(&c->scan->own_stream, hipStreamNonBlocking) != hipSuccess) {
      if (c->scan) rj_scan_destroy(c->scan);
      c->scan = nullptr;
      (void)hipHostFree(c->buf[0]);
      c->buf[0] = c->buf[1] = nullptr;
      return 0;
    }
  }
  // room in the open buffer (when it is full its batch is led by one of the callers in it: wait for the swap)
  while (c->used + n + 1 > kCombineBuffer) c->room.wait(lk);
  const int b = c->open;
  me.off = c->used;
  c->used += n + 1;
  c->calls.push_back(&me);
  c->copying[b]++;
  char* dst = c->buf[b] + me.off;
  lk.unlock();
  if (n) memcpy(dst, text, n);
  dst[n] = static_cast<char>(prog->batch_separator);
  lk.lock();
  if (--c->copying[b] == 0) c->copied.notify_all();
  while (!me.done) {
    if (c->leader_active || c->open != b) {  // (c->open != b: my batch is closed already, its leader will wake me)
      me.cv.wait(lk);
      continue;
    }
    // lead the open batch (this call is in it)
    c->leader_active = true;
    std::vector<PendingCall*> batch;
    batch.swap(c->calls);
    const uint64_t used = c->used;
    c->open ^= 1;
    c->used = 0;
    c->room.notify_all();
    while (c->copying[b] > 0) c->copied.wait(lk);
    lk.unlock();
    run_combined(prog, c->scan, c->buf[b], used, batch);
    lk.lock();
    c->leader_active = false;
    for (PendingCall* p : batch) {
      p->done = true;
      if (p != &me) p->cv.notify_one();
    }
    // the batch that filled meanwhile: its first caller leads it
    if (!c->calls.empty()) c->calls.front()->cv.notify_one();
  }
  lk.unlock();
  if (me.retry_alone) {
    *result = rj_match_range_host(prog, text, n, 0, n + 1, 0, 0, 0, spans);
    return 1;
  }
  if (me.result < 0) g_error = me.error;
  if (spans) *spans = me.spans;
  *result = me.result;
  return 1;
}

}  // namespace

// rj_program_free (engine.hip)
void rejit_amd::forget_combiner(uint64_t program_id) {
  std::lock_guard<std::mutex> lk(g_combiners_mu);
  g_combiners.erase(program_id);
}

extern "C" {

int64_t rj_match_all(const rj_program* prog, const char* text, size_t n, uint64_t** spans) {
  ErrnoGuard errno_guard;
  if (spans) *spans = nullptr;
  if (!prog || (!text && n)) return fail(RJ_BAD_ARGUMENT, "null argument");
  int64_t result = 0;
  if (multi_device_match_all(prog, text, n, spans, &result)) return result;  // large text, several GPUs: one range per device
  static const bool no_combine = getenv("RJ_NO_COMBINE") != nullptr;  // measurement override
  if (n <= kCombineMaxText && prog->batch_separator >= 0 && !no_combine && combined_match_all(prog, text, n, spans, &result) == 1) return result;
  return rj_match_range_host(prog, text, n, 0, n + 1, 0, 0, 0, spans);
}

int64_t rj_match_all_batch(const rj_program* prog, const char* const* texts, const size_t* sizes, size_t n_texts,
                           uint64_t* counts, uint64_t** spans) {
  ErrnoGuard errno_guard;
  if (spans) *spans = nullptr;
  if (!prog || (n_texts && (!texts || !sizes || !counts))) return fail(RJ_BAD_ARGUMENT, "null argument");
  for (size_t i = 0; i < n_texts; i++)
    if (!texts[i] && sizes[i]) return fail(RJ_BAD_ARGUMENT, "null text in batch");
  if (n_texts == 0) return 0;
  int64_t result = 0;
  if (multi_device_match_all_batch(prog, texts, sizes, n_texts, counts, spans, &result)) return result;  // files spread over the GPUs
  return rj_match_all_batch_one_device(prog, texts, sizes, n_texts, counts, spans);
}

}  // extern "C"

int64_t rejit_amd::rj_match_all_batch_one_device(const rj_program* prog, const char* const* texts, const size_t* sizes, size_t n_texts,
                                                 uint64_t* counts, uint64_t** spans) {
  if (spans) *spans = nullptr;
  DeviceGuard on_device(prog->device);
  if (prog->batch_separator < 0 || n_texts == 1) {
    // no byte can safely end a text inside a concatenation (or nothing to batch): text by text
    std::vector<uint64_t> all;
    uint64_t total = 0;
    for (size_t i = 0; i < n_texts; i++) {
      uint64_t* one = nullptr;
      int64_t c = rj_match_all(prog, texts[i], sizes[i], spans ? &one : nullptr);
      if (c < 0) return c;
      counts[i] = static_cast<uint64_t>(c);
      if (spans && c) all.insert(all.end(), one, one + 2 * c);
      rj_free_spans(one);
      total += static_cast<uint64_t>(c);
    }
    if (spans && total) {
      uint64_t* h = static_cast<uint64_t*>(malloc(all.size() * sizeof(uint64_t)));
      if (!h) return fail(RJ_DEVICE_ERROR, "out of host memory");
      memcpy(h, all.data(), all.size() * sizeof(uint64_t));
      *spans = h;
    }
    return static_cast<int64_t>(total);
  }
  rj_scan* s = nullptr;
  int rc = host_scan_for(prog, &s);
  if (rc != RJ_OK) return rc;
  // text i occupies [off[i], off[i] + sizes[i]); position off[i] + sizes[i] holds the separator and
  // is text i's end position (an empty match there belongs to text i)
  std::vector<uint64_t> off(n_texts + 1);
  uint64_t total_bytes = 0;
  for (size_t i = 0; i < n_texts; i++) {
    off[i] = total_bytes;
    if (sizes[i] >= (1ull << 62) || total_bytes + sizes[i] + 1 < total_bytes) return fail(RJ_BAD_ARGUMENT, "batch: sizes overflow");
    total_bytes += sizes[i] + 1;
  }
  off[n_texts] = total_bytes;
  // uploads are queued slice by slice from s->pinned: whatever way this function is left, no copy may still
  // be in flight (the next call may free or refill the staging buffer)
  struct DrainStream {
    hipStream_t st;
    ~DrainStream() { (void)hipStreamSynchronize(st); }
  } drain{s->own_stream};
  const uint64_t n = total_bytes - 1;  // the last separator is the end of the buffer
  if (total_bytes > s->pinned_cap) {
    if (s->pinned) (void)hipHostFree(s->pinned);
    s->pinned = nullptr;
    s->pinned_cap = 0;
    const size_t want = ((total_bytes + (1u << 20)) / 4096 + 1) * 4096;
    RJ_HIP(hipHostMalloc(reinterpret_cast<void**>(&s->pinned), want));
    s->pinned_cap = want;
  }
  const char sep = static_cast<char>(prog->batch_separator);
  RJ_HIP(s->text.reserve(((total_bytes + 64 + 4095) / 4096) * 4096));
  const double t_pack = trace_host() ? now_ms() : 0;
  {
    // Packing is a host memcpy of the whole batch (one core moves ~10 GB/s, PCIe takes 50+): it is
    // spread over a few threads and done slice by slice, each slice's DMA starting as soon as it is
    // packed, so packing slice k+1 overlaps the upload of slice k.
    CopyPool& pool = CopyPool::get();
    const unsigned n_thr = total_bytes > (8u << 20) ? pool.width() : 1u;
    // bytes [a, b) of the packed buffer: the pieces of the texts that lie there, and the separators.  The work is dealt out
    // by BYTES, not by texts: a tree's sizes have a heavy tail (bench.py's: log-normal, sigma 1.8), and a 20 MiB file copied
    // by one thread held a whole 32 MiB slice up (round 6, RJ_TRACE_HOST: 7.5 ms to pack 256 MiB whatever the thread count).
    auto pack_bytes = [&](uint64_t a, uint64_t b) {
      size_t i = static_cast<size_t>(std::upper_bound(off.begin(), off.begin() + static_cast<long>(n_texts), a) - off.begin());
      i = i > 0 ? i - 1 : 0;   // the text that holds byte a (or its separator)
      for (; i < n_texts && off[i] < b; i++) {
        const uint64_t t_lo = off[i], t_hi = off[i] + sizes[i];   // text bytes [t_lo, t_hi), separator at t_hi
        const uint64_t lo = std::max(a, t_lo), hi = std::min(b, t_hi);
        if (hi > lo) stream_copy(s->pinned + lo, texts[i] + (lo - t_lo), hi - lo);
        if (t_hi >= a && t_hi < b) s->pinned[t_hi] = sep;
      }
    };
    static const uint64_t kSlice = (getenv("RJ_BATCH_SLICE_MB") ? static_cast<uint64_t>(atoi(getenv("RJ_BATCH_SLICE_MB"))) : 16ull) << 20;  // measurement override (jrep_10gb: 4 MiB 37.4, 8 38.7, 16 38.9, 32 35.5, 64 34.7 GB/s end to end)
    size_t first = 0;
    double t_copy = 0, t_queue = 0;
    unsigned slice_no = 0;
    bool used_second = false;
    while (first < n_texts) {
      // texts [first, last) make up about one slice
      size_t last = first;
      while (last < n_texts && off[last] - off[first] < kSlice) last++;
      const uint64_t lo = off[first], hi = off[last];
      const double tp0 = trace_host() ? now_ms() : 0;
      if (n_thr == 1 || hi - lo < (4u << 20)) {
        pack_bytes(lo, hi);
      } else {
        const uint64_t per = ((hi - lo + n_thr - 1) / n_thr + 63) & ~63ull;
        pool.run(n_thr, [&](unsigned t) {
          const uint64_t a = std::min(hi, lo + per * t), b = std::min(hi, a + per);
          if (b > a) pack_bytes(a, b);
        });
      }
      const double tp1 = trace_host() ? now_ms() : 0;
      // the slices' DMAs alternate between two streams (measured, jrep_10gb: the uploads still running when the packing ends
      // 2.6 -> 1.05 ms per 256 MiB batch, 42.0 -> 43.0 GB/s end to end; RJ_BATCH_ONE_STREAM: measurement override)
      static const bool two_streams = getenv("RJ_BATCH_ONE_STREAM") == nullptr;
      hipStream_t up = s->own_stream;
      if (two_streams && (slice_no++ & 1u)) {
        if (!s->tail_stream) RJ_HIP(hipStreamCreateWithFlags(&s->tail_stream, hipStreamNonBlocking));
        up = s->tail_stream;
        used_second = true;
      }
      RJ_HIP(hipMemcpyAsync(static_cast<char*>(s->text.p) + lo, s->pinned + lo, hi - lo, hipMemcpyHostToDevice, up));
      if (trace_host()) {
        t_copy += tp1 - tp0;
        t_queue += now_ms() - tp1;
      }
      first = last;
    }
    if (used_second) {   // the pipeline on own_stream waits for the other stream's copies
      RJ_HIP(hipEventRecord(s->ev[3], s->tail_stream));
      RJ_HIP(hipStreamWaitEvent(s->own_stream, s->ev[3], 0));
    }
    if (trace_host())
      fprintf(stderr, "rejit batch: packed + queued %llu bytes in %.3f ms (host copies %.3f ms on %u threads, hipMemcpyAsync calls %.3f ms)\n",
              static_cast<unsigned long long>(total_bytes), now_ms() - t_pack, t_copy, n_thr, t_queue);
  }
  return finish_packed(s, off.data(), sizes, n_texts, total_bytes, counts, spans);
}

extern "C" {

int rj_batch_separator(const rj_program* prog) { return prog ? prog->batch_separator : -1; }

int rj_host_stats(const rj_program* prog, void* stats, size_t struct_size) {
  if (!prog || !stats) return fail(RJ_BAD_ARGUMENT, "null argument");
  for (auto& p : g_host_scans.v)
    if (p.first == prog->id) return rj_scan_stats_sized(p.second, stats, struct_size);
  return fail(RJ_BAD_ARGUMENT, "rj_host_stats: this thread has made no host-text call of the pattern (or the call was combined with other threads')");
}

void* rj_host_alloc(size_t bytes) {
  ErrnoGuard errno_guard;
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1) != hipSuccess) return nullptr;
  return p;
}

void rj_host_free(void* p) {
  ErrnoGuard errno_guard;
  if (p) (void)hipHostFree(p);
}

int64_t rj_match_all_packed(const rj_program* prog, const char* packed, const uint64_t* offsets, const size_t* sizes, size_t n_texts,
                            uint64_t total_bytes, uint64_t* counts, uint64_t** spans) {
  ErrnoGuard errno_guard;
  if (spans) *spans = nullptr;
  if (!prog || (n_texts && (!packed || !offsets || !sizes || !counts))) return fail(RJ_BAD_ARGUMENT, "null argument");
  if (n_texts == 0) return 0;
  for (size_t i = 0; i < n_texts; i++) {
    const uint64_t end = offsets[i] + sizes[i];
    if (end < offsets[i] || end >= total_bytes || (i + 1 < n_texts && end >= offsets[i + 1]))
      return fail(RJ_BAD_ARGUMENT, "packed batch: text %zu overlaps the next one or leaves no room for its separator", i);
  }
  if (prog->batch_separator < 0 || n_texts == 1) {
    // (no byte can end a text inside a concatenation, or nothing to batch: text by text)
    std::vector<const char*> texts(n_texts);
    for (size_t i = 0; i < n_texts; i++) texts[i] = packed + offsets[i];
    return rj_match_all_batch_one_device(prog, texts.data(), sizes, n_texts, counts, spans);
  }
  int64_t result = 0;
  {
    // several GPUs: the files are spread over them (every device packs its share itself)
    std::vector<const char*> texts(n_texts);
    for (size_t i = 0; i < n_texts; i++) texts[i] = packed + offsets[i];
    if (multi_device_match_all_batch(prog, texts.data(), sizes, n_texts, counts, spans, &result)) return result;
  }
  return run_packed(prog, packed, offsets, sizes, n_texts, total_bytes, counts, spans);
}

void rj_free_spans(uint64_t* spans) { free(spans); }

// kMatchFirst / kMatchAnywhere with early exit (the reference's generated code returns at the
// first match, codegen-x64.cc:401-446).  kMatchFirst is the first element of kMatchAll (left-most
// longest; reference behaviour Q6), and the greedy selection takes the left-most candidate whatever
// follows it, so it is enough to look at growing prefixes of the START positions: [0, 256 KiB),
// then 8x more each time.  When the longest match is bounded only the bytes a block's candidates
// can reach are uploaded first (a walk from s < hi ends before hi + max_len), so a hit near the
// start of a large host buffer costs microseconds instead of the whole PCIe copy.
static int first_match(const rj_program* prog, const char* text, size_t n, uint64_t* begin, uint64_t* end) {
  if (!prog || (!text && n)) return fail(RJ_BAD_ARGUMENT, "null argument");
  rj_scan* s = nullptr;
  int rc = host_scan_for(prog, &s);
  if (rc != RJ_OK) return rc;
  RJ_HIP(s->text.reserve(((n + 64 + 4095) / 4096) * 4096));
  const uint8_t* d_text = s->text.as<uint8_t>();
  const uint64_t max_len = prog->host->max_len;
  const bool bounded = max_len < (1ull << 20);
  uint64_t uploaded = 0;
  auto upload_to = [&](uint64_t upto) -> hipError_t {
    if (upto > n) upto = n;
    if (upto <= uploaded) return hipSuccess;
    hipError_t e = hipMemcpyAsync(static_cast<char*>(s->text.p) + uploaded, text + uploaded, upto - uploaded,
                                  hipMemcpyHostToDevice, s->own_stream);
    uploaded = upto;
    return e;
  };
  uint64_t lo = 0, block = 256u << 10;
  // Patterns at risk of the reference's ring artefact: a RANGE of starts owns whole segments between
  // synchronisation points of the reference's loop and needs the whole text to find them (rj_scan_run's
  // contract); on a truncated buffer the starts of a sync-free stretch that crosses a block boundary would be
  // owned by no round (`x{0,2}yz` over 300 000 x + "yz" reported no match).  One round over the whole text:
  // MatchFirst is the first element of the exact MatchAll.
  if (prog->host->q8_risk) block = n + 1;
  for (;;) {
    uint64_t hi = lo + block;
    const bool last = hi >= n;
    if (last) hi = n;
    // text the automaton may see in this round: everything for the last block or an unbounded
    // pattern, else up to the furthest byte a candidate of the block can reach (+1 for the
    // end-of-line context)
    const uint64_t visible = (last || !bounded) ? n : std::min<uint64_t>(n, hi + max_len + 1);
    RJ_HIP(upload_to(visible));
    rc = run_pipeline(s, d_text, visible, lo, (last && visible == n) ? n + 1 : hi, 0, 0, 0, s->own_stream);
    if (rc != RJ_OK) return rc;
    if (s->result_count > 0) {
      uint64_t pair[2];
      RJ_HIP(copy_result_pairs(s, pair, 0, 1, s->own_stream));
      if (begin) *begin = pair[0];
      if (end) *end = pair[1];
      return 1;
    }
    if (last) return 0;
    lo = hi;
    block *= 8;
  }
}

int rj_match_first(const rj_program* prog, const char* text, size_t n, uint64_t* begin, uint64_t* end) {
  ErrnoGuard errno_guard;
  return first_match(prog, text, n, begin, end);
}

int rj_match_anywhere(const rj_program* prog, const char* text, size_t n) {
  ErrnoGuard errno_guard;
  return first_match(prog, text, n, nullptr, nullptr);
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
