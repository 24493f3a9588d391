// samples/regexdna_gpu.cc -- regex-dna with the text resident in HBM for the whole program: the native counterpart of the
// reference's sample/regexdna.cc:25-94 (and, with its two independent halves in flight together, of
// sample/regexdna-multithread.cc:56-116), written over the C ABI (include/rejit_hip.h) only.
//
//   regexdna_gpu [--serial] [--timing] < input.fasta
//
//   1. stdin is read straight into pinned host memory (rj_host_alloc) while a second thread brings the device up and
//      compiles the patterns (the ~250 ms of a ROCm process's start-up overlap the read);
//   2. ONE upload; `>.*\n|\n` -> "" on the device (rj_scan_run + rj_scan_replace): the stripped sequence never leaves HBM;
//   3. the nine counts in ONE pass over the sequence (rj_multi_start on its own stream) WHILE
//   4. the eleven IUB replacements run one after the other on a second stream, ping-ponging between two device buffers
//      (each is rj_scan_run + rj_scan_replace: MatchAll + Replace, src/rejit.cc:97-112,220-226) -- the two halves only
//      read the stripped sequence, exactly what lets the reference's multithread sample run them on different threads;
//   5. twelve lines of output.  Nothing but the input crosses PCIe.
// The reference's own program also runs unchanged on this library (oracle/_ref/regexdna_hip), but pays a PCIe round trip
// of the whole text per ReplaceAll / MatchAllCount call.
#include <hip/hip_runtime_api.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "rejit_hip.h"

namespace {

const char* const kPatterns[] = {"agggtaaa|tttaccct",         "[cgt]gggtaaa|tttaccc[acg]", "a[act]ggtaaa|tttacc[agt]t",
                                 "ag[act]gtaaa|tttac[agt]ct", "agg[act]taaa|ttta[agt]cct", "aggg[acg]aaa|ttt[cgt]ccct",
                                 "agggt[cgt]aa|tt[acg]accct", "agggta[cgt]a|t[acg]taccct", "agggtaa[cgt]|[acg]ttaccct"};
const char* const kIub[][2] = {{"B", "(c|g|t)"}, {"D", "(a|g|t)"}, {"H", "(a|c|t)"}, {"K", "(g|t)"},   {"M", "(a|c)"}, {"N", "(a|c|g|t)"},
                               {"R", "(a|g)"},   {"S", "(c|g)"},   {"V", "(a|c|g)"}, {"W", "(a|t)"},   {"Y", "(c|t)"}};
constexpr int kN = 9, kNIub = 11;

[[noreturn]] void die(const char* what) {
  fprintf(stderr, "regexdna_gpu: %s: %s\n", what, rj_last_error());
  _exit(2);
}
#define HIP_OK(call)                                                              \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "regexdna_gpu: %s: %s\n", #call, hipGetErrorString(e_));    \
      _exit(2);                                                                   \
    }                                                                             \
  } while (0)

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Programs {
  rj_program* strip = nullptr;
  rj_program* count[kN] = {};
  rj_program* iub[kNIub] = {};
};

}  // namespace

int main(int argc, char** argv) {
  bool serial = false, timing = false;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--serial")) serial = true;
    else if (!strcmp(argv[i], "--timing")) timing = true;
    else {
      fprintf(stderr, "usage: regexdna_gpu [--serial] [--timing] < input.fasta\n");
      return 2;
    }
  }
  const double t_start = now_ms();
  // ---- the device comes up and the 21 patterns are compiled while stdin is read
  Programs P;
  std::thread device([&] {
    if (rj_compile(">.*\n|\n", &P.strip) != RJ_OK) die("compile");
    for (int i = 0; i < kN; i++)
      if (rj_compile(kPatterns[i], &P.count[i]) != RJ_OK) die("compile");
    for (int i = 0; i < kNIub; i++)
      if (rj_compile(kIub[i][0], &P.iub[i]) != RJ_OK) die("compile");
  });
  // ---- stdin, in growing ordinary chunks (the pinned allocation needs the runtime, which is still starting)
  std::vector<char> input;
  {
    size_t cap = 64u << 20, len = 0;
    input.resize(cap);
    for (;;) {
      if (len == cap) input.resize(cap *= 2);
      const ssize_t got = read(0, input.data() + len, cap - len);
      if (got <= 0) break;
      len += static_cast<size_t>(got);
    }
    input.resize(len);
  }
  const size_t raw_size = input.size();
  const double t_read = now_ms();
  device.join();
  const double t_dev_up = now_ms();

  hipStream_t s_main, s_side;
  HIP_OK(hipStreamCreateWithFlags(&s_main, hipStreamNonBlocking));
  HIP_OK(hipStreamCreateWithFlags(&s_side, hipStreamNonBlocking));
  // device buffers: the raw text, and two for the sequence / the replaced texts (the IUB replacements grow the 500 MB
  // sequence to 668 MB; a code letter becomes at most 9 bytes, the letters are rare: 1.5 x + slack, checked per step)
  char *d_raw = nullptr, *d_a = nullptr, *d_b = nullptr;
  const size_t cap = raw_size + raw_size / 2 + (1u << 20);
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&d_raw), raw_size + 64));
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&d_a), cap));
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&d_b), cap));
  HIP_OK(hipHostRegister(input.data(), raw_size ? raw_size : 1, hipHostRegisterDefault));   // (pinned in place: the upload runs at the PCIe rate)
  HIP_OK(hipMemcpyAsync(d_raw, input.data(), raw_size, hipMemcpyHostToDevice, s_main));
  HIP_OK(hipStreamSynchronize(s_main));
  const double t_up = now_ms();

  // ---- strip: ReplaceAll(">.*\n|\n", text, "")  (sample/regexdna.cc:49)
  rj_scan* strip = nullptr;
  if (rj_scan_create(P.strip, &strip) != RJ_OK) die("scan");
  if (rj_scan_run(strip, d_raw, raw_size, 0, raw_size + 1, 0, 0, 0, s_main) < 0) die("strip");
  const int64_t seq_size = rj_scan_replace(strip, d_raw, raw_size, "", 0, d_a, cap, s_main);
  if (seq_size < 0) die("strip replace");
  const double t_strip = now_ms();

  // ---- the nine counts: one pass over the sequence, queued on its own stream ...
  rj_multi* multi = nullptr;
  if (rj_multi_create(P.count, kN, &multi) != RJ_OK) die("multi");
  (void)rj_multi_set_counts_only(multi, 1);   // regexdna asks MatchAllCount: scan + classification + counts in one kernel (plane_count.hip)
  uint64_t counts[kN] = {};
  if (serial) {
    if (rj_multi_run(multi, d_a, static_cast<uint64_t>(seq_size), counts, s_main) < 0) die("counts");
  } else {
    if (rj_multi_start(multi, d_a, static_cast<uint64_t>(seq_size), 0, static_cast<uint64_t>(seq_size) + 1, s_side) != RJ_OK) die("counts");
  }
  const double t_counts_queued = now_ms();
  // ---- ... while the eleven replacements run on the main stream; d_a holds the sequence (read by the counts): the first
  // replacement writes d_b, the second reads d_b and writes d_raw's buffer ... so that d_a stays intact until the counts
  // are collected
  char* bufs[3] = {d_a, d_b, nullptr};
  // a third buffer for the ping-pong (the raw text is no longer needed; it is large enough only if cap fits: allocate)
  HIP_OK(hipFree(d_raw));
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&bufs[2]), cap));
  const char* src = d_a;
  uint64_t n = static_cast<uint64_t>(seq_size);
  int next = 1;
  for (int i = 0; i < kNIub; i++) {
    rj_scan* sc = nullptr;
    if (rj_scan_create(P.iub[i], &sc) != RJ_OK) die("scan");
    const int64_t m = rj_scan_run(sc, src, n, 0, n + 1, 0, 0, 0, s_main);
    if (m < 0) die("replace scan");
    const size_t repl_len = strlen(kIub[i][1]);
    if (n + static_cast<uint64_t>(m) * repl_len + 64 > cap) {
      fprintf(stderr, "regexdna_gpu: the replaced text outgrew its buffer (%llu matches of %s)\n", static_cast<unsigned long long>(m), kIub[i][0]);
      _exit(2);
    }
    char* dst = bufs[next];
    const int64_t new_len = rj_scan_replace(sc, src, n, kIub[i][1], repl_len, dst, cap, s_main);
    if (new_len < 0) die("replace");
    rj_scan_destroy(sc);
    src = dst;
    n = static_cast<uint64_t>(new_len);
    next = next == 1 ? 2 : 1;   // (never d_a: the counts read it)
  }
  const double t_repl = now_ms();
  if (!serial && rj_multi_finish(multi, counts) < 0) die("counts");
  const double t_done = now_ms();

  for (int i = 0; i < kN; i++) printf("%s %llu\n", kPatterns[i], static_cast<unsigned long long>(counts[i]));
  printf("\n%zu\n%lld\n%llu\n", raw_size, static_cast<long long>(seq_size), static_cast<unsigned long long>(n));
  fflush(stdout);
  if (timing)
    fprintf(stderr,
            "read %.0f ms (device up after %.0f ms) | upload %.1f ms | strip %.1f ms | counts queued %.1f ms | 11 replaces %.1f ms | counts collected +%.1f ms | "
            "device pipeline %.1f ms | total %.0f ms\n",
            t_read - t_start, t_dev_up - t_start, t_up - t_dev_up, t_strip - t_up, t_counts_queued - t_strip, t_repl - t_counts_queued, t_done - t_repl,
            t_done - t_up, t_done - t_start);
  // (no teardown: the process ends here and takes the runtime with it)
  _exit(0);
}
