// samples/regexdna_rccl.cc -- the nine regexdna counts over EVERY GPU of the node from one process: a thread, a
// shard of the sequence and an RCCL rank per device, the exchange step behind one C call
// (rj_multi_device_counts, include/rejit_hip.h).  The reference's counterpart keeps one CPU thread per pattern busy
// (sample/regexdna-multithread.cc:65-78); across devices the unit of work is the byte range instead -- every device
// answers all nine patterns over its part in one pass, and the left-most-longest selection is carried over the cuts
// by an all-gather of 8 integers per pattern.
//
//   regexdna_rccl [--devices N] < fasta.txt     (make -C samples rccl; with one GPU visible: a one-rank communicator)
//
// Output: the nine "pattern count" lines of the Benchmarks-Game program (sample/regexdna.cc:56-70) and the
// lengths of the input and of the stripped sequence.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "rejit_hip.h"

namespace {

const char* const kPatterns[] = {"agggtaaa|tttaccct",         "[cgt]gggtaaa|tttaccc[acg]", "a[act]ggtaaa|tttacc[agt]t",
                                 "ag[act]gtaaa|tttac[agt]ct", "agg[act]taaa|ttta[agt]cct", "aggg[acg]aaa|ttt[cgt]ccct",
                                 "agggt[cgt]aa|tt[acg]accct", "agggta[cgt]a|t[acg]taccct", "agggtaa[cgt]|[acg]ttaccct"};
constexpr int kN = 9;
constexpr uint64_t kCutAlign = 1024;  // cuts at multiples of a scan chunk (any cut is correct; aligned ones cost nothing)
constexpr uint64_t kLeftHalo = 64;    // bytes before the own range a shard sees (context of ^ / $; 16-byte alignment)

#define HIP_OK(call)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                       \
      exit(2);                                                                         \
    }                                                                                  \
  } while (0)

struct Shard {
  int status = 0;
  std::string error;
  uint64_t counts[kN] = {};
};

void run_rank(int rank, int world, ncclComm_t comm, const char* seq, uint64_t n, Shard* out) {
  HIP_OK(hipSetDevice(rank));
  // programs are per device: compiled with that device current
  rj_program* progs[kN];
  uint64_t max_len = 0;
  for (int i = 0; i < kN; i++) {
    if (rj_compile(kPatterns[i], &progs[i]) != RJ_OK) {
      out->status = 2;
      out->error = rj_last_error();
      return;
    }
    rj_info info;
    rj_program_info(progs[i], &info);
    max_len = std::max<uint64_t>(max_len, info.max_len);
  }
  // match begins [own_b, own_e) belong to this rank (the last one owns the empty match at n as well); it sees the
  // bytes a match beginning there can reach, and a little before the range
  const auto cut = [&](int r) { return r == 0 ? 0 : r == world ? n + 1 : (n * static_cast<uint64_t>(r) / world) / kCutAlign * kCutAlign; };
  const uint64_t own_b = cut(rank), own_e = cut(rank + 1);
  const uint64_t lo = own_b > kLeftHalo ? (own_b - kLeftHalo) & ~15ull : 0;
  const uint64_t hi = std::min<uint64_t>(n, own_e + max_len);
  const uint64_t n_local = hi - lo;
  void* d_text = nullptr;
  HIP_OK(hipMalloc(&d_text, n_local + 16));
  HIP_OK(hipMemcpy(d_text, seq + lo, n_local, hipMemcpyHostToDevice));
  hipStream_t stream;
  HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  rj_multi* multi = nullptr;
  if (rj_multi_create(progs, kN, &multi) != RJ_OK) {
    out->status = 2;
    out->error = rj_last_error();
    return;
  }
  // one pass over the shard for all nine patterns, then the exchange: every rank gets the counts over the WHOLE text
  (void)rj_multi_set_counts_only(multi, 1);   // counts are all that is asked: one kernel per shard, rows of the exchange from its first / last match
  const int how = rj_multi_device_counts(multi, d_text, n_local, own_b - lo, std::min(own_e, n + 1) - lo, static_cast<int64_t>(lo), comm, rank,
                                         world, out->counts, stream);
  if (how < 0) {
    out->status = 2;
    out->error = rj_last_error();
  }
  rj_multi_destroy(multi);
  for (rj_program* p : progs) rj_program_free(p);
  HIP_OK(hipStreamDestroy(stream));
  HIP_OK(hipFree(d_text));
}

}  // namespace

int main(int argc, char** argv) {
  int want_devices = 0;  // --devices N: at most N of the visible devices (default: all of them)
  for (int i = 1; i < argc; i++)
    if (std::string(argv[i]) == "--devices" && i + 1 < argc) want_devices = atoi(argv[++i]);
  std::string input;
  {
    char buf[1 << 16];
    size_t got;
    while ((got = fread(buf, 1, sizeof(buf), stdin)) > 0) input.append(buf, got);
  }
  // strip the FASTA headers and the line breaks (sample/regexdna.cc:49: ReplaceAll(">.*\n|\n", ""))
  rj_program* strip = nullptr;
  if (rj_compile(">.*\n|\n", &strip) != RJ_OK) {
    fprintf(stderr, "%s\n", rj_last_error());
    return 2;
  }
  char* seq = nullptr;
  size_t n = 0;
  if (rj_replace_all(strip, input.data(), input.size(), "", 0, &seq, &n) < 0) {
    fprintf(stderr, "%s\n", rj_last_error());
    return 2;
  }
  rj_program_free(strip);

  int world = 0;
  HIP_OK(hipGetDeviceCount(&world));
  if (world < 1) {
    fprintf(stderr, "no GPU\n");
    return 2;
  }
  const int visible = world;
  if (want_devices > 0) world = std::min(world, want_devices);
  if (n < static_cast<size_t>(world) * (1u << 20)) world = 1;  // (a shard per device only pays from megabytes up)
  fprintf(stderr, "regexdna_rccl: %d device(s) of %d visible\n", world, visible);
  std::vector<ncclComm_t> comms(static_cast<size_t>(world));
  std::vector<int> devs(static_cast<size_t>(world));
  for (int r = 0; r < world; r++) devs[static_cast<size_t>(r)] = r;
  if (ncclCommInitAll(comms.data(), world, devs.data()) != ncclSuccess) {
    fprintf(stderr, "ncclCommInitAll failed\n");
    return 2;
  }
  std::vector<Shard> shards(static_cast<size_t>(world));
  std::vector<std::thread> threads;
  for (int r = 0; r < world; r++)
    threads.emplace_back(run_rank, r, world, comms[static_cast<size_t>(r)], seq, static_cast<uint64_t>(n), &shards[static_cast<size_t>(r)]);
  for (std::thread& t : threads) t.join();
  int status = 0;
  for (int r = 0; r < world; r++)
    if (shards[static_cast<size_t>(r)].status) {
      fprintf(stderr, "device %d: %s\n", r, shards[static_cast<size_t>(r)].error.c_str());
      status = 2;
    }
  if (!status) {
    for (int i = 0; i < kN; i++) printf("%s %llu\n", kPatterns[i], static_cast<unsigned long long>(shards[0].counts[i]));
    printf("\n%zu\n%zu\n", input.size(), n);
  }
  for (ncclComm_t c : comms) ncclCommDestroy(c);
  rj_free_text(seq);
  return status;
}
