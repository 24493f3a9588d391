// samples/complex_rccl.cc -- MatchAll of ONE pattern over a synthetic text sharded across EVERY GPU of the node, the ordered
// match list gathered on device 0 over RCCL: BASELINE configs[3] (the reference's benchmark regex
// `([complex]|(regexp)){2,7}abcdefgh(at|the|[e-nd]as well)` over random ASCII, tools/benchmarks/run.py:350) as a native
// caller of the C ABI.  One process; a thread, a shard of the text and an RCCL rank per device; the shard's scan, the carry
// of the left-most-longest selection over the cuts and the gather of the (begin, end) pairs are ONE call,
// rj_scan_gather_spans (include/rejit_hip.h).  The reference has no counterpart across devices (its callers hand one
// buffer to one MatchAll, include/rejit.h:65-68); its result -- the ordered list -- is what device 0 ends up with.
//
//   complex_rccl [--devices N] [--bytes B] [--pattern RX] [--needle S] [--plant K] [--dump-text FILE] [--print-spans]
//
// The text: byte i = '0' + mix(seed, i) % 74 (the harness's range ['0','z'), tools/benchmarks/run.py:313), generated on
// each device for its own range, with K copies of the needle planted at evenly spread offsets and across every cut.
// Output: devices, bytes, matches, a digest of the list (sum of begins, sum of ends, xor of begin * 31 + end), the time of
// the second call.  `make -C samples rccl`.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "rejit_hip.h"

namespace {

#define HIP_OK(call)                                                 \
  do {                                                               \
    hipError_t e_ = (call);                                          \
    if (e_ != hipSuccess) {                                          \
      fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));     \
      exit(2);                                                       \
    }                                                                \
  } while (0)

__device__ __host__ inline uint64_t mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__global__ void fill_text(uint8_t* out, uint64_t first, uint64_t count, uint64_t seed) {
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += stride)
    out[i] = static_cast<uint8_t>('0' + mix(seed ^ ((first + i) * 0x2545F4914F6CDD1Dull)) % 74);
}

struct Options {
  int devices = 0;
  uint64_t bytes = 64ull << 20;
  std::string pattern = "([complex]|(regexp)){2,7}abcdefgh(at|the|[e-nd]as well)";
  std::string needle = "complexregexpabcdefghthe";
  uint64_t plant = 1000;
  std::string dump;
  bool print_spans = false;
};

constexpr uint64_t kCutAlign = 1024, kLeftHalo = 64, kSeed = 0xC0FFEE;

uint64_t cut(uint64_t n, int r, int world) { return r == 0 ? 0 : r == world ? n + 1 : (n * static_cast<uint64_t>(r) / world) / kCutAlign * kCutAlign; }

// where the needles go: evenly spread, and one across every cut (3 bytes before it) -- the same list on every rank
std::vector<uint64_t> plant_offsets(const Options& o, int world) {
  std::vector<uint64_t> at;
  const uint64_t L = o.needle.size();
  if (L == 0 || o.bytes < 4 * L) return at;
  for (uint64_t j = 0; j < o.plant; j++) at.push_back((j + 1) * (o.bytes - L) / (o.plant + 1));
  for (int r = 1; r < world; r++) {
    const uint64_t c = cut(o.bytes, r, world);
    if (c >= 3 && c - 3 + L <= o.bytes) at.push_back(c - 3);
  }
  std::sort(at.begin(), at.end());
  std::vector<uint64_t> kept;
  for (uint64_t a : at)
    if (kept.empty() || a >= kept.back() + L) kept.push_back(a);  // (no overlapping plants)
  return kept;
}

struct RankResult {
  int status = 0;
  std::string error;
  int64_t total = 0;
  double ms = 0;
  std::vector<uint64_t> spans;  // root only
};

void run_rank(int rank, int world, ncclComm_t comm, const Options& o, RankResult* out) {
  HIP_OK(hipSetDevice(rank));
  rj_program* prog = nullptr;
  if (rj_compile(o.pattern.c_str(), &prog) != RJ_OK) {
    out->status = 2;
    out->error = rj_last_error();
    return;
  }
  rj_info info;
  rj_program_info(prog, &info);
  const uint64_t n = o.bytes;
  const uint64_t own_b = cut(n, rank, world), own_e = cut(n, rank + 1, world);
  const uint64_t lo = own_b > kLeftHalo ? (own_b - kLeftHalo) & ~15ull : 0;
  // the bytes a match beginning in the own range can reach (an unbounded or ring-artefact-risk pattern: the rest of the text)
  const bool to_the_end = info.max_len == ~0ull || info.ring_artefact_risk;
  const uint64_t hi = to_the_end ? n : std::min<uint64_t>(n, own_e + info.max_len);
  const uint64_t n_local = hi - lo;
  uint8_t* d_text = nullptr;
  HIP_OK(hipMalloc(reinterpret_cast<void**>(&d_text), n_local + 16));
  hipStream_t stream;
  HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  hipLaunchKernelGGL(fill_text, dim3(4096), dim3(256), 0, stream, d_text, lo, n_local, kSeed);
  HIP_OK(hipStreamSynchronize(stream));
  const std::vector<uint64_t> plants = plant_offsets(o, world);
  for (uint64_t a : plants) {
    const uint64_t b = std::max(a, lo), e = std::min<uint64_t>(a + o.needle.size(), hi);
    if (b < e) HIP_OK(hipMemcpy(d_text + (b - lo), o.needle.data() + (b - a), e - b, hipMemcpyHostToDevice));
  }
  if (!o.dump.empty() && world == 1) {
    std::vector<uint8_t> h(n);
    HIP_OK(hipMemcpy(h.data(), d_text, n, hipMemcpyDeviceToHost));
    FILE* f = fopen(o.dump.c_str(), "wb");
    if (f) {
      fwrite(h.data(), 1, n, f);
      fclose(f);
    }
  }
  rj_scan* scan = nullptr;
  if (rj_scan_create(prog, &scan) != RJ_OK) {
    out->status = 2;
    out->error = rj_last_error();
    return;
  }
  for (int call = 0; call < 2; call++) {  // the second call is the timed one (the first sizes the device lists)
    const auto t0 = std::chrono::steady_clock::now();
    out->total = rj_scan_gather_spans(scan, d_text, n_local, own_b - lo, std::min(own_e, n + 1) - lo, static_cast<int64_t>(lo), comm, rank, world, 0,
                                      stream);
    out->ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (out->total < 0) {
      out->status = 2;
      out->error = rj_last_error();
      return;
    }
  }
  if (rank == 0) {
    uint64_t count = 0;
    const uint64_t* d = rj_scan_gathered_spans(scan, &count);
    out->spans.resize(2 * count);
    if (count) HIP_OK(hipMemcpy(out->spans.data(), d, 16 * count, hipMemcpyDeviceToHost));
  }
  rj_scan_destroy(scan);
  rj_program_free(prog);
  HIP_OK(hipStreamDestroy(stream));
  HIP_OK(hipFree(d_text));
}

}  // namespace

int main(int argc, char** argv) {
  Options o;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
    if (a == "--devices") o.devices = atoi(next());
    else if (a == "--bytes") o.bytes = strtoull(next(), nullptr, 10);
    else if (a == "--pattern") o.pattern = next();
    else if (a == "--needle") o.needle = next();
    else if (a == "--plant") o.plant = strtoull(next(), nullptr, 10);
    else if (a == "--dump-text") o.dump = next();
    else if (a == "--print-spans") o.print_spans = true;
    else {
      fprintf(stderr, "usage: complex_rccl [--devices N] [--bytes B] [--pattern RX] [--needle S] [--plant K] [--dump-text FILE] [--print-spans]\n");
      return 2;
    }
  }
  int visible = 0;
  HIP_OK(hipGetDeviceCount(&visible));
  if (visible < 1) {
    fprintf(stderr, "no GPU\n");
    return 2;
  }
  int world = o.devices > 0 ? std::min(o.devices, visible) : visible;
  if (o.bytes < static_cast<uint64_t>(world) * (1u << 16)) world = 1;
  std::vector<ncclComm_t> comms(static_cast<size_t>(world));
  std::vector<int> devs(static_cast<size_t>(world));
  for (int r = 0; r < world; r++) devs[static_cast<size_t>(r)] = r;
  if (ncclCommInitAll(comms.data(), world, devs.data()) != ncclSuccess) {
    fprintf(stderr, "ncclCommInitAll failed\n");
    return 2;
  }
  std::vector<RankResult> res(static_cast<size_t>(world));
  std::vector<std::thread> threads;
  for (int r = 0; r < world; r++) threads.emplace_back(run_rank, r, world, comms[static_cast<size_t>(r)], std::cref(o), &res[static_cast<size_t>(r)]);
  for (std::thread& t : threads) t.join();
  int status = 0;
  for (int r = 0; r < world; r++)
    if (res[static_cast<size_t>(r)].status) {
      fprintf(stderr, "device %d: %s\n", r, res[static_cast<size_t>(r)].error.c_str());
      status = 2;
    }
  if (!status) {
    const RankResult& R = res[0];
    uint64_t sb = 0, se = 0, x = 0;
    for (size_t i = 0; i + 1 < R.spans.size(); i += 2) {
      sb += R.spans[i];
      se += R.spans[i + 1];
      x ^= mix(R.spans[i] * 31 + R.spans[i + 1]);
    }
    double ms = 0;
    for (int r = 0; r < world; r++) ms = std::max(ms, res[static_cast<size_t>(r)].ms);
    printf("devices %d (visible %d)\nbytes %llu\nmatches %lld\ndigest %llu %llu %llx\n", world, visible, static_cast<unsigned long long>(o.bytes),
           static_cast<long long>(R.total), static_cast<unsigned long long>(sb), static_cast<unsigned long long>(se), static_cast<unsigned long long>(x));
    printf("ms %.3f (%.1f GB/s over all devices, list gathered on device 0)\n", ms, o.bytes / ms / 1e6);
    if (o.print_spans)
      for (size_t i = 0; i + 1 < R.spans.size(); i += 2) printf("%llu %llu\n", static_cast<unsigned long long>(R.spans[i]), static_cast<unsigned long long>(R.spans[i + 1]));
  }
  for (ncclComm_t c : comms) ncclCommDestroy(c);
  return status;
}
