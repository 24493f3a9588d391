// tools/probes/startup_probe.cc -- where does a GPU process's start-up go?  (run on the GPU box)
//   g++ -O2 -std=c++17 -Iinclude tools/probes/startup_probe.cc -Lrejit_amd -lrejit_hip -Wl,-rpath,$PWD/rejit_amd -L/opt/rocm/lib -lamdhip64 -o /tmp/startup_probe
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "rejit_hip.h"
extern "C" int hipInit(unsigned);
extern "C" int hipFree(void*);
extern "C" int hipHostMalloc(void**, size_t, unsigned);
extern "C" int hipMalloc(void**, size_t);
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  double t0 = now();
  hipInit(0);
  double t1 = now();
  hipFree(nullptr);
  double t2 = now();
  void* p = nullptr;
  hipMalloc(&p, 64 << 20);
  double t3 = now();
  void* h = nullptr;
  hipHostMalloc(&h, 64 << 20, 0);
  double t4 = now();
  rj_program* re = nullptr;
  rj_compile("regexp", &re);
  double t5 = now();
  std::string text(20000, 'x');
  uint64_t* sp = nullptr;
  rj_match_all(re, text.data(), text.size(), &sp);
  double t6 = now();
  rj_match_all(re, text.data(), text.size(), &sp);
  double t7 = now();
  std::string big(64 << 20, 'y');
  std::vector<const char*> texts(3000);
  std::vector<size_t> sizes(3000, 20000);
  for (size_t i = 0; i < 3000; i++) texts[i] = big.data() + i * 20000;
  std::vector<uint64_t> counts(3000);
  rj_match_all_batch(re, texts.data(), sizes.data(), 3000, counts.data(), &sp);
  double t8 = now();
  rj_match_all_batch(re, texts.data(), sizes.data(), 3000, counts.data(), &sp);
  double t9 = now();
  printf("hipInit %.1f ms, hipFree(0) %.1f, hipMalloc 64M %.1f, hipHostMalloc 64M %.1f, rj_compile %.1f, first match_all 20KB %.1f, second %.3f, first batch 60MB %.1f, second %.1f\n",
         t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6, t8 - t7, t9 - t8);
  return 0;
}
