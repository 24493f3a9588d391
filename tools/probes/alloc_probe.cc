// tools/probes/alloc_probe.cc -- what do hipMalloc / hipFree / hipHostMalloc cost by size?  (run on the GPU box)
//   g++ -O2 -std=c++17 tools/probes/alloc_probe.cc -L/opt/rocm/lib -lamdhip64 -o /tmp/alloc_probe
#include <chrono>
#include <cstdio>
#include <initializer_list>
extern "C" int hipMalloc(void**, size_t);
extern "C" int hipFree(void*);
extern "C" int hipHostMalloc(void**, size_t, unsigned);
extern "C" int hipHostFree(void*);
extern "C" int hipMemset(void*, int, size_t);
extern "C" int hipDeviceSynchronize();
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  void* w = nullptr;
  hipMalloc(&w, 1 << 20);
  hipMemset(w, 0, 1 << 20);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 2; rep++)
    for (size_t mb : {1ul, 16ul, 64ul, 256ul, 1024ul, 4096ul, 16384ul}) {
      void* p = nullptr;
      double t0 = now();
      hipMalloc(&p, mb << 20);
      double t1 = now();
      hipMemset(p, 0, mb << 20);
      hipDeviceSynchronize();
      double t2 = now();
      hipFree(p);
      double t3 = now();
      printf("%6zu MiB: hipMalloc %.3f ms, first touch (memset) %.3f ms, hipFree %.3f ms\n", mb, t1 - t0, t2 - t1, t3 - t2);
    }
  for (size_t mb : {1ul, 16ul, 64ul, 256ul}) {
    void* p = nullptr;
    double t0 = now();
    hipHostMalloc(&p, mb << 20, 0);
    double t1 = now();
    hipHostFree(p);
    double t2 = now();
    printf("%6zu MiB: hipHostMalloc %.3f ms, hipHostFree %.3f ms\n", mb, t1 - t0, t2 - t1);
  }
  return 0;
}
