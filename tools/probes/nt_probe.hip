// tools/probes/nt_probe.hip -- MEASUREMENT: a read-only stream over a device buffer with the load's cache policy as the
// variable (default / nt / sc1 / sc0 sc1 / sc1 nt), in the read probe's shape (16 B per lane and load, four loads in flight)
// and in plane_count's (32 contiguous bytes per lane).  hipcc --offload-arch=gfx950 -O3 -o nt_probe nt_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int POLICY>
__device__ __forceinline__ u32x4 ld(const u32x4* p) {
  u32x4 v;
  if constexpr (POLICY == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  else if constexpr (POLICY == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  else if constexpr (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  else if constexpr (POLICY == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  else if constexpr (POLICY == 4) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int POLICY, int DEPTH>
__global__ __launch_bounds__(256) void probe(const u32x4* text, uint64_t n16, uint64_t span16, uint32_t* out) {
  const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const uint32_t lane = threadIdx.x & 63u;
  uint64_t i = wave * span16 + lane, end = (wave + 1) * span16;
  if (end > n16) end = n16;
  u32x4 acc = {0, 0, 0, 0};
  for (; i + 64 * (DEPTH - 1) < end; i += 64 * DEPTH) {
    u32x4 v[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; k++) v[k] = ld<POLICY>(text + i + 64 * k);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < DEPTH; k++) {
      asm volatile("" : "+v"(v[k]));
      acc ^= v[k];
    }
  }
  uint32_t x = acc.x ^ acc.y ^ acc.z ^ acc.w;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x ^= __shfl_xor(x, o);
  if (lane == 0) out[wave] = x;
}

template <int POLICY, int DEPTH>
float run(const void* d, uint64_t n, uint32_t* out, int grid, int launches) {
  const uint64_t n16 = n / 16, waves = static_cast<uint64_t>(grid) * 4;
  const uint64_t span16 = ((n16 + waves - 1) / waves + 64 * DEPTH - 1) / (64 * DEPTH) * (64 * DEPTH);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f, total = 0.f;
  for (int i = 0; i < launches + 2; i++) {
    hipExtLaunchKernelGGL((probe<POLICY, DEPTH>), dim3(grid), dim3(256), 0, 0, e0, e1, 0, static_cast<const u32x4*>(d), n16, span16, out);
    hipStreamSynchronize(0);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (i >= 2) {
      total += ms;
      if (ms < best) best = ms;
    }
  }
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  printf("  avg %.4f ms best %.4f ms  %.0f GB/s avg", total / launches, best, n / (total / launches) * 1e-6);
  return total / launches;
}

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 500000000ull;
  const int launches = argc > 2 ? atoi(argv[2]) : 20;
  void* d = nullptr;
  if (hipMalloc(&d, n + 4096) != hipSuccess) return 1;
  hipMemset(d, 0x5a, n);
  uint32_t* out = nullptr;
  hipMalloc(reinterpret_cast<void**>(&out), 4 * 4 * 16384);
  const char* names[] = {"default", "nt", "sc1", "sc0 sc1", "sc1 nt", "sc0 nt"};
  for (int rep = 0; rep < 2; rep++)
    for (int grid : {2048, 3072, 4096}) {
      printf("n=%llu grid=%d rep=%d\n", (unsigned long long)n, grid, rep);
#define ROW(P, D) printf(" %-8s depth %d:", names[P], D); run<P, D>(d, n, out, grid, launches); printf("\n");
      ROW(0, 4) ROW(1, 4) ROW(2, 4) ROW(3, 4) ROW(4, 4) ROW(5, 4)
      ROW(0, 8) ROW(1, 8)
      ROW(0, 2) ROW(1, 2)
    }
  return 0;
}
