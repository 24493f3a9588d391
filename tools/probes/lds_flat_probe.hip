// tools/probes/lds_flat_probe.hip -- latency of a dependent load from LDS through a generic (flat) pointer vs ds_read.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_flat_probe.hip -o /tmp/lds_flat_probe && /tmp/lds_flat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chase(const uint32_t* global_tab, int use_lds, int steps, uint32_t* out, long long* cycles, int mode) {
  extern __shared__ uint32_t tab[];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) tab[i] = (i * 97 + 13) & 4095;
  __syncthreads();
  uint32_t idx = threadIdx.x;
  long long t0 = clock64();
  if (mode == 0) {            // ds_read: the compiler knows it is LDS
    for (int s = 0; s < steps; s++) idx = tab[idx];
  } else {                    // generic pointer: LDS or global, decided at run time -> flat_load
    const uint32_t* p = use_lds ? tab : global_tab;
    for (int s = 0; s < steps; s++) idx = p[idx];
  }
  long long t1 = clock64();
  out[threadIdx.x] = idx;
  if (threadIdx.x == 0) *cycles = t1 - t0;
}
int main() {
  uint32_t *g, *out; long long* cyc;
  hipMalloc(&g, 4096 * 4); hipMalloc(&out, 1024 * 4); hipMalloc(&cyc, 8);
  uint32_t h[4096]; for (int i = 0; i < 4096; i++) h[i] = (i * 97 + 13) & 4095;
  hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice);
  const int steps = 4096;
  for (int mode = 0; mode < 3; mode++) {
    long long c = 0;
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(chase, dim3(1), dim3(64), 16384, 0, g, mode != 2, steps, out, cyc, mode == 0 ? 0 : 1);
      hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    }
    printf("%s: %.1f cycles per dependent load\n", mode == 0 ? "ds_read (LDS known)" : mode == 1 ? "flat_load -> LDS" : "flat_load -> global (L2/L1 hit)", (double)c / steps);
  }
  return 0;
}
