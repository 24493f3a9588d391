// tools/mqsad_probe.hip -- semantics + throughput probe for v_mqsad_u32_u8 on gfx950.
// hipcc --offload-arch=gfx950 -O3 tools/mqsad_probe.hip -o /tmp/mqsad_probe && /tmp/mqsad_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void sem(const uint64_t* s0, const uint32_t* s1, u32x4* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32x4 z = {0, 0, 0, 0};
  out[i] = __builtin_amdgcn_mqsad_u32_u8(s0[i], s1[i], z);
}

__global__ void sem1(const uint64_t* s0, const uint32_t* s1, uint32_t* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = __builtin_amdgcn_msad_u8((uint32_t)s0[i], s1[i], 7u);
}

template <int MODE>
__global__ void thr(uint32_t* out, int iters) {
  uint32_t a = threadIdx.x * 2654435761u, b = a ^ 0x12345678u, c = b + 77;
  u32x4 acc = {0, 0, 0, 0};
  uint32_t m = 0xFFFFFFFFu;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      if (MODE == 0) {
        uint64_t s0 = ((uint64_t)b << 32) | a;
        acc = __builtin_amdgcn_mqsad_u32_u8(s0, c, acc);
        a += acc.x;
      } else if (MODE == 2) {   // v_msad_u8 x2 + v_min: the 8-byte masked window test
        uint32_t t = __builtin_amdgcn_msad_u8(a, c, 0u);
        t = __builtin_amdgcn_msad_u8(b, c ^ 0x5a5a5a5au, t);
        m = m < t ? m : t;
        a += m + u;
      } else if (MODE == 3) {   // current form: xor/bitop3 + xor + and_or + min
        uint32_t t = (a ^ c) & 0xFFFFFF00u;
        uint32_t v = ((b ^ (c ^ 0x5a5a5a5au)) & 0x00FFFFFFu) | t;
        m = m < v ? m : v;
        a += m + u;
      } else {
        uint32_t t = (a ^ c) & b;
        m = m < t ? m : t;
        a += m + u;
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + acc.x + acc.y + acc.z + acc.w + m;
}

int main() {
  const int n = 4096;
  uint64_t* h0 = (uint64_t*)malloc(n * 8);
  uint32_t* h1 = (uint32_t*)malloc(n * 4);
  srand(1);
  for (int i = 0; i < n; i++) {
    uint64_t v = 0;
    for (int k = 0; k < 8; k++) v |= (uint64_t)(rand() & 0xFF) << (8 * k);
    h0[i] = v;
    uint32_t r = 0;
    for (int k = 0; k < 4; k++) r |= (uint32_t)((rand() % 4 == 0) ? 0 : (rand() & 0xFF)) << (8 * k);
    if (i % 3 == 0) r = (uint32_t)(v >> (8 * (i % 4)));  // plant exact matches at offset i%4
    h1[i] = r;
  }
  uint64_t* d0; uint32_t* d1; u32x4* dout;
  hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 4); hipMalloc(&dout, n * 16);
  hipMemcpy(d0, h0, n * 8, hipMemcpyHostToDevice); hipMemcpy(d1, h1, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(sem, dim3(n / 256), dim3(256), 0, 0, d0, d1, dout, n);
  uint32_t* ho = (uint32_t*)malloc(n * 16);
  hipMemcpy(ho, dout, n * 16, hipMemcpyDeviceToHost);
  // hypotheses: mask on S1 byte == 0 (A) or on S0 byte == 0 (B) or none (C)
  long badA = 0, badB = 0, badC = 0;
  for (int i = 0; i < n; i++)
    for (int p = 0; p < 4; p++) {
      uint32_t a = 0, b = 0, c = 0;
      for (int k = 0; k < 4; k++) {
        int x = (h0[i] >> (8 * (p + k))) & 0xFF, y = (h1[i] >> (8 * k)) & 0xFF;
        int d = abs(x - y);
        if (y != 0) a += d;
        if (x != 0) b += d;
        c += d;
      }
      badA += ho[4 * i + p] != a; badB += ho[4 * i + p] != b; badC += ho[4 * i + p] != c;
    }
  printf("semantics mismatches: maskS1=%ld maskS0=%ld nomask=%ld (of %d)\n", badA, badB, badC, n * 4);
  for (int i = 0; i < 3; i++) printf("  s0=%016llx s1=%08x -> %u %u %u %u\n", (unsigned long long)h0[i], h1[i], ho[4*i], ho[4*i+1], ho[4*i+2], ho[4*i+3]);
  uint32_t* dt; hipMalloc(&dt, 2048 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  {
    // semantics of the single (non-quad) v_msad_u8: mask on zero bytes of src1?
    long bad = 0;
    uint32_t* dres; hipMalloc(&dres, n * 4);
    hipLaunchKernelGGL(sem1, dim3(n / 256), dim3(256), 0, 0, d0, d1, dres, n);
    uint32_t* hr = (uint32_t*)malloc(n * 4);
    hipMemcpy(hr, dres, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) {
      uint32_t a = 0;
      for (int k = 0; k < 4; k++) { int x = (h0[i] >> (8 * k)) & 0xFF, y = (h1[i] >> (8 * k)) & 0xFF; if (y != 0) a += abs(x - y); }
      bad += hr[i] != a + 7;
    }
    printf("v_msad_u8 semantics (mask on src1 zero bytes, + src2): mismatches %ld of %d\n", bad, n);
  }
  for (int mode = 0; mode < 4; mode++) {
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(thr<0>, dim3(2048), dim3(256), 0, 0, dt, 2000);
      else if (mode == 1) hipLaunchKernelGGL(thr<1>, dim3(2048), dim3(256), 0, 0, dt, 2000);
      else if (mode == 2) hipLaunchKernelGGL(thr<2>, dim3(2048), dim3(256), 0, 0, dt, 2000);
      else hipLaunchKernelGGL(thr<3>, dim3(2048), dim3(256), 0, 0, dt, 2000);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double ops = 2048.0 * 4 * 2000 * 16;  // wave-instructions of the probed op
    printf("mode %d (%s): %.3f ms, %.1f G wave-instr/s (dependent chain + 1-3 helper ops)\n", mode, mode == 0 ? "mqsad" : mode == 1 ? "xor/and/min" : mode == 2 ? "msad x2 + min (+2 adds)" : "xor,and,xor,and_or,min (+2 adds)", best, ops / best / 1e6);
  }
  return 0;
}
