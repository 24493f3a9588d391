// tools/probes/read_probe.hip -- MEASUREMENT, not part of the matching path and not part of librejit_hip.so (round 6: it was an
// exported symbol of the product's C ABI until then): the average duration of a READ-ONLY kernel over a device buffer, the
// achievable ceiling a scan kernel is quoted against in the same run (bench.py: `hbm_ceiling`).  Built
// by rejit_amd.build() into rejit_amd/librejit_bench.so (in tree: it travels to the GPU box with the snapshot).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdint>
#include <cstdlib>

#include "../../rejit_amd/csrc/stream_load.h"

namespace {

// NT: the scans' own load policy (stream_load.h: non-temporal, 6.9-7.0 TB/s); false: the default policy (5.9-6.2 TB/s), what
// rounds 1-5 quoted as the ceiling
template <bool NT>
__device__ __forceinline__ uint4 probe_load(const uint4* p) {
  if constexpr (NT) return rejit_amd::stream_load16(p);
  else return *p;
}

// The achievable ceiling of a READ-ONLY stream on this device, measured in the run that quotes a scan against it
// ("measure a plain device read-only kernel in the same run as the achievable ceiling"): every lane reads
// 16 bytes per load, four loads in flight per lane, XORs them together and the wave leaves one word -- nothing else.
// Same launch shape as the scans (workgroups of four waves over contiguous spans).
template <bool NT>
__global__ __launch_bounds__(256) void stream_read_probe(const uint4* text, uint64_t n16, uint64_t span16, uint32_t* out) {
  const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const uint32_t lane = threadIdx.x & 63u;
  uint64_t i = wave * span16 + lane, end = (wave + 1) * span16;
  if (end > n16) end = n16;
  uint4 acc{0, 0, 0, 0};
  for (; i + 192 < end; i += 256) {
    const uint4 a = probe_load<NT>(text + i), b = probe_load<NT>(text + i + 64), c = probe_load<NT>(text + i + 128), d = probe_load<NT>(text + i + 192);
    acc.x ^= a.x ^ b.x ^ c.x ^ d.x;
    acc.y ^= a.y ^ b.y ^ c.y ^ d.y;
    acc.z ^= a.z ^ b.z ^ c.z ^ d.z;
    acc.w ^= a.w ^ b.w ^ c.w ^ d.w;
  }
  for (; i < end; i += 64) {
    const uint4 a = probe_load<NT>(text + i);
    acc.x ^= a.x;
    acc.y ^= a.y;
    acc.z ^= a.z;
    acc.w ^= a.w;
  }
  uint32_t v = acc.x ^ acc.y ^ acc.z ^ acc.w;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o);
  if (lane == 0) out[wave] = v;
}

// (measurement only, RJ_PROBE_PATTERN=1: the same reads in plane_count's layout -- a lane takes 32 CONTIGUOUS bytes as two
// 16-byte loads, so a load instruction of the wave touches every second 16 bytes of 2 KiB)
template <bool NT>
__global__ __launch_bounds__(256) void stream_read_probe_pairs(const uint4* text, uint64_t n16, uint64_t span16, uint32_t* out) {
  const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const uint32_t lane = threadIdx.x & 63u;
  uint64_t i = wave * span16 + 2 * lane, end = (wave + 1) * span16;
  if (end > n16) end = n16;
  uint4 acc{0, 0, 0, 0};
  for (; i + 129 < end; i += 256) {
    const uint4 a = probe_load<NT>(text + i), b = probe_load<NT>(text + i + 1), c = probe_load<NT>(text + i + 128), d = probe_load<NT>(text + i + 129);
    acc.x ^= a.x ^ b.x ^ c.x ^ d.x;
    acc.y ^= a.y ^ b.y ^ c.y ^ d.y;
    acc.z ^= a.z ^ b.z ^ c.z ^ d.z;
    acc.w ^= a.w ^ b.w ^ c.w ^ d.w;
  }
  uint32_t v = acc.x ^ acc.y ^ acc.z ^ acc.w;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o);
  if (lane == 0) out[wave] = v;
}

template <bool NT>
static void launch_stream_read_probe(const void* d_text, uint64_t n, uint32_t* d_out, int grid, hipEvent_t t0, hipEvent_t t1, hipStream_t st) {
  const uint64_t n16 = n / 16, waves = static_cast<uint64_t>(grid) * 4;
  const uint64_t span16 = ((n16 + waves - 1) / waves + 63) / 64 * 64;
  static const bool pairs = getenv("RJ_PROBE_PATTERN") && atoi(getenv("RJ_PROBE_PATTERN")) == 1;
  if (pairs) {
    const uint64_t span = (span16 + 255) / 256 * 256;
    hipExtLaunchKernelGGL(stream_read_probe_pairs<NT>, dim3(grid), dim3(256), 0, st, t0, t1, 0, static_cast<const uint4*>(d_text), n16, span, d_out);
    return;
  }
  hipExtLaunchKernelGGL(stream_read_probe<NT>, dim3(grid), dim3(256), 0, st, t0, t1, 0, static_cast<const uint4*>(d_text), n16, span16, d_out);
}


}  // namespace

// average ms of `launches` launches (two untimed ones first) over d_text[0..n); < 0: an error.  policy 0: non-temporal loads (the
// scans' own), 1: the default policy
extern "C" float rjb_stream_read_probe_policy(const void* d_text, uint64_t n, int launches, void* hip_stream, int policy) {
  if (!d_text || n < (1u << 20) || launches < 1 || (reinterpret_cast<uintptr_t>(d_text) & 15u) != 0) return -1.f;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  // the scans' own launch shape: a workgroup per 128 KiB, at most 16 Ki of them
  uint64_t blocks = n / 1024 / 128;
  if (blocks > 16384) blocks = 16384;
  if (blocks < 256) blocks = 256;
  const int grid = static_cast<int>(blocks);
  uint32_t* out = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&out), sizeof(uint32_t) * 4 * static_cast<size_t>(grid)) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
      hipEventCreate(&e1) != hipSuccess)
    return -1.f;
  float total = 0.f;
  for (int i = 0; i < launches + 2; i++) {
    if (policy == 0) launch_stream_read_probe<true>(d_text, n, out, grid, e0, e1, st);
    else launch_stream_read_probe<false>(d_text, n, out, grid, e0, e1, st);
    if (hipStreamSynchronize(st) != hipSuccess) {
      total = -1.f;
      break;
    }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (i >= 2) total += ms;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(out);
  return total < 0.f ? -1.f : total / static_cast<float>(launches);
}

extern "C" float rjb_stream_read_probe(const void* d_text, uint64_t n, int launches, void* hip_stream) {
  return rjb_stream_read_probe_policy(d_text, n, launches, hip_stream, 0);
}
