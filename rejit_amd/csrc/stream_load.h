// rejit_amd/csrc/stream_load.h -- loads of TEXT THAT A KERNEL READS ONCE, with the non-temporal cache policy
// (`global_load_dwordx4 ... nt`).  Round 6, measured (tools/probes/nt_probe.hip: a read-only stream over 500 MB and 5 GB, 16
// bytes per lane and load, 2 / 4 / 8 loads in flight, grids of 2048 .. 4096 workgroups): default policy 5.9-6.2 TB/s, nt
// 6.9-7.0 TB/s; sc0 / sc1 beside either change nothing.  A streamed line that claims no place among the lines the L2 and the
// MALL try to keep is the whole difference, and the main stream of every scan kernel is such a line: the reference's fast
// forward (src/x64/codegen-x64.cc:1292-1403) reads its text once, front to back, as well.  What a kernel re-reads soon (a
// neighbour lane's first bytes, a candidate's window) keeps the default policy.
// RJ_NO_NT_LOADS (build time) switches the policy off: A/B builds.
#ifndef REJIT_AMD_STREAM_LOAD_H_
#define REJIT_AMD_STREAM_LOAD_H_

#include <hip/hip_runtime.h>

#include <cstdint>

namespace rejit_amd {

typedef uint32_t rj_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t rj_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint4 stream_load16(const void* p) {
#ifdef RJ_NO_NT_LOADS
  return *reinterpret_cast<const uint4*>(p);
#else
  const rj_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const rj_u32x4*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
#endif
}

__device__ __forceinline__ uint2 stream_load8(const void* p) {
#ifdef RJ_NO_NT_LOADS
  return *reinterpret_cast<const uint2*>(p);
#else
  const rj_u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const rj_u32x2*>(p));
  return make_uint2(v.x, v.y);
#endif
}

}  // namespace rejit_amd

#endif  // REJIT_AMD_STREAM_LOAD_H_
