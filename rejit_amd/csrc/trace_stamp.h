// rejit_amd/csrc/trace_stamp.h -- phase time stamps of a DEBUG build (-DRJ_TRACE_VERIFY; tools/verify_trace.py builds
// a separate library that way, the product never contains them).  Include FIRST in a .hip file that wants stamps
// (before the walker headers, which default RJ_STAMP to nothing) and place RJ_TRACE_EXPORT(name) once in it.
// Stamp 0 keeps the EARLIEST time any lane passed it (the kernel's start), every other stamp the LATEST: with many
// hits the numbers are the critical path of the slowest one.  name##_reset() clears the buffer between runs.
#ifndef REJIT_AMD_TRACE_STAMP_H_
#define REJIT_AMD_TRACE_STAMP_H_
#ifdef RJ_TRACE_VERIFY
#include <hip/hip_runtime.h>
static __device__ unsigned long long rj_trace_buf[64];  // wall_clock64(): 10 ns units
#define RJ_STAMP(i) ((i) == 0 ? atomicMin(&rj_trace_buf[0], static_cast<unsigned long long>(wall_clock64())) \
                              : atomicMax(&rj_trace_buf[i], static_cast<unsigned long long>(wall_clock64())))
// per-workgroup stamps with plain stores (no atomics: nothing for a lane to wait for): slot = blockIdx.x, i < 8
static __device__ unsigned long long rj_trace_wide[8 * 4096];
#define RJ_STAMP_AT(slot, i) (rj_trace_wide[((slot) & 4095u) * 8u + (i)] = wall_clock64())
#define RJ_TRACE_EXPORT(name)                                                                                        \
  extern "C" int name##_wide(unsigned long long* out) { return static_cast<int>(hipMemcpyFromSymbol(out, HIP_SYMBOL(rj_trace_wide), sizeof(rj_trace_wide))); } \
  extern "C" int name(unsigned long long* out) { return static_cast<int>(hipMemcpyFromSymbol(out, HIP_SYMBOL(rj_trace_buf), sizeof(rj_trace_buf))); } \
  extern "C" int name##_reset() {                                                                                    \
    unsigned long long z[64] = {~0ull};                                                                              \
    return static_cast<int>(hipMemcpyToSymbol(HIP_SYMBOL(rj_trace_buf), z, sizeof(z)));                              \
  }
#else
#define RJ_STAMP(i) ((void)0)
#define RJ_STAMP_AT(slot, i) ((void)0)
#define RJ_TRACE_EXPORT(name)
#endif
#endif
