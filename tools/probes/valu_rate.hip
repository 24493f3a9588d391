// tools/probes/valu_rate.hip -- how many cycles a SIMD of gfx950 spends per wave64 VALU instruction, by opcode (round 5):
// eight independent chains of one instruction per lane, 8 waves per SIMD on every SIMD of the device, HIP events around the
// launch.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHAIN8(OP)                                                          \
  asm volatile(OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" \
               OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n" \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(k))
#define CHAIN8_3(OP, TAIL)                                                  \
  asm volatile(OP " %0, %0, %8, %8" TAIL "\n" OP " %1, %1, %8, %8" TAIL "\n" OP " %2, %2, %8, %8" TAIL "\n" OP " %3, %3, %8, %8" TAIL "\n" \
               OP " %4, %4, %8, %8" TAIL "\n" OP " %5, %5, %8, %8" TAIL "\n" OP " %6, %6, %8, %8" TAIL "\n" OP " %7, %7, %8, %8" TAIL "\n" \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(k))
#define CHAIN8_DPP()                                                        \
  asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n" \
               "v_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n" \
               "v_mov_b32_dpp %4, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %5, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n" \
               "v_mov_b32_dpp %6, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\nv_mov_b32_dpp %7, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n" \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]))

template <int OP>
__global__ __launch_bounds__(256) void rate(unsigned* out, int iters, unsigned k) {
  unsigned a[8];
  for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 2654435761u + i;
  for (int it = 0; it < iters; it++) {
    if (OP == 0) CHAIN8("v_xor_b32");
    if (OP == 1) CHAIN8("v_add_u32");
    if (OP == 2) CHAIN8("v_lshlrev_b32");
    if (OP == 3) CHAIN8_3("v_alignbit_b32", "");
    if (OP == 4) CHAIN8_3("v_dot4_u32_u8", "");
    if (OP == 5) CHAIN8_3("v_bitop3_b32", " bitop3:0x48");
    if (OP == 6) CHAIN8_3("v_and_or_b32", "");
    if (OP == 7) CHAIN8_3("v_bfi_b32", "");
    if (OP == 8) CHAIN8_3("v_fma_f32", "");
    if (OP == 9) CHAIN8_DPP();
    if (OP == 10) CHAIN8_3("v_add3_u32", "");
    if (OP == 11) CHAIN8("v_and_b32");
  }
  unsigned v = 0;
  for (int i = 0; i < 8; i++) v ^= a[i];
  if (v == 0x12345678u) out[0] = v;
}

template <int OP>
double run(const char* name, unsigned* d_out, int cus, double ghz) {
  const int iters = 4096, grid = cus * 8;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(rate<OP>, dim3(grid), dim3(256), 0, 0, d_out, iters, 0x01010101u);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < 3; r++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate<OP>, dim3(grid), dim3(256), 0, 0, d_out, iters, 0x01010101u);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double wave_instr_per_simd = 8.0 /*waves*/ * iters * 8.0;
  const double cycles = best * 1e-3 * ghz * 1e9;
  printf("%-16s %8.3f ms  %.2f cycles per wave64 instruction per SIMD (at %.2f GHz)  %.1f T lane-ops/s\n", name, best, cycles / wave_instr_per_simd, ghz,
         wave_instr_per_simd * cus * 4 * 64 / (best * 1e-3) / 1e12);
  return best;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const double ghz = p.clockRate / 1e6;
  printf("%s: %d CUs, %.2f GHz\n", p.gcnArchName, p.multiProcessorCount, ghz);
  unsigned* d_out;
  hipMalloc(&d_out, 64);
  run<0>("v_xor_b32", d_out, p.multiProcessorCount, ghz);
  run<11>("v_and_b32", d_out, p.multiProcessorCount, ghz);
  run<1>("v_add_u32", d_out, p.multiProcessorCount, ghz);
  run<2>("v_lshlrev_b32", d_out, p.multiProcessorCount, ghz);
  run<3>("v_alignbit_b32", d_out, p.multiProcessorCount, ghz);
  run<4>("v_dot4_u32_u8", d_out, p.multiProcessorCount, ghz);
  run<5>("v_bitop3_b32", d_out, p.multiProcessorCount, ghz);
  run<6>("v_and_or_b32", d_out, p.multiProcessorCount, ghz);
  run<7>("v_bfi_b32", d_out, p.multiProcessorCount, ghz);
  run<10>("v_add3_u32", d_out, p.multiProcessorCount, ghz);
  run<8>("v_fma_f32", d_out, p.multiProcessorCount, ghz);
  run<9>("v_mov_b32_dpp", d_out, p.multiProcessorCount, ghz);
  return 0;
}
