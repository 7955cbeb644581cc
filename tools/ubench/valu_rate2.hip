// Second VALU microbenchmark: operand-source effects (SGPR / literal / inline constants) and misc ops.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define OPS8(INS, TAIL) INS " %0, %0" TAIL "\n" INS " %1, %1" TAIL "\n" INS " %2, %2" TAIL "\n" INS " %3, %3" TAIL "\n" \
                        INS " %4, %4" TAIL "\n" INS " %5, %5" TAIL "\n" INS " %6, %6" TAIL "\n" INS " %7, %7" TAIL "\n"
#define OPS8L(INS, LIT) INS " %0, " LIT ", %0\n" INS " %1, " LIT ", %1\n" INS " %2, " LIT ", %2\n" INS " %3, " LIT ", %3\n" \
                       INS " %4, " LIT ", %4\n" INS " %5, " LIT ", %5\n" INS " %6, " LIT ", %6\n" INS " %7, " LIT ", %7\n"
#define REGS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)

template <int OP>
__global__ void __launch_bounds__(512, 2) bench(float* out, int iters, float seed) {
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  float cv = seed * 1.0001f;
  for (int i = 0; i < iters; ++i) {
    if constexpr (OP == 0) { REP8(asm volatile(OPS8("v_mul_f32", ", %8") : REGS : "v"(cv));) }
    else if constexpr (OP == 1) { REP8(asm volatile(OPS8("v_mul_f32", ", %8") : REGS : "s"(seed));) }
    else if constexpr (OP == 2) { REP8(asm volatile(OPS8L("v_mul_f32", "0x3f3504f3") : REGS);) }   // literal
    else if constexpr (OP == 3) { REP8(asm volatile(OPS8("v_mul_f32", ", 0.5") : REGS);) }          // inline const
    else if constexpr (OP == 4) { REP8(asm volatile(OPS8("v_add_f32", ", %8") : REGS : "s"(seed));) }
    else if constexpr (OP == 5) { REP8(asm volatile(OPS8("v_max_f32", ", %8") : REGS : "v"(cv));) }
    else if constexpr (OP == 6) { REP8(asm volatile(OPS8("v_min_f32", ", %8") : REGS : "v"(cv));) }
    else if constexpr (OP == 7) { REP8(asm volatile(OPS8("v_sub_f32", ", %8") : REGS : "v"(cv));) }
    else if constexpr (OP == 8) { REP8(asm volatile(OPS8("v_xor_b32", ", %8") : REGS : "v"(cv));) }
    else if constexpr (OP == 9) { REP8(asm volatile(OPS8("v_mov_b32", "") : REGS);) }
    else if constexpr (OP == 10) { REP8(asm volatile(OPS8("v_cvt_f32_ubyte1", "") : REGS);) }
    else if constexpr (OP == 11) { REP8(asm volatile(OPS8("v_cvt_f32_u32", "") : REGS);) }
    else if constexpr (OP == 12) { REP8(asm volatile(OPS8("v_dot4_u32_u8", ", %8, %0") : REGS : "v"(cv));) }
    else if constexpr (OP == 13) { REP8(asm volatile(OPS8("v_fmamk_f32", ", 0x3f3504f3, %8") : REGS : "v"(cv));) }
    else if constexpr (OP == 14) { REP8(asm volatile(OPS8("v_max_i32", ", %8") : REGS : "v"(cv));) }
    else if constexpr (OP == 15) { REP8(asm volatile(OPS8L("v_and_b32", "0xff") : REGS);) }
    else if constexpr (OP == 16) { REP8(asm volatile(OPS8("v_bfe_u32", ", 8, 8") : REGS);) }
    else if constexpr (OP == 17) { REP8(asm volatile(OPS8("v_exp_f32", "") : REGS);) }
    else if constexpr (OP == 18) { REP8(asm volatile(OPS8("v_rcp_f32", "") : REGS);) }
    else if constexpr (OP == 19) { REP8(asm volatile(OPS8("v_add_u32", ", %8") : REGS : "v"(cv));) }
    else if constexpr (OP == 20) { REP8(asm volatile(OPS8("v_mul_legacy_f32", ", %8") : REGS : "v"(cv));) }
    else if constexpr (OP == 21) { REP8(asm volatile(OPS8("v_max3_f32", ", %8, %8") : REGS : "v"(cv));) }
    else if constexpr (OP == 22) { REP8(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                                                     "v_permlane32_swap_b32 %1, %2\n v_permlane32_swap_b32 %3, %4\n v_permlane32_swap_b32 %5, %6\n v_permlane32_swap_b32 %7, %0\n" : REGS);) }
    else if constexpr (OP == 23) { REP8(asm volatile(OPS8("v_cndmask_b32_e64", ", %8, vcc") : REGS : "v"(cv) : "vcc");) }
    else if constexpr (OP == 24) { REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n"
                                                     "v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n" : REGS : "v"(cv) : "vcc");) }
    else if constexpr (OP == 25) { REP8(asm volatile(OPS8("v_add_u32_dpp", ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") : REGS : "v"(cv));) }
    else if constexpr (OP == 26) { REP8(asm volatile(OPS8("v_cndmask_b32_e32", ", %8, vcc") : REGS : "v"(cv) : "vcc");) }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int OP>
void run(const char* name, float* d) {
  const int iters = 2000, grid = 256, block = 512;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  bench<OP><<<grid, block>>>(d, 10, 1.0f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  bench<OP><<<grid, block>>>(d, iters, 1.0f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %7.3f ms  %5.2f ns per wave-instr per SIMD\n", name, ms, ms * 1e6 / (2000.0 * 64.0 * 2.0));
}

int main() {
  float* d; (void)hipMalloc(&d, 256 * 512 * 4);
  run<0>("v_mul_f32 v,v", d);   run<1>("v_mul_f32 v,s", d);   run<2>("v_mul_f32 v,literal", d); run<3>("v_mul_f32 v,inline", d);
  run<4>("v_add_f32 v,s", d);   run<5>("v_max_f32 v,v", d);   run<6>("v_min_f32 v,v", d);       run<7>("v_sub_f32 v,v", d);
  run<8>("v_xor_b32", d);       run<9>("v_mov_b32", d);       run<10>("v_cvt_f32_ubyte1", d);   run<11>("v_cvt_f32_u32", d);
  run<12>("v_dot4_u32_u8", d);  run<13>("v_fmamk_f32 literal", d); run<14>("v_max_i32", d);     run<15>("v_and_b32 lit", d);
  run<16>("v_bfe_u32", d);      run<17>("v_exp_f32", d);      run<18>("v_rcp_f32", d);          run<19>("v_add_u32", d);
  run<20>("v_mul_legacy_f32", d); run<21>("v_max3_f32", d);
  run<22>("v_permlane32_swap_b32", d); run<23>("v_cndmask_b32_e64 (vcc)", d); run<26>("v_cndmask_b32_e32 (vcc)", d); run<24>("v_cmp_lt_f32 -> vcc", d); run<25>("v_add_u32_dpp quad_perm", d);
  return 0;
}
