// Microbenchmark: issue rate of the VALU ops the spectrum kernel is made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ void __launch_bounds__(512, 2) bench(float* out, int iters, float seed) {
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  v2f p0 = {seed, seed}, p1 = p0 + 1.f, p2 = p0 + 2.f, p3 = p0 + 3.f, p4 = p0 + 4.f, p5 = p0 + 5.f, p6 = p0 + 6.f, p7 = p0 + 7.f;
  v2f c = {1.0001f, 0.9999f};
  float cs = 1.0001f;
  for (int i = 0; i < iters; ++i) {
    if constexpr (OP == 0) {  // v_fma_f32 x8 independent chains, 64 per iter
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                        "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cs));)
    } else if constexpr (OP == 1) {  // v_pk_fma_f32
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                        "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c));)
    } else if constexpr (OP == 2) {  // v_add_f32
      REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                        "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cs));)
    } else if constexpr (OP == 3) {  // v_pk_add_f32 with op_sel swap + neg
      REP8(asm volatile("v_pk_add_f32 %0, %0, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n"
                        "v_pk_add_f32 %2, %2, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %3, %3, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n"
                        "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c));)
    } else if constexpr (OP == 4) {  // v_log_f32
      REP8(asm volatile("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n"
                        "v_log_f32 %4, %4\n v_log_f32 %5, %5\n v_log_f32 %6, %6\n v_log_f32 %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if constexpr (OP == 5) {  // v_sqrt_f32
      REP8(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                        "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if constexpr (OP == 6) {  // v_cvt_f32_ubyte0
      REP8(asm volatile("v_cvt_f32_ubyte0 %0, %0\n v_cvt_f32_ubyte1 %1, %1\n v_cvt_f32_ubyte2 %2, %2\n v_cvt_f32_ubyte3 %3, %3\n"
                        "v_cvt_f32_ubyte0 %4, %4\n v_cvt_f32_ubyte1 %5, %5\n v_cvt_f32_ubyte2 %6, %6\n v_cvt_f32_ubyte3 %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if constexpr (OP == 7) {  // mixed: log interleaved with 3 fma
      REP8(asm volatile("v_log_f32 %0, %0\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                        "v_log_f32 %4, %4\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cs));)
    } else if constexpr (OP == 9) {  // v_fmac_f32 (VOP2): d += a*b
      REP8(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                        "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cs), "v"(seed));)
    } else if constexpr (OP == 10) {  // v_mul_f32
      REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                        "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cs));)
    } else if constexpr (OP == 11) {  // v_fma_f32 with three distinct sources
      REP8(asm volatile("v_fma_f32 %0, %1, %8, %2\n v_fma_f32 %1, %2, %8, %3\n v_fma_f32 %2, %3, %8, %4\n v_fma_f32 %3, %4, %8, %5\n"
                        "v_fma_f32 %4, %5, %8, %6\n v_fma_f32 %5, %6, %8, %7\n v_fma_f32 %6, %7, %8, %0\n v_fma_f32 %7, %0, %8, %1\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cs));)
    } else if constexpr (OP == 12) {  // v_max_f32
      REP8(asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
                        "v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(cs));)
    } else if constexpr (OP == 13) {  // v_pk_fma_f32 with op_sel / neg (complex multiply second half)
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %1, %1, %8, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n"
                        "v_pk_fma_f32 %2, %2, %8, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %3, %3, %8, %4 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n"
                        "v_pk_fma_f32 %4, %4, %8, %5 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %5, %5, %8, %6 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n"
                        "v_pk_fma_f32 %6, %6, %8, %7 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %7, %7, %8, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c));)
    } else if constexpr (OP == 14) {  // v_fma_f32 with an SGPR operand
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %1\n v_fma_f32 %1, %1, %8, %2\n v_fma_f32 %2, %2, %8, %3\n v_fma_f32 %3, %3, %8, %4\n"
                        "v_fma_f32 %4, %4, %8, %5\n v_fma_f32 %5, %5, %8, %6\n v_fma_f32 %6, %6, %8, %7\n v_fma_f32 %7, %7, %8, %0\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(seed));)
    } else if constexpr (OP == 8) {  // v_pk_mul_f32
      REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                        "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c));)
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

template <int OP>
void run(const char* name, float* d) {
  const int iters = 2000, grid = 256, block = 512;   // 2 waves per SIMD on every CU
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  bench<OP><<<grid, block>>>(d, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  bench<OP><<<grid, block>>>(d, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_wave = double(iters) * 64.0;
  const double waves_per_simd = 2.0;
  const double ns_per_instr = ms * 1e6 / (instr_per_wave * waves_per_simd);
  printf("%-34s %8.3f ms  %6.2f ns per wave-instr per SIMD  (= %.2f clk @2.4GHz, %.2f clk @2.0GHz)\n", name, ms,
         ns_per_instr, ns_per_instr * 2.4, ns_per_instr * 2.0);
}

int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  run<0>("v_fma_f32", d);
  run<1>("v_pk_fma_f32", d);
  run<2>("v_add_f32", d);
  run<3>("v_pk_add_f32 (op_sel/neg mods)", d);
  run<8>("v_pk_mul_f32", d);
  run<4>("v_log_f32", d);
  run<5>("v_sqrt_f32", d);
  run<6>("v_cvt_f32_ubyteN", d);
  run<7>("1 v_log : 3 v_fma", d);
  run<9>("v_fmac_f32 (VOP2)", d);
  run<10>("v_mul_f32", d);
  run<11>("v_fma_f32 distinct srcs", d);
  run<14>("v_fma_f32 v,s,v", d);
  run<12>("v_max_f32", d);
  run<13>("v_pk_fma_f32 op_sel+neg", d);
  return 0;
}
