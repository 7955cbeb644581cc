#include <hip/hip_runtime.h>
#include <cstdio>
// DIST independent chains per wave: instruction i depends on instruction i-DIST.
template <int DIST, int OPK>
__global__ void __launch_bounds__(256) bench(float* out, int iters, float seed) {
  typedef float v2 __attribute__((ext_vector_type(2)));
  float a[8];
  v2 p[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = seed + j; p[j] = v2{seed + j, seed - j}; }
  v2 pc = {seed, seed * 0.5f};
  float c = seed * 1.0001f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 64; ++r) {
      float& x = a[r % DIST];
      if constexpr (OPK == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c));
      if constexpr (OPK == 1) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(x) : "v"(c));
      if constexpr (OPK == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
      v2& y = p[r % DIST];
      if constexpr (OPK == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y) : "v"(pc));
      if constexpr (OPK == 4) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(y) : "v"(pc));
      if constexpr (OPK == 5) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(y) : "v"(pc));
      if constexpr (OPK == 6) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(y) : "v"(pc));
      if constexpr (OPK == 7) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(c));
    }
  }
  float s = 0; for (int j = 0; j < 8; ++j) s += a[j] + p[j].x + p[j].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int DIST, int OPK>
void run(float* d, int wps) {   // wps waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
  const int iters = 2000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  bench<DIST, OPK><<<256 * wps, 256>>>(d, 10, 1.f); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); bench<DIST, OPK><<<256 * wps, 256>>>(d, iters, 1.f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const char* nm[] = {"v_add_f32", "v_fmac_f32", "v_fma_f32", "v_pk_add_f32", "v_pk_fma_f32", "pk_add swz", "pk_mul", "v_mul_f32"};
  printf("%-11s dist %d  waves/SIMD %d : %6.2f ns per wave-instr per wave, %5.2f ns per wave-instr per SIMD\n", nm[OPK], DIST, wps,
         ms * 1e6 / (iters * 64.0), ms * 1e6 / (iters * 64.0 * wps));
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
  for (int wps : {1, 2, 4, 8}) { run<1, 0>(d, wps); run<2, 0>(d, wps); run<4, 0>(d, wps); run<8, 0>(d, wps); }
  for (int wps : {1, 2, 4}) { run<1, 1>(d, wps); run<4, 1>(d, wps); run<1, 2>(d, wps); run<4, 2>(d, wps); }
  for (int wps : {1, 2, 4}) { run<1, 3>(d, wps); run<2, 3>(d, wps); run<4, 3>(d, wps); run<8, 3>(d, wps); run<1, 4>(d, wps); run<4, 4>(d, wps); run<8, 4>(d, wps); run<4, 5>(d, wps); run<4, 6>(d, wps); run<4, 7>(d, wps); run<8, 7>(d, wps);}
  return 0;
}
