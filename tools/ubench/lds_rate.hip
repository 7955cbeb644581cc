// LDS pipe rates on gfx950: bytes/clk/CU for the access shapes the frame kernel uses.
// 1024-thread workgroup (16 waves), one per CU, every wave issues REP back-to-back LDS ops.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(1024) void k(float* out, int iters, int stride_b) {
  extern __shared__ char lds[];
  const int t = threadIdx.x;
  // lane address: wave w owns a 8 KiB-ish region; lanes at `stride_b` bytes
  unsigned a = (unsigned)((t >> 6) * 8448 + (t & 63) * stride_b);
  f2 v = {(float)t, 1.f}; f4 q = {(float)t, 1.f, 2.f, 3.f};
  float acc = 0.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      unsigned ar = a + (r & 3) * ((OP == 2 || OP == 3) ? 16 : 8);
      if constexpr (OP == 0) { asm volatile("ds_write_b64 %0, %1" :: "v"(ar), "v"(v) : "memory"); }
      if constexpr (OP == 1) { f2 x; asm volatile("ds_read_b64 %0, %1" : "=v"(x) : "v"(ar) : "memory"); asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); acc += 0.f; (void)x; }
      if constexpr (OP == 2) { f4 x; asm volatile("ds_read_b128 %0, %1" : "=v"(x) : "v"(ar) : "memory"); asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); (void)x; }
      if constexpr (OP == 3) { asm volatile("ds_write_b128 %0, %1" :: "v"(ar), "v"(q) : "memory"); }
      if constexpr (OP == 4) { f4 x; asm volatile("ds_read2_b64 %0, %1 offset1:33" : "=v"(x) : "v"(ar) : "memory"); asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); (void)x; }
      if constexpr (OP == 5) { asm volatile("ds_write2_b64 %0, %1, %2 offset1:33" :: "v"(ar), "v"(v), "v"(v) : "memory"); }
      if constexpr (OP == 6) { float x; asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"(ar) : "memory"); asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); (void)x; }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (acc == 123.f) out[t] = acc;
}

template <int OP>
int run(const char* name, int bytes_per_lane, int stride_b, float* d) {
  const int iters = 2000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute((const void*)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  k<OP><<<256, 1024, 150 * 1024>>>(d, 10, stride_b);
  CK(hipEventRecord(e0));
  k<OP><<<256, 1024, 150 * 1024>>>(d, iters, stride_b);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double bytes = 16.0 * 64 * bytes_per_lane * 16.0 * iters;      // per CU
  printf("%-28s stride %4d B  %7.3f ms  %6.1f B/ns/CU  (%.2f ns per wave-instr per CU)\n", name, stride_b, ms,
         bytes / (ms * 1e6), ms * 1e6 / (16.0 * 16 * iters));
  return 0;
}

int main() {
  float* d; CK(hipMalloc(&d, 1 << 20));
  for (int s : {8, 264}) {
    run<0>("ds_write_b64", 8, s, d);
    run<1>("ds_read_b64", 8, s, d);
    run<4>("ds_read2_b64", 16, s, d);
    run<5>("ds_write2_b64", 16, s, d);
  }
  for (int s : {16, 272, 528, 48}) { run<2>("ds_read_b128", 16, s, d); run<3>("ds_write_b128", 16, s, d); }
  run<6>("ds_read_b32", 4, 4, d);
  return 0;
}
