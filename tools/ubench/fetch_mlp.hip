// How fast can ONE workgroup of 1024 threads per CU (the frame kernel's shape: 16 waves, most of the LDS) pull a stream
// of 128 KB rows, as the row pass of the long-frame chain does?  Per-CU fetch rate against loads in flight per thread,
// load flavour and working set (512 MiB: HBM, 128 MiB: Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned v4 __attribute__((ext_vector_type(4)));
template <int K, bool NT>
__global__ void __launch_bounds__(1024, 1) k_fetch(const v4* __restrict__ in, unsigned* out, size_t rows_total, int rows_per_wg) {
  extern __shared__ unsigned char smem[];
  // row r of this workgroup: 128 KB = 8192 v4; thread t reads v4 index i*1024 + t, i < 8; K rows kept in flight
  unsigned acc = 0;
  const size_t base_row = (size_t)blockIdx.x * rows_per_wg;
  v4 buf[K][8];
  for (int k = 0; k < K; ++k)
    for (int i = 0; i < 8; ++i) {
      const v4* p = in + ((base_row + k) % rows_total) * 8192 + i * 1024 + threadIdx.x;
      buf[k][i] = NT ? __builtin_nontemporal_load(p) : *p;
    }
  for (int r = 0; r < rows_per_wg; r += K) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc ^= buf[k][i].x ^ buf[k][i].y ^ buf[k][i].z ^ buf[k][i].w;
      const size_t nr = (base_row + r + K + k) % rows_total;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const v4* p = in + nr * 8192 + i * 1024 + threadIdx.x;
        buf[k][i] = NT ? __builtin_nontemporal_load(p) : *p;
      }
    }
  }
  if (acc == 0x12345678u) out[0] = acc + smem[0];
}
template <int K, bool NT> void run(const char* name, const v4* buf, unsigned* out, size_t mib) {
  const size_t rows_total = (mib << 20) / (128 << 10);
  const int rows_per_wg = 64;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipFuncSetAttribute((const void*)k_fetch<K, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 139520);
  for (int w = 0; w < 2; ++w) k_fetch<K, NT><<<256, 1024, 139520>>>(buf, out, rows_total, rows_per_wg);
  (void)hipEventRecord(e0);
  for (int w = 0; w < 5; ++w) k_fetch<K, NT><<<256, 1024, 139520>>>(buf, out, rows_total, rows_per_wg);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double bytes = 256.0 * rows_per_wg * 131072.0;
  printf("%-34s %5zu MiB: %7.3f ms  %6.2f TB/s  %6.1f B/ns per CU  %5.2f us per 128 KB row\n", name, mib, ms, bytes / ms / 1e9, bytes / 256 / (ms * 1e6), ms * 1e3 / rows_per_wg);
}
int main() {
  void *buf, *out;
  if (hipMalloc(&buf, size_t(2) << 30) != hipSuccess || hipMalloc(&out, 4096) != hipSuccess) return 1;
  (void)hipMemset(buf, 1, size_t(2) << 30);
  for (size_t mib : {2048, 128}) {
    run<1, true>("1 row in flight, non-temporal", (const v4*)buf, (unsigned*)out, mib);
    run<1, false>("1 row in flight, plain", (const v4*)buf, (unsigned*)out, mib);
    run<2, true>("2 rows in flight, non-temporal", (const v4*)buf, (unsigned*)out, mib);
    run<2, false>("2 rows in flight, plain", (const v4*)buf, (unsigned*)out, mib);
  }
  return 0;
}
