// Write / read bandwidth against working-set size: does the 256 MiB Infinity Cache take a write stream (the long-frame
// path's Z array) faster than HBM does?  Each kernel sweeps the same buffer `reps` times back to back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void __launch_bounds__(256) k_read(const uint4* __restrict__ in, uint32_t* out, size_t n) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint4 v = in[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_write(uint4* out, size_t n, unsigned tag) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = uint4{1u, 2u, tag, unsigned(i)};
}
__global__ void __launch_bounds__(256) k_write_nt(uint4* out, size_t n, unsigned tag) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { typedef unsigned v4 __attribute__((ext_vector_type(4))); __builtin_nontemporal_store(v4{1u, 2u, tag, unsigned(i)}, reinterpret_cast<v4*>(&out[i])); }
}
template <class F> float timeit(F f, int reps) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) f(r); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); for (int r = 0; r < reps; ++r) f(r); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
int main() {
  void *buf, *small;
  if (hipMalloc(&buf, size_t(4) << 30) != hipSuccess || hipMalloc(&small, 4096) != hipSuccess) return 1;
  (void)hipMemset(buf, 1, size_t(4) << 30);
  for (size_t mib : {32, 64, 128, 192, 256, 384, 512, 1024, 4096}) {
    const size_t n = (mib << 20) / 16;
    const int reps = mib <= 512 ? 40 : 8;
    const float w = timeit([&](int r) { k_write<<<4096, 256>>>((uint4*)buf, n, r); }, reps);
    const float wn = timeit([&](int r) { k_write_nt<<<4096, 256>>>((uint4*)buf, n, r); }, reps);
    const float rd = timeit([&](int) { k_read<<<4096, 256>>>((const uint4*)buf, (uint32_t*)small, n); }, reps);
    const float wr = timeit([&](int r) { k_write<<<4096, 256>>>((uint4*)buf, n, r); k_read<<<4096, 256>>>((const uint4*)buf, (uint32_t*)small, n); }, reps);
    printf("%5zu MiB: write %6.2f TB/s | write nt %6.2f TB/s | read %6.2f TB/s | write then read %6.2f TB/s (bytes moved / time)\n",
           mib, (mib << 20) / w / 1e9, (mib << 20) / wn / 1e9, (mib << 20) / rd / 1e9, 2.0 * (mib << 20) / wr / 1e9);
  }
  return 0;
}
