// Practical HBM roof for the path's traffic shape: read 2 B per point (int8 IQ), write 4 B per point (float dB),
// no arithmetic beyond a convert.  Also pure read and pure write streams for reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_mix(const uint32_t* __restrict__ in, float2* __restrict__ out, size_t n_dw) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_dw; i += (size_t)gridDim.x * 256) {
    const uint32_t u = in[i];                                   // two IQ samples
    out[2 * i] = float2{float(u & 0xff), float((u >> 8) & 0xff)};
    out[2 * i + 1] = float2{float((u >> 16) & 0xff), float(u >> 24)};   // 4 B out per 1 B... (2 floats per sample pair)
  }
}
// same byte ratio as the path: 2 B in -> 4 B out per point
__global__ void __launch_bounds__(256) k_path(const uint32_t* __restrict__ in, float2* __restrict__ out, size_t n_dw) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_dw; i += (size_t)gridDim.x * 256) {
    const uint32_t u = in[i];                                   // 4 B = two points
    out[i] = float2{float(u & 0xffff), float(u >> 16)};         // 8 B = two points' dB values
  }
}
__global__ void __launch_bounds__(256) k_read(const uint4* __restrict__ in, uint32_t* out, size_t n) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint4 v = in[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_write(uint4* out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = uint4{1u, 2u, 3u, unsigned(i)};
}

template <class F> float timeit(F f, int reps) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); for (int r = 0; r < reps; ++r) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
int main() {
  const size_t pts = size_t(1) << 29;                   // 512 M points: 1 GiB in, 2 GiB out (>> 256 MiB MALL)
  void *in, *out;
  CK(hipMalloc(&in, pts * 2)); CK(hipMalloc(&out, pts * 4));
  CK(hipMemset(in, 1, pts * 2)); CK(hipMemset(out, 0, pts * 4));
  for (int grid : {2048, 8192}) {
    float ms = timeit([&] { k_path<<<grid, 256>>>((const uint32_t*)in, (float2*)out, pts / 2); }, 5);
    printf("path-shaped copy (2 B in + 4 B out per point), grid %5d: %.3f ms  %.2f TB/s  = %.3f Tpts/s\n", grid, ms, pts * 6 / ms / 1e9, pts / ms / 1e9);
    ms = timeit([&] { k_read<<<grid, 256>>>((const uint4*)out, (uint32_t*)in, pts * 4 / 16); }, 5);
    printf("pure read  16 B/lane, grid %5d: %.3f ms  %.2f TB/s\n", grid, ms, pts * 4 / ms / 1e9);
    ms = timeit([&] { k_write<<<grid, 256>>>((uint4*)out, pts * 4 / 16); }, 5);
    printf("pure write 16 B/lane, grid %5d: %.3f ms  %.2f TB/s\n", grid, ms, pts * 4 / ms / 1e9);
  }
  return 0;
}
