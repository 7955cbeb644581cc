// Microbenchmark: scalar (tdsa_fft.hpp) vs packed-fp32 (tdsa_fft_pk.hpp) radix-16 pass body in registers:
// 16 twiddle products + radix-16 DIF per iteration, 4 waves per SIMD (1024 threads, 1 block per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "tdsa_fft_pk.hpp"
using namespace tdsa;

__device__ __forceinline__ void opq(c32& w) { asm volatile("" : "+v"(w.x), "+v"(w.y)); }
__device__ __forceinline__ void opq(p32& w) { asm volatile("" : "+v"(w)); }

template <bool TW>
__global__ void __launch_bounds__(1024, 1) k_scalar(const c32* in, c32* out, int iters) {
  c32 v[16], w[16];
  for (int i = 0; i < 16; ++i) { v[i] = in[threadIdx.x + 1024 * i]; w[i] = in[threadIdx.x + 7 + 1024 * i]; }
  for (int it = 0; it < iters; ++it) {
    if (TW) static_for<0, 16>([&](auto ic) { constexpr int i = decltype(ic)::value; opq(w[i]); v[i] = cmul(v[i], w[i]); });
    dif<16, 0, 16>(v);
    static_for<0, 16>([&](auto ic) { opq(v[decltype(ic)::value]); });
  }
  for (int i = 0; i < 16; ++i) out[(blockIdx.x * 1024 + threadIdx.x) * 16 + i] = v[i];
}
template <bool TW>
__global__ void __launch_bounds__(1024, 1) k_packed(const c32* in, c32* out, int iters) {
  p32 v[16], w[16];
  for (int i = 0; i < 16; ++i) { v[i] = to_p(in[threadIdx.x + 1024 * i]); w[i] = to_p(in[threadIdx.x + 7 + 1024 * i]); }
  for (int it = 0; it < iters; ++it) {
    if (TW) static_for<0, 16>([&](auto ic) { constexpr int i = decltype(ic)::value; opq(w[i]); v[i] = pmul(v[i], w[i]); });
    difp<16, 0, 16>(v);
    static_for<0, 16>([&](auto ic) { opq(v[decltype(ic)::value]); });
  }
  for (int i = 0; i < 16; ++i) out[(blockIdx.x * 1024 + threadIdx.x) * 16 + i] = to_c(v[i]);
}
// straight-line body of REP passes (code size ~ REP x 1.8 KB), 16 waves per CU, all at their own PC
template <int REP>
__global__ void __launch_bounds__(1024, 1) k_big(const c32* in, c32* out, int iters) {
  c32 v[16], w[16];
  for (int i = 0; i < 16; ++i) { v[i] = in[threadIdx.x + 1024 * i]; w[i] = in[threadIdx.x + 7 + 1024 * i]; }
  // de-phase the waves so that they do not walk the code in lock step
  for (int d = 0; d < (threadIdx.x >> 6) * 37; ++d) asm volatile("s_nop 7");
  for (int it = 0; it < iters; ++it) {
    static_for<0, REP>([&](auto rc) {
      static_for<0, 16>([&](auto ic) { constexpr int i = decltype(ic)::value; opq(w[i]); v[i] = cmul(v[i], w[i]); });
      dif<16, 0, 16>(v);
      static_for<0, 16>([&](auto ic) { opq(v[decltype(ic)::value]); });
    });
  }
  for (int i = 0; i < 16; ++i) out[(blockIdx.x * 1024 + threadIdx.x) * 16 + i] = v[i];
}
// correctness: one pass of each on the same data
__global__ void k_check(const c32* in, float* err) {
  c32 a[16]; p32 b[16];
  for (int i = 0; i < 16; ++i) { a[i] = in[threadIdx.x * 16 + i]; b[i] = to_p(a[i]); }
  const c32 w = in[threadIdx.x + 4096];
  static_for<0, 16>([&](auto ic) { constexpr int i = decltype(ic)::value; a[i] = cmul(a[i], w); b[i] = pmul(b[i], to_p(w)); });
  dif<16, 0, 16>(a); difp<16, 0, 16>(b);
  float e = 0.f, m = 0.f;
  for (int i = 0; i < 16; ++i) { e = fmaxf(e, fmaxf(fabsf(a[i].x - b[i].x), fabsf(a[i].y - b[i].y))); m = fmaxf(m, fabsf(a[i].x)); }
  err[threadIdx.x] = e / m;
}

template <class F> float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(2); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); f(4000); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  c32 *in, *out; float* err;
  (void)hipMalloc(&in, 1 << 22); (void)hipMalloc(&out, 256 * 1024 * 16 * 8); (void)hipMalloc(&err, 4096);
  float* h = (float*)malloc(1 << 22);
  for (int i = 0; i < (1 << 20); ++i) h[i] = float((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  (void)hipMemcpy(in, h, 1 << 22, hipMemcpyHostToDevice);
  k_check<<<1, 256>>>(in, err);
  float he[256]; (void)hipMemcpy(he, err, 1024, hipMemcpyDeviceToHost);
  float me = 0; for (int i = 0; i < 256; ++i) me = he[i] > me ? he[i] : me;
  printf("max rel diff packed vs scalar radix-16: %.3g\n", me);
  float ms;
  ms = timeit([&](int it) { k_scalar<false><<<256, 1024>>>(in, out, it); });
  printf("scalar dif<16>          : %.3f ms -> %6.1f ns per radix-16 per SIMD-wave-slot (x4 waves)\n", ms, ms * 1e6 / 4000 / 4);
  ms = timeit([&](int it) { k_packed<false><<<256, 1024>>>(in, out, it); });
  printf("packed difp<16>         : %.3f ms -> %6.1f ns\n", ms, ms * 1e6 / 4000 / 4);
  ms = timeit([&](int it) { k_scalar<true><<<256, 1024>>>(in, out, it); });
  printf("scalar 16 cmul + dif<16>: %.3f ms -> %6.1f ns\n", ms, ms * 1e6 / 4000 / 4);
  ms = timeit([&](int it) { k_packed<true><<<256, 1024>>>(in, out, it); });
  printf("packed 16 pmul + difp   : %.3f ms -> %6.1f ns\n", ms, ms * 1e6 / 4000 / 4);
  ms = timeit([&](int it) { k_big<1><<<256, 1024>>>(in, out, it); });
  printf("straight-line x1  (%4.1f KB): %.3f ms -> %6.1f ns per pass\n", 1.8, ms, ms * 1e6 / 4000 / 4);
  ms = timeit([&](int it) { k_big<4><<<256, 1024>>>(in, out, it / 4); });
  printf("straight-line x4  (%4.1f KB): %.3f ms -> %6.1f ns per pass\n", 7.2, ms, ms * 1e6 / 4000 / 4);
  ms = timeit([&](int it) { k_big<8><<<256, 1024>>>(in, out, it / 8); });
  printf("straight-line x8  (%4.1f KB): %.3f ms -> %6.1f ns per pass\n", 14.4, ms, ms * 1e6 / 4000 / 4);
  ms = timeit([&](int it) { k_big<16><<<256, 1024>>>(in, out, it / 16); });
  printf("straight-line x16 (%4.1f KB): %.3f ms -> %6.1f ns per pass\n", 28.8, ms, ms * 1e6 / 4000 / 4);
  ms = timeit([&](int it) { k_big<32><<<256, 1024>>>(in, out, it / 32); });
  printf("straight-line x32 (%4.1f KB): %.3f ms -> %6.1f ns per pass\n", 57.6, ms, ms * 1e6 / 4000 / 4);
  return 0;
}
