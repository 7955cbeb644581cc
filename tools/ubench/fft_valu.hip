// Microbenchmark: the compiler-generated radix butterfly code in isolation (registers only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../topdogspectrumanalyser_amd/csrc/tdsa_fft.hpp"
using namespace tdsa;

__device__ __forceinline__ void opaque2(c32& w) { asm volatile("" : "+v"(w.x), "+v"(w.y)); }

template <int VARIANT>
__global__ void __launch_bounds__(512, 2) k32(const c32* in, c32* out, int iters) {
  c32 v[32];
  for (int i = 0; i < 32; ++i) v[i] = in[threadIdx.x + 512 * i];
  c32 lo[3], hi[7];
  for (int i = 0; i < 3; ++i) lo[i] = in[threadIdx.x + i];
  for (int i = 0; i < 7; ++i) hi[i] = in[threadIdx.x + 8 + i];
  for (int it = 0; it < iters; ++it) {
    if (VARIANT >= 1) {
      static_for<0, 3>([&](auto ic) { opaque2(lo[decltype(ic)::value]); });
      static_for<0, 7>([&](auto ic) { opaque2(hi[decltype(ic)::value]); });
      twiddle32(v, lo, hi);
    }
    dif<32, 0, 32>(v);
    static_for<0, 32>([&](auto ic) { opaque2(v[decltype(ic)::value]); });
  }
  for (int i = 0; i < 32; ++i) out[(blockIdx.x * 512 + threadIdx.x) * 32 + i] = v[i];
}

template <int WPS>
__global__ void __launch_bounds__(256 * WPS, WPS) k16(const c32* in, c32* out, int iters) {
  c32 v[16];
  for (int i = 0; i < 16; ++i) v[i] = in[threadIdx.x + 512 * i];
  for (int it = 0; it < iters; ++it) {
    dif<16, 0, 16>(v);
    static_for<0, 16>([&](auto ic) { opaque2(v[decltype(ic)::value]); });
  }
  for (int i = 0; i < 16; ++i) out[(blockIdx.x * 1024 + threadIdx.x) * 16 + i] = v[i];
}

// body of increasing size: REP copies of (twiddle32 + dif<32>) per loop iteration, straight-line
template <int REP>
__global__ void __launch_bounds__(512, 2) kbig(const c32* in, c32* out, int iters) {
  c32 v[32];
  for (int i = 0; i < 32; ++i) v[i] = in[threadIdx.x + 512 * i];
  c32 lo[3], hi[7];
  for (int i = 0; i < 3; ++i) lo[i] = in[threadIdx.x + i];
  for (int i = 0; i < 7; ++i) hi[i] = in[threadIdx.x + 8 + i];
  for (int it = 0; it < iters; ++it) {
    static_for<0, REP>([&](auto rc) {
      static_for<0, 3>([&](auto ic) { opaque2(lo[decltype(ic)::value]); });
      static_for<0, 7>([&](auto ic) { opaque2(hi[decltype(ic)::value]); });
      twiddle32(v, lo, hi);
      dif<32, 0, 32>(v);
      static_for<0, 32>([&](auto ic) { opaque2(v[decltype(ic)::value]); });
    });
  }
  for (int i = 0; i < 32; ++i) out[(blockIdx.x * 512 + threadIdx.x) * 32 + i] = v[i];
}

template <class F>
float timeit(F f) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(2);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  f(2000);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  c32 *in, *out;
  (void)hipMalloc(&in, 1 << 22); (void)hipMalloc(&out, 256 * 1024 * 32 * 8);
  (void)hipMemset(in, 0x3c, 1 << 22);
  float ms;
  ms = timeit([&](int it) { k32<0><<<256, 512>>>(in, out, it); });
  printf("dif<32> only, 2 waves/SIMD          : %.3f ms  -> %.1f ns per radix-32 per wave\n", ms, ms * 1e6 / 2000);
  ms = timeit([&](int it) { k32<1><<<256, 512>>>(in, out, it); });
  printf("twiddle32 + dif<32>, 2 waves/SIMD   : %.3f ms  -> %.1f ns per pass per wave\n", ms, ms * 1e6 / 2000);
  ms = timeit([&](int it) { k16<2><<<256, 512>>>(in, out, it); });
  printf("dif<16>, 2 waves/SIMD               : %.3f ms  -> %.1f ns per radix-16 per wave\n", ms, ms * 1e6 / 2000);
  ms = timeit([&](int it) { k16<4><<<256, 1024>>>(in, out, it); });
  printf("dif<16>, 4 waves/SIMD               : %.3f ms  -> %.1f ns per radix-16 per wave (x2 waves)\n", ms, ms * 1e6 / 2000);
  ms = timeit([&](int it) { kbig<1><<<256, 512>>>(in, out, it); });
  printf("kbig<1> (1 pass / iter)             : %.3f ms  -> %.1f ns per pass per wave\n", ms, ms * 1e6 / 2000);
  ms = timeit([&](int it) { kbig<2><<<256, 512>>>(in, out, it / 2); });
  printf("kbig<2> (2 passes / iter)           : %.3f ms  -> %.1f ns per pass per wave\n", ms, ms * 1e6 / 2000);
  ms = timeit([&](int it) { kbig<4><<<256, 512>>>(in, out, it / 4); });
  printf("kbig<4> (4 passes / iter)           : %.3f ms  -> %.1f ns per pass per wave\n", ms, ms * 1e6 / 2000);
  ms = timeit([&](int it) { kbig<8><<<256, 512>>>(in, out, it / 8); });
  printf("kbig<8> (8 passes / iter)           : %.3f ms  -> %.1f ns per pass per wave\n", ms, ms * 1e6 / 2000);
  ms = timeit([&](int it) { kbig<16><<<256, 512>>>(in, out, it / 16); });
  printf("kbig<16> (16 passes / iter)         : %.3f ms  -> %.1f ns per pass per wave\n", ms, ms * 1e6 / 2000);
  return 0;
}
