// Microbenchmark: do VALU work and LDS traffic of DIFFERENT waves of one CU overlap on gfx950?
// 16 waves per workgroup (4 per SIMD), one workgroup per CU.  mode 0: all waves run the VALU loop,
// mode 1: all waves run the LDS loop, mode 2: even waves VALU / odd waves LDS (half the work of each).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../topdogspectrumanalyser_amd/csrc/tdsa_fft.hpp"
using namespace tdsa;
__device__ __forceinline__ void opq(c32& w) { asm volatile("" : "+v"(w.x), "+v"(w.y)); }

__global__ void __launch_bounds__(1024, 4) k(const c32* in, c32* out, int iters, int mode) {
  extern __shared__ c32 lds[];
  const int tid = threadIdx.x, wave = tid >> 6;
  c32 v[16];
  for (int i = 0; i < 16; ++i) v[i] = in[tid + 1024 * i];
  const bool do_valu = mode == 0 || (mode == 2 && (wave & 1) == 0) || (mode == 3 && (wave & 4) == 0);
  const bool do_lds = mode == 1 || (mode == 2 && (wave & 1) == 1) || (mode == 3 && (wave & 4) != 0);
  for (int it = 0; it < iters; ++it) {
    if (do_valu) {
      dif<16, 0, 16>(v); dif<16, 0, 16>(v);
      static_for<0, 16>([&](auto ic) { opq(v[decltype(ic)::value]); });
    }
    if (do_lds) {
      static_for<0, 16>([&](auto ic) { constexpr int i = decltype(ic)::value; lds[33 * (tid >> 1) * 0 + tid * 17 + i] = v[i]; });
      __builtin_amdgcn_wave_barrier();
      static_for<0, 16>([&](auto ic) { constexpr int i = decltype(ic)::value; v[i] = lds[tid + i * 1025]; });
      static_for<0, 16>([&](auto ic) { opq(v[decltype(ic)::value]); });
    }
  }
  for (int i = 0; i < 16; ++i) out[(blockIdx.x * 1024 + tid) * 16 + i] = v[i];
}

int main() {
  c32 *in, *out;
  (void)hipMalloc(&in, 1 << 22); (void)hipMalloc(&out, 256 * 1024 * 16 * 8);
  (void)hipMemset(in, 0, 1 << 22);
  const size_t ldsb = 139264;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, int(ldsb));
  const char* names[] = {"all 16 waves VALU (2 x dif<16> / iter)", "all 16 waves LDS (16 st + 16 ld b64 / iter)",
                         "even waves VALU, odd waves LDS", "waves 0-3,8-11 VALU, 4-7,12-15 LDS"};
  for (int mode = 0; mode < 4; ++mode) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<<<256, 1024, ldsb>>>(in, out, 10, mode);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<<<256, 1024, ldsb>>>(in, out, 2000, mode);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.3f ms  (%.1f ns / iteration)\n", names[mode], ms, ms * 1e6 / 2000);
  }
  return 0;
}
