// r16_proto.hip - prototype of a 16384-point frame kernel WITHOUT the half-thread trick:
//   N = 16 x 16 x 16 x 4, 16 points per thread, 1024 threads per frame, every radix pass entirely in one lane
//   (no v_permlane32_swap, no selects), three exchanges through LDS of which only two cross waves:
//     n = 1024 a + 64 b + 4 c + d ,  k = ka + 16 kb + 256 kc + 4096 kd
//     P1: thread j = 64 b + 4 c + d    : DFT16 over a, * W_256^(b ka)            -> E1[ka][j]        (all-to-all)
//     P2: wave ka, lane 4c + d         : DFT16 over b                            -> E2 (own block)
//     P3: wave ka, lane 4kb + d        : * W_4096^(c (ka + 16 kb)), DFT16 over c -> E3 (own block)
//     P4: thread t, m = t + 1024 j     : * W_N^(d m), DFT4 over d                -> bins m + 4096 kd
// Stand-alone: builds a .so with r16_run() for tools/ubench/r16_check.py (parity against numpy + timing).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "../../topdogspectrumanalyser_amd/csrc/tdsa_fft.hpp"

using namespace tdsa;

namespace {
constexpr int N = 16384, NT = 1024, BS = 1090;             // BS = 2 (mod 32): see the E3 read pattern
constexpr size_t LDS_BYTES = size_t(16) * BS * 8 + 256 * 8 + 64 * 4;
constexpr float k10Log10_2 = 3.01029995663981195214f;

typedef float lds_v2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) volatile lds_v2* lds_vptr;
__device__ __forceinline__ c32 lds_ld(const c32* p) { const lds_v2 x = *(lds_vptr)(p); return c32{x.x, x.y}; }
__device__ __forceinline__ void lds_st(c32* p, c32 v) { *(lds_vptr)(p) = lds_v2{v.x, v.y}; }
__device__ __forceinline__ float hw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ void opaque(c32& w) { asm volatile("" : "+v"(w.x), "+v"(w.y)); }
__device__ __forceinline__ int dpp_wave_sum(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
  return x;
}
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, int(bytes), 0x00020000);
}

struct P {
  const unsigned char* in;   // int8 interleaved
  long long frame_stride;    // bytes
  int n_frames;
  const float* window;       // [N] window / 128
  const float2* tw;          // [N] W_N^m
  float* out_db;             // [F][N] fftshift-ed
  float* hold_max;           // [N]
  float cal_db;
};

__global__ void __launch_bounds__(NT, 4) r16_kernel(const P p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  c32* buf = reinterpret_cast<c32*>(smem);
  c32* t1 = reinterpret_cast<c32*>(smem + size_t(16) * BS * 8);          // W_256^m
  int* redi = reinterpret_cast<int*>(smem + size_t(16) * BS * 8 + 256 * 8);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int u0 = int((long long)blockIdx.x * p.n_frames / gridDim.x);
  const int u1 = int((long long)(blockIdx.x + 1) * p.n_frames / gridDim.x);

  if (tid < 256) t1[tid] = p.tw[tid * 64];
  // P3 pre-twiddle seeds: W_4096^(c e), e = ka + 16 kb, c = 4a + j -> hi[a] = W^(4 a e) (a = 1..3), lo[j] = W^(j e)
  const int ka3 = wave, kb3 = lane >> 2, d3 = lane & 3;
  const int e3 = ka3 + 16 * kb3;
  c32 s3lo[3], s3hi[3];
  static_for<0, 3>([&](auto ic) { constexpr int j = decltype(ic)::value + 1; s3lo[j - 1] = p.tw[((j * e3) & 4095) * 4]; });
  static_for<0, 3>([&](auto ic) { constexpr int a = decltype(ic)::value + 1; s3hi[a - 1] = p.tw[((4 * a * e3) & 4095) * 4]; });
  // P4 pre-twiddle seeds W_N^(d t), d = 1..3
  c32 s4[3];
  static_for<0, 3>([&](auto ic) { constexpr int d = decltype(ic)::value + 1; s4[d - 1] = p.tw[(d * tid) & (N - 1)]; });
  float hmax[16];
  static_for<0, 16>([&](auto ic) { hmax[decltype(ic)::value] = -INFINITY; });
  float cal_v = p.cal_db; asm volatile("" : "+v"(cal_v));

  // addresses (complex elements)
  const int e1w = tid;                                   // + ka * BS
  const int e1r = wave * BS + lane;                      // + b * 64
  const int e2w = wave * BS + lane;                      // + kb * 68
  const int e2r = wave * BS + kb3 * 68 + d3;             // + 4 c
  const int e3w = wave * BS + (kb3 & 3) + 4 * d3 + 16 * (kb3 >> 2);                    // + 64 kc
  const int ka4 = tid & 15, kb4 = (tid >> 4) & 15, kcl4 = tid >> 8;
  const int e3r = ka4 * BS + (kb4 & 3) + 16 * (kb4 >> 2) + 64 * kcl4;                    // + 4 d + 256 j

  const rsrc_t win_rsrc = make_rsrc(p.window, N * 4u);
  const unsigned col_off = unsigned(tid) * 2u;
  uint32_t raw[16];
  auto load_raw = [&](int frame) {
    const rsrc_t r = make_rsrc(p.in + (long long)frame * p.frame_stride, N * 2u);
    static_for<0, 16>([&](auto ac) {
      constexpr int a = decltype(ac)::value;
      raw[a] = __builtin_amdgcn_raw_buffer_load_b16(r, col_off, a * 2048u, 0);
    });
  };
  float win[16];
  auto load_window = [&] {
    static_for<0, 16>([&](auto ac) {
      constexpr int a = decltype(ac)::value;
      win[a] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(win_rsrc, unsigned(tid) * 4u, a * 4096u, 0));
    });
  };
  if (u0 < u1) load_raw(u0);
  load_window();
  __syncthreads();

  for (int frame = u0; frame < u1; ++frame) {
    // ---- DC sums (exact integers), wave reduce, publish ----
    static_for<0, 16>([&](auto ac) { raw[decltype(ac)::value] ^= 0x8080u; });
    unsigned si = 0, sq = 0;
    static_for<0, 16>([&](auto ac) {
      constexpr int a = decltype(ac)::value;
      si = __builtin_amdgcn_udot4(raw[a], 0x00000001u, si, false);
      sq = __builtin_amdgcn_udot4(raw[a], 0x00000100u, sq, false);
    });
    const int wi = dpp_wave_sum(int(si)), wq = dpp_wave_sum(int(sq));
    if (lane == 63) *reinterpret_cast<int2*>(&redi[wave * 2]) = int2{wi, wq};
    __syncthreads();                                      // B0: also WAR for E1 against the previous P4 reads
    const int2 q = *reinterpret_cast<const int2*>(&redi[(lane & 15) * 2]);
    int ti = q.x, tq = q.y;
    ti += __builtin_amdgcn_update_dpp(0, ti, 0xB1, 0xf, 0xf, true);  tq += __builtin_amdgcn_update_dpp(0, tq, 0xB1, 0xf, 0xf, true);
    ti += __builtin_amdgcn_update_dpp(0, ti, 0x4E, 0xf, 0xf, true);  tq += __builtin_amdgcn_update_dpp(0, tq, 0x4E, 0xf, 0xf, true);
    ti += __builtin_amdgcn_update_dpp(0, ti, 0x141, 0xf, 0xf, true); tq += __builtin_amdgcn_update_dpp(0, tq, 0x141, 0xf, 0xf, true);
    ti += __builtin_amdgcn_update_dpp(0, ti, 0x140, 0xf, 0xf, true); tq += __builtin_amdgcn_update_dpp(0, tq, 0x140, 0xf, 0xf, true);
    float sub_re = float(ti) * (1.0f / N), sub_im = float(tq) * (1.0f / N);
    asm volatile("" : "+v"(sub_re), "+v"(sub_im));

    // ---- unpack + DC + window; P1 ----
    c32 v[16];
    static_for<0, 16>([&](auto ac) {
      constexpr int a = decltype(ac)::value;
      const uint32_t u = raw[a];
      v[a] = c32{(float(u & 0xffu) - sub_re) * win[a], (float((u >> 8) & 0xffu) - sub_im) * win[a]};
    });
    if (frame + 1 < u1) load_raw(frame + 1);
    dit<16, 0, 16>(v);
    {
      const int b = wave;                                  // W_256^(b ka): wave-uniform index, broadcast reads
      static_for<0, 16>([&](auto kc_) {
        constexpr int ka = decltype(kc_)::value;
        c32 x = v[bitrev(ka, 4)];
        if constexpr (ka > 0) x = cmul(x, lds_ld(&t1[(b * ka) & 255]));
        lds_st(&buf[ka * BS + e1w], x);
      });
    }
    __syncthreads();                                      // B1
    // ---- P2 (wave ka) ----
    static_for<0, 16>([&](auto bc) { constexpr int b = decltype(bc)::value; v[b] = lds_ld(&buf[e1r + b * 64]); });
    dit<16, 0, 16>(v);
    static_for<0, 16>([&](auto kc_) { constexpr int kb = decltype(kc_)::value; lds_st(&buf[e2w + kb * 68], v[bitrev(kb, 4)]); });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- P3 (wave ka) ----
    static_for<0, 16>([&](auto cc) { constexpr int c = decltype(cc)::value; v[c] = lds_ld(&buf[e2r + 4 * c]); });
    static_for<0, 3>([&](auto ic) { opaque(s3lo[decltype(ic)::value]); opaque(s3hi[decltype(ic)::value]); });
    static_for<1, 16>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      constexpr int a = c >> 2, j = c & 3;
      if constexpr (a == 0) v[c] = cmul(v[c], s3lo[j - 1]);
      else if constexpr (j == 0) v[c] = cmul(v[c], s3hi[a - 1]);
      else v[c] = cmul(v[c], cmul(s3hi[a - 1], s3lo[j - 1]));
    });
    dit<16, 0, 16>(v);
    static_for<0, 16>([&](auto kc_) { constexpr int kc = decltype(kc_)::value; lds_st(&buf[e3w + 64 * kc], v[bitrev(kc, 4)]); });
    __syncthreads();                                      // B3
    // ---- P4: four radix-4 butterflies, m = tid + 1024 j ----
    static_for<0, 4>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      static_for<0, 4>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        v[4 * j + d] = lds_ld(&buf[e3r + 4 * d + 256 * j]);
      });
    });
    static_for<0, 3>([&](auto ic) { opaque(s4[decltype(ic)::value]); });
    static_for<0, 4>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      static_for<1, 4>([&](auto dc) {
        constexpr int d = decltype(dc)::value;                 // W_N^(d (t + 1024 j)) = s4[d-1] * W_16^(d j)
        v[4 * j + d] = cmul(v[4 * j + d], mul_w<d * j, 16>(s4[d - 1]));
      });
      // radix-4 DIT on v[4j .. 4j+3] (natural input order d): outputs kd in bit-reversed registers
      bf_w<0, 2>(v[4 * j], v[4 * j + 2]);
      bf_w<0, 2>(v[4 * j + 1], v[4 * j + 3]);
      bf_w<0, 4, true>(v[4 * j], v[4 * j + 1]);
      bf_w<1, 4, true>(v[4 * j + 2], v[4 * j + 3]);
    });
    // registers: X[m + 4096 kd] at v[4j + bitrev(kd, 2)]
    // ---- epilogue ----
    float db[16];
    static_for<0, 16>([&](auto qc) {
      constexpr int qq = decltype(qc)::value;               // qq = j + 4 kd  -> bin t + 1024 qq
      constexpr int j = qq & 3, kd = qq >> 2;
      const c32 X = v[4 * j + bitrev(kd, 2)];
      db[qq] = fmaf(k10Log10_2, __builtin_amdgcn_logf(X.x * X.x + X.y * X.y), cal_v);
    });
    load_window();
    {
      const rsrc_t r = make_rsrc(p.out_db + (long long)frame * N, N * 4u);
      static_for<0, 16>([&](auto qc) {
        constexpr int qq = decltype(qc)::value;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(db[qq]), r, unsigned(tid) * 4u, ((qq ^ 8) * 1024u) * 4u, 0);
      });
    }
    static_for<0, 16>([&](auto qc) { constexpr int qq = decltype(qc)::value; hmax[qq] = hw_max(hmax[qq], db[qq]); });
  }
  if (p.hold_max != nullptr && u1 > u0) {
    static_for<0, 16>([&](auto qc) {
      constexpr int qq = decltype(qc)::value;
      float* a = p.hold_max + tid + (qq ^ 8) * 1024;
      if (hmax[qq] > *a) {
        if (hmax[qq] >= 0.f) atomicMax(reinterpret_cast<int*>(a), __float_as_int(hmax[qq]));
        else atomicMin(reinterpret_cast<unsigned*>(a), __float_as_uint(hmax[qq]));
      }
    });
  }
}
}  // namespace

extern "C" int r16_run(const void* in_dev, long long frame_stride, int n_frames, const float* window_dev,
                       const void* tw_dev, float* out_dev, float* hold_dev, int iters, float* ms_per_iter) {
  P p{static_cast<const unsigned char*>(in_dev), frame_stride, n_frames, window_dev, static_cast<const float2*>(tw_dev),
      out_dev, hold_dev, 0.0f};
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(r16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            int(LDS_BYTES)) != hipSuccess) return 1;
    attr = true;
  }
  const int grid = n_frames < 256 ? n_frames : 256;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(r16_kernel, dim3(grid), dim3(NT), LDS_BYTES, 0, p);
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  (void)hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(r16_kernel, dim3(grid), dim3(NT), LDS_BYTES, 0, p);
  (void)hipEventRecord(e1);
  if (hipEventSynchronize(e1) != hipSuccess) return 3;
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  if (ms_per_iter) *ms_per_iter = iters > 0 ? ms / iters : 0.f;
  return hipGetLastError() == hipSuccess ? 0 : 4;
}
