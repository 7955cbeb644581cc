// tdsa_big.hip - the 2^20-point path (BASELINE.json config C5: 1M-pt FFT, Welch averaging, cal offset).
//
// A 2^20-point frame (8 MiB as complex64) cannot live in LDS, so it is done as a four-step FFT over
// N = N1 * N2 = 1024 * 1024 with n = n1*N2 + n2 and k = k1 + N1*k2:
//   (0) transpose_in : raw IQ bytes [n1][n2] -> [n2][n1]                (2N bytes read, 2N written)
//   (1) cols_kernel  : per n2: window, DC, 1024-pt FFT over n1, * W_N^(n2*k1); written as Y[k1][n2]
//                      through an LDS tile so every store is a full 128-byte line (8N bytes written)
//   (2) rows_kernel  : per k1: 1024-pt FFT over n2, |X|^2, accumulated over the K Welch segments in
//                      float64 registers, one atomic add per bin per workgroup (8N bytes read)
//   (3) finish_kernel: mean -> 10*log10(. * scale + floor) + cal (- tare) -> fftshift-ed dB row, hold
// Replaces np.fft.fft on a 2^20 frame + TraceAverager("lin") + _apply_cal_offset of the reference
// (hackrf_samples.py:370, utils/signal_processing.py:56-59, core/display_data_processor.py:317-327).
// Every 1024-point FFT is one half-wave (32 lanes x 32 points): radix-32, wave-local LDS transpose,
// twiddle, radix-32.
#include "tdsa_fft.hpp"
#include "tdsa_kernels.hpp"

namespace tdsa {

constexpr int kB1 = 1024;            // N1 = N2
constexpr int kBigN = kB1 * kB1;

// 1024-point FFT held by a half-wave: lane j (0..31) enters with v[i] = x[j + 32*i] and leaves with
// v[m] = X[j + 32*m].  xch: this half-wave's private 33*32 complex LDS scratch.  tw1k: W_1024^m table.
__device__ __forceinline__ void fft1024_halfwave(c32 (&v)[32], c32* xch, const c32* __restrict__ tw1k, int j) {
  dif<32, 0, 32>(v);                                        // X1[k] (k = 0..31) at v[bitrev(k)]
  static_for<0, 32>([&](auto ic) {
    constexpr int k = decltype(ic)::value;
    xch[j * 33 + k] = v[bitrev(k, 5)];
  });
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  // lane j now takes output residue k = j: needs X1_{j'}[k = j] for all j' = 0..31, times W_1024^(j' * j)
  static_for<0, 32>([&](auto ic) {
    constexpr int jp = decltype(ic)::value;
    const c32 x = xch[jp * 33 + j];
    v[jp] = jp == 0 ? x : cmul(x, tw1k[(jp * j) & 1023]);
  });
  __builtin_amdgcn_wave_barrier();
  dif<32, 0, 32>(v);                                        // X[j + 32*m] at v[bitrev(m)]
  c32 t[32];
  static_for<0, 32>([&](auto ic) { constexpr int m = decltype(ic)::value; t[m] = v[bitrev(m, 5)]; });
  static_for<0, 32>([&](auto ic) { constexpr int m = decltype(ic)::value; v[m] = t[m]; });
}

// (0) raw bytes: in[seg*stride + (n1*1024 + n2)*2 .. +1]  ->  xt[seg][n2][n1] (2 bytes per sample)
__global__ void __launch_bounds__(256) big_transpose_in(const unsigned char* in, long long seg_stride,
                                                        uint16_t* xt) {
  __shared__ uint16_t tile[32][33];
  const int seg = blockIdx.z, n1b = blockIdx.y * 32, n2b = blockIdx.x * 32;
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;   // 32 x 8
  const unsigned char* src = in + (long long)seg * seg_stride;
#pragma unroll
  for (int r = 0; r < 32; r += 8) {
    const long long s = (long long)(n1b + ly + r) * kB1 + n2b + lx;
    tile[ly + r][lx] = uint16_t(src[2 * s]) | (uint16_t(src[2 * s + 1]) << 8);   // byte loads: any alignment
  }
  __syncthreads();
  uint16_t* dst = xt + (long long)seg * kBigN;
#pragma unroll
  for (int r = 0; r < 32; r += 8) dst[(long long)(n2b + ly + r) * kB1 + n1b + lx] = tile[lx][ly + r];
}

struct BigColsParams {
  const uint16_t* xt;        // [K][n2][n1]
  const float* wt;           // [n2][n1] window * input scale (transposed)
  const float2* tw1k;        // W_1024^m
  const float2* twlo;        // W_N^m, m < 1024
  const float2* dc_sub;      // [K] per-segment subtract value (raw units) or null
  float2* y;                 // [K][k1][n2]
  unsigned xor_mask;
  float in_off;
};

constexpr int kColRows = 16;  // n2 values per workgroup -> 128-byte output lines
__global__ void __launch_bounds__(kColRows * 32, 2) big_cols_kernel(const BigColsParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  c32* lds = reinterpret_cast<c32*>(smem);                 // 16 * 33*32 exchange; later the [1024][16] tile
  const int tid = threadIdx.x, j = tid & 31, row = tid >> 5;
  const int seg = blockIdx.y, n2 = blockIdx.x * kColRows + row;
  const uint16_t* x = p.xt + ((long long)seg * kB1 + n2) * kB1;
  const float* w = p.wt + (long long)n2 * kB1;
  float sub_re = p.in_off, sub_im = p.in_off;
  if (p.dc_sub != nullptr) { const c32 s = p.dc_sub[seg]; sub_re = s.x; sub_im = s.y; }
  c32 v[32];
  static_for<0, 32>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    const unsigned u = (unsigned(x[j + 32 * i]) ^ p.xor_mask) & 0xffffu;
    const float ww = w[j + 32 * i];
    v[i] = c32{(float(u & 0xffu) - sub_re) * ww, (float(u >> 8) - sub_im) * ww};
  });
  fft1024_halfwave(v, lds + row * (33 * 32), p.tw1k, j);    // v[m] = Y[k1 = j + 32m] of column n2
  __syncthreads();                                          // all exchanges done: reuse LDS as the tile
  static_for<0, 32>([&](auto ic) {
    constexpr int m = decltype(ic)::value;
    const int k1 = j + 32 * m;
    const unsigned e = unsigned(n2) * unsigned(k1);          // < 2^20 : W_N^e = W_1024^(e >> 10) * W_N^(e & 1023)
    const c32 tw = cmul(p.tw1k[e >> 10], p.twlo[e & 1023]);
    lds[k1 * (kColRows + 1) + row] = cmul(v[m], tw);     // 17-element lines: conflict-free stores
  });
  __syncthreads();
  // 1024 lines of 16 complex (one full 128-byte line each): 16 threads per line
  c32* yo = p.y + (long long)seg * kBigN + (long long)blockIdx.x * kColRows;
#pragma unroll 4
  for (int it = 0; it < 32; ++it) {
    const int k1 = it * 32 + (tid >> 4), q = tid & 15;
    yo[(long long)k1 * kB1 + q] = lds[k1 * (kColRows + 1) + q];
  }
}

constexpr int kRowRows = 4;   // k1 rows per workgroup
__global__ void __launch_bounds__(kRowRows * 32, 2) big_rows_kernel(const float2* y, const float2* tw1k, int n_seg,
                                                                    double* sum) {
  __shared__ __attribute__((aligned(16))) c32 lds[kRowRows * 33 * 32];
  const int tid = threadIdx.x, j = tid & 31, row = tid >> 5;
  const int k1 = blockIdx.x * kRowRows + row;
  const int s0 = int((long long)blockIdx.y * n_seg / gridDim.y), s1 = int((long long)(blockIdx.y + 1) * n_seg / gridDim.y);
  double acc[32];
  static_for<0, 32>([&](auto ic) { acc[decltype(ic)::value] = 0.0; });
  for (int seg = s0; seg < s1; ++seg) {
    const c32* yr = y + ((long long)seg * kB1 + k1) * kB1;
    c32 v[32];
    static_for<0, 32>([&](auto ic) { constexpr int i = decltype(ic)::value; v[i] = yr[j + 32 * i]; });
    fft1024_halfwave(v, lds + row * (33 * 32), tw1k, j);   // v[m] = X[k1 + 1024*(j + 32m)]
    static_for<0, 32>([&](auto ic) {
      constexpr int m = decltype(ic)::value;
      acc[m] += double(v[m].x * v[m].x + v[m].y * v[m].y);
    });
  }
  if (s1 > s0) {
    static_for<0, 32>([&](auto ic) {
      constexpr int m = decltype(ic)::value;
      const int k2 = j + 32 * m;
      atomicAdd(&sum[(long long)k2 * kB1 + k1], acc[m]);    // natural order k = k1 + 1024*k2
    });
  }
}

struct BigFinishParams {
  const double* sum;   // [N] natural order, sum over all segments seen so far
  double* mean_out;    // [N] fftshift-ed running mean (TraceAverager._buffer) or null
  int count;
  int db_mode;         // 0: 20log10(sqrt(mean)+floor), 1: 10log10(mean*scale+floor)
  float pscale, log_floor, cal_db;
  const float* tare;
  float* out_db;       // [N] or null
  float* hold_max;     // or null
  float* hold_min;
  int max_first, min_first;
};
__global__ void __launch_bounds__(256) big_finish_kernel(const BigFinishParams p) {
  const int ks = blockIdx.x * 256 + threadIdx.x;            // shifted index
  const int k = ks ^ (kBigN / 2);
  const double mean = p.sum[k] / double(p.count);
  if (p.mean_out != nullptr) p.mean_out[ks] = mean;
  float db;
  if (p.db_mode == 0) db = 20.0f * log10f(sqrtf(float(mean)) + p.log_floor);
  else db = 10.0f * log10f(float(mean * double(p.pscale) + double(p.log_floor)));
  db += p.cal_db;
  if (p.tare != nullptr) db -= p.tare[ks];
  if (p.out_db != nullptr) p.out_db[ks] = db;
  if (p.hold_max != nullptr) p.hold_max[ks] = p.max_first ? ((db != db) ? -500.f : db) : fmaxf(p.hold_max[ks], db);
  if (p.hold_min != nullptr) p.hold_min[ks] = p.min_first ? ((db != db) ? 500.f : db) : fminf(p.hold_min[ks], db);
}

// ---- host launchers ------------------------------------------------------------------------------------
hipError_t launch_big_transpose(const void* in, long long seg_stride, int n_seg, uint16_t* xt, hipStream_t s) {
  hipLaunchKernelGGL(big_transpose_in, dim3(32, 32, n_seg), dim3(256), 0, s, static_cast<const unsigned char*>(in),
                     seg_stride, xt);
  return hipGetLastError();
}

hipError_t launch_big_cols(const uint16_t* xt, const float* wt, const float2* tw1k, const float2* twlo,
                           const float2* dc_sub, float2* y, unsigned xor_mask, float in_off, int n_seg,
                           hipStream_t s) {
  BigColsParams p{xt, wt, tw1k, twlo, dc_sub, y, xor_mask, in_off};
  const size_t lds = size_t(kB1) * (kColRows + 1) * sizeof(c32);    // 136 KiB tile (>= 16 exchange areas)
  static std::atomic<unsigned long long> attr_done{0};
  const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(big_cols_kernel), int(lds), attr_done);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(big_cols_kernel, dim3(kB1 / kColRows, n_seg), dim3(kColRows * 32), lds, s, p);
  return hipGetLastError();
}

hipError_t launch_big_rows(const float2* y, const float2* tw1k, int n_seg, double* sum, hipStream_t s) {
  int split = n_seg >= 4 ? 4 : n_seg;
  hipLaunchKernelGGL(big_rows_kernel, dim3(kB1 / kRowRows, split), dim3(kRowRows * 32), 0, s, y, tw1k, n_seg, sum);
  return hipGetLastError();
}

hipError_t launch_big_finish(const double* sum, double* mean_out, int count, int db_mode, float pscale,
                             float log_floor, float cal_db, const float* tare, float* out_db, float* hold_max,
                             float* hold_min, int max_first, int min_first, hipStream_t s) {
  BigFinishParams p{sum, mean_out, count, db_mode, pscale, log_floor, cal_db, tare, out_db, hold_max, hold_min,
                    max_first, min_first};
  hipLaunchKernelGGL(big_finish_kernel, dim3(kBigN / 256), dim3(256), 0, s, p);
  return hipGetLastError();
}

}  // namespace tdsa
