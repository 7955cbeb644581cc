// tdsa_chirp.hip - frames whose length is NOT a power of two (and the powers of two below 64): 2 <= N <= 2^19.
//
// np.fft.fft / scipy.fft.fft take any N (hackrf_samples.py:370, rtl_samples.py:170) and
// HackrfSamplesDataSource.set_num_samples / RtlSamplesDataSource.set_fft_size accept any positive size
// (hackrf_samples.py:392-405, rtl_samples.py:208-214).  Such a frame is transformed as a chirp-z (Bluestein)
// convolution on the power-of-two frame kernel:
//
//   X[k] = a[k] * sum_n (x[n] w[n] a[n]) * conj(a)[k - n],     a[n] = exp(-i pi n^2 / N)
//
//   (0) chirp_sums   : exact per-frame I / Q sums -> the frame mean as a small residual on top of the format's
//                      zero level (dc_alpha = 1: that IS the subtract value; 0 <= dc_alpha < 1: the ordinary DC
//                      tracker of tdsa_trace.hip runs on the residuals)
//   (1) chirp_pre    : unpack + DC + window, times a[n] -> the first N entries of the rows U[f][M] complex64,
//                      M = 2^ceil(log2(2N-1)); the zero padding is implied (SpecParams::in_valid), not stored
//   (2) frame kernel : FFT_M(U), stored as conj(FFT_M(U) * B) - B = FFT_M(conj(a) wrapped), a plan-time table
//                      made in double (SpecParams::out_mul: the multiply rides the transform's stores)
//   (3) frame kernel : FFT_M of that = M * conj(convolution)      (inverse transform through conjugation)
//   (4) chirp_post   : |X[k]|^2 = |./M|^2 for k < N (|a[k]| = 1) -> fftshift by N/2 (np.fft.fftshift for any N)
//                      -> dB (+cal, -tare) rows and hold traces, or linear power rows for the averager scan
// N > 8192 (M = 2^15 .. 2^20, round 4): steps (2) and (3) run on the long-frame kernels of tdsa_big.hip - (2) as column
// pass + rows through the frame kernel, which leaves conj(X B) in its own [k1][k2] order (B is stored in that order),
// (3) transposed: rows first, then big_cols_out_kernel's per-column N1-point DFT, which leaves natural order.
//
// M <= 16384 and frames of whole waves (M >= 1024), round 5: steps (1) and (4) ride the transforms - the first
// transform unpacks the raw samples and multiplies them by window x a[n] on load, the second stores the dB / power rows
// of the N wanted bins itself (instantiations spectrum_kernel<L, true, 0, 1 | 2>); the rows U and the complex result
// are never stored; hold traces are folded from the finished rows (chirp_hold_kernel).
//
// Cost: two M-point complex-to-complex transforms (plus, for M < 1024 or M > 16384, two element-wise passes over
// [F][M] complex64 rows); the point of this path is that every size the reference accepts has a device path.
#include "tdsa_fft.hpp"
#include "tdsa_kernels.hpp"

namespace tdsa {

namespace {

constexpr float kTenLog10Of2 = 3.01029995663981195214f;   // 10*log10(2): dB = kTenLog10Of2 * log2(power)

// float max/min through integer atomics (IEEE-754 order trick); NaN candidates are skipped (np.fmax / np.fmin)
__device__ __forceinline__ void chirp_atomic_fmax(float* addr, float v) {
  if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else if (v < 0.f) atomicMin(reinterpret_cast<unsigned*>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void chirp_atomic_fmin(float* addr, float v) {
  if (v >= 0.f) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
  else if (v < 0.f) atomicMax(reinterpret_cast<unsigned*>(addr), __float_as_uint(v));
}

}  // namespace

// ---- (0) frame means ------------------------------------------------------------------------------------
// res[f] = mean of the frame's raw samples MINUS the format's zero level (128 / 127.5 / 0), raw units.  Byte
// formats: integer sums (exact), the small numerator sum - zero * n formed in integers, one division in double -
// a float32 "sum / n - 128" would cancel to ~1e-5 LSB, visible in the DC bin.
// Long frames (part != null): gridDim.y workgroups per frame, kChirpSumChunk samples each, leave their partial sums in
// part[f][chunk][2] and chirp_sums_finish_kernel adds them in chunk order (one workgroup per frame read a 10^6-point frame at
// 48 GB/s: 420 of the 780 us of a ten-frame call).
constexpr int kChirpSumChunk = 32768;
template <bool IN_C64>
__global__ void __launch_bounds__(256) chirp_sums_kernel(const void* in, unsigned xor_mask, long long frame_stride, int n_all,
                                                         int twice_zero, float2* res, float2* dc_state, float in_scale,
                                                         double* part) {
  __shared__ double red[8];
  const int f = blockIdx.x;
  const int i0 = part != nullptr ? int(blockIdx.y) * kChirpSumChunk : 0;
  const int n = part != nullptr ? (n_all - i0 < kChirpSumChunk ? n_all - i0 : kChirpSumChunk) : n_all;   // this workgroup's samples
  const unsigned char* fb = static_cast<const unsigned char*>(in) + (long long)f * frame_stride +
                            (long long)i0 * (IN_C64 ? 8 : 2);
  double sr = 0.0, si = 0.0;
  if constexpr (IN_C64) {
    const float2* x = reinterpret_cast<const float2*>(fb);
    for (int i = threadIdx.x; i < n; i += 256) { sr += double(x[i].x); si += double(x[i].y); }
  } else {
    // four samples (8 bytes) per lane and load (frame starts are only sample aligned: packed struct), the tail
    // sample by sample; sums of bytes: exact in 32-bit integers for any n this path takes
    struct __attribute__((packed, aligned(2))) Quad { unsigned x, y; };
    const Quad* xq = reinterpret_cast<const Quad*>(fb);
    const uint16_t* x = reinterpret_cast<const uint16_t*>(fb);
    unsigned ui = 0, uq = 0;
    const int nq = n / 4;
    for (int i = threadIdx.x; i < nq; i += 256) {
      const Quad q = xq[i];
      const unsigned a = q.x ^ xor_mask, b = q.y ^ xor_mask;
      ui = __builtin_amdgcn_udot4(a, 0x00010001u, ui, false);
      uq = __builtin_amdgcn_udot4(a, 0x01000100u, uq, false);
      ui = __builtin_amdgcn_udot4(b, 0x00010001u, ui, false);
      uq = __builtin_amdgcn_udot4(b, 0x01000100u, uq, false);
    }
    for (int i = 4 * nq + threadIdx.x; i < n; i += 256) {
      const unsigned u = unsigned(x[i]) ^ (xor_mask & 0xffffu);
      ui += u & 0xffu;
      uq += u >> 8;
    }
    sr = double(ui); si = double(uq);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { sr += __shfl_xor(sr, off); si += __shfl_xor(si, off); }
  if ((threadIdx.x & 63) == 0) { red[(threadIdx.x >> 6) * 2] = sr; red[(threadIdx.x >> 6) * 2 + 1] = si; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double tr = red[0] + red[2] + red[4] + red[6], ti = red[1] + red[3] + red[5] + red[7];
    if (part != nullptr) {
      double* o = part + ((long long)f * gridDim.y + blockIdx.y) * 2;
      o[0] = tr;
      o[1] = ti;
      return;
    }
    // (2 sum - twice_zero n) / (2 n): integer-valued numerator for the byte formats
    const double dn = double(n), tz = double(twice_zero);
    const float2 r = float2{float((2.0 * tr - tz * dn) / (2.0 * dn)), float((2.0 * ti - tz * dn) / (2.0 * dn))};
    res[f] = r;
    // per-frame mean mode (dc_alpha = 1): the estimate the plan carries is the last frame's mean, in units of x
    if (dc_state != nullptr && f == int(gridDim.x) - 1) *dc_state = float2{r.x * in_scale, r.y * in_scale};
  }
}

// Short frames (n <= kChirpSumWaveMax): a wave per frame, four frames per workgroup - a workgroup per 300-sample frame made
// this a launch of 8192 workgroups reading 600 bytes each (11 of the 42 us of such a call).
constexpr int kChirpSumWaveMax = 4096;
template <bool IN_C64>
__global__ void __launch_bounds__(256) chirp_sums_wave_kernel(const void* in, unsigned xor_mask, long long frame_stride, int n,
                                                              int n_frames, int twice_zero, float2* res, float2* dc_state,
                                                              float in_scale) {
  const int lane = threadIdx.x & 63;
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (f >= n_frames) return;
  const unsigned char* fb = static_cast<const unsigned char*>(in) + (long long)f * frame_stride;
  double sr = 0.0, si = 0.0;
  if constexpr (IN_C64) {
    const float2* x = reinterpret_cast<const float2*>(fb);
    for (int i = lane; i < n; i += 64) { sr += double(x[i].x); si += double(x[i].y); }
  } else {
    struct __attribute__((packed, aligned(2))) Quad { unsigned x, y; };
    const Quad* xq = reinterpret_cast<const Quad*>(fb);
    const uint16_t* x = reinterpret_cast<const uint16_t*>(fb);
    unsigned ui = 0, uq = 0;
    const int nq = n / 4;
    for (int i = lane; i < nq; i += 64) {
      const Quad q = xq[i];
      const unsigned a = q.x ^ xor_mask, b = q.y ^ xor_mask;
      ui = __builtin_amdgcn_udot4(a, 0x00010001u, ui, false);
      uq = __builtin_amdgcn_udot4(a, 0x01000100u, uq, false);
      ui = __builtin_amdgcn_udot4(b, 0x00010001u, ui, false);
      uq = __builtin_amdgcn_udot4(b, 0x01000100u, uq, false);
    }
    for (int i = 4 * nq + lane; i < n; i += 64) {
      const unsigned u = unsigned(x[i]) ^ (xor_mask & 0xffffu);
      ui += u & 0xffu;
      uq += u >> 8;
    }
    sr = double(ui); si = double(uq);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { sr += __shfl_xor(sr, off); si += __shfl_xor(si, off); }
  if (lane == 0) {
    const double dn = double(n), tz = double(twice_zero);
    const float2 r = float2{float((2.0 * sr - tz * dn) / (2.0 * dn)), float((2.0 * si - tz * dn) / (2.0 * dn))};
    res[f] = r;
    if (dc_state != nullptr && f == n_frames - 1) *dc_state = float2{r.x * in_scale, r.y * in_scale};
  }
}

// one wave per frame: the partial sums are fetched side by side, thread 0 adds them in chunk order (reproducible; exact for
// the byte formats: integers)
__global__ void __launch_bounds__(64) chirp_sums_finish_kernel(const double* part, int chunks, int n, int n_frames,
                                                               int twice_zero, float2* res, float2* dc_state, float in_scale) {
  __shared__ double pr[64], pi[64];
  const int f = blockIdx.x;
  double tr = 0.0, ti = 0.0;
  for (int c0 = 0; c0 < chunks; c0 += 64) {
    const int c = c0 + int(threadIdx.x);
    if (c < chunks) {
      pr[threadIdx.x] = part[((long long)f * chunks + c) * 2];
      pi[threadIdx.x] = part[((long long)f * chunks + c) * 2 + 1];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const int nc = chunks - c0 < 64 ? chunks - c0 : 64;
      for (int u = 0; u < nc; ++u) { tr += pr[u]; ti += pi[u]; }
    }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  const double dn = double(n), tz = double(twice_zero);
  const float2 r = float2{float((2.0 * tr - tz * dn) / (2.0 * dn)), float((2.0 * ti - tz * dn) / (2.0 * dn))};
  res[f] = r;
  if (dc_state != nullptr && f == n_frames - 1) *dc_state = float2{r.x * in_scale, r.y * in_scale};
}

int chirp_sum_chunks(int n) { return n > 2 * kChirpSumChunk ? (n + kChirpSumChunk - 1) / kChirpSumChunk : 1; }

hipError_t launch_chirp_sums(const void* in, int in_c64, unsigned xor_mask, long long frame_stride, int n, int n_frames,
                             int twice_zero, float2* res, float2* dc_state, float in_scale, hipStream_t s, double* part) {
  if (n <= kChirpSumWaveMax && n_frames >= 256) {   // (few frames - a GUI tick has one - keep a workgroup each: latency)
    const dim3 grid((n_frames + 3) / 4);
    if (in_c64)
      hipLaunchKernelGGL(chirp_sums_wave_kernel<true>, grid, dim3(256), 0, s, in, xor_mask, frame_stride, n, n_frames,
                         twice_zero, res, dc_state, in_scale);
    else
      hipLaunchKernelGGL(chirp_sums_wave_kernel<false>, grid, dim3(256), 0, s, in, xor_mask, frame_stride, n, n_frames,
                         twice_zero, res, dc_state, in_scale);
    return hipGetLastError();
  }
  const int chunks = part != nullptr ? chirp_sum_chunks(n) : 1;
  double* const pp = chunks > 1 ? part : nullptr;
  const dim3 grid(n_frames, chunks);
  if (in_c64)
    hipLaunchKernelGGL(chirp_sums_kernel<true>, grid, dim3(256), 0, s, in, xor_mask, frame_stride, n, twice_zero,
                       res, dc_state, in_scale, pp);
  else
    hipLaunchKernelGGL(chirp_sums_kernel<false>, grid, dim3(256), 0, s, in, xor_mask, frame_stride, n, twice_zero,
                       res, dc_state, in_scale, pp);
  if (pp != nullptr)
    hipLaunchKernelGGL(chirp_sums_finish_kernel, dim3(n_frames), dim3(64), 0, s, pp, chunks, n, n_frames,
                       twice_zero, res, dc_state, in_scale);
  return hipGetLastError();
}

// ---- (1) unpack, DC, window, chirp, zero padding ------------------------------------------------------------
struct ChirpPreParams {
  const void* in;
  int in_c64;
  long long frame_stride;    // bytes
  int n, m, n_frames;
  const float* window;       // [n] window * input scale
  const float2* chirp;       // [n] a[n]
  const float2* dc_sub;      // [F] DC estimate minus the zero level, raw units, or null
  unsigned xor_mask;
  float in_off;
  float2* u;                 // [F][m]
  int split_h;               // > 0 (frames of more than 2^19 points): sample i goes to row 2f + (i >= split_h), index i mod split_h,
                             // of the rows [2F][m]; the second row is zero-filled up to split_h
};

// two consecutive samples per thread: 4 / 16 bytes in (frame starts are only sample aligned: packed loads),
// one 16-byte store out; m is a power of two >= 64, so a pair never straddles the end of a row
struct __attribute__((packed, aligned(2))) ChirpRaw2 { uint32_t v; };
struct __attribute__((packed, aligned(4))) ChirpC64x2 { float a, b, c, d; };

__global__ void __launch_bounds__(256) chirp_pre_kernel(const ChirpPreParams p) {
  const int i = 2 * (blockIdx.x * 256 + threadIdx.x);
  if (i >= p.n) return;                  // the padding up to M is never read (SpecParams::in_valid): not written
  const unsigned xm = p.xor_mask;
  const bool in0 = i < p.n, in1 = i + 1 < p.n;
  float w0 = 0.f, w1 = 0.f;
  c32 a0 = c32{0.f, 0.f}, a1 = c32{0.f, 0.f};
  if (in0) { w0 = p.window[i]; a0 = p.chirp[i]; }
  if (in1) { w1 = p.window[i + 1]; a1 = p.chirp[i + 1]; }
  for (int f = blockIdx.y; f < p.n_frames; f += gridDim.y) {
    c32 o0 = c32{0.f, 0.f}, o1 = c32{0.f, 0.f};
    if (in0) {
      const unsigned char* fb = static_cast<const unsigned char*>(p.in) + (long long)f * p.frame_stride;
      float r0, i0, r1 = 0.f, i1 = 0.f;
      if (p.in_c64) {
        const float* x = reinterpret_cast<const float*>(fb) + 2 * i;
        if (in1) { const ChirpC64x2 q = *reinterpret_cast<const ChirpC64x2*>(x); r0 = q.a; i0 = q.b; r1 = q.c; i1 = q.d; }
        else { r0 = x[0]; i0 = x[1]; }
      } else {
        unsigned v;
        if (in1) v = reinterpret_cast<const ChirpRaw2*>(fb + 2 * i)->v ^ xm;
        else v = unsigned(reinterpret_cast<const uint16_t*>(fb)[i]) ^ (xm & 0xffffu);
        r0 = float(v & 0xffu) - p.in_off;      // exact: small integers / halves
        i0 = float((v >> 8) & 0xffu) - p.in_off;
        r1 = float((v >> 16) & 0xffu) - p.in_off;
        i1 = float(v >> 24) - p.in_off;
      }
      if (p.dc_sub != nullptr) { const float2 d = p.dc_sub[f]; r0 -= d.x; i0 -= d.y; r1 -= d.x; i1 -= d.y; }
      o0 = cmul(c32{r0 * w0, i0 * w0}, a0);
      if (in1) o1 = cmul(c32{r1 * w1, i1 * w1}, a1);
    }
    if (p.split_h == 0) {
      *reinterpret_cast<float4*>(p.u + (long long)f * p.m + i) = float4{o0.x, o0.y, o1.x, o1.y};
    } else {
      // two half-length rows per frame (an odd split_h: a pair may straddle the halves - element by element)
      float2* rows = p.u + (long long)(2 * f) * p.m;
      const int h = p.split_h;
      rows[i < h ? i : p.m + i - h] = float2{o0.x, o0.y};
      if (in1) rows[i + 1 < h ? i + 1 : p.m + i + 1 - h] = float2{o1.x, o1.y};
      if (i + 2 >= p.n) {                       // the thread that holds the frame's end pads the second row up to split_h
        for (int z = p.n - h; z < h; ++z) rows[p.m + z] = float2{0.f, 0.f};
      }
    }
  }
}

hipError_t launch_chirp_pre(const void* in, int in_c64, long long frame_stride, int n, int m, int n_frames,
                            const float* window, const float2* chirp, const float2* dc_sub, unsigned xor_mask,
                            float in_off, float2* u, hipStream_t s, int split_h) {
  ChirpPreParams p{in, in_c64, frame_stride, n, m, n_frames, window, chirp, dc_sub, xor_mask, in_off, u, split_h};
  const int gy = n_frames < 2048 ? n_frames : 2048;
  hipLaunchKernelGGL(chirp_pre_kernel, dim3(((n + 1) / 2 + 255) / 256, gy), dim3(256), 0, s, p);
  return hipGetLastError();
}

// ---- (4) back to N bins: power, fftshift, dB / linear rows, hold traces ---------------------------------------
struct ChirpPostParams {
  const float2* y;           // [F][m]  M * conj(convolution)
  int n, m, n_frames, first_frame_index;
  float inv_m;
  int db_mode;               // 0: 20 log10(|X| + floor), 1: 10 log10(|X|^2 * pscale + floor)
  float pscale, log_floor, cal_db;
  const float* tare;         // [n] or null
  float* out_db;             // [F][n] or null
  float* out_lin;            // [F][n] linear power * pscale (averaging modes) or null
  float* hold_max;           // [n] or null
  float* hold_min;
  int split_h;               // > 0: y holds [2F][m] rows, bin k of frame f at row 2f + (k >= split_h), index k mod split_h
};

constexpr int kChirpFramesPerBlock = 8;    // one hold atomic (when the trace moves) per bin and this many frames

__global__ void __launch_bounds__(256) chirp_post_kernel(const ChirpPostParams p) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= p.n) return;
  int j = k + p.n / 2;                       // np.fft.fftshift: bin k lands at (k + N/2) mod N, any N
  if (j >= p.n) j -= p.n;
  const float tare = p.tare != nullptr ? p.tare[j] : 0.f;
  float hmax = -INFINITY, hmin = INFINITY;
  const int f0 = blockIdx.y * kChirpFramesPerBlock;
  const int f1 = f0 + kChirpFramesPerBlock < p.n_frames ? f0 + kChirpFramesPerBlock : p.n_frames;
#pragma unroll 4
  for (int f = f0; f < f1; ++f) {
    // X[k] = a[k] * conj(w) / M with |a[k]| = 1: only |X|^2 is needed, the last chirp factor drops out
    const c32 w = p.split_h == 0 ? p.y[(long long)f * p.m + k]
                                 : p.y[(long long)(2 * f + (k >= p.split_h ? 1 : 0)) * p.m + (k >= p.split_h ? k - p.split_h : k)];
    const float xr = w.x * p.inv_m, xi = w.y * p.inv_m;
    const float pw = xr * xr + xi * xi;
    if (p.out_lin != nullptr) {
      p.out_lin[(long long)f * p.n + j] = pw * p.pscale;
      continue;
    }
    float db;
    if (p.db_mode == 0) db = fmaf(2.0f * kTenLog10Of2, __builtin_amdgcn_logf(__builtin_amdgcn_sqrtf(pw) + p.log_floor), p.cal_db);
    else db = fmaf(kTenLog10Of2, __builtin_amdgcn_logf(fmaf(pw, p.pscale, p.log_floor)), p.cal_db);
    db -= tare;
    if (p.out_db != nullptr) p.out_db[(long long)f * p.n + j] = db;
    float dmx = db, dmn = db;
    if (p.first_frame_index + f == 0 && db != db) { dmx = -500.f; dmn = 500.f; }     // _nan_safe, first frame ever
    hmax = fmaxf(hmax, dmx);                 // a NaN operand is ignored, as by np.fmax
    hmin = fminf(hmin, dmn);
  }
  if (p.out_lin == nullptr) {
    if (p.hold_max != nullptr && hmax > p.hold_max[j]) chirp_atomic_fmax(p.hold_max + j, hmax);
    if (p.hold_min != nullptr && hmin < p.hold_min[j]) chirp_atomic_fmin(p.hold_min + j, hmin);
  }
}

hipError_t launch_chirp_post(const float2* y, int n, int m, int n_frames, int first_frame_index,
                             int db_mode, float pscale, float log_floor, float cal_db, const float* tare, float* out_db,
                             float* out_lin, float* hold_max, float* hold_min, hipStream_t s, int split_h) {
  ChirpPostParams p{y, n, m, n_frames, first_frame_index, 1.0f / float(m), db_mode, pscale, log_floor, cal_db,
                    tare, out_db, out_lin, hold_max, hold_min, split_h};
  const int gy = (n_frames + kChirpFramesPerBlock - 1) / kChirpFramesPerBlock;
  hipLaunchKernelGGL(chirp_post_kernel, dim3((n + 255) / 256, gy), dim3(256), 0, s, p);
  return hipGetLastError();
}

// Frames of more than 2^19 points that are not a power of two: M = 2^ceil(log2(2N - 1)) would be 2^21, more than the
// long-frame kernels transform.  The convolution is split instead - samples into halves A = [0, H), B = [H, N), bins into
// K0 = [0, H), K1 = [H, N), H = ceil(N / 2) - into four sub-convolutions of H inputs and H outputs, each of which fits a
// circular convolution of 2^20 points (2 H - 1 <= 2^20):
//   y[k]     = sum_A u[n] b[k - n]     + sum_B u[H + n'] b[k - n' - H]        k  in K0
//   y[H + k'] = sum_A u[n] b[k' - n + H] + sum_B u[H + n'] b[k' - n']         k' in [0, N - H)
// i.e. with the spectra of the two half-rows UA, UB and of three filter segments B0 (b[m]), Bm (b[m - H]), Bp (b[m + H]):
//   Y0 = UA B0 + UB Bm,   Y1 = UA Bp + UB B0  - two forward and two inverse 2^20-point transforms per frame.
// This kernel forms conj(Y0), conj(Y1) in place of UA, UB (rows 2f, 2f + 1 of [2F][m], any element order: the tables are
// stored in the order the first transform leaves its bins in).
__global__ void __launch_bounds__(256) chirp_split_combine_kernel(float2* u, long long m, int n_frames, const float2* b0,
                                                                  const float2* bm, const float2* bp) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const c32 f0 = b0[i], fm = bm[i], fp = bp[i];
  for (int f = blockIdx.y; f < n_frames; f += gridDim.y) {
    float2* ra = u + (long long)(2 * f) * m;
    float2* rb = ra + m;
    const c32 ua = ra[i], ub = rb[i];
    const c32 y0 = cadd(cmul(ua, f0), cmul(ub, fm));
    const c32 y1 = cadd(cmul(ua, fp), cmul(ub, f0));
    ra[i] = float2{y0.x, -y0.y};
    rb[i] = float2{y1.x, -y1.y};
  }
}

hipError_t launch_chirp_split_combine(float2* u, long long m, int n_frames, const float2* b0, const float2* bm,
                                      const float2* bp, hipStream_t s) {
  const int gy = n_frames < 64 ? n_frames : 64;
  hipLaunchKernelGGL(chirp_split_combine_kernel, dim3(unsigned((m + 255) / 256), gy), dim3(256), 0, s, u, m, n_frames, b0, bm, bp);
  return hipGetLastError();
}

// Hold traces from finished dB rows [F][n] (plans with M <= 16384, whose second transform stores the rows itself):
// column max / min over blocks of frames, one atomic per bin and block where the trace moves.
__global__ void __launch_bounds__(256) chirp_hold_kernel(const float* rows, int n, int n_frames, int first_frame_index,
                                                         float* hold_max, float* hold_min) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  float hmax = -INFINITY, hmin = INFINITY;
  const int f0 = blockIdx.y * kChirpFramesPerBlock;
  const int f1 = f0 + kChirpFramesPerBlock < n_frames ? f0 + kChirpFramesPerBlock : n_frames;
#pragma unroll 4
  for (int f = f0; f < f1; ++f) {
    const float db = rows[(long long)f * n + j];
    float dmx = db, dmn = db;
    if (first_frame_index + f == 0 && db != db) { dmx = -500.f; dmn = 500.f; }     // _nan_safe, first frame ever
    hmax = fmaxf(hmax, dmx);                 // a NaN operand is ignored, as by np.fmax
    hmin = fminf(hmin, dmn);
  }
  if (hold_max != nullptr && hmax > hold_max[j]) chirp_atomic_fmax(hold_max + j, hmax);
  if (hold_min != nullptr && hmin < hold_min[j]) chirp_atomic_fmin(hold_min + j, hmin);
}

hipError_t launch_chirp_hold(const float* rows, int n, int n_frames, int first_frame_index, float* hold_max, float* hold_min,
                             hipStream_t s) {
  const int gy = (n_frames + kChirpFramesPerBlock - 1) / kChirpFramesPerBlock;
  hipLaunchKernelGGL(chirp_hold_kernel, dim3((n + 255) / 256, gy), dim3(256), 0, s, rows, n, n_frames, first_frame_index,
                     hold_max, hold_min);
  return hipGetLastError();
}

// Real-input (audio) frames on a chirp plan: the signal went in as signal + 0i, so X is its full spectrum and the
// one-sided power is |X[k]|^2 for k = 0 .. N/2, every bin but the first and the last doubled
// (power[1:-1] *= 2, audio_samples.py:131 - also for odd N, where the last bin is not a Nyquist bin).
// lin[(f * rows_per_frame + row) * (N/2 + 1) + k]: the layout of real_fold_kernel.
__global__ void __launch_bounds__(256) chirp_post_real_kernel(const float2* y, int n, int m, int n_frames, int rows_per_frame,
                                                              int row, float inv_m, float pscale, float* lin) {
  const int nb = n / 2 + 1;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= nb) return;
  const float two = (k >= 1 && k <= nb - 2) ? 2.0f : 1.0f;
  for (int f = blockIdx.y; f < n_frames; f += gridDim.y) {
    const c32 w = y[(long long)f * m + k];
    const float xr = w.x * inv_m, xi = w.y * inv_m;
    lin[((long long)f * rows_per_frame + row) * nb + k] = (xr * xr + xi * xi) * pscale * two;
  }
}

hipError_t launch_chirp_post_real(const float2* y, int n, int m, int n_frames, int rows_per_frame, int row, float pscale,
                                  float* lin, hipStream_t s) {
  const int gy = n_frames < 4096 ? n_frames : 4096;
  hipLaunchKernelGGL(chirp_post_real_kernel, dim3((n / 2 + 1 + 255) / 256, gy), dim3(256), 0, s, y, n, m, n_frames,
                     rows_per_frame, row, 1.0f / float(m), pscale, lin);
  return hipGetLastError();
}

}  // namespace tdsa
