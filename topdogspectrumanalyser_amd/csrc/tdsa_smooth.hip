// Frame lengths whose only prime factors are 2, 3 and 5 (1000, 1500, 3000, 6000, ... - what a user types into
// set_num_samples / set_fft_size: hackrf_samples.py:392-405, rtl_samples.py:208-214): a mixed-radix
// Stockham FFT of exactly N points - in LDS up to 10 000 points, in two passes above (up to 2^20) - instead of the chirp-z convolution's two transforms of M >= 2N - 1 points
// (tdsa_chirp.hip, which keeps every other size).  np.fft.fft / scipy.fft.fft run the same kind of factorisation on the host
// (hackrf_samples.py:370, rtl_samples.py:170).
//
// One workgroup takes FPW frames; every stage of radix r in {4, 2, 3, 5} reads the frames from one LDS buffer and writes
// them to the other (autosort: the last stage leaves natural bin order), a barrier between stages:
//   n = current sub-length, s = N / n, m = n / r;  butterfly (p < m, q < s):
//     a_t = x[q + s (p + t m)],  b = DFT_r(a),  y[q + s (r p + u)] = b_u W_N^(p u s)
// Around it the path of the chirp-z plans, operation for operation: unpack, minus the frame mean / tracked DC estimate
// (chirp_sums_kernel + dc_track_kernel), times window x input scale on load; |X|^2, fftshift by floor(N / 2), dB + cal
// offset - tare or linear power rows on store; hold traces folded from the rows by chirp_hold_kernel.
#include <hip/hip_runtime.h>

#include "tdsa_kernels.hpp"

namespace tdsa {

namespace {

using c32 = float2;
__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return c32{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return c32{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ c32 cmul_(c32 a, c32 b) { return c32{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ c32 mul_mi(c32 a) { return c32{a.y, -a.x}; }      // a * (-i)

constexpr float kTenLog10Of2 = 3.01029995663981195214f;

template <int R>
__device__ __forceinline__ void dft(c32 (&a)[R]) {
  if constexpr (R == 2) {
    const c32 t = a[0];
    a[0] = cadd(t, a[1]);
    a[1] = csub(t, a[1]);
  } else if constexpr (R == 3) {
    constexpr float s = 0.86602540378443864676f;
    const c32 t1 = cadd(a[1], a[2]);
    const c32 t2 = c32{a[0].x - 0.5f * t1.x, a[0].y - 0.5f * t1.y};
    const c32 d = csub(a[1], a[2]);
    const c32 t3 = c32{s * d.y, -s * d.x};                                  // -i s (a1 - a2)
    a[0] = cadd(a[0], t1);
    a[1] = cadd(t2, t3);
    a[2] = csub(t2, t3);
  } else if constexpr (R == 4) {
    const c32 t0 = cadd(a[0], a[2]), t1 = csub(a[0], a[2]), t2 = cadd(a[1], a[3]), t3 = mul_mi(csub(a[1], a[3]));
    a[0] = cadd(t0, t2);
    a[1] = cadd(t1, t3);
    a[2] = csub(t0, t2);
    a[3] = csub(t1, t3);
  } else {
    static_assert(R == 5, "radix");
    constexpr float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;   // cos(2 pi / 5), cos(4 pi / 5)
    constexpr float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;    // sin(2 pi / 5), sin(4 pi / 5)
    const c32 t1 = cadd(a[1], a[4]), t2 = cadd(a[2], a[3]), t3 = csub(a[1], a[4]), t4 = csub(a[2], a[3]);
    const c32 m1 = c32{a[0].x + c1 * t1.x + c2 * t2.x, a[0].y + c1 * t1.y + c2 * t2.y};
    const c32 m2 = c32{a[0].x + c2 * t1.x + c1 * t2.x, a[0].y + c2 * t1.y + c1 * t2.y};
    const c32 n1 = c32{s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y};
    const c32 n2 = c32{s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y};
    a[0] = cadd(a[0], cadd(t1, t2));
    a[1] = c32{m1.x + n1.y, m1.y - n1.x};                                    // m1 - i n1
    a[4] = c32{m1.x - n1.y, m1.y + n1.x};
    a[2] = c32{m2.x + n2.y, m2.y - n2.x};
    a[3] = c32{m2.x - n2.y, m2.y + n2.x};
  }
}


// i / d for i d < 2^32 (here: i < 2^17, d <= 10 000) without a division: magic = floor(2^32 / d) + 1
// (magic 0: d = 1)
__device__ __forceinline__ int div_magic(int i, unsigned magic) { return magic == 0u ? i : int(__umulhi(unsigned(i), magic)); }

// one stage over the workgroup's FPW frames: butterflies i = (frame, p, q) spread over the threads.  LD(frame, index) /
// ST(frame, index, value): LDS for the stages in the middle; the first stage takes its inputs straight from the raw frames
// (unpack, DC, window) and the last one turns its outputs into the dB / power row - two LDS round trips and two barriers less
// FRFAST: neighbouring threads take the same butterfly of neighbouring slots (two-pass transforms: the slots are adjacent
// columns / rows, whose global elements sit side by side) instead of neighbouring butterflies of one slot
template <int R, bool FRFAST, typename LD, typename ST>
__device__ __forceinline__ void stage(const c32* __restrict__ tw, int tw_step, int N, int s, int fpw, int threads,
                                      unsigned magic_per, unsigned magic_s, unsigned magic_fpw, LD ld, ST st) {
  const int per = N / R, m = per / s;
  for (int i = threadIdx.x; i < fpw * per; i += threads) {
    int fr, b;
    if constexpr (FRFAST) { b = div_magic(i, magic_fpw); fr = i - b * fpw; }
    else { fr = div_magic(i, magic_per); b = i - fr * per; }
    const int pp = s == 1 ? b : div_magic(b, magic_s), q = b - pp * s;
    c32 a[R];
#pragma unroll
    for (int t = 0; t < R; ++t) a[t] = ld(fr, q + s * (pp + t * m));
    dft<R>(a);
    const int ws = pp * s * tw_step;                     // W_n^(p u) = W_N^(p u s); (two passes: W_N = W_Ntotal^tw_step)
    st(fr, q + s * (R * pp), a[0]);
#pragma unroll
    for (int u = 1; u < R; ++u) st(fr, q + s * (R * pp + u), m == 1 ? a[u] : cmul_(a[u], tw[ws * u]));   // (p u s < N)
  }
}

template <bool FRFAST, typename LD, typename ST>
__device__ __forceinline__ void stage_r(int r, const c32* tw, int tw_step, int N, int s, int fpw, int threads, unsigned mp, unsigned ms,
                                        unsigned mf, LD ld, ST st) {
  switch (r) {
    case 5: stage<5, FRFAST>(tw, tw_step, N, s, fpw, threads, mp, ms, mf, ld, st); break;
    case 4: stage<4, FRFAST>(tw, tw_step, N, s, fpw, threads, mp, ms, mf, ld, st); break;
    case 3: stage<3, FRFAST>(tw, tw_step, N, s, fpw, threads, mp, ms, mf, ld, st); break;
    default: stage<2, FRFAST>(tw, tw_step, N, s, fpw, threads, mp, ms, mf, ld, st); break;
  }
}

}  // namespace

// MODE 0: frames of p.n points, a workgroup's slots are fpw consecutive frames.
// Frames of n_total = n1 n2 points above the LDS limit (10 000 < N <= 2^20, still 2^a 3^b 5^c) take two passes through `z`
// (n = n1' n2 + n2', k = k1 + n1 k2; blockIdx.y = frame):
// MODE 1: column pass - slots are fpw adjacent columns n2', transform over n1' (p.n = n1), times W_N^(n2' k1), z[k1][n2'];
// MODE 2: row pass - slots are fpw adjacent rows k1, transform over n2' (p.n = n2), bins k1 + n1 k2 leave as the dB rows.
template <int MODE>
__global__ void __launch_bounds__(1024) smooth_kernel(const SmoothParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int N = p.n, fpw = p.fpw, T = blockDim.x;
  const int Ns = MODE == 0 ? N : (N | 1);                // LDS stride of a slot (two passes: odd, neighbouring slots on other banks)
  c32* buf0 = reinterpret_cast<c32*>(smem);
  c32* buf1 = buf0 + fpw * Ns;
  const int f0 = blockIdx.x * fpw;                       // first frame (MODE 0) / column (1) / row (2) of this workgroup
  const int count = MODE == 0 ? p.n_frames : (MODE == 1 ? p.n2 : p.n1);
  const int nf = count - f0 < fpw ? count - f0 : fpw;
  const int NT = MODE == 0 ? N : p.n_total;              // points of a frame
  const int half = NT / 2;
  const int fy = MODE == 0 ? 0 : int(blockIdx.y);        // (two passes) the frame
  // per-frame mean removal of byte samples (dc_alpha = 1, the HackRF branch's default) without the sums kernel's launch:
  // exact integer I / Q sums of this workgroup's frames, the residual (2 sum - twice_zero n) / (2 n) formed in double exactly
  // as chirp_sums_kernel forms it - the same bits
  __shared__ float dc_re[16], dc_im[16];
  __shared__ unsigned red_i[16], red_q[16];
  if (MODE == 0 && p.dc_own) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = T >> 6;
    for (int fr = 0; fr < nf; ++fr) {
      const uint16_t* xs = reinterpret_cast<const uint16_t*>(static_cast<const unsigned char*>(p.in) + (long long)(f0 + fr) * p.frame_stride);
      unsigned ui = 0, uq = 0;
      for (int k = threadIdx.x; k < N; k += T) {
        const unsigned u = unsigned(xs[k]) ^ (p.xor_mask & 0xffffu);
        ui += u & 0xffu;
        uq += u >> 8;
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) { ui += __shfl_xor(ui, off); uq += __shfl_xor(uq, off); }
      if (lane == 0) { red_i[wave] = ui; red_q[wave] = uq; }
      __syncthreads();
      if (threadIdx.x == 0) {
        unsigned ti = 0, tq = 0;
        for (int w = 0; w < waves; ++w) { ti += red_i[w]; tq += red_q[w]; }
        const double dn = double(N), tz = double(p.twice_zero);
        const float rr = float((2.0 * double(ti) - tz * dn) / (2.0 * dn)), ri = float((2.0 * double(tq) - tz * dn) / (2.0 * dn));
        dc_re[fr] = rr;
        dc_im[fr] = ri;
        if (p.dc_state != nullptr && f0 + fr == p.n_frames - 1) *p.dc_state = float2{rr * p.in_scale, ri * p.in_scale};
      }
      __syncthreads();
    }
  }
  // sample k of frame slot fr: unpack, DC, window x input scale (slots past the call's last frame: zeros)
  const auto load_raw = [&](int fr, int kk) -> c32 {
    if (fr >= nf) return c32{0.f, 0.f};
    if constexpr (MODE == 2) return p.z[(long long)fy * NT + (long long)(f0 + fr) * p.n2 + kk];
    const int f = MODE == 0 ? f0 + fr : fy;
    const int k = MODE == 0 ? kk : kk * p.n2 + f0 + fr;  // (column pass: sample n1' n2 + n2')
    const unsigned char* fb = static_cast<const unsigned char*>(p.in) + (long long)f * p.frame_stride;
    float re, im;
    if (p.in_c64) {
      const c32 z = reinterpret_cast<const c32*>(fb)[k];
      re = z.x;
      im = z.y;
    } else {
      const unsigned u = unsigned(reinterpret_cast<const uint16_t*>(fb)[k]) ^ (p.xor_mask & 0xffffu);
      re = float(u & 0xffu) - p.in_off;                  // exact: small integers / halves
      im = float(u >> 8) - p.in_off;
    }
    if (MODE == 0 && p.dc_own) { re -= dc_re[fr]; im -= dc_im[fr]; }
    else if (p.dc_sub != nullptr) { const c32 d = p.dc_sub[f]; re -= d.x; im -= d.y; }
    const float w = p.window[k];
    return c32{re * w, im * w};
  };
  // bin k of frame slot fr: |X|^2, fftshift, dB + cal - tare / linear power (chirp_post_kernel's arithmetic)
  const auto store_bin = [&](int fr, int kk, c32 X) {
    if (fr >= nf) return;
    if constexpr (MODE == 1) {                           // column pass: times W_N^(n2' k1), to z[k1][n2']
      const int n2 = f0 + fr;
      p.z[(long long)fy * NT + (long long)kk * p.n2 + n2] = cmul_(X, p.tw[n2 * kk]);
      return;
    }
    const int k = MODE == 0 ? kk : f0 + fr + p.n1 * kk;  // (row pass: bin k1 + n1 k2)
    int j = k + half;                                    // np.fft.fftshift: bin k lands at (k + N / 2) mod N, any N
    if (j >= NT) j -= NT;
    const float pw = X.x * X.x + X.y * X.y;
    const long long o = (long long)(MODE == 0 ? f0 + fr : fy) * NT + j;
    if (p.out_lin != nullptr) {
      p.out_lin[o] = pw * p.pscale;
    } else {
      float db;
      if (p.db_mode == 0) db = fmaf(2.0f * kTenLog10Of2, __builtin_amdgcn_logf(__builtin_amdgcn_sqrtf(pw) + p.log_floor), p.cal_db);
      else db = fmaf(kTenLog10Of2, __builtin_amdgcn_logf(fmaf(pw, p.pscale, p.log_floor)), p.cal_db);
      if (p.tare != nullptr) db -= p.tare[j];
      p.out_db[o] = db;
    }
  };
  c32* x = buf0;
  c32* y = buf1;
  int sd = 1;                                            // s = N / n of the stage
  const int last = p.n_stages - 1;
  for (int st = 0; st <= last; ++st) {
    const int r = p.radix[st];
    const unsigned mp = p.magic_per[st], ms = p.magic_s[st];
    const c32* xs = x;
    c32* ys = y;
    const auto ld_lds = [=](int fr, int k) -> c32 { return xs[fr * Ns + k]; };
    const auto st_lds = [=](int fr, int k, c32 v) { ys[fr * Ns + k] = v; };
    const unsigned mf = p.magic_fpw;
    // column pass: its first stage reads and its last stage writes elements that are adjacent across the slots; row pass:
    // the last stage's bins are (the first stage reads along a row: adjacent across the butterflies)
    constexpr bool F0 = MODE == 1, FL = MODE != 0;
    if (st == 0 && st == last) stage_r<FL>(r, p.tw, p.tw_step, N, sd, fpw, T, mp, ms, mf, load_raw, store_bin);
    else if (st == 0) stage_r<F0>(r, p.tw, p.tw_step, N, sd, fpw, T, mp, ms, mf, load_raw, st_lds);
    else if (st == last) stage_r<FL>(r, p.tw, p.tw_step, N, sd, fpw, T, mp, ms, mf, ld_lds, store_bin);
    else stage_r<false>(r, p.tw, p.tw_step, N, sd, fpw, T, mp, ms, mf, ld_lds, st_lds);
    sd *= r;
    c32* t = x; x = y; y = t;
    if (st != last) __syncthreads();
  }
}

// about 2048 points per workgroup, two frames up to 2048 points each (N = 1500: 63 us per 4096 frames against 77 with one),
// one above (measured: profiles/r05_chirp.txt)
int smooth_frames_per_workgroup(int n) {
  int fpw = 2048 / n;
  if (fpw < 2 && n <= 2048) fpw = 2;
  if (fpw < 1) fpw = 1;
  if (fpw > 16) fpw = 16;
  return fpw;
}

static hipError_t launch_mode(const SmoothParams& p, int mode, dim3 grid, int threads, size_t lds, hipStream_t s) {
  // the largest request any plan makes (160 000 bytes: a 10 000-point frame), raised once per kernel and DEVICE
  // (ensure_dynamic_lds: function attributes belong to the device the module is loaded on; a process may hold plans on
  // several GPUs and launch from several threads)
  static std::atomic<unsigned long long> attr_done[3];
  const void* fn = mode == 0 ? reinterpret_cast<const void*>(smooth_kernel<0>)
                             : (mode == 1 ? reinterpret_cast<const void*>(smooth_kernel<1>) : reinterpret_cast<const void*>(smooth_kernel<2>));
  const hipError_t e = ensure_dynamic_lds(fn, 160 * 1000, attr_done[mode]);
  if (e != hipSuccess) return e;
  if (lds > size_t(160 * 1000)) return hipErrorInvalidValue;
  if (mode == 0) hipLaunchKernelGGL(smooth_kernel<0>, grid, dim3(threads), lds, s, p);
  else if (mode == 1) hipLaunchKernelGGL(smooth_kernel<1>, grid, dim3(threads), lds, s, p);
  else hipLaunchKernelGGL(smooth_kernel<2>, grid, dim3(threads), lds, s, p);
  return hipGetLastError();
}

// mode 0: p.n = frame length; modes 1 / 2: p.n = n1 / n2 with radix[] the stages of THAT length
hipError_t launch_smooth(SmoothParams p, hipStream_t s, int mode) {
  p.fpw = smooth_frames_per_workgroup(p.n);
  const auto magic = [](int d) { return d == 1 ? 0u : unsigned((1ull << 32) / unsigned(d)) + 1u; };
  int sd = 1;
  for (int st = 0; st < p.n_stages; ++st) {
    p.magic_per[st] = magic(p.n / p.radix[st]);
    p.magic_s[st] = magic(sd);
    sd *= p.radix[st];
  }
  // a thread per four points, whole waves, at most 1024
  int threads = ((p.fpw * p.n / 4 + 63) / 64) * 64;
  threads = threads < 64 ? 64 : (threads > 1024 ? 1024 : threads);
  p.magic_fpw = magic(p.fpw);
  const size_t lds = size_t(2) * p.fpw * (mode == 0 ? p.n : (p.n | 1)) * sizeof(float2);
  const int count = mode == 0 ? p.n_frames : (mode == 1 ? p.n2 : p.n1);
  const dim3 grid((count + p.fpw - 1) / p.fpw, mode == 0 ? 1 : p.n_frames);
  return launch_mode(p, mode, grid, threads, lds, s);
}

}  // namespace tdsa
