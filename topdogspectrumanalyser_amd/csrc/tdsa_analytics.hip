// tdsa_analytics.hip - what happens to the dB rows after the IQ -> spectrum path, kept on the device
// (SURVEY.md 8(f) f-3 / f-4): per-row peak / argmax / band power, the top-N peak list, the density
// histogram and the waterfall ring.  All kernels stream [rows][n] float32 rows that already sit in HBM
// (the frame kernel's output) and hand back scalars or a small image: HBM-bound by construction.
//
//   rows_stats_kernel     np.max / np.argmax (core/duty_cycle.py:36, core/marker_manager.py:97) and
//                         MarkerManager._band_power (core/marker_manager.py:308-319)
//   top_peaks_kernel      DataProcessor._find_top_peaks (core/display_data_processor.py:432-471)
//   density_kernel        DensityDisplay._update_hist (displays/density_display.py:306-318)
//   rows_differ_kernel /  Waterfall new-row test + _add_row (displays/waterfall.py:171-175, 330-336)
//   waterfall_scatter_kernel
#include "tdsa_kernels.hpp"

#include <math.h>

namespace tdsa {

namespace {

struct PeakPair {
  float v;
  int i;
};
// np.max / np.argmax order: NaN beats everything, then larger value, then smaller index
__device__ __forceinline__ bool better(PeakPair a, PeakPair b) {
  const bool an = a.v != a.v, bn = b.v != b.v;
  if (an || bn) return an && (!bn || a.i < b.i);
  return a.v > b.v || (a.v == b.v && a.i < b.i);
}
__device__ __forceinline__ PeakPair wave_best(PeakPair p) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    PeakPair q{__shfl_xor(p.v, o), __shfl_xor(p.i, o)};
    if (better(q, p)) p = q;
  }
  return p;
}
__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o);
  return x;
}
// the same reductions with DPP moves (no LDS round trip per step as with ds_bpermute); the result is valid in
// lane 63 only.  Lanes without a source lane keep their own value, which is the identity of min / "better".
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float x) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(x), __float_as_uint(x), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int x) {
  return __builtin_amdgcn_update_dpp(x, x, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ float wave_min_to_lane63(float x) {
  x = fminf(x, dpp_f<0xB1, 0xf>(x));     // quad_perm [1,0,3,2]
  x = fminf(x, dpp_f<0x4E, 0xf>(x));     // quad_perm [2,3,0,1]
  x = fminf(x, dpp_f<0x141, 0xf>(x));    // row_half_mirror
  x = fminf(x, dpp_f<0x140, 0xf>(x));    // row_mirror: every lane = row minimum
  x = fminf(x, dpp_f<0x142, 0xa>(x));    // row_bcast:15 -> rows 1, 3
  x = fminf(x, dpp_f<0x143, 0xc>(x));    // row_bcast:31 -> rows 2, 3
  return x;
}
// strongest candidate of the wave, equal values: larger index (the order of the reference's reversed ascending sort)
__device__ __forceinline__ PeakPair wave_strongest_to_lane63(PeakPair p) {
  auto step = [&](float qv, int qi) {
    if (qv > p.v || (qv == p.v && qi > p.i)) p = PeakPair{qv, qi};
  };
  step(dpp_f<0xB1, 0xf>(p.v), dpp_i<0xB1, 0xf>(p.i));
  step(dpp_f<0x4E, 0xf>(p.v), dpp_i<0x4E, 0xf>(p.i));
  step(dpp_f<0x141, 0xf>(p.v), dpp_i<0x141, 0xf>(p.i));
  step(dpp_f<0x140, 0xf>(p.v), dpp_i<0x140, 0xf>(p.i));
  step(dpp_f<0x142, 0xa>(p.v), dpp_i<0x142, 0xa>(p.i));
  step(dpp_f<0x143, 0xc>(p.v), dpp_i<0x143, 0xc>(p.i));
  return p;
}

// ---- per-row peak / argmax / band power ------------------------------------------------------------
__global__ void __launch_bounds__(256) rows_stats_kernel(const float* __restrict__ rows, int n, int band_lo,
                                                         int band_hi, double bin_width, float* peak_db,
                                                         int* peak_bin, double* band_db) {
  const float* row = rows + (size_t)blockIdx.x * n;
  PeakPair best{-INFINITY, 0x7fffffff};
  double bsum = 0.0;
  auto take = [&](float v, int i) {
    const PeakPair c{v, i};
    if (better(c, best)) best = c;
    // 10 ** (levels / 10) in float32 like numpy does for a float32 trace, accumulated wider
    if (i >= band_lo && i <= band_hi) bsum += (double)exp10f(v / 10.0f);
  };
  if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(row) & 15) == 0) {     // 16-byte loads when the row allows them
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4* row4 = reinterpret_cast<const f4*>(row);
    for (int j = threadIdx.x; j < n / 4; j += 256) {
      const f4 q = __builtin_nontemporal_load(row4 + j);                  // read once
      take(q.x, 4 * j); take(q.y, 4 * j + 1); take(q.z, 4 * j + 2); take(q.w, 4 * j + 3);
    }
  } else {
    for (int i = threadIdx.x; i < n; i += 256) take(row[i], i);
  }
  __shared__ PeakPair s_best[4];
  __shared__ double s_sum[4];
  best = wave_best(best);
  bsum = wave_sum(bsum);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    s_best[w] = best;
    s_sum[w] = bsum;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) {
      if (better(s_best[k], best)) best = s_best[k];
      bsum += s_sum[k];
    }
    if (peak_db) peak_db[blockIdx.x] = best.v;
    if (peak_bin) peak_bin[blockIdx.x] = best.i;
    if (band_db) {
      const double total = bsum * bin_width;
      // Python's max(total, 1e-30) keeps a NaN total (1e-30 > nan is False): so does `total < 1e-30 ? ... : total`
      band_db[blockIdx.x] = band_lo > band_hi ? NAN : 10.0 * log10(total < 1e-30 ? 1e-30 : total);
    }
  }
}

// ---- top-N peak list ----------------------------------------------------------------------------------
// One workgroup per row; the row and the candidate values live in LDS.  Candidates (strict interior local
// maxima) are visited from the strongest down exactly like the reference's sorted loop: every round is a
// block argmax over the live candidates plus one pass that forms the valley minimum against each peak
// accepted so far.  Ends as soon as n_peaks are accepted or the candidates run out.
constexpr int kPeakThreads = 1024;
constexpr int kMaxPeaks = 8;

__global__ void __launch_bounds__(kPeakThreads) top_peaks_kernel(const float* __restrict__ rows, int n, int n_peaks,
                                                                 int min_sep, float excursion, int* out_bins,
                                                                 float* out_db) {
  extern __shared__ float smem[];
  float* row = smem;             // [n]; the only large LDS array, so two rows are in flight per CU
  __shared__ PeakPair s_best[kPeakThreads / 64];
  __shared__ float s_min[kMaxPeaks][kPeakThreads / 64];
  __shared__ int s_sel[kMaxPeaks];
  __shared__ float s_selv[kMaxPeaks];
  __shared__ int s_nsel;

  const float* src = rows + (size_t)blockIdx.x * n;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < n; i += kPeakThreads) row[i] = src[i];
  if (tid == 0) s_nsel = 0;
  __syncthreads();
  // live candidates (strict interior local maxima) of this thread's elements i = tid + 1024 k: bit k of a register
  unsigned live = 0u;
  for (int i = tid, k = 0; i < n; i += kPeakThreads, ++k) {
    const bool is_max = i > 0 && i < n - 1 && row[i] > row[i - 1] && row[i] > row[i + 1];
    live |= is_max ? (1u << k) : 0u;
  }
  // strongest live candidate of this thread; equal values: larger index first (reversed ascending sort).  Only a
  // thread whose mask changed looks at the row again.
  PeakPair mine{-INFINITY, -1};
  auto rescan = [&] {
    mine = PeakPair{-INFINITY, -1};
    for (unsigned m = live; m != 0u; m &= m - 1u) {
      const int i = tid + kPeakThreads * __builtin_ctz(m);
      const float v = row[i];
      if (v > mine.v || (v == mine.v && i > mine.i)) mine = PeakPair{v, i};
    }
  };
  rescan();

  for (;;) {
    PeakPair best = wave_strongest_to_lane63(mine);
    if (lane == 63) s_best[w] = best;
    __syncthreads();
    // every thread folds the 16 wave results itself (LDS broadcasts): no serial section, one barrier less
    PeakPair gb = s_best[0];
#pragma unroll
    for (int k = 1; k < kPeakThreads / 64; ++k) {
      const PeakPair q = s_best[k];
      if (q.v > gb.v || (q.v == gb.v && q.i > gb.i)) gb = q;
    }
    const int cur = gb.v == -INFINITY ? -1 : gb.i;
    if (cur < 0) break;
    if (tid == (cur & (kPeakThreads - 1))) {      // the candidate leaves the list whatever happens to it
      live &= ~(1u << (cur / kPeakThreads));
      rescan();
    }
    const float curv = gb.v;
    const int nsel = s_nsel;
    // the separation test needs no valley: a candidate too close to an accepted peak is dropped right away
    bool too_close = false;
#pragma unroll
    for (int k = 0; k < kMaxPeaks; ++k) {
      if (k < nsel) {
        const int sk = s_sel[k];
        const int d = cur > sk ? cur - sk : sk - cur;
        too_close |= d < min_sep;
      }
    }
    if (too_close) {
      __syncthreads();                       // s_best is rewritten next round
      continue;
    }
    // valley minimum between the candidate and every accepted peak, one pass over the row; the ranges are formed
    // once per round (registers), peaks beyond nsel are skipped by uniform branches
    int lo[kMaxPeaks], hi[kMaxPeaks];
    float vmin[kMaxPeaks];
#pragma unroll
    for (int k = 0; k < kMaxPeaks; ++k) {
      const int sk = k < nsel ? s_sel[k] : cur;
      lo[k] = cur < sk ? cur : sk;
      hi[k] = cur < sk ? sk : cur;
      vmin[k] = INFINITY;
    }
    for (int i = tid; i < n; i += kPeakThreads) {
      const float v = row[i];
#pragma unroll
      for (int k = 0; k < kMaxPeaks; ++k) {
        if (k < nsel) {
          if (i >= lo[k] && i <= hi[k]) vmin[k] = fminf(vmin[k], v);   // (a NaN in the range would make np.min NaN
        }                                                               //  and never reject; rows here carry none)
      }
    }
#pragma unroll
    for (int k = 0; k < kMaxPeaks; ++k) {
      if (k < nsel) {
        const float m = wave_min_to_lane63(vmin[k]);
        if (lane == 63) s_min[k][w] = m;
      }
    }
    __syncthreads();
    if (w == 0) {
      // wave 0: lane l folds the 16 wave minima of every accepted peak (one LDS read + shuffles per peak)
      bool reject = false;
      for (int k = 0; k < nsel; ++k) {
        const float valley = __shfl(wave_min_to_lane63(lane < kPeakThreads / 64 ? s_min[k][lane] : INFINITY), 63);
        // reference arithmetic: power[idx] - valley is float32 - Python float (float32 under numpy >= 2),
        // sel_pwr - valley is Python float - Python float (double)
        if (curv - valley < excursion || (double)s_selv[k] - (double)valley < (double)excursion) reject = true;
      }
      if (lane == 0) {
        if (!reject) {
          s_sel[nsel] = cur;
          s_selv[nsel] = curv;
          s_nsel = nsel + 1;
        }
      }
    }
    __syncthreads();
    const int nsel_now = s_nsel;
    if (nsel_now >= n_peaks) break;
    if (nsel_now != nsel && min_sep > 1) {
      // accepted: every candidate closer than min_sep would be turned down when its turn came (the accepted set
      // only grows and a rejected candidate leaves no trace), so they go now instead of costing a round each
      const unsigned before = live;
      for (unsigned m = live; m != 0u; m &= m - 1u) {
        const int k = __builtin_ctz(m), i = tid + kPeakThreads * k;
        if (i > cur - min_sep && i < cur + min_sep) live &= ~(1u << k);
      }
      if (live != before) rescan();
    }
  }
  if (tid < n_peaks) {
    const bool have = tid < s_nsel;
    out_bins[(size_t)blockIdx.x * n_peaks + tid] = have ? s_sel[tid] : -1;
    if (out_db) out_db[(size_t)blockIdx.x * n_peaks + tid] = have ? s_selv[tid] : NAN;
  }
}

// ---- density histogram ----------------------------------------------------------------------------------
// hist[f][a] for kDensFreq neighbouring frequency bins per workgroup; thread a owns amplitude bin a of each.
// Rows are applied in order with the reference's float32 arithmetic: hist *= decay (when decay < 1), then
// += 1 at int32(((v - AMP_MIN) / AMP_RNG) * AMP_BINS) (truncation toward zero, NaN / out of range dropped).
constexpr int kAmpBins = 512;
constexpr int kDensFreq = 16;
constexpr float kAmpMin = -200.0f, kAmpRng = 300.0f;

constexpr int kDensRows = kAmpBins / kDensFreq;   // rows whose bin indices one pass of the workgroup forms

// Cost is VALU: every cell takes one multiply per row whatever happens, the question is what finding the ONE cell per
// (row, frequency bin) that also gets +1 costs.  The first version had every thread compare its amplitude bin with
// all 16 indices of every row (16 LDS reads + 16 compares + 16 selects per row: 2.1 ms per second of C3 spectra).
// Now the 512 threads of the workgroup scatter the 512 indices of a 32-row chunk into a bit table in LDS -
// word [row][a / 2] holds, for amplitude bins a and a + 1, one bit per frequency bin that hit them - and a thread
// reads ONE word per row; only waves in which some lane was hit (the few whose 64 amplitude bins cover the trace)
// run the 16 adds, the others do their 16 multiplies and move on.
__global__ void __launch_bounds__(kAmpBins) density_kernel(const float* __restrict__ rows, int n_rows, int n,
                                                           float decay, float* hist) {
  __shared__ unsigned s_hit[kDensRows][kAmpBins / 2];
  const int f0 = blockIdx.x * kDensFreq;
  const int a = threadIdx.x;
  const int rr = a / kDensFreq, jj = a % kDensFreq;     // this thread forms the index of (row r0 + rr, bin f0 + jj)
  const int sh = 16 * (a & 1);
  float h[kDensFreq];
#pragma unroll
  for (int j = 0; j < kDensFreq; ++j) h[j] = (f0 + j < n) ? hist[(size_t)(f0 + j) * kAmpBins + a] : 0.0f;
  for (int i = a; i < kDensRows * (kAmpBins / 2); i += kAmpBins) (&s_hit[0][0])[i] = 0u;
  const bool do_decay = decay < 1.0f;
  float decay_v = decay;
  asm volatile("" : "+v"(decay_v));      // a VALU op with an SGPR source issues at half rate on gfx950
  const bool col_ok = f0 + jj < n;
  float v_next = (rr < n_rows && col_ok) ? rows[(size_t)rr * n + f0 + jj] : NAN;
  for (int r0 = 0; r0 < n_rows; r0 += kDensRows) {
    int idx = -1;
    const float v = v_next;                       // fetched while the previous chunk was applied
    v_next = (r0 + kDensRows + rr < n_rows && col_ok) ? rows[(size_t)(r0 + kDensRows + rr) * n + f0 + jj] : NAN;
    if (r0 + rr < n_rows && col_ok) {
      const float x = (v - kAmpMin) / kAmpRng * float(kAmpBins);
      // astype(int32) truncates toward zero; NaN and anything outside [0, AMP_BINS) is dropped
      if (v == v && x > -1.0f && x < float(kAmpBins)) idx = int(x);
    }
    __syncthreads();                 // previous chunk fully consumed (every word read and cleared by its owners)
    if (idx >= 0) atomicOr(&s_hit[rr][idx >> 1], 1u << (jj + 16 * (idx & 1)));
    __syncthreads();
    const int lim = n_rows - r0 < kDensRows ? n_rows - r0 : kDensRows;
    constexpr int RB = 4;                                   // rows whose words are fetched together
    for (int rb = 0; rb < kDensRows; rb += RB) {
      if (rb >= lim) break;
      unsigned w[RB];
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        // lanes a and a ^ 1 share a word and a wave: both have read it before either clears it
        w[u] = s_hit[rb + u][a >> 1];
        s_hit[rb + u][a >> 1] = 0u;
      }
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        if (rb + u < lim) {
          const unsigned m = (w[u] >> sh) & 0xffffu;
          if (do_decay) {
#pragma unroll
            for (int j = 0; j < kDensFreq; ++j) h[j] = __fmul_rn(h[j], decay_v);   // `hist *= d`
          }
          if (__builtin_amdgcn_ballot_w64(m != 0u) != 0) {                        // `hist[f, idx] += 1`, its own rounding
            // bit j of m -> 0.0f or 1.0f without a select (v_cndmask with an implicit vcc mask is the slowest
            // VALU instruction of the chip); h + 0.0f leaves h as it is (h >= 0)
#pragma unroll
            for (int j = 0; j < kDensFreq; ++j) {
              const unsigned all = unsigned(__builtin_amdgcn_sbfe(int(m), j, 1));   // 0 or 0xffffffff
              h[j] = __fadd_rn(h[j], __uint_as_float(all & 0x3f800000u));
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < kDensFreq; ++j)
    if (f0 + j < n) hist[(size_t)(f0 + j) * kAmpBins + a] = h[j];
}

__global__ void log1p_kernel(const float* __restrict__ in, float* out, size_t count) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = log1pf(in[i]);
}

// ---- waterfall ring -------------------------------------------------------------------------------------
// differs[r] = !np.array_equal(rows[r], previous row)  (previous of row 0 = `last`, or "no previous")
__global__ void __launch_bounds__(256) rows_differ_kernel(const float* __restrict__ rows, const float* __restrict__ last,
                                                          int have_last, int n, int* differs) {
  const int r = blockIdx.x;
  const float* cur = rows + (size_t)r * n;
  const float* prev = r > 0 ? cur - n : last;
  int diff = (r == 0 && !have_last) ? 1 : 0;
  if (!diff)
    for (int i = threadIdx.x; i < n; i += 256) diff |= (cur[i] != prev[i]) ? 1 : 0;    // NaN != NaN, as in numpy
  diff = __syncthreads_or(diff);
  if (threadIdx.x == 0) differs[r] = diff;
}
// copy row r to ring rows dst[r] and dst[r] + H (dst[r] < 0: duplicate, skipped)
__global__ void __launch_bounds__(256) waterfall_scatter_kernel(const float* __restrict__ rows, const int* __restrict__ dst,
                                                                int n, int history, float* ring) {
  const int r = blockIdx.x;
  const int d = dst[r];
  if (d < 0) return;
  const float* src = rows + (size_t)r * n;
  float* a = ring + (size_t)d * n;
  float* b = ring + (size_t)(d + history) * n;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float v = src[i];
    a[i] = v;
    b[i] = v;
  }
}

}  // namespace

hipError_t launch_rows_stats(const float* rows, int n_rows, int n, int band_lo, int band_hi, double bin_width,
                             float* peak_db, int* peak_bin, double* band_db, hipStream_t s) {
  if (n_rows <= 0) return hipSuccess;
  rows_stats_kernel<<<n_rows, 256, 0, s>>>(rows, n, band_lo, band_hi, bin_width, peak_db, peak_bin, band_db);
  return hipGetLastError();
}

hipError_t launch_top_peaks(const float* rows, int n_rows, int n, int n_peaks, int min_sep, float excursion,
                            int* out_bins, float* out_db, hipStream_t s) {
  if (n_rows <= 0) return hipSuccess;
  const size_t lds = size_t(n) * sizeof(float);
  static std::atomic<unsigned long long> attr_done{0};
  const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(top_peaks_kernel), 72 * 1024, attr_done);
  if (e != hipSuccess) return e;
  top_peaks_kernel<<<n_rows, kPeakThreads, lds, s>>>(rows, n, n_peaks, min_sep, excursion, out_bins, out_db);
  return hipGetLastError();
}

hipError_t launch_density(const float* rows, int n_rows, int n, float decay, float* hist, hipStream_t s) {
  if (n_rows <= 0) return hipSuccess;
  density_kernel<<<(n + kDensFreq - 1) / kDensFreq, kAmpBins, 0, s>>>(rows, n_rows, n, decay, hist);
  return hipGetLastError();
}

hipError_t launch_log1p(const float* in, float* out, size_t count, hipStream_t s) {
  if (count == 0) return hipSuccess;
  log1p_kernel<<<unsigned((count + 255) / 256), 256, 0, s>>>(in, out, count);
  return hipGetLastError();
}

hipError_t launch_rows_differ(const float* rows, const float* last, int have_last, int n_rows, int n, int* differs,
                              hipStream_t s) {
  if (n_rows <= 0) return hipSuccess;
  rows_differ_kernel<<<n_rows, 256, 0, s>>>(rows, last, have_last, n, differs);
  return hipGetLastError();
}

hipError_t launch_waterfall_scatter(const float* rows, const int* dst, int n_rows, int n, int history, float* ring,
                                    hipStream_t s) {
  if (n_rows <= 0) return hipSuccess;
  waterfall_scatter_kernel<<<n_rows, 256, 0, s>>>(rows, dst, n, history, ring);
  return hipGetLastError();
}

}  // namespace tdsa
