// tdsa_analytics.hip - what happens to the dB rows after the IQ -> spectrum path, kept on the device
// (SURVEY.md 8(f) f-3 / f-4): per-row peak / argmax / band power, the top-N peak list, the density
// histogram and the waterfall ring.  All kernels stream [rows][n] float32 rows that already sit in HBM
// (the frame kernel's output) and hand back scalars or a small image: HBM-bound by construction.
//
//   rows_stats_kernel     np.max / np.argmax (core/duty_cycle.py:36, core/marker_manager.py:97) and
//                         MarkerManager._band_power (core/marker_manager.py:308-319)
//   top_peaks_kernel      DataProcessor._find_top_peaks (core/display_data_processor.py:432-471)
//   marker_peaks_kernel   MarkerManager.snap_to_peak / snap_to_next_peak = scipy find_peaks(height, prominence,
//                         distance) (core/marker_manager.py:74-127)
//   density_kernel        DensityDisplay._update_hist (displays/density_display.py:306-318)
//   rows_differ_kernel /  Waterfall new-row test + _add_row (displays/waterfall.py:171-175, 330-336)
//   waterfall_plan_kernel / waterfall_scatter_kernel
#include "tdsa_fft.hpp"        // static_for
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
__device__ __forceinline__ PeakPair row_strongest(PeakPair p) {      // every lane: the strongest of its row of 16 lanes
  auto step = [&](float qv, int qi) {
    if (qv > p.v || (qv == p.v && qi > p.i)) p = PeakPair{qv, qi};
  };
  step(dpp_f<0xB1, 0xf>(p.v), dpp_i<0xB1, 0xf>(p.i));
  step(dpp_f<0x4E, 0xf>(p.v), dpp_i<0x4E, 0xf>(p.i));
  step(dpp_f<0x141, 0xf>(p.v), dpp_i<0x141, 0xf>(p.i));
  step(dpp_f<0x140, 0xf>(p.v), dpp_i<0x140, 0xf>(p.i));
  return p;
}
__device__ __forceinline__ float row_min(float x) {                   // every lane: the minimum of its row of 16 lanes
  x = fminf(x, dpp_f<0xB1, 0xf>(x));
  x = fminf(x, dpp_f<0x4E, 0xf>(x));
  x = fminf(x, dpp_f<0x141, 0xf>(x));
  x = fminf(x, dpp_f<0x140, 0xf>(x));
  return x;
}
__device__ __forceinline__ PeakPair wave_strongest_to_lane63(PeakPair p) {
  auto step = [&](float qv, int qi) {
    if (qv > p.v || (qv == p.v && qi > p.i)) p = PeakPair{qv, qi};
  };
  p = row_strongest(p);
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
    if (band_db && bin_width < 0.0) {
      band_db[blockIdx.x] = bsum;           // the band's linear sum itself (per-frame scalars of plans without the fused epilogue)
    } else if (band_db) {
      const double total = bsum * bin_width;
      // Python's max(total, 1e-30) keeps a NaN total (1e-30 > nan is False): so does `total < 1e-30 ? ... : total`
      band_db[blockIdx.x] = band_lo > band_hi ? NAN : 10.0 * log10(total < 1e-30 ? 1e-30 : total);
    }
  }
}

// ---- per-frame scalars left by the frame kernel's STATS epilogue ------------------------------------------
// One 16-byte record per wave and frame: {max dB of the wave's bins, the first display position among them that holds it
// (bit 30 set: the position of the first NaN instead - np.max is then NaN, np.argmax that bin), linear band power of its
// bins, 0}.  Thread f folds its frame's waves in wave order: np.max / np.argmax rules (`better`), the band in float64
// times 10^(cal / 10) (the records carry the power BEFORE the calibration offset, the rows after).  Reads nothing but
// the records: it runs when the scalars are asked for (tdsa_get_frame_stats), not behind every frame-kernel launch.
__global__ void __launch_bounds__(256) frame_stats_finish_kernel(const uint4* __restrict__ parts, int n_frames,
                                                                 int wpf, double cal_lin, float* peak_db,
                                                                 int* peak_bin, double* band_lin) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= n_frames) return;
  PeakPair best{-INFINITY, 0x7fffffff};
  double bsum = 0.0;
  for (int w = 0; w < wpf; ++w) {
    const uint4 h = parts[(size_t)f * wpf + w];
    bsum += (double)__uint_as_float(h.z);
    const PeakPair c{(h.y & 0x40000000u) ? NAN : __uint_as_float(h.x), int(h.y & 0x3fffffffu)};
    if (better(c, best)) best = c;
  }
  peak_db[f] = best.v;
  peak_bin[f] = best.i;
  band_lin[f] = bsum * cal_lin;
}

// ---- top-N peak list ----------------------------------------------------------------------------------
// One workgroup per row; the row and the candidate values live in LDS.  Candidates (strict interior local
// maxima) are visited from the strongest down exactly like the reference's sorted loop: every round is a
// block argmax over the live candidates plus one pass that forms the valley minimum against each peak
// accepted so far.  Ends as soon as n_peaks are accepted or the candidates run out.
constexpr int kPeakThreads = 1024;                  // threads per row at the largest rows; smaller rows take fewer (below)
constexpr int kMarkMaxNTop = 16384;                 // rows_top_peaks: n_bins <= 16384 (the row lives in LDS)
constexpr int kMaxPeaks = 8;
constexpr int kPeakVals = kMarkMaxNTop / kPeakThreads;   // bins per thread

// LOG2T: 2^LOG2T threads per row, 16 bins per thread at most - 64 threads for rows up to 1024 bins, ... 256 up to 4096 (eight rows per
// CU), 512 up to 8192, 1024 above: the rounds are latency, and the rows in flight per CU are what hides it (a row of 1024 bins took as long as one
// of 8192 while every row had 1024 threads)
template <int LOG2T>
__global__ void __launch_bounds__(1 << LOG2T) __attribute__((amdgpu_waves_per_eu(8, 8))) top_peaks_kernel(const float* __restrict__ rows, int n, int n_peaks,
                                                                 int min_sep, float excursion, int* out_bins,
                                                                 float* out_db) {
  constexpr int T = 1 << LOG2T, W = T / 64;
  extern __shared__ float smem[];
  float* row = smem;             // [n]; the only large LDS array, so two rows are in flight per CU
  __shared__ PeakPair s_best[kPeakThreads / 64];
  __shared__ float s_min[kMaxPeaks][kPeakThreads / 64];
  __shared__ int s_sel[kMaxPeaks];
  __shared__ float s_selv[kMaxPeaks];
  __shared__ float s_sel_upto[kMaxPeaks], s_sel_from[kMaxPeaks];   // minimum of an accepted peak's block up to / from the peak
  __shared__ int s_nsel;

  __shared__ float s_bmin[kMarkMaxNTop / 32];       // minimum of every 32 bins (a half-wave's run while the row is loaded)
  const float* src = rows + (size_t)blockIdx.x * n;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  float vals[kPeakVals];                              // the thread's own bins i = tid + 1024 k stay in registers
#pragma unroll
  for (int k = 0; k < kPeakVals; ++k) {
    const int i = tid + T * k;
    vals[k] = i < n ? src[i] : INFINITY;
  }
#pragma unroll
  for (int k = 0; k < kPeakVals; ++k) {
    const int i = tid + T * k;
    if (T * k < n) {                       // (uniform)
      if (i < n) row[i] = vals[k];
      float mn = row_min(vals[k]);
      mn = fminf(mn, dpp_f<0x142, 0xa>(mn));          // row_bcast:15: lanes 31 and 63 hold the minimum of their 32 bins
      if ((lane & 31) == 31 && (i & ~31) < n) s_bmin[i >> 5] = mn;
    }
  }
  if (tid == 0) s_nsel = 0;
  __syncthreads();
  // live candidates (strict interior local maxima) of this thread's elements i = tid + 1024 k: bit k of a register
  unsigned live = 0u;
#pragma unroll
  for (int k = 0; k < kPeakVals; ++k) {
    const int i = tid + T * k;
    const bool is_max = i > 0 && i < n - 1 && vals[k] > row[i - 1] && vals[k] > row[i + 1];
    live |= is_max ? (1u << k) : 0u;
  }
  // strongest live candidate of this thread; equal values: larger index first (reversed ascending sort).  Only a
  // thread whose mask changed looks at the row again.
  PeakPair mine{-INFINITY, -1};
  auto rescan = [&] {                                 // from the registers, highest bin first: a strict compare keeps the larger index
    mine = PeakPair{-INFINITY, -1};
#pragma unroll
    for (int k = kPeakVals - 1; k >= 0; --k) {
      if (((live >> k) & 1u) && vals[k] > mine.v) mine = PeakPair{vals[k], tid + T * k};
    }
  };
  rescan();

  for (;;) {
    PeakPair best = wave_strongest_to_lane63(mine);
    if (lane == 63) s_best[w] = best;
    __syncthreads();
    // every row of 16 lanes folds the 16 wave results itself (one LDS read per lane, four DPP steps): no serial section
    const PeakPair gb = row_strongest(s_best[lane & (W - 1)]);
    const int cur = __builtin_amdgcn_readfirstlane(gb.v == -INFINITY ? -1 : gb.i);   // (the same in every lane: scalar registers)
    if (cur < 0) break;
    if (tid == (cur & (T - 1))) {      // the candidate leaves the list whatever happens to it
      live &= ~(1u << (cur / T));
      rescan();
    }
    const float curv = gb.v;
    const int nsel = __builtin_amdgcn_readfirstlane(s_nsel);
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
    int lo[kMaxPeaks], hi[kMaxPeaks];                 // in blocks of 32 bins: the blocks that hold the two peaks
    float vmin[kMaxPeaks];
#pragma unroll
    for (int k = 0; k < kMaxPeaks; ++k) {
      const int sk = k < nsel ? __builtin_amdgcn_readfirstlane(s_sel[k]) : cur;
      lo[k] = (cur < sk ? cur : sk) >> 5;
      hi[k] = (cur < sk ? sk : cur) >> 5;
      vmin[k] = INFINITY;
    }
    // (until round 6 every thread walked its 16 bins of the row against every range: 200 of a round's ~400 instructions
    //  per wave; then one thread per block of 32 bins took the block's minimum where the block lay inside a range and walked
    //  it at a range's two ends - up to 62 dependent LDS reads by one thread, the longest chain of a round.  Now the threads
    //  take whole blocks strictly between the two peaks' blocks only; the two end blocks are wave 0's, below)
    for (int b = tid; b < (n + 31) / 32; b += T) {
      const float bm = s_bmin[b];
#pragma unroll
      for (int k = 0; k < kMaxPeaks; ++k) {
        if (k < nsel && b > lo[k] && b < hi[k]) vmin[k] = fminf(vmin[k], bm);   // (a NaN in the range would make np.min
      }                                                                          //  NaN and never reject; rows here carry none)
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
      // the candidate's own block, a half-wave per side: lanes 0 - 31 the bins up to the candidate, lanes 32 - 63 the bins
      // from it on (the part of a valley that lies in this block, whichever side the accepted peak is on); an accepted peak
      // keeps the two minima it had as a candidate
      const int cidx = (cur & ~31) + (lane & 31);
      const float cv = cidx < n ? row[cidx] : INFINITY;
      float part = (lane < 32 ? cidx <= cur : cidx >= cur) ? cv : INFINITY;
      part = fminf(part, dpp_f<0xB1, 0xf>(part));
      part = fminf(part, dpp_f<0x4E, 0xf>(part));
      part = fminf(part, dpp_f<0x141, 0xf>(part));
      part = fminf(part, dpp_f<0x140, 0xf>(part));
      part = fminf(part, dpp_f<0x142, 0xa>(part));          // row_bcast:15: lanes 31 and 63 hold their half's minimum
      const float cur_upto = __shfl(part, 31), cur_from = __shfl(part, 63);
      // four accepted peaks at a time, one per row of 16 lanes: the row folds the peak's 16 wave minima, adds the end blocks
      bool rej = false;
      for (int base = 0; base < nsel; base += 4) {
        const int kq = base + (lane >> 4);
        const bool valid = kq < nsel;
        const int k = valid ? kq : 0;
        float valley = row_min(valid && (lane & 15) < W ? s_min[k][lane & 15] : INFINITY);
        const int sk = s_sel[k];
        valley = fminf(valley, cur < sk ? fminf(cur_from, s_sel_upto[k]) : fminf(cur_upto, s_sel_from[k]));
        const bool same = valid && (sk >> 5) == (cur >> 5);
        if (__builtin_amdgcn_ballot_w64(same) != 0ull) {      // both peaks in one block of 32: the bins between them
          for (int q = 0; q < 4; ++q) {
            const int k2 = base + q;
            if (k2 < nsel && (s_sel[k2] >> 5) == (cur >> 5)) {
              const int s2 = s_sel[k2], a = cur < s2 ? cur : s2, z = cur < s2 ? s2 : cur;
              const float m = __shfl(wave_min_to_lane63(cidx >= a && cidx <= z ? cv : INFINITY), 63);
              if ((lane >> 4) == q) valley = m;
            }
          }
        }
        // reference arithmetic: power[idx] - valley is float32 - Python float (float32 under numpy >= 2),
        // sel_pwr - valley is Python float - Python float (double)
        if (valid && (curv - valley < excursion || (double)s_selv[k] - (double)valley < (double)excursion)) rej = true;
      }
      const bool reject = __builtin_amdgcn_ballot_w64(rej) != 0ull;
      if (lane == 0) {
        if (!reject) {
          s_sel[nsel] = cur;
          s_selv[nsel] = curv;
          s_sel_upto[nsel] = cur_upto;
          s_sel_from[nsel] = cur_from;
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
      if (2 * min_sep - 1 <= T) {
        // the window (cur - min_sep, cur + min_sep) is no wider than the stride of a thread's bins: it holds at most one of them
        const int last = cur + min_sep - 1, k = (last - tid) >> LOG2T, i = tid + T * k;
        if (last >= tid && k < kPeakVals && i > cur - min_sep) live &= ~(1u << k);
      } else {
        for (unsigned m = live; m != 0u; m &= m - 1u) {
          const int k = __builtin_ctz(m), i = tid + T * k;
          if (i > cur - min_sep && i < cur + min_sep) live &= ~(1u << k);
        }
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

// ---- marker peak search ----------------------------------------------------------------------------------
// MarkerManager.snap_to_peak / snap_to_next_peak (core/marker_manager.py:74-127), i.e.
// scipy.signal.find_peaks(levels, height=threshold, prominence=excursion, distance=3) and what the two methods pick
// from its result.  One workgroup per row, the row in LDS.  scipy's conditions in scipy's order:
//   1. local maxima: strict rise before, strict fall after, a flat top counts once at (left + right) / 2; the first
//      and last sample never count (scipy _local_maxima_1d)
//   2. height: x[peak] >= height
//   3. distance: from the highest peak down, a kept peak removes every peak closer than `distance`
//      (_select_by_peak_distance).  That greedy order is a fixed point: a peak is removed iff a KEPT peak of higher
//      priority is within reach, kept iff every higher-priority peak within reach is removed - rounds of local
//      decisions reach it without sorting (a round settles at least the highest undecided peak; random traces take
//      3-4 rounds).  Priority = (value, index): between equal peaks the larger index first (a stable sort walked
//      backwards; numpy's default argsort, which scipy calls, leaves that order to the CPU's sorting network).
//   4. prominence: walk outwards while the samples are <= the peak, lowest sample met on each side = its base,
//      prominence = peak - higher base >= prominence (_peak_prominences, wlen = None), in float64 like scipy.
//      The walk skips whole blocks of 32 / 1024 samples whose maximum does not exceed the peak (block maxima and
//      minima formed while the row is loaded), so the one strongest peak of a row costs ~150 steps instead of N.
constexpr int kMarkThreads = 512;
constexpr int kMarkMaxN = 16384;
constexpr int kMarkWords = kMarkMaxN / 32;

struct MarkerLds {
  float bmax1[kMarkWords], bmin1[kMarkWords];         // per 32 samples; a NaN counts as +inf in the maximum
  float bmax2[kMarkMaxN / 1024], bmin2[kMarkMaxN / 1024];
  unsigned cand[kMarkWords], kept[kMarkWords], gone[kMarkWords], fin[kMarkWords];
  int red_i[kMarkThreads / 64][4];
  float red_f[kMarkThreads / 64], red_a[kMarkThreads / 64];
  int scan[kMarkThreads / 64];
};

__device__ __forceinline__ double marker_prominence(const float* row, const MarkerLds& L, int n, int p) {
  const float xp = row[p];
  float lmin = xp, rmin = xp;
  int i = p - 1;
  while (i >= 0) {
    if ((i & 31) == 31) {                                   // a whole block of 32 (1024) ends here
      if ((i & 1023) == 1023 && L.bmax2[i >> 10] <= xp) { lmin = fminf(lmin, L.bmin2[i >> 10]); i -= 1024; continue; }
      if (L.bmax1[i >> 5] <= xp) { lmin = fminf(lmin, L.bmin1[i >> 5]); i -= 32; continue; }
    }
    const float v = row[i];
    if (!(v <= xp)) break;
    lmin = fminf(lmin, v);
    --i;
  }
  i = p + 1;
  while (i < n) {
    if ((i & 31) == 0) {                                    // a whole block starts here (a ragged last block is padded
      if ((i & 1023) == 0 && L.bmax2[i >> 10] <= xp) { rmin = fminf(rmin, L.bmin2[i >> 10]); i += 1024; continue; }   // with the identities)
      if (L.bmax1[i >> 5] <= xp) { rmin = fminf(rmin, L.bmin1[i >> 5]); i += 32; continue; }
    }
    const float v = row[i];
    if (!(v <= xp)) break;
    rmin = fminf(rmin, v);
    ++i;
  }
  return (double)xp - (double)fmaxf(lmin, rmin);
}

// The FILTER needs only `prominence >= threshold`, and that is decided long before the walk ends: the difference
// (double)xp - (double)v falls as v rises, so `xp - min(side) >= threshold` holds iff SOME sample of the side's walk has
// `xp - v >= threshold` - the walk of a side stops at the first such sample (side passed), or at the first sample above
// the peak / NaN / the end of the row (side failed: the peak is out).  A noise peak is settled in a handful of steps,
// a carrier in two or three (the window's skirt); only ripple of less than the threshold walks far, in blocks.
//
// The test per sample is ONE float compare: `(double)xp - (double)v >= threshold` holds for exactly the floats v <= thr, thr =
// the largest float that passes (the difference is monotone in v) - found once per peak next to float(xp - threshold).  And the
// first kMarkNear steps of a side run as a fixed-length predicated loop (every lane the same trip count, no exec juggling):
// the walks of the lanes of a wave ended within it in all but a few percent of the peaks, and the divergent loop with its
// block skipping - some 40 instructions per step - was where the kernel's time went.
constexpr int kMarkNear = 6;
__device__ __forceinline__ bool marker_prominent(const float* row, const MarkerLds& L, int n, int p, double prominence) {
  if (0.0 >= prominence) return true;                        // the bases start at the peak itself: prominence >= 0
  const float xp = row[p];
  const double dxp = (double)xp;
  auto passes = [&](float v) { return dxp - (double)v >= prominence; };
  float thr = (float)(dxp - prominence);
  if (!passes(thr)) thr = nextafterf(thr, -INFINITY);
  if (!passes(thr)) thr = nextafterf(thr, -INFINITY);
  if (!passes(thr)) return false;                            // (a NaN threshold, or none of the floats around it: nothing can pass)
  for (int k = 0; k < 2; ++k) {
    const float up = nextafterf(thr, INFINITY);
    if (up != thr && passes(up)) thr = up;
  }
  // state of a side: 0 walking, 1 passed, 2 failed
  int st = 0, i = p - 1;
#pragma unroll
  for (int k = 0; k < kMarkNear; ++k) {
    const float v = row[i < 0 ? 0 : i];
    const int now = (i < 0 || !(v <= xp)) ? 2 : (v <= thr ? 1 : 0);
    st = st == 0 ? now : st;
    --i;
  }
  if (st == 0) {
    while (i >= 0) {
      if ((i & 31) == 31) {
        if ((i & 1023) == 1023 && L.bmax2[i >> 10] <= xp) {
          if (L.bmin2[i >> 10] <= thr) { st = 1; break; }
          i -= 1024; continue;
        }
        if (L.bmax1[i >> 5] <= xp) {
          if (L.bmin1[i >> 5] <= thr) { st = 1; break; }
          i -= 32; continue;
        }
      }
      const float v = row[i];
      if (!(v <= xp)) break;
      if (v <= thr) { st = 1; break; }
      --i;
    }
  }
  if (st != 1) return false;
  st = 0;
  i = p + 1;
#pragma unroll
  for (int k = 0; k < kMarkNear; ++k) {
    const float v = row[i >= n ? n - 1 : i];
    const int now = (i >= n || !(v <= xp)) ? 2 : (v <= thr ? 1 : 0);
    st = st == 0 ? now : st;
    ++i;
  }
  if (st == 0) {
    while (i < n) {
      if ((i & 31) == 0) {
        if ((i & 1023) == 0 && L.bmax2[i >> 10] <= xp) {
          if (L.bmin2[i >> 10] <= thr) return true;
          i += 1024; continue;
        }
        if (L.bmax1[i >> 5] <= xp) {
          if (L.bmin1[i >> 5] <= thr) return true;
          i += 32; continue;
        }
      }
      const float v = row[i];
      if (!(v <= xp)) return false;
      if (v <= thr) return true;
      ++i;
    }
    return false;
  }
  return st == 1;
}

__global__ void __launch_bounds__(kMarkThreads) marker_peaks_kernel(const float* __restrict__ rows, int n, double height,
                                                                    double prominence, int distance, int current_idx,
                                                                    int max_list, int* out_count, int* out_snap,
                                                                    int* out_next, int* out_bins, double* out_prom) {
  extern __shared__ __attribute__((aligned(16))) float mark_smem[];
  float* row = mark_smem;                                        // [n]
  __shared__ MarkerLds L;
  const float* src = rows + (size_t)blockIdx.x * n;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nw = (n + 31) >> 5, nw2 = (n + 1023) >> 10;

  // load + block extrema of 32: a half-wave holds one block
  if ((n & 3) == 0) {
    // four samples per lane, every load of the row in flight before the first is used; eight lanes hold a block of 32
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4* src4 = reinterpret_cast<const f4*>(src);
    f4* row4 = reinterpret_cast<f4*>(row);
    const int nc = n >> 2;
    constexpr int kChunks = kMarkMaxN / 4 / kMarkThreads;
    f4 q[kChunks];
#pragma unroll
    for (int k = 0; k < kChunks; ++k) {
      const int c = tid + kMarkThreads * k;
      q[k] = c < nc ? __builtin_nontemporal_load(src4 + c) : f4{0.0f, 0.0f, 0.0f, 0.0f};
    }
#pragma unroll
    for (int k = 0; k < kChunks; ++k) {
      const int c = tid + kMarkThreads * k;
      const bool in = c < nc;
      float mx = -INFINITY, mn = INFINITY;
      if (in) {
        row4[c] = q[k];
        const float e[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          mx = fmaxf(mx, e[j] != e[j] ? INFINITY : e[j]);
          mn = fminf(mn, e[j]);
        }
      }
      // eight lanes hold a block of 32 bins: two quad steps and the mirror of each half-row (DPP, no LDS round trips)
      mx = fmaxf(mx, dpp_f<0xB1, 0xf>(mx)); mn = fminf(mn, dpp_f<0xB1, 0xf>(mn));
      mx = fmaxf(mx, dpp_f<0x4E, 0xf>(mx)); mn = fminf(mn, dpp_f<0x4E, 0xf>(mn));
      mx = fmaxf(mx, dpp_f<0x141, 0xf>(mx)); mn = fminf(mn, dpp_f<0x141, 0xf>(mn));
      if ((lane & 7) == 0 && (c >> 3) < nw) {
        L.bmax1[c >> 3] = mx;
        L.bmin1[c >> 3] = mn;
      }
    }
  } else
  for (int i0 = 0; i0 < n; i0 += kMarkThreads) {
    const int i = i0 + tid;
    const bool in = i < n;
    const float v = in ? src[i] : 0.0f;
    if (in) {
      row[i] = v;
    }
    float mx = in ? (v != v ? INFINITY : v) : -INFINITY, mn = in ? v : INFINITY;
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) {
      mx = fmaxf(mx, __shfl_xor(mx, o));
      mn = fminf(mn, __shfl_xor(mn, o));                   // fminf skips NaN: such a block is never skipped anyway
    }
    if ((lane & 31) == 0 && (i >> 5) < nw) {
      L.bmax1[i >> 5] = mx;
      L.bmin1[i >> 5] = mn;
    }
  }
  for (int w = tid; w < kMarkWords; w += kMarkThreads) L.cand[w] = L.kept[w] = L.gone[w] = L.fin[w] = 0u;
  __syncthreads();
  for (int b = tid; b < nw2; b += kMarkThreads) {
    float mx = -INFINITY, mn = INFINITY;
    for (int k = 32 * b; k < 32 * b + 32 && k < nw; ++k) {
      mx = fmaxf(mx, L.bmax1[k]);
      mn = fminf(mn, L.bmin1[k]);
    }
    L.bmax2[b] = mx;
    L.bmin2[b] = mn;
  }
  // 1 + 2: a bin above both neighbours is a peak (one ballot per 64 bins, a half-wave = one word of the bit set); the thread
  // of the rising left edge of a FLAT top - rare - walks it (whole equal blocks at a time) and marks the middle
  for (int i0 = 0; i0 < n; i0 += kMarkThreads) {
    const int i = i0 + tid;
    bool simple = false, flat = false;
    float v = 0.0f;
    if (i >= 1 && i <= n - 2) {
      v = row[i];
      const float l = row[i - 1], r = row[i + 1];
      simple = l < v && r < v && (double)v >= height;
      flat = l < v && r == v;
    }
    const unsigned long long hits = __builtin_amdgcn_ballot_w64(simple);
    const unsigned half = unsigned(hits >> (lane & 32));
    if ((lane & 31) == 0 && half != 0u) atomicOr(&L.cand[i >> 5], half);     // (bins beyond n never hit)
    if (flat) {
      int a = i + 1;
      while (a < n - 1) {
        if ((a & 31) == 0 && a + 32 < n - 1 && L.bmax1[a >> 5] == v && L.bmin1[a >> 5] == v) { a += 32; continue; }
        if (row[a] != v) break;
        ++a;
      }
      if (row[a] < v && (double)v >= height) {
        const int mid = (i + a - 1) >> 1;
        atomicOr(&L.cand[mid >> 5], 1u << (mid & 31));
      }
    }
  }
  __syncthreads();
  // 3: distance rule (peaks are never adjacent: distance <= 2 removes nothing)
  if (distance > 2) {
    const int reach = distance - 1;
    if (reach < 32) {
      // a peak with no other peak within reach stays, whatever its height: settled by shifts of the bit set before the rounds
      for (int w = tid; w < nw; w += kMarkThreads) {
        const unsigned cw = L.cand[w], below = w > 0 ? L.cand[w - 1] : 0u, above = w + 1 < nw ? L.cand[w + 1] : 0u;
        unsigned near = 0u;
        for (int d = 2; d <= reach; ++d) near |= (cw << d) | (cw >> d) | (below >> (32 - d)) | (above << (32 - d));
        L.kept[w] = cw & ~near;
      }
      __syncthreads();
    }
    for (;;) {
      int open = 0;
      for (int w = tid; w < nw; w += kMarkThreads) {
        unsigned k = L.kept[w], g = L.gone[w];
        for (unsigned m = L.cand[w] & ~k & ~g; m != 0u; m &= m - 1u) {
          const int b = __builtin_ctz(m), p = 32 * w + b;
          const float xp = row[p];
          bool removed = false, wait = false;
          const int q0 = p - reach < 0 ? 0 : p - reach, q1 = p + reach > n - 1 ? n - 1 : p + reach;
          auto look = [&](int q) {
            if (!((L.cand[q >> 5] >> (q & 31)) & 1u)) return;
            const float xq = row[q];
            if (!(xq > xp || (xq == xp && q > p))) return;              // lower priority: it waits for us
            if ((L.kept[q >> 5] >> (q & 31)) & 1u) removed = true;
            else if (!((L.gone[q >> 5] >> (q & 31)) & 1u)) wait = true;
          };
          // (the bins next to a peak are never peaks: with scipy's default distance 3 one bin is looked at on each side)
          for (int q = q0; q <= p - 2; ++q) look(q);
          for (int q = p + 2; q <= q1; ++q) look(q);
          if (removed) g |= 1u << b;
          else if (!wait) k |= 1u << b;
          else open = 1;
        }
        // a thread writes only its own words; a neighbour that still reads the old ones decides a round later
        L.kept[w] = k;
        L.gone[w] = g;
      }
      if (!__syncthreads_or(open)) break;
    }
  } else {
    for (int w = tid; w < nw; w += kMarkThreads) L.kept[w] = L.cand[w];
    __syncthreads();
  }
  // 4: prominence
  int mine = 0;                                              // peaks of this thread's words
  int first = 0x7fffffff, next = 0x7fffffff;
  PeakPair top{-INFINITY, 0x7fffffff};
  for (int w = tid; w < nw; w += kMarkThreads) {
    unsigned f = 0u;
    for (unsigned m = L.kept[w]; m != 0u; m &= m - 1u) {
      const int b = __builtin_ctz(m), p = 32 * w + b;
      if (marker_prominent(row, L, n, p, prominence)) {
        f |= 1u << b;
        const PeakPair c{row[p], p};
        if (c.v > top.v || (c.v == top.v && c.i < top.i)) top = c;        // np.argmax of the heights: first of equals
        first = p < first ? p : first;
        if (p > current_idx && p < next) next = p;
      }
    }
    L.fin[w] = f;
    mine += __builtin_popcount(f);
  }
  // block results: count (and the exclusive scan the list needs), highest peak, first peak, first peak right of the marker
  int incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const PeakPair q{__shfl_xor(top.v, o), __shfl_xor(top.i, o)};
    if (q.v > top.v || (q.v == top.v && q.i < top.i)) top = q;
    const int f2 = __shfl_xor(first, o), n2 = __shfl_xor(next, o);
    first = f2 < first ? f2 : first;
    next = n2 < next ? n2 : next;
  }
  if (lane == 63) L.scan[wv] = incl;
  if (lane == 0) {
    L.red_i[wv][0] = top.i;
    L.red_i[wv][1] = first;
    L.red_i[wv][2] = next;
    L.red_f[wv] = top.v;
  }
  __syncthreads();
  int base = 0, total = 0;
#pragma unroll
  for (int k = 0; k < kMarkThreads / 64; ++k) {
    const int c = L.scan[k];
    base += k < wv ? c : 0;
    total += c;
  }
  if (total == 0 && out_snap) {
    // no peak at all: snap_to_peak falls back to np.argmax of the row (marker_manager.py:97) - formed only then
    PeakPair amax{-INFINITY, 0x7fffffff};
    for (int i = tid; i < n; i += kMarkThreads) {
      const PeakPair c{row[i], i};
      if (better(c, amax)) amax = c;
    }
    amax = wave_best(amax);
    if (lane == 0) { L.red_i[wv][3] = amax.i; L.red_a[wv] = amax.v; }
    __syncthreads();                                        // (total is the same in every thread)
  }
  if (tid == 0) {
    PeakPair am{-INFINITY, 0x7fffffff};
    if (total == 0 && out_snap) {
      am = PeakPair{L.red_a[0], L.red_i[0][3]};
      for (int k = 1; k < kMarkThreads / 64; ++k) {
        const PeakPair a{L.red_a[k], L.red_i[k][3]};
        if (better(a, am)) am = a;
      }
    }
    for (int k = 1; k < kMarkThreads / 64; ++k) {
      const PeakPair t{L.red_f[k], L.red_i[k][0]};
      if (t.v > top.v || (t.v == top.v && t.i < top.i)) top = t;
      first = L.red_i[k][1] < first ? L.red_i[k][1] : first;
      next = L.red_i[k][2] < next ? L.red_i[k][2] : next;
    }
    if (out_count) out_count[blockIdx.x] = total;
    if (out_snap) out_snap[blockIdx.x] = total > 0 ? top.i : am.i;                       // marker_manager.py:93-97
    if (out_next) out_next[blockIdx.x] = total == 0 ? -1 : (next != 0x7fffffff ? next : first);   // :120-126, wraps
  }
  if (out_bins && max_list > 0) {
    int rank = base + incl - mine;                          // peaks before this thread's first one, in index order
    // (with more than one word per thread the order inside a thread is not the global one: n <= 16384 has one)
    for (int w = tid; w < nw && rank < max_list; w += kMarkThreads) {
      for (unsigned m = L.fin[w]; m != 0u && rank < max_list; m &= m - 1u, ++rank) {
        const int p = 32 * w + __builtin_ctz(m);
        out_bins[(size_t)blockIdx.x * max_list + rank] = p;
        if (out_prom) out_prom[(size_t)blockIdx.x * max_list + rank] = marker_prominence(row, L, n, p);
      }
    }
    for (int r = total + tid; r < max_list; r += kMarkThreads) {
      out_bins[(size_t)blockIdx.x * max_list + r] = -1;
      if (out_prom) out_prom[(size_t)blockIdx.x * max_list + r] = NAN;
    }
  }
}

// ---- density histogram ----------------------------------------------------------------------------------
// hist[f][a]: rows are applied in order with the reference's float32 arithmetic: hist *= decay (when decay < 1), then
// += 1 at int32(((v - AMP_MIN) / AMP_RNG) * AMP_BINS) (truncation toward zero, NaN / out of range dropped).
//
// The arithmetic floor of touching every cell is one multiply per cell and row: 16384 x 512 x 2440 on 1024 SIMDs of 16
// lanes = 0.52 ms per second of C3 spectra (a wave64 instruction holds its SIMD for four clocks) - and rounds 2 - 5 sat at
// 0.93 ms with amplitude bins across the lanes: every wave multiplies every row, and the wave that owns the trace's amplitude
// bins places sixteen +1 on top (48 instructions).  But most cells are ZERO and 0 x decay = 0 exactly: a trace lives in a few
// dozen of the 512 amplitude bins.  Round 6, third attempt (the first two lost: profiles/r06_analytics.txt): FREQUENCY bins
// across the lanes, 16 amplitude cells per lane in registers, a workgroup = 64 frequency bins x 256 amplitude cells (16
// waves).  Neighbouring frequency bins sit at similar levels, so a wave's 64 x 16 cells are all zero or not together: only
// the 3 - 4 waves per 64 frequency bins that hold the traces multiply (16 per row) and place hits (four 128-bit LDS reads of
// the lane's row of a one-hot table + 16 adds), the others pass a row in three scalar instructions.  The 1024 threads of a
// workgroup form the cell indices of a chunk of 32 rows once (LDS, int16: the byte offset of the cell's one-hot row) and mark,
// per row, which waves they hit.
constexpr int kAmpBins = 512;
constexpr float kAmpMin = -200.0f, kAmpRng = 300.0f;
constexpr int kDensLanes = 64;                      // frequency bins per workgroup (one per lane)
constexpr int kDensCells = 16;                      // amplitude cells per lane
constexpr int kDensWaves = 16;                      // waves per workgroup: 256 amplitude cells
constexpr int kDensRows = 32;                       // rows per chunk
constexpr int kDensOneStride = 20;                  // floats per row of the one-hot table: 80-byte rows put the sixteen rows' 128-bit
                                                    // reads on sixteen different quads of banks
constexpr int kDensNone = 0x7fff;                   // s_idx of a sample that hits no cell of the workgroup

__global__ void __launch_bounds__(kDensWaves * 64) __attribute__((amdgpu_waves_per_eu(8, 8))) density_kernel(const float* __restrict__ rows, int n_rows, int n,
                                                                  float decay, float* hist) {
  __shared__ __attribute__((aligned(8))) short s_idx[2][kDensLanes][kDensRows + 4];   // cell index of (frequency bin, row), -1: none; 72-byte rows:
                                                                                        // a lane fetches its chunk as eight 8-byte reads
  __shared__ unsigned s_wh[2][kDensRows];               // per row: which waves of this workgroup are hit
  __shared__ __attribute__((aligned(16))) float s_one[kDensCells + 1][kDensOneStride];   // row k: 1.0f at cell k, row 16: zeros
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int f = blockIdx.x * kDensLanes + lane;
  const int a_wg = blockIdx.y * (kDensWaves * kDensCells);       // first amplitude cell of the workgroup
  const int a0 = a_wg + w * kDensCells;                          // ... of this wave
  const bool f_ok = f < n;
  typedef float f4 __attribute__((ext_vector_type(4)));
  float h[kDensCells];
  bool nz = false;
  if (f_ok) {
    const f4* src = reinterpret_cast<const f4*>(hist + (size_t)f * kAmpBins + a0);
#pragma unroll
    for (int k = 0; k < kDensCells / 4; ++k) {
      const f4 q = src[k];
      h[4 * k] = q.x; h[4 * k + 1] = q.y; h[4 * k + 2] = q.z; h[4 * k + 3] = q.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < kDensCells; ++k) h[k] = 0.0f;
  }
#pragma unroll
  for (int k = 0; k < kDensCells; ++k) nz |= h[k] != 0.0f;        // (a NaN cell counts: it has to keep being multiplied)
  bool wave_nz = __builtin_amdgcn_ballot_w64(nz) != 0ull;
  if (tid < 2 * kDensRows) (&s_wh[0][0])[tid] = 0u;
  if (tid < (kDensCells + 1) * kDensOneStride) (&s_one[0][0])[tid] = (tid % kDensOneStride == tid / kDensOneStride) ? 1.0f : 0.0f;
  const bool do_decay = decay < 1.0f;
  float decay_v = decay;
  asm volatile("" : "+v"(decay_v));      // a VALU op with an SGPR source issues at half rate on gfx950
  float decay_one = do_decay ? decay : 1.0f;   // h x 1.0f is h: the branch-free path multiplies always
  asm volatile("" : "+v"(decay_one));
  // the workgroup's fetch of a chunk: thread t takes rows t / 64 and t / 64 + 16 at frequency bin lane (256-byte runs)
  const int lr = tid >> 6;
  auto fetch = [&](int r) -> float { return (r < n_rows && f_ok) ? rows[(size_t)r * n + f] : NAN; };
  float v0 = fetch(lr), v1 = fetch(lr + 16);
  __syncthreads();                       // s_wh is zero
  int par = 0;
  for (int r0 = 0; r0 < n_rows; r0 += kDensRows, par ^= 1) {
    auto cell = [&](float v, int r) -> unsigned {
      int idx = -1;
      if (r0 + r < n_rows && f_ok) {
        const float x = (v - kAmpMin) / kAmpRng * float(kAmpBins);
        // astype(int32) truncates toward zero; NaN and anything outside [0, AMP_BINS) is dropped
        if (v == v && x > -1.0f && x < float(kAmpBins)) idx = int(x);
      }
      // what the waves read: the byte offset of the cell's one-hot row counted from the workgroup's first cell (wave w subtracts
      // its 16 x 80 x w and clamps: anything outside its sixteen cells lands on the row of zeros)
      const int rel = idx - a_wg;
      const bool mine = idx >= 0 && rel >= 0 && rel < kDensWaves * kDensCells;
      s_idx[par][lane][r] = short(mine ? rel * (kDensOneStride * 4) : kDensNone);
      return mine ? 1u << (rel / kDensCells) : 0u;        // the wave of this workgroup the sample hits
    };
    // which waves rows lr and lr + 16 hit: the 64 lanes of THIS wave are the rows' 64 frequency bins - an OR across the wave (DPP,
    // both rows in one word) and plain stores by their only writer (sixty-four atomics on one LDS word took the LDS pipe as long
    // as a hit wave's reads)
    unsigned bits = cell(v0, lr) | (cell(v1, lr + 16) << 16);
    bits |= unsigned(dpp_i<0xB1, 0xf>(int(bits)));
    bits |= unsigned(dpp_i<0x4E, 0xf>(int(bits)));
    bits |= unsigned(dpp_i<0x141, 0xf>(int(bits)));
    bits |= unsigned(dpp_i<0x140, 0xf>(int(bits)));
    bits |= unsigned(dpp_i<0x142, 0xa>(int(bits)));
    bits |= unsigned(dpp_i<0x143, 0xc>(int(bits)));
    if (lane == 63) {
      s_wh[par][lr] = bits & 0xffffu;
      s_wh[par][lr + 16] = bits >> 16;
    }
    v0 = fetch(r0 + kDensRows + lr);      // the next chunk's values travel while this one is applied
    v1 = fetch(r0 + kDensRows + lr + 16);
    __syncthreads();
    const int lim = n_rows - r0 < kDensRows ? n_rows - r0 : kDensRows;
    const unsigned whv = lane < kDensRows ? s_wh[par][lane] : 0u;
    const unsigned hit_rows = unsigned(__builtin_amdgcn_ballot_w64(((whv >> w) & 1u) != 0u));
    // (one barrier per chunk: s_idx / s_wh of this parity are written again two chunks on, behind the next chunk's barrier, which
    //  no wave passes before every wave has left this chunk)
    if (hit_rows != 0u && lim == kDensRows) {
      // a wave that is hit in a full chunk takes every row the same way - multiply, fetch the one-hot row (the row of zeros where
      // the sample belongs to other cells), add - as one block without branches: the reads of later rows are issued under the
      // arithmetic of earlier ones
      typedef unsigned u2 __attribute__((ext_vector_type(2)));
      const u2* ip = reinterpret_cast<const u2*>(&s_idx[par][lane][0]);
      u2 iwq = {0u, 0u};                  // four rows' offsets at a time
      static_for<0, kDensRows>([&](auto rc) {
#pragma clang fp contract(off)            // two roundings per row and cell, like the reference's `*=` and `+=`: never an fma
        constexpr int r = decltype(rc)::value;
        if constexpr (r % 4 == 0) iwq = ip[r / 4];
        const unsigned iwr = (r & 2) ? iwq.y : iwq.x;
        const unsigned off = (r & 1) ? (iwr >> 16) : (iwr & 0xffffu);
        const unsigned mine = off - unsigned(w * kDensCells * kDensOneStride * 4);
        const unsigned row_off = mine < unsigned(kDensCells * kDensOneStride * 4) ? mine : unsigned(kDensCells * kDensOneStride * 4);
        const f4* one = reinterpret_cast<const f4*>(reinterpret_cast<const char*>(&s_one[0][0]) + row_off);
#pragma unroll
        for (int k = 0; k < kDensCells / 4; ++k) {
          const f4 q = one[k];
          // (plain operators: __fmul_rn / __fadd_rn are inline functions of a header compiled with contraction on, and carry it here)
          const float m0 = h[4 * k] * decay_one, m1 = h[4 * k + 1] * decay_one, m2 = h[4 * k + 2] * decay_one, m3 = h[4 * k + 3] * decay_one;
          h[4 * k] = m0 + q.x;
          h[4 * k + 1] = m1 + q.y;
          h[4 * k + 2] = m2 + q.z;
          h[4 * k + 3] = m3 + q.w;
        }
      });
      wave_nz = true;
    } else if (hit_rows == 0u && lim == kDensRows) {
      // a wave whose cells only fade through a full chunk: 32 x 16 multiplies in a short loop, nothing else
      if (wave_nz && do_decay) {
#pragma unroll 2
        for (int r = 0; r < kDensRows; ++r) {
#pragma unroll
          for (int k = 0; k < kDensCells; ++k) h[k] = __fmul_rn(h[k], decay_v);
        }
      }
    } else if (hit_rows != 0u || wave_nz) {
      // the ragged last chunk: row by row
      static_for<0, kDensRows>([&](auto rc) {
#pragma clang fp contract(off)
        constexpr int r = decltype(rc)::value;
        const bool hit = (hit_rows >> r) & 1u;
        if (r < lim && (hit || wave_nz)) {
          if (do_decay && wave_nz) {
#pragma unroll
            for (int k = 0; k < kDensCells; ++k) h[k] = __fmul_rn(h[k], decay_v);   // `hist *= d`
          }
          if (hit) {
            // `hist[f, idx] += 1`, its own rounding: the lane's cell as a one-hot word, bit k -> 0.0f or 1.0f without a select
            // (h + 0.0f leaves h as it is: h >= 0)
            const unsigned off = (unsigned short)s_idx[par][lane][r];
            const unsigned mine = off - unsigned(w * kDensCells * kDensOneStride * 4);
            const unsigned row_off = mine < unsigned(kDensCells * kDensOneStride * 4) ? mine : unsigned(kDensCells * kDensOneStride * 4);
            const f4* one = reinterpret_cast<const f4*>(reinterpret_cast<const char*>(&s_one[0][0]) + row_off);
#pragma unroll
            for (int k = 0; k < kDensCells / 4; ++k) {
              const f4 q = one[k];
              h[4 * k] = __fadd_rn(h[4 * k], q.x);
              h[4 * k + 1] = __fadd_rn(h[4 * k + 1], q.y);
              h[4 * k + 2] = __fadd_rn(h[4 * k + 2], q.z);
              h[4 * k + 3] = __fadd_rn(h[4 * k + 3], q.w);
            }
            // (exec narrowed per cell by v_cmpx + a plain add - two vector instructions per cell instead of three - measured
            //  0.81 ms against 0.63: every write of exec drains the vector pipe)
            wave_nz = true;
          }
        }
      });
    }
  }
  if (f_ok) {
    f4* dst = reinterpret_cast<f4*>(hist + (size_t)f * kAmpBins + a0);
#pragma unroll
    for (int k = 0; k < kDensCells / 4; ++k) dst[k] = f4{h[4 * k], h[4 * k + 1], h[4 * k + 2], h[4 * k + 3]};
  }
}

__global__ void log1p_kernel(const float* __restrict__ in, float* out, size_t count) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = log1pf(in[i]);
}

// ---- images as the display takes them: levels -> uint8 ----------------------------------------------------------
// What ImageItem.setImage(img, levels=(lo, hi)) makes of a float image before the colour table
// (displays/waterfall.py:353-356; displays/density_display.py:318 with autoLevels): float32 (v - lo) / (hi - lo) * 255,
// clipped to [0, 255], truncated - one byte per pixel over PCIe instead of four.  A NaN pixel -> 0.
__global__ void __launch_bounds__(256) quantize_u8_kernel(const float* __restrict__ in, unsigned char* out, size_t count,
                                                          float lo, float span) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= count) return;
  unsigned pk = 0u;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (i + k < count) {
      float t = __fmul_rn(__fdiv_rn(__fsub_rn(in[i + k], lo), span), 255.0f);
      t = t < 0.f ? 0.f : (t > 255.f ? 255.f : t);              // NaN: both tests fail
      const unsigned b = t == t ? unsigned(t) : 0u;
      pk |= b << (8 * k);
    }
  }
  if (i + 4 <= count && (reinterpret_cast<uintptr_t>(out) & 3) == 0) *reinterpret_cast<unsigned*>(out + i) = pk;
  else for (int k = 0; k < 4 && i + k < count; ++k) out[i + k] = (unsigned char)(pk >> (8 * k));
}
// min / max of a NaN-free image of non-negative values (log1p of a histogram): positive floats order like their bits
__global__ void __launch_bounds__(256) minmax_pos_kernel(const float* __restrict__ in, size_t count, unsigned* mm) {
  unsigned lo = 0x7f800000u, hi = 0u;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
    const unsigned b = __float_as_uint(in[i]);
    lo = b < lo ? b : lo;
    hi = b > hi ? b : hi;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const unsigned l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&mm[0], lo);
    atomicMax(&mm[1], hi);
  }
}

// ---- waterfall ring -------------------------------------------------------------------------------------
// The ring holds every line ONCE ([H][n]; the reference's doubled buffer exists to make its view one slice -
// here the view is two copies and a pushed row crosses HBM twice instead of three times).  A push is three
// launches and one host wait: new-row flags, the pointer walk as a scan over the flags, the scatter.
// differs[r] = !np.array_equal(rows[r], previous row)  (previous of row 0 = `last`, or "no previous").
// Rows that differ do so within their first bins as a rule: a block leaves at the first 1024 bins that differ.
__global__ void __launch_bounds__(256) rows_differ_kernel(const float* __restrict__ rows, const float* __restrict__ last,
                                                          int have_last, int n, int vec, int* differs) {
  const int r = blockIdx.x;
  if (r == 0 && !have_last) {
    if (threadIdx.x == 0) differs[0] = 1;
    return;
  }
  const float* cur = rows + (size_t)r * n;
  const float* prev = r > 0 ? cur - n : last;
  int diff = 0;
  for (int base = 0; base < n; base += 1024) {                       // uniform trip count: the exit below is block-wide
    const int i = base + threadIdx.x * 4;
    if (vec) {
      if (i < n) {
        const float4 a = *reinterpret_cast<const float4*>(cur + i), b = *reinterpret_cast<const float4*>(prev + i);
        diff = (a.x != b.x) | (a.y != b.y) | (a.z != b.z) | (a.w != b.w);      // NaN != NaN, as in numpy
      }
    } else {
      for (int k = 0; k < 4; ++k)
        if (i + k < n) diff |= (cur[i + k] != prev[i + k]) ? 1 : 0;
    }
    diff = __syncthreads_or(diff);
    if (diff) break;
  }
  if (threadIdx.x == 0) differs[r] = diff;
}
// Waterfall._add_row for a batch (displays/waterfall.py:171-175): ptr = (ptr - 1) % H for every new row, the row goes to
// line ptr.  With c(r) = new rows among 0 ... r the line of a new row is (ptr0 - c(r)) mod H; when more than H rows are
// new, later rows overwrite earlier ones - only the last H new rows are written (no write races in the scatter).
// info = {new rows, index of the last new row or -1}, left in device memory for the scatter and in pinned host memory.
__global__ void __launch_bounds__(1024) waterfall_plan_kernel(const int* __restrict__ differs, int n_rows, int ptr0,
                                                              int history, int* __restrict__ dst, int* info_dev,
                                                              int* info_host) {
  __shared__ int s_cnt[16], s_last[16];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int per = (n_rows + 1023) / 1024;
  const int r0 = min(t * per, n_rows), r1 = min(r0 + per, n_rows);
  int c = 0, last = -1;
  for (int r = r0; r < r1; ++r)
    if (differs[r]) {
      ++c;
      last = r;
    }
  int inc = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(inc, o);
    if (lane >= o) inc += v;
    last = max(last, __shfl_xor(last, o));
  }
  if (lane == 63) s_cnt[wv] = inc;
  if (lane == 0) s_last[wv] = last;
  __syncthreads();
  int before = 0, total = 0, last_all = -1;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int v = s_cnt[k];
    before += k < wv ? v : 0;
    total += v;
    last_all = max(last_all, s_last[k]);
  }
  int run = before + inc - c;
  for (int r = r0; r < r1; ++r) {
    int d = -1;
    if (differs[r]) {
      ++run;
      if (total - run < history) d = (ptr0 - run % history + history) % history;
    }
    dst[r] = d;
  }
  if (t == 0) {
    info_dev[0] = total;
    info_dev[1] = last_all;
    info_host[0] = total;
    info_host[1] = last_all;
  }
}
// copy row r to ring line dst[r] (dst[r] < 0: duplicate or overwritten within the batch, skipped); the last new row is
// also Waterfall._last_row of the next push
__global__ void __launch_bounds__(256) waterfall_scatter_kernel(const float* __restrict__ rows, const int* __restrict__ dst,
                                                                const int* __restrict__ info, int n, int vec,
                                                                float* __restrict__ ring, float* __restrict__ last) {
  const int r = blockIdx.x;
  const int d = dst[r];
  if (d < 0) return;
  const bool is_last = info[1] == r;
  const float* src = rows + (size_t)r * n;
  float* a = ring + (size_t)d * n;
  if (vec) {
    for (int i = threadIdx.x * 4; i < n; i += 1024) {
      const float4 v = *reinterpret_cast<const float4*>(src + i);
      *reinterpret_cast<float4*>(a + i) = v;
      if (is_last) *reinterpret_cast<float4*>(last + i) = v;
    }
  } else {
    for (int i = threadIdx.x; i < n; i += 256) {
      const float v = src[i];
      a[i] = v;
      if (is_last) last[i] = v;
    }
  }
}

}  // namespace

hipError_t launch_rows_stats(const float* rows, int n_rows, int n, int band_lo, int band_hi, double bin_width,
                             float* peak_db, int* peak_bin, double* band_db, hipStream_t s) {
  if (n_rows <= 0) return hipSuccess;
  rows_stats_kernel<<<n_rows, 256, 0, s>>>(rows, n, band_lo, band_hi, bin_width, peak_db, peak_bin, band_db);
  return hipGetLastError();
}

hipError_t launch_frame_stats_finish(const void* parts, int n_frames, int wpf, double cal_lin, float* peak_db,
                                     int* peak_bin, double* band_lin, hipStream_t s) {
  if (n_frames <= 0) return hipSuccess;
  frame_stats_finish_kernel<<<(n_frames + 255) / 256, 256, 0, s>>>(static_cast<const uint4*>(parts), n_frames, wpf, cal_lin,
                                                                   peak_db, peak_bin, band_lin);
  return hipGetLastError();
}

hipError_t launch_top_peaks(const float* rows, int n_rows, int n, int n_peaks, int min_sep, float excursion,
                            int* out_bins, float* out_db, hipStream_t s) {
  if (n_rows <= 0) return hipSuccess;
  const size_t lds = size_t(n) * sizeof(float);
  if (n <= 1024) {
    top_peaks_kernel<6><<<n_rows, 64, lds, s>>>(rows, n, n_peaks, min_sep, excursion, out_bins, out_db);
  } else if (n <= 2048) {
    top_peaks_kernel<7><<<n_rows, 128, lds, s>>>(rows, n, n_peaks, min_sep, excursion, out_bins, out_db);
  } else if (n <= 4096) {
    top_peaks_kernel<8><<<n_rows, 256, lds, s>>>(rows, n, n_peaks, min_sep, excursion, out_bins, out_db);
  } else if (n <= 8192) {
    top_peaks_kernel<9><<<n_rows, 512, lds, s>>>(rows, n, n_peaks, min_sep, excursion, out_bins, out_db);
  } else {
    static std::atomic<unsigned long long> attr_done{0};
    const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(top_peaks_kernel<10>), 72 * 1024, attr_done);
    if (e != hipSuccess) return e;
    top_peaks_kernel<10><<<n_rows, kPeakThreads, lds, s>>>(rows, n, n_peaks, min_sep, excursion, out_bins, out_db);
  }
  return hipGetLastError();
}

hipError_t launch_marker_peaks(const float* rows, int n_rows, int n, double height, double prominence, int distance,
                               int current_idx, int max_list, int* out_count, int* out_snap, int* out_next,
                               int* out_bins, double* out_prom, hipStream_t s) {
  if (n_rows <= 0) return hipSuccess;
  const size_t lds = size_t(n) * sizeof(float);
  static std::atomic<unsigned long long> attr_done{0};
  const hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(marker_peaks_kernel), 64 * 1024, attr_done);
  if (e != hipSuccess) return e;
  marker_peaks_kernel<<<n_rows, kMarkThreads, lds, s>>>(rows, n, height, prominence, distance, current_idx, max_list,
                                                        out_count, out_snap, out_next, out_bins, out_prom);
  return hipGetLastError();
}

hipError_t launch_density(const float* rows, int n_rows, int n, float decay, float* hist, hipStream_t s) {
  if (n_rows <= 0) return hipSuccess;
  density_kernel<<<dim3((n + kDensLanes - 1) / kDensLanes, kAmpBins / (kDensWaves * kDensCells)), kDensWaves * 64, 0, s>>>(rows, n_rows, n, decay, hist);
  return hipGetLastError();
}

hipError_t launch_log1p(const float* in, float* out, size_t count, hipStream_t s) {
  if (count == 0) return hipSuccess;
  log1p_kernel<<<unsigned((count + 255) / 256), 256, 0, s>>>(in, out, count);
  return hipGetLastError();
}

hipError_t launch_quantize_u8(const float* in, unsigned char* out, size_t count, float lo, float hi, hipStream_t s) {
  if (count == 0) return hipSuccess;
  const float span = float(double(hi) - double(lo));            // Python forms (hi - lo) in double; the division is float32
  quantize_u8_kernel<<<unsigned((count + 1023) / 1024), 256, 0, s>>>(in, out, count, lo, span);
  return hipGetLastError();
}
hipError_t launch_minmax_pos(const float* in, size_t count, unsigned* mm_dev, hipStream_t s) {
  if (count == 0) return hipSuccess;
  const unsigned init[2] = {0x7f800000u, 0u};
  hipError_t e = hipMemcpyAsync(mm_dev, init, sizeof(init), hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return e;
  const size_t blocks = (count + 255) / 256;
  minmax_pos_kernel<<<unsigned(blocks < 2048 ? blocks : 2048), 256, 0, s>>>(in, count, mm_dev);
  return hipGetLastError();
}

static inline int rows_vec_ok(const void* p0, const void* p1, const void* p2, int n) {
  return ((reinterpret_cast<uintptr_t>(p0) | reinterpret_cast<uintptr_t>(p1) | reinterpret_cast<uintptr_t>(p2)) & 15) == 0 &&
         (n & 3) == 0;
}
hipError_t launch_waterfall_push(const float* rows, int n_rows, int n, int have_last, int ptr0, int history, int* differs,
                                 int* dst, int* info_dev, int* info_host, float* ring, float* last, hipStream_t s) {
  if (n_rows <= 0) return hipSuccess;
  const int vec = rows_vec_ok(rows, ring, last, n);
  rows_differ_kernel<<<n_rows, 256, 0, s>>>(rows, last, have_last, n, vec, differs);
  waterfall_plan_kernel<<<1, 1024, 0, s>>>(differs, n_rows, ptr0, history, dst, info_dev, info_host);
  waterfall_scatter_kernel<<<n_rows, 256, 0, s>>>(rows, dst, info_dev, n, vec, ring, last);
  return hipGetLastError();
}

}  // namespace tdsa
