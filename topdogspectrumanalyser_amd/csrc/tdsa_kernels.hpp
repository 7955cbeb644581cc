// tdsa_kernels.hpp - kernel parameter blocks and launcher prototypes shared by the kernel TUs
// and the C-ABI layer (tdsa_capi.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>

namespace tdsa {

// Raise a kernel's dynamic-LDS limit once per DEVICE (function attributes belong to the device the
// module is loaded on; a process may hold plans on several GPUs).  `done` is a per-kernel bit mask.
inline hipError_t ensure_dynamic_lds(const void* fn, int bytes, std::atomic<unsigned long long>& done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
  return e;
}

constexpr int kMinLog2N = 6;    // 64
constexpr int kMaxLog2N = 14;   // 16384 : largest frame whose c64 working set fits the 160 KiB LDS

// dc_mode
constexpr int DC_NONE = 0, DC_FRAME_MEAN = 1, DC_TRACKED = 2;

struct SpecParams {
  const void* in;            // device: interleaved int8/uint8 I,Q or float2 samples
  long long frame_stride;    // bytes between consecutive frame starts (hop * bytes per sample)
  int n_frames;
  int first_frame_index;     // global index of frame 0 of this launch (nan_safe semantics of frame 0)
  const float* window;       // [N] window * input scale (natural order: sizes that keep it in LDS, N <= 1024)
  const float* window_perm;  // [N] the same values in the frame kernel's thread order (launch_window_perm), N >= 2048
  const float2* tw;          // [N] exp(-2 pi i m / N)
  float* out_db;             // [F][N] fftshift-ed dB, or null
  float* out_lin;            // [F][N] fftshift-ed linear power * pscale (averaging modes), or null
  float2* out_cplx;          // [F][N] complex spectrum X[k] in natural bin order (real-input path), or null
  const float2* out_mul;     // [N] with out_cplx, complex64 input, no hold: conj(X[k] * out_mul[k]) is stored instead (chirp-z), or null
  int out_mul_rows;          // > 1: out_mul holds this many rows of N, frame f takes row f mod out_mul_rows (long chirp-z frames)
  int in_valid;              // complex64 input, no hold: samples from this index on are zeros and are not read (0: all N are read)
  int out_valid;             // with out_cplx, complex64 input, no hold: only bins below this index are stored (0: all N)
  const float2* dc_sub;      // [F] per-frame DC estimate in raw-sample units, WITHOUT in_off (DC_TRACKED), or null
  float2* dc_state;          // last frame's mean in units of x is stored here (DC_FRAME_MEAN), or null
  float* part_max;           // [N] the plan's max-hold trace (merged into with float atomics), or null
  float* part_min;           // [N] the plan's min-hold trace
  const float* tare;         // [N] dB baseline to subtract, or null
  unsigned xor_mask;         // 0x80808080 for int8 (-> offset binary), 0 for uint8
  float in_off;              // 128 (int8), 127.5 (uint8), 0 (c64): value of "zero" in raw units
  float in_scale;            // 1/128, 1/127.5, 1 : raw unit -> x
  int dc_mode;
  int db_mode;               // TDSA_DB_MAG / TDSA_DB_POW
  float pscale;              // linear power scale (PSD) for DB_POW / out_lin
  float log_floor;
  float cal_db;              // calibration offset added to dB
  int hold_flags;            // bit0 max, bit1 min
  unsigned long long* dbg;   // developer timeline buffer (TDSA_TIMELINE builds only), else null
  // ---- several captures ("segments") in one launch (tdsa_process_dev_batch) ----
  // frame f belongs to segment s = floor(f / seg_frames) = umulhi(f, seg_magic), frame fi = f - s * seg_frames of it:
  // it reads from in + s * seg_in_stride + fi * frame_stride and its row goes to out + (s * seg_out_stride + fi * N)
  // elements.  seg_magic = 0 (one segment): s = 0, fi = f.
  // ---- chunk aggregates of the TraceAverager scan (out_lin launches of the sizes with one frame per workgroup slot) ----
  const float* agg_w;        // [F] per frame: weight of P_f in its workgroup's aggregate (avg_weights_kernel), or null
  float* agg_out;            // [grid][N] sum over the workgroup's frames of agg_w[f] P_f, display order; null: not wanted
  int agg_only;              // 1: the linear rows themselves are not wanted (no dB rows, no hold: only the averager's state)
  unsigned seg_magic;        // ceil(2^32 / seg_frames); exact for f * seg_frames < 2^32 (checked by the host)
  unsigned seg_frames;
  long long seg_in_stride;   // bytes
  long long seg_out_stride;  // output elements
  // ---- chirp-z plans with M <= 16384 (tdsa_chirp.hip; complex64 instantiation without hold): the two element-wise passes
  //      of the convolution ride the two transforms (appended in round 5: the offsets of everything above stay) ----
  const void* pre_raw;       // FIRST transform: the raw frames (bytes, or complex64 if pre_c64) - sample n of frame f is
                             // unpacked, DC-freed (dc_sub[f], or null) and multiplied by pre_aw[n] on load; `in` is not read.
                             // null: `in` holds complex64 rows
  long long pre_stride;      // bytes between raw frames
  const float2* pre_aw;      // [in_valid] window * input scale * chirp a[n] (made in double, rounded once)
  int pre_c64;
  unsigned pre_xor;          // 0x80808080 for int8 (-> offset binary; the kernel uses the low 16 bits: one sample per load), 0 for uint8
  float pre_off;             // 128 / 127.5 / 0: the raw format's zero level (in_off / xor_mask above stay those of complex64 rows)
  int post_n;                // SECOND transform: > 0: bins k < post_n leave as |X / M|^2 -> power (out_lin) or dB (out_db, tare,
                             // part_max / part_min) rows of post_n values, bin k at (k + post_n / 2) mod post_n; out_cplx is not written
  float post_inv_m;          // 1 / M
  int rows_twice;            // long chirp-z frames (size 14 only): the row passes of the first and of the (transposed) second
                             // transform run back to back on each row: X -> conj(X out_mul) -> through LDS -> transform -> out_cplx
  // ---- per-frame scalars from the epilogue (STATS instantiations: N >= 1024, dB rows, no tare, hold none / max; appended in
  //      round 6) ----
  void* stats_part;          // [F][waves per frame] records of kStatsRecBytes (see the STATS epilogue of the frame kernel), or null
  unsigned band_lohi;        // band power: inclusive display-bin range lo | hi << 16; lo > hi: none
};
constexpr int kStatsRecBytes = 16;   // {max dB, its first display position among the wave's bins (bit 30: a NaN), band power, 0}
// waves of one frame in the frame kernel (frames of whole waves: N >= 1024)
inline int spectrum_waves_per_frame(int log2n) { return log2n <= 10 ? 1 : (1 << log2n) / 1024; }

struct LaunchGeom {
  int grid, block, fpw;
  size_t lds_bytes;
};

// format: 0 = bytes (int8/uint8), 1 = complex64
LaunchGeom spectrum_geometry(int log2n, int n_frames, int num_cu);
hipError_t launch_spectrum(int log2n, int in_c64, const SpecParams& p, const LaunchGeom& g,
                           hipStream_t s);
// window table in the order the frame kernel's threads consume it: thread (t, h) of a frame finds its 16 values as
// 64 contiguous bytes (4 x 16-byte loads per frame instead of 8 x 8 at N = 16384); no-op for N <= 1024
hipError_t launch_window_perm(int log2n, const float* w, float* w_perm, hipStream_t s);

// frames [u0, u1) workgroup b of a persistent frame-kernel grid takes (one frame per workgroup slot): the formula of
// tdsa_spectrum_kernel.hpp, shared with the averager's chained scan whose chunks are those ranges
__host__ __device__ inline void spectrum_unit_range(unsigned b, unsigned n_units, unsigned grid, int& u0, int& u1) {
  const unsigned upw = n_units / grid, urem = n_units - upw * grid;
  u0 = int(b * upw + b * urem / grid);
  u1 = int((b + 1) * upw + (b + 1) * urem / grid);
}

struct AvgParams {
  const float* lin;       // [F][N] linear power (already PSD-scaled)
  int n_frames, n;
  double* state;          // [N] TraceAverager._buffer
  int count_in;           // TraceAverager._count before this batch (0: buffer is None)
  int mode, avg_n;        // TDSA_AVG_EXP / TDSA_AVG_LIN, n >= 2
  float log_floor, cal_db;
  const float* tare;      // or null
  float* out_db;          // [F][N] or null
  float* state_max;       // hold state (updated in place) or null
  float* state_min;
  int first_frame_index;
  // chunks = the frame ranges of the frame kernel's wg_chunks workgroups, whose aggregates (float32, agg[c][n]) the frame
  // kernel has already formed (SpecParams::agg_out): the scan then needs no pass of its own for them.  0: chunks of 64.
  int wg_chunks;
  int wg_fold;            // consecutive workgroup ranges per chunk of the scan (1 .. 4): ceil(wg_chunks / wg_fold) <= 256 chunks
  int wg_fpw;             // frames per workgroup unit of that grid (Cfg::FPW; 0 or 1: one): range b = frames [u0 fpw, u1 fpw)
  const float* agg;
  const double* chunk_a;  // [kAvgMaxWgChunks + 64] per workgroup range: product of its frames' a_f (1 for an empty one and past the end)
  const float* chunk_v;   // [kAvgMaxWgChunks + 64] 1: the range has frames (its aggregate row was written), else 0
  // Sizes below 4096 (several frames per workgroup slot of the frame kernel: its workgroups' frames are not a range): the
  // chunks are wg_chunks equal ranges of the batch and a pass of the scan's own forms their aggregates with these weights
  // (agg_w_local[f], as launch_avg_weights makes them) into agg.  Null: the frame kernel has formed them.
  const float* agg_w_local;
  int state_only;         // 1: no dB rows and no hold traces wanted and the frame kernel wrote no linear rows: chain only
};
constexpr int kAvgMaxWgChunks = 1024;
// frames [f0, f1) of workgroup range b (wg_chunks > 0)
__host__ __device__ inline void avg_wg_range(const AvgParams& p, int b, int& f0, int& f1) {
  const int fpw = p.wg_fpw > 1 ? p.wg_fpw : 1;
  int u0, u1;
  spectrum_unit_range(unsigned(b), unsigned((p.n_frames + fpw - 1) / fpw), unsigned(p.wg_chunks), u0, u1);
  f0 = u0 * fpw < p.n_frames ? u0 * fpw : p.n_frames;
  f1 = u1 * fpw < p.n_frames ? u1 * fpw : p.n_frames;
}
// carry: [chunks][n] doubles of scratch for the chunked scan (null: sequential kernel)
hipError_t launch_avg_scan(const AvgParams& p, hipStream_t s, double* carry);
int avg_scan_chunks(int n_frames);
// weights of the frames in their workgroup's aggregate (see SpecParams::agg_w): w[f] = b_f * prod of a_g over the later
// frames g of the same chunk, chunks = spectrum_unit_range(c, n_frames, wg_chunks)
// ... and the per-chunk multipliers / validity flags the chain reads (chunk_a, chunk_v above)
hipError_t launch_avg_weights(const AvgParams& p, float* w, double* chunk_a, float* chunk_v, hipStream_t s);

// per-frame sums for the DC tracker (dc_alpha in (0,1)): sums[f] = sum of raw I, raw Q (float2)
hipError_t launch_frame_sums(const void* in, int in_c64, unsigned xor_mask, long long frame_stride,
                             int n, int n_frames, float2* sums, hipStream_t s);
// sequential tracker: dc <- (1-a) dc + a mean ; writes dc_sub[f] (raw units) and the final state
hipError_t launch_dc_track(const float2* sums, int n, int n_frames, float alpha, float in_off,
                           float in_scale, float2* dc_state, float2* dc_sub, hipStream_t s, int parts = 1);

struct TraceParams {
  const float* db_in;   // [n]
  int n;
  float cal_db;
  float* tare_acc;      // [n] linear accumulator (collecting) or null
  int tare_collect;     // accumulate this frame
  int tare_first;       // first collected frame: acc = linear (not +=)
  int tare_finish;      // this frame completes the baseline: baseline = 10log10(max(acc/cnt,1e-30))
  int tare_count;       // frames collected including this one
  float* tare_base;     // [n] baseline (written when finishing; read when active)
  int tare_active;      // subtract baseline (after a finish in the same call too)
  float* live;          // [n] out
  float* state_max;     // or null
  float* state_min;
  int max_first, min_first;  // adopt (nan_safe) instead of fmax/fmin
  float* max_copy;      // [n] second destination of the new max / min trace (a host-visible buffer), or null
  float* min_copy;
};
hipError_t launch_trace_update(const TraceParams& p, hipStream_t s);

hipError_t launch_avg_host_frame(const double* lin, int n, double* state, int count_in, int mode,
                                 int avg_n, hipStream_t s);

hipError_t launch_fill(float* p, size_t n, float v, hipStream_t s);

// ---- real-input (audio) path: two real channels ride one complex FFT (z = left + i*right) ----
// channel: 0 mono ((L+R)/2), 1 left, 2 right, 3 stereo (rows 2f = left, 2f+1 = right)
hipError_t launch_real_select(const float2* lr, size_t count, int channel, float2* za, float2* zb, hipStream_t s);
hipError_t launch_real_fold(const float2* spec, int n, int n_frames, int rows_per_frame, int row, float pscale,
                            float* lin, hipStream_t s);
hipError_t launch_lin_to_db(const float* lin, size_t count, float log_floor, float cal_db, float* out_db,
                            hipStream_t s);

// ---- long frames, N = 2^15 .. 2^20 = N1 x 16384 (tdsa_big.hip) ----
constexpr int kBigMinLog2N = 15, kBigMaxLog2N = 20;
// column pass: raw IQ of n_seg segments -> Z[seg][k1][n2] (complex64).  tw_seed[s][n2], n2 < 16384: the per-column seeds
// of W_N^(n2 k1), k1 = a + 8 b: rows s = 0 .. NA-2 hold W_N^(n2 (s + 1)), rows NA-1 .. NA+NB-3 hold W_N^(n2 8 (s - NA + 2)),
// with N1 = N / 16384, NA = min(N1, 8), NB = N1 / NA (big_seed_rows(log2n) rows in all)
constexpr int big_seed_rows(int log2n) {
  const int n1 = 1 << (log2n - 14), na = n1 < 8 ? n1 : 8, nb = n1 / na;
  return (na - 1) + (nb - 1);
}
// The window as the column threads get it.  mode 0: `table` [N] (window * input scale), one load per sample.
// mode 2: `flat` for every sample (rectangular windows, the chirp-z path's all-ones window): no loads at all.
// tdsa_set_window decides (tdsa_capi.cpp).
struct BigWindow {
  int mode;
  const float* table;
  float flat;
};
// long chirp-z frames: the column pass reads the RAW frames (seg_stride = bytes between frames, dc_sub per FRAME) and
// multiplies the unpacked samples by aw[n] = window x input scale x chirp on load
struct BigChirpPre {
  const float2* aw;          // [n] or null (off)
  int n;                     // samples per frame
  int c64;                   // raw samples are complex64
  int split_h;               // > 0: segment 2f + s is the half [s H, ...) of frame f (frames above 2^19 points)
};
// ... and the transposed transform's column pass writes the wanted bins as power / dB rows [F][n]
struct BigChirpPost {
  int n;                     // bins per frame, 0: off
  int split_h;
  float inv_m;
  int db_mode;
  float pscale, log_floor, cal_db;
  const float* tare;
  float* out_db;
  float* out_lin;
};
hipError_t launch_big_cols(int log2n, const void* in, int in_c64, long long seg_stride, int n_seg, const BigWindow& win,
                           const float2* tw_seed, const float2* dc_sub, float2* z,
                           unsigned xor_mask, float in_off, hipStream_t s, unsigned in_valid = 0,
                           const BigChirpPre* pre = nullptr);
// transposed four-step, second half: R[seg][k1][m2] (row transforms of T[k1][k2] = V[k1 + N1 k2]) -> Y[seg][m] in natural
// order, M = N1 * 16384 points; bins from out_valid on are not stored (tdsa_big.hip)
hipError_t launch_big_cols_out(int log2m, const float2* r, long long seg_stride, int n_seg, const float2* tw_seed, float2* y,
                               unsigned out_valid, hipStream_t s, const BigChirpPost* post = nullptr);
// exact per-frame sums + DC tracker in double; dc_res[f] = DC estimate in raw units MINUS in_off (small)
hipError_t launch_big_dc(const void* in, int in_c64, unsigned xor_mask, long long frame_stride, int n, int n_frames,
                         double alpha, double in_off, double in_scale, double* sums, float2* dc_state, float2* dc_res,
                         hipStream_t s);
// row pass: 16384-point transforms of the rows, power summed per workgroup (tdsa_big.hip)
hipError_t launch_big_rows(const float2* z, long long seg_stride, int group, int n1, int act, float* acc, int acc_split,
                           int acc_add, const float2* tw, hipStream_t s);
// the row pass of one round and gather + finish as ONE launch with a dependency-counted ticket queue (tdsa_big.hip);
// queue: 32 zeroed bytes of device memory owned by the plan, ticket_base / rows_target: the counters' values this launch starts from / waits for
hipError_t launch_big_rows_gather(int log2n, const float2* z, long long seg_stride, int group, int n1, int act, float* acc,
                                  const float2* tw, double* dst, int add, double* mean_out, int count, int db_mode, float pscale,
                                  float log_floor, float cal_db, const float* tare, float* out_db, float* hold_max,
                                  float* hold_min, int max_first, int min_first, void* queue, unsigned long long ticket_base,
                                  unsigned long long rows_target, unsigned long long* tickets_used, int variant, hipStream_t s);
// P[k1 * split + j][k2] float partial sums -> dst[(k1 + N1*k2) ^ N/2] double (overwrite or accumulate)
hipError_t launch_big_gather(int log2n, const float* s_rows, int split, double* dst, int add, hipStream_t s);
// the same + mean = dst / count -> dB (+cal, -tare) row and hold traces, one launch
hipError_t launch_big_gather_finish(int log2n, const float* s_rows, int split, double* dst, int add, double* mean_out, int count,
                                    int db_mode, float pscale, float log_floor, float cal_db, const float* tare,
                                    float* out_db, float* hold_max, float* hold_min, int max_first, int min_first,
                                    hipStream_t s);
hipError_t launch_big_finish(const double* src, long long n, double* mean_out, int count, int db_mode, float pscale,
                             float log_floor, float cal_db, const float* tare, float* out_db, float* hold_max,
                             float* hold_min, int max_first, int min_first, hipStream_t s);

// ---- Welch partials across GPUs (tdsa_welch_export / tdsa_welch_combine) ----
constexpr int kWelchMaxParts = 64;
// dst[i] = src[i] / div as float32 (as_f32) or float64
hipError_t launch_welch_export(const double* src, double div, void* dst, int as_f32, long long n, hipStream_t s);
// sum_r counts[r] * parts[r][i] (double, rank order) -> sum_out (or null), mean -> mean_out (or null), dB row, hold traces;
// native_db: the dB arithmetic of the LDS-resident sizes' averager (else that of the long-frame finish)
hipError_t launch_welch_combine(const void* const* parts, const int* counts, int n_parts, int as_f32, long long n,
                                double* sum_out, double* mean_out, int total, int native_db, int db_mode, float pscale,
                                float log_floor, float cal_db, const float* tare, float* out_db, float* hold_max,
                                float* hold_min, int max_first, int min_first, hipStream_t s);
#ifdef TDSA_DEV
hipError_t launch_xcd_shift(int wgs, hipStream_t s);
#endif
// n_cu * 4 workgroups of 256 threads, each wave 64 * iters independent v_add_f32 (8 chains): the SIMDs' saturated VALU rate
hipError_t launch_valu_clock(float* scratch, int n_cu, int iters, hipStream_t s);

// ---- frame lengths made of the factors 2, 3, 5 only, up to 10 000 points (tdsa_smooth.hip): mixed-radix Stockham FFT in LDS ----
constexpr int kSmoothMaxN = 10000;     // two LDS buffers of N complex64 values: 160 000 of the 163 840 bytes a workgroup may have
constexpr int kSmoothMaxStages = 16;
struct SmoothParams {
  const void* in;            // raw frames
  int in_c64;
  long long frame_stride;    // bytes
  int n, n_frames, fpw;      // fpw: frames per workgroup (set by the launcher)
  int n_stages;
  int radix[kSmoothMaxStages];   // 4, 2, 3, 5 in any order, product n
  unsigned magic_per[kSmoothMaxStages], magic_s[kSmoothMaxStages], magic_fpw;   // division constants (set by the launcher)
  const float2* tw;          // [n] exp(-2 pi i k / n)  (two passes: [n_total], the stages use every tw_step-th entry)
  int tw_step;               // 1; two passes: n_total / n of the pass
  int n_total, n1, n2;       // two passes: frame length n1 * n2 (column pass: n = n1, row pass: n = n2)
  float2* z;                 // two passes: [F][n1][n2] between them
  const float* window;       // [n] window * input scale
  const float2* dc_sub;      // [F] DC estimate minus the zero level, raw units, or null
  int dc_own;                // byte samples, per-frame mean removal: the kernel forms the frame means itself (dc_sub unused)
  int twice_zero;            // 256 int8 after the xor, 255 uint8
  float in_scale;
  float2* dc_state;          // (dc_own) receives the last frame's mean in units of x, or null
  unsigned xor_mask;
  float in_off;
  int db_mode;
  float pscale, log_floor, cal_db;
  const float* tare;
  float* out_db;             // [F][n] or null
  float* out_lin;            // [F][n] linear power * pscale (averaging modes) or null
};
hipError_t launch_smooth(SmoothParams p, hipStream_t s, int mode = 0);

// ---- frame lengths that are not a power of two, 2 <= N <= 8192 (tdsa_chirp.hip): chirp-z on the frame kernel ----
constexpr int kChirpMaxN = (1 << 20) - 1; // M = 2^ceil(log2(2N-1)) <= 2^20 up to N = 2^19 (M > 16384: the long-frame kernels); longer
                                         // frames as four half-length sub-convolutions of 2^20 points (tdsa_chirp.hip)
// res[f] = frame mean minus the format's zero level, raw units (twice_zero: 256 int8 after the xor, 255 uint8, 0 c64)
// dc_state (or null): receives the last frame's mean in units of x - the per-frame mean mode needs no tracker pass
// part: scratch of n_frames * chirp_sum_chunks(n) * 2 doubles (frames long enough to be summed by several workgroups) or null
int chirp_sum_chunks(int n);
hipError_t launch_chirp_sums(const void* in, int in_c64, unsigned xor_mask, long long frame_stride, int n, int n_frames,
                             int twice_zero, float2* res, float2* dc_state, float in_scale, hipStream_t s,
                             double* part = nullptr);
hipError_t launch_chirp_pre(const void* in, int in_c64, long long frame_stride, int n, int m, int n_frames,
                            const float* window, const float2* chirp, const float2* dc_sub, unsigned xor_mask,
                            float in_off, float2* u, hipStream_t s, int split_h = 0);
hipError_t launch_chirp_post(const float2* y, int n, int m, int n_frames, int first_frame_index,
                             int db_mode, float pscale, float log_floor, float cal_db, const float* tare, float* out_db,
                             float* out_lin, float* hold_max, float* hold_min, hipStream_t s, int split_h = 0);
// split convolution of frames above 2^19 points: rows (2f, 2f + 1) of u[2F][m] <- conj(UA b0 + UB bm), conj(UA bp + UB b0)
hipError_t launch_chirp_split_combine(float2* u, long long m, int n_frames, const float2* b0, const float2* bm,
                                      const float2* bp, hipStream_t s);
// hold traces folded from finished dB rows [F][n] (chirp-z plans whose second transform stores the rows itself)
hipError_t launch_chirp_hold(const float* rows, int n, int n_frames, int first_frame_index, float* hold_max, float* hold_min,
                             hipStream_t s);
// real-input frames: one-sided linear power rows in the layout of launch_real_fold
hipError_t launch_chirp_post_real(const float2* y, int n, int m, int n_frames, int rows_per_frame, int row, float pscale,
                                  float* lin, hipStream_t s);

// ---- trace analytics / accumulators (tdsa_analytics.hip) ---------------------------------------------
hipError_t launch_rows_stats(const float* rows, int n_rows, int n, int band_lo, int band_hi, double bin_width,
                             float* peak_db, int* peak_bin, double* band_db, hipStream_t s);
// bin_width < 0: band_db receives the band's linear sum (no width, no log)
hipError_t launch_frame_stats_finish(const void* parts, int n_frames, int wpf, double cal_lin, float* peak_db,
                                     int* peak_bin, double* band_lin, hipStream_t s);
hipError_t launch_top_peaks(const float* rows, int n_rows, int n, int n_peaks, int min_sep, float excursion,
                            int* out_bins, float* out_db, hipStream_t s);
hipError_t launch_marker_peaks(const float* rows, int n_rows, int n, double height, double prominence, int distance,
                               int current_idx, int max_list, int* out_count, int* out_snap, int* out_next,
                               int* out_bins, double* out_prom, hipStream_t s);
hipError_t launch_density(const float* rows, int n_rows, int n, float decay, float* hist, hipStream_t s);
hipError_t launch_log1p(const float* in, float* out, size_t count, hipStream_t s);
hipError_t launch_quantize_u8(const float* in, unsigned char* out, size_t count, float lo, float hi, hipStream_t s);
hipError_t launch_minmax_pos(const float* in, size_t count, unsigned* mm_dev, hipStream_t s);
hipError_t launch_waterfall_push(const float* rows, int n_rows, int n, int have_last, int ptr0, int history, int* differs,
                                 int* dst, int* info_dev, int* info_host, float* ring, float* last, hipStream_t s);

}  // namespace tdsa
