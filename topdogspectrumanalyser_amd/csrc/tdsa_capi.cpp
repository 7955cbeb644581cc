// tdsa_capi.cpp - the extern "C" boundary of libtdsa_hip.so (declared in include/tdsa_hip.h).
// Plain C types only; every entry point returns a status and never throws.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/tdsa_hip.h"
#include "tdsa_kernels.hpp"

using namespace tdsa;

namespace {

thread_local char g_err[512] = "ok";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail(TDSA_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

int ilog2i(int x) {
  int l = 0;
  while ((1 << l) < x) ++l;
  return l;
}

}  // namespace

struct tdsa_plan_s {
  int device = 0, nfft = 0, log2n = 0, max_frames = 0, num_cu = 256;
  hipStream_t stream = nullptr;
  // tdsa_set_overlap: extra streams consecutive order-independent launches rotate over, so the ragged
  // tail of one persistent launch (and the inter-kernel gap) is filled by the head of the next
  static constexpr int kMaxOverlap = 4;
  hipStream_t aux[kMaxOverlap - 1] = {};
  hipEvent_t ev_aux[kMaxOverlap - 1] = {};
  hipEvent_t ev_state = nullptr;
  int n_overlap = 1, rr = 0;
  int overlap_share = 50;                // percent of the CUs an overlapped launch is sized for (3+ streams)
  bool aux_busy = false, state_dirty = true;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  tdsa_mode mode{};
  bool window_set = false;
  float* d_window[3] = {nullptr, nullptr, nullptr};   // window * input scale, per input format
  float* d_window_perm[3] = {nullptr, nullptr, nullptr};   // the same in the frame kernel's thread order (N >= 2048)
  float2* d_tw = nullptr;
  float* d_hold_max = nullptr;
  float* d_hold_min = nullptr;
  long long held_max = 0, held_min = 0;
  double* d_avg = nullptr;
  int avg_count = 0;
  float* d_lin = nullptr;                // [max_frames][N] linear power scratch (averaging modes)
  double* d_carry = nullptr;             // [chunks][N] chunk carries of the averager scan
  size_t carry_chunks = 0;               // chunks d_carry has room for
  float* d_agg = nullptr;                // [grid][N] chunk aggregates formed by the frame kernel's workgroups (sizes >= 4096)
  float* d_agg_w = nullptr;              // [max_frames] weight of each frame in its workgroup's aggregate
  double* d_chunk_a = nullptr;           // [kAvgMaxWgChunks + 64] per chunk: product of its frames' multipliers
  float* d_chunk_v = nullptr;            // [kAvgMaxWgChunks + 64] per chunk: 1 = has frames
  long long agg_w_key[7] = {-1, -1, -1, -1, -1, -1, -1};   // (count, mode, n, frames, chunks, frames per unit, path) the weights in d_agg_w were made for
  float2* d_cplx = nullptr;              // [max_frames][N] complex spectra (real-input path)
  float2* d_real = nullptr;              // real-input path: the selected signal(s) as complex streams (two for stereo)
  size_t real_bytes = 0;
  float* d_lin1 = nullptr;               // [max_frames][2][N/2+1] one-sided linear power (real-input path)
  float* d_db1 = nullptr;                // same shape, dB
  float2* d_dc_state = nullptr;
  float2* d_sums = nullptr;
  float2* d_dc_sub = nullptr;
  float* d_tare_base = nullptr;
  float* d_tare_acc = nullptr;
  bool tare_active = false;
  int tare_count = 0;
  void* d_in_stage = nullptr;
  size_t in_stage_bytes = 0;
  // pinned bounce buffers of the host entry points for small calls (the one-frame-per-GUI-tick case): a copy from / to
  // pageable memory costs ~10 us each way in the runtime's own staging, a memcpy through pinned memory ~2 us
  void* h_in_pin = nullptr;
  void* h_out_pin = nullptr;
  size_t in_pin_bytes = 0, out_pin_bytes = 0;
  float* d_out_stage = nullptr;
  float* d_trace_in = nullptr;
  float* d_trace_live = nullptr;
  long long frames_seen = 0;             // frames processed since the last hold reset (nan_safe rule)
  unsigned long long* d_dbg = nullptr;   // TDSA_TIMELINE developer builds
  // long-frame plans, N = 2^15 .. 2^20 = N1 x 16384 (tdsa_big.hip)
  bool big = false;
  float2* d_z = nullptr;                 // [group][N1][16384] complex64 rows after the column pass
  float* d_acc = nullptr;                // [N1][16384] power sums of the current call (row pass output)
  double* d_sum = nullptr;               // [N] fftshift-ed sums over the segments averaged so far
  void* d_welch = nullptr;               // staging of tdsa_welch_export (one partial) / tdsa_welch_combine (all of them)
  size_t welch_bytes = 0;
  float* d_clock = nullptr;              // scratch of tdsa_shader_clock
  bool big_mean_in_sum = false;          // Welch calls leave the running mean as d_sum / avg_count; d_avg is formed when someone asks
  double* d_lin64 = nullptr;             // [N] fftshift-ed power of one frame (exp / capped lin averaging)
  double* d_sums64 = nullptr;            // [max_frames][2] exact I / Q sums of the frames of a call
  float2* d_tw_seed = nullptr;           // [big_seed_rows][16384] per-column twiddle seeds of the column pass
  float2* d_tw_row = nullptr;            // W_16384^m : the row pass's twiddle table
  float* d_ones = nullptr;               // [16384] unit window for the row pass
  BigWindow big_win[3] = {};             // the column pass's window per input format (tdsa_set_window: table or one value)
  int avg_wg_min = 128;                  // batches of more frames than this take the workgroup-chunk scan (tdsa_debug_knob "avg_wg_min")
  bool avg_f64_chunks = false;           // tdsa_debug_knob "avg_f64_chunks": always the scan over fixed 64-frame chunks with float64 aggregates
  // frame lengths made of 2, 3, 5 only (up to 10 000 points in one LDS pass, two passes above): mixed-radix FFT of exactly nfft points (tdsa_smooth.hip) for the
  // complex path; the plan stays a chirp-z plan for everything else (real input)
  bool smooth = false;
  int smooth_on = 1;                     // tdsa_debug_knob "smooth": 0 = such sizes run as chirp-z convolutions like every other
  int smooth_stages = 0;
  int smooth_radix[kSmoothMaxStages] = {0};
  // ... above 10 000 points (up to 2^20): two passes, nfft = smooth_n1 * smooth_n2, both within the LDS limit
  int smooth_n1 = 0, smooth_n2 = 0;
  int smooth_stages2 = 0;
  int smooth_radix2[kSmoothMaxStages] = {0};   // the stages of smooth_n2 (smooth_radix: those of smooth_n1)
  float2* d_smooth_z = nullptr;          // [max_frames][n1][n2] between the passes
  float2* d_smooth_tw = nullptr;         // [nfft] exp(-2 pi i k / nfft)
  int chirp_fuse_big = 1;                // tdsa_debug_knob "chirp_fuse_big": 0 = long chirp-z frames run chirp_pre / chirp_post as their own passes
  int chirp_single = 1;                  // tdsa_debug_knob "chirp_single": 0 = chirp-z plans run chirp_pre / two transforms / chirp_post as separate
                                         // kernels (M <= 16384; developer builds: two launches that carry the passes), separate row passes (M > 16384)
  int big_pre_wgs = 0;                   // developer builds, tdsa_debug_knob "big_pre_wgs": empty workgroups launched ahead of every column pass
  int big_group = 64;                    // segments per column-pass / row-pass round (one round for the K = 64 Welch capture)
  int big_fuse_gather = 0;               // tdsa_debug_knob "big_fuse_gather": 1 = Welch captures of one round run row pass + gather + finish as ONE
                                         // launch with a ticket queue (measured: profiles/r06_c5_fused_gather.txt)
  void* d_bigq = nullptr;                // the queue's counters (32 bytes, zeroed once; they only grow)
  unsigned long long bigq_tickets = 0, bigq_rows = 0;   // what the next launch starts from
  // frame lengths that are not a power of two (tdsa_chirp.hip): chirp-z on the m_fft-point frame kernel
  bool chirp = false;
  int m_fft = 0, log2m = 0;              // M = 2^log2m >= 2 nfft - 1
  bool chirp_big = false;                // M > 16384: the two M-point transforms run on the long-frame kernels (N1 x 16384)
  int chirp_split = 0;                   // frames above 2^19 points: H = ceil(N / 2); the convolution runs as four half-length
                                         // sub-convolutions of M = 2^20 points on rows [2F][M] (tdsa_chirp.hip)
  float2* d_chirp_bm = nullptr;          // [M] spectra of the filter segments b[m - H], b[m + H] (d_chirp_b: b[m]), split plans only
  float2* d_chirp_bp = nullptr;
  float2* d_chirp_a = nullptr;           // [nfft] a[n] = exp(-i pi n^2 / nfft)
  float2* d_chirp_b = nullptr;           // [M]    FFT_M of conj(a) wrapped around M
  float2* d_chirp_aw[3] = {nullptr, nullptr, nullptr};   // [nfft] window * input scale * a[n] per input format (M <= 16384: the
                                         // unpack / window / chirp pass rides the first transform's loads)
  float2* d_u0 = nullptr;                // [max_frames][M] work rows (allocated on first use)
  float2* d_u1 = nullptr;
  void* d_scratch = nullptr;             // grows on demand: results of tdsa_rows_stats / tdsa_rows_top_peaks
  size_t scratch_bytes = 0;
  // per-frame scalars (tdsa_set_frame_stats): the results of the last kFsKeep calls.  A call takes the next of ITS stream's
  // kFsKeep slots, so a slot is only ever rewritten by later work of the stream that wrote it (in order, no events between
  // the frame-kernel launches), and the last kFsKeep calls overall are always still there.
  static constexpr int kFsKeep = 4;
  struct FsSlot {
    void* d_part = nullptr;              // [frames][waves per frame] records of the frame kernel's STATS epilogue
    float* d_peak = nullptr;             // [frames]
    int* d_bin = nullptr;
    double* d_band = nullptr;
    size_t cap = 0;
    int n_frames = 0;
    int state = 0;                       // 0: nothing, 1: results (in flight on `stream`), 2: the call produced no rows to take them from
    bool pending = false;                // the frame kernel's records are there, frame_stats_finish_kernel has not run yet
    int wpf = 1;
    double cal_lin = 1.0;
    hipStream_t stream = nullptr;
  } fs[kMaxOverlap][kFsKeep];
  unsigned fs_count[kMaxOverlap] = {};   // calls that took a slot, per stream
  FsSlot* fs_hist[kFsKeep] = {};         // the slots of the last calls, newest at fs_seq - 1
  bool fs_on = false;
  int fs_lo = 1, fs_hi = 0;              // band: inclusive display-bin range, lo > hi = none
  unsigned long long fs_seq = 0;         // calls that left (or tried to leave) statistics
  hipStream_t fs_stream = nullptr;       // folds and reads back a slot
  bool profiling = false;
  bool sync_call = false;                // set by the synchronous host entry points around their device call
  std::vector<hipEvent_t> prof_events;   // pairs (begin, end) around frame-kernel launches
  size_t prof_used = 0;
};

namespace {

int bytes_per_sample(int fmt) { return fmt == TDSA_IN_C64 ? 8 : 2; }

constexpr size_t kPinnedBounceMax = size_t(1) << 20;    // host calls up to 1 MiB each way go through pinned bounce buffers
constexpr size_t kZeroCopyMax = size_t(256) << 10;       // ... and up to 256 KiB in + out are read / written in place by the kernels

bool avg_active(const tdsa_mode& m) { return m.avg_mode != TDSA_AVG_OFF && m.avg_n > 1; }

// Order the plan's main stream after everything in flight on the auxiliary streams.  Every entry point
// that touches plan state or enqueues on the main stream calls this first; work it enqueues afterwards
// is in turn waited for by the next overlapped launch (state_dirty).
int join_streams(tdsa_plan p) {
  if (p->aux_busy) {
    for (int i = 0; i < p->n_overlap - 1; ++i) {
      HIPCHK(hipEventRecord(p->ev_aux[i], p->aux[i]));
      HIPCHK(hipStreamWaitEvent(p->stream, p->ev_aux[i], 0));
    }
    p->aux_busy = false;
  }
  p->state_dirty = true;
  return TDSA_OK;
}
#define JOIN(p)                                  \
  do {                                           \
    int rc_join_ = join_streams(p);              \
    if (rc_join_ != TDSA_OK) return rc_join_;    \
  } while (0)

// stream for the next frame-kernel launch of an order-independent call
int pick_stream(tdsa_plan p, hipStream_t* out) {
  *out = p->stream;
  if (p->n_overlap <= 1) return TDSA_OK;
  const int k = p->rr++ % p->n_overlap;
  if (k == 0) return TDSA_OK;
  if (p->state_dirty) {   // state set up on the main stream (window, fills, tare ...) must be visible
    HIPCHK(hipEventRecord(p->ev_state, p->stream));
    for (int i = 0; i < p->n_overlap - 1; ++i) HIPCHK(hipStreamWaitEvent(p->aux[i], p->ev_state, 0));
    p->state_dirty = false;
  }
  *out = p->aux[k - 1];
  p->aux_busy = true;
  return TDSA_OK;
}

int reset_hold(tdsa_plan p, bool mx, bool mn) {
  if (mx) {
    HIPCHK(launch_fill(p->d_hold_max, p->nfft, -INFINITY, p->stream));
    p->held_max = 0;
  }
  if (mn) {
    HIPCHK(launch_fill(p->d_hold_min, p->nfft, INFINITY, p->stream));
    p->held_min = 0;
  }
  return TDSA_OK;
}

int launch_spectrum_profiled(tdsa_plan p, int in_c64, const SpecParams& sp, const LaunchGeom& g,
                             hipStream_t s = nullptr) {
  if (!p->profiling) {
    HIPCHK(launch_spectrum(p->log2n, in_c64, sp, g, s ? s : p->stream));
    return TDSA_OK;
  }
  if (p->prof_used + 2 > p->prof_events.size()) {
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    p->prof_events.push_back(a);
    p->prof_events.push_back(b);
  }
  HIPCHK(hipEventRecord(p->prof_events[p->prof_used], p->stream));
  HIPCHK(launch_spectrum(p->log2n, in_c64, sp, g, p->stream));
  HIPCHK(hipEventRecord(p->prof_events[p->prof_used + 1], p->stream));
  p->prof_used += 2;
  return TDSA_OK;
}

// Long averaged batches of the sizes whose frame kernel cannot form the scan's chunk aggregates itself (several frames per
// workgroup slot below 4096 points; chirp-z plans): the chunks become up to 256 equal ranges of the batch, a pass of the
// scan's own forms their float32 aggregates (avg_agg_local_kernel) and the two-level chain of the workgroup-chunk path
// takes it from there.  With fixed chunks of 64 frames one thread per bin walked hundreds of chunks in order: 75 of the
// 169 us of a 19 531-frame batch at N = 1024.  Short batches keep the 64-frame chunks.
static int avg_use_ranges(tdsa_plan p, AvgParams& ap, int n_frames, hipStream_t s) {
  if (p->avg_f64_chunks || n_frames <= 1024 || p->d_carry == nullptr) return TDSA_OK;
  const int ranges = (n_frames + 63) / 64 < 256 ? (n_frames + 63) / 64 : 256;
  if (size_t(ranges) > p->carry_chunks) return TDSA_OK;
  // a range's aggregate is a float32 sum: kept to the ~160 frames the workgroup-chunk path allows itself; longer
  // ranges (batches of more than 40 960 frames here) take the 64-frame chunks with their float64 aggregates - the
  // reference's TraceAverager is float64 throughout (utils/signal_processing.py:48)
  if ((n_frames + ranges - 1) / ranges > 160) return TDSA_OK;
  const size_t row = size_t(p->nfft);                  // (rows of the real-input path are shorter: nfft / 2 + 1)
  if (!p->d_agg) {      // (a native plan may also take the workgroup-chunk path on another call: one row per workgroup of its grid)
    const size_t grid_rows = (p->chirp || p->big) ? 0 : size_t(spectrum_geometry(p->log2n, p->max_frames, p->num_cu).grid);
    HIPCHK(hipMalloc(&p->d_agg, (grid_rows > 256 ? grid_rows : size_t(256)) * row * sizeof(float)));
  }
  if (!p->d_agg_w) HIPCHK(hipMalloc(&p->d_agg_w, size_t(p->max_frames) * sizeof(float)));
  if (!p->d_chunk_a) HIPCHK(hipMalloc(&p->d_chunk_a, size_t(kAvgMaxWgChunks + 64) * sizeof(double)));
  if (!p->d_chunk_v) HIPCHK(hipMalloc(&p->d_chunk_v, size_t(kAvgMaxWgChunks + 64) * sizeof(float)));
  ap.chunk_a = p->d_chunk_a;
  ap.chunk_v = p->d_chunk_v;
  ap.wg_chunks = ranges;
  ap.wg_fold = 1;
  ap.agg = p->d_agg;
  ap.agg_w_local = p->d_agg_w;
  const long long key[7] = {p->avg_count, ap.mode, ap.avg_n, n_frames, ranges, 1, 1};          // path 1: equal ranges of the batch
  if (std::memcmp(key, p->agg_w_key, sizeof(key)) != 0) {
    HIPCHK(launch_avg_weights(ap, p->d_agg_w, p->d_chunk_a, p->d_chunk_v, s));
    std::memcpy(p->agg_w_key, key, sizeof(key));
  }
  return TDSA_OK;
}

// Long-frame plans.  Modes (decided by the plan's averaging settings):
//   * "lin" with avg_n >= frames seen so far + n_frames: Welch - the K segments of the call (and of earlier
//     calls since the last reset) are averaged, out_db_dev receives ONE row, the dB of the running mean;
//   * otherwise one frame per call (n_frames == 1): plain dB, or TraceAverager exp / capped lin on the
//     float64 state exactly as for the LDS-resident sizes.
// TraceAverager._buffer of a long-frame plan after Welch calls: the gather leaves the float64 SUM of the segments and the
// dB row; the mean itself (8 N bytes more per capture) is only formed for whoever reads the state - tdsa_get_avg,
// tdsa_welch_export, or the capped running mean that takes over once avg_n frames have been seen.
static int big_materialize_mean(tdsa_plan p) {
  if (p->big_mean_in_sum && p->avg_count > 0)
    HIPCHK(launch_big_finish(p->d_sum, (long long)p->nfft, p->d_avg, p->avg_count, TDSA_DB_POW, 1.0f, 1.0f, 0.0f, nullptr,
                             nullptr, nullptr, nullptr, 0, 0, p->stream));
  p->big_mean_in_sum = false;
  return TDSA_OK;
}

int process_big(tdsa_plan p, int in_format, const void* iq_dev, int hop, int n_frames, float* out_db_dev) {
  const tdsa_mode& m = p->mode;
  const bool averaging = avg_active(m);
  const bool welch = averaging && m.avg_mode == TDSA_AVG_LIN && (long long)p->avg_count + n_frames <= m.avg_n;
  if (!welch && n_frames != 1)
    return fail(TDSA_ERR_ARG, "a %d-point plan takes one frame per call unless it is Welch-averaging "
                "(avg lin with avg_n >= total frames: count %d + %d > %d)", p->nfft, p->avg_count, n_frames,
                averaging ? m.avg_n : 0);
  const size_t N = size_t(p->nfft);
  const int n1 = p->nfft >> 14;
  const int group = n_frames < p->big_group ? n_frames : p->big_group;
  if (!p->d_z) HIPCHK(hipMalloc(&p->d_z, size_t(p->max_frames < p->big_group ? p->max_frames : p->big_group) * N * sizeof(float2)));
  const int in_c64 = in_format == TDSA_IN_C64;
  const unsigned xor_mask = in_format == TDSA_IN_I8 ? 0x80808080u : 0u;
  const float in_off = in_format == TDSA_IN_I8 ? 128.0f : (in_c64 ? 0.0f : 127.5f);
  const float in_scale = in_format == TDSA_IN_I8 ? 1.0f / 128.0f : (in_c64 ? 1.0f : 1.0f / 127.5f);
  const long long stride = (long long)hop * bytes_per_sample(in_format);
  const float2* dc_sub = nullptr;
  if (m.dc_alpha >= 0.0f) {   // per-segment mean (alpha = 1) or tracker (alpha < 1)
    if (!p->d_sums64) HIPCHK(hipMalloc(&p->d_sums64, size_t(p->max_frames) * 2 * sizeof(double)));
    HIPCHK(launch_big_dc(iq_dev, in_c64, xor_mask, stride, p->nfft, n_frames, m.dc_alpha > 1.0f ? 1.0 : double(m.dc_alpha),
                         double(in_off), double(in_scale), p->d_sums64, p->d_dc_state, p->d_dc_sub, p->stream));
    dc_sub = p->d_dc_sub;
  }
  // column pass and row pass alternate over rounds of segments.  The row pass leaves per-workgroup partial power
  // sums in d_acc (P[k1 * split + j][k2]): the first round of a call overwrites its rows, later rounds add to them,
  // the gather sums over j - nothing is carried from call to call, so a failed call leaves no residue
  const int split_max = p->num_cu / n1 > 1 ? p->num_cu / n1 : 1;
  int split_layout = 1;
  bool fused_tail = false;
  for (int s0 = 0; s0 < n_frames; s0 += group) {
    const int ns = n_frames - s0 < group ? n_frames - s0 : group;
    const int act = ns < split_max ? ns : split_max;
    if (s0 == 0) split_layout = act;
#ifdef TDSA_DEV
    if (p->big_pre_wgs > 0) HIPCHK(launch_xcd_shift(p->big_pre_wgs, p->stream));
#endif
    HIPCHK(launch_big_cols(p->log2n, static_cast<const unsigned char*>(iq_dev) + (long long)s0 * stride, in_c64, stride, ns,
                           p->big_win[in_format], p->d_tw_seed, dc_sub ? dc_sub + s0 : nullptr, p->d_z,
                           xor_mask, in_off, p->stream));
    if (p->profiling) {
      if (p->prof_used + 2 > p->prof_events.size()) {
        hipEvent_t a, b;
        HIPCHK(hipEventCreate(&a));
        HIPCHK(hipEventCreate(&b));
        p->prof_events.push_back(a);
        p->prof_events.push_back(b);
      }
      HIPCHK(hipEventRecord(p->prof_events[p->prof_used], p->stream));
    }
    fused_tail = welch && p->big_fuse_gather && !p->profiling && n_frames <= group;
    if (fused_tail) {
      if (!p->d_bigq) {
        HIPCHK(hipMalloc(&p->d_bigq, 32));
        HIPCHK(hipMemsetAsync(p->d_bigq, 0, 32, p->stream));
        p->bigq_tickets = p->bigq_rows = 0;
      }
      unsigned long long used = 0;
      p->bigq_rows += (unsigned long long)n1 * act;
      HIPCHK(launch_big_rows_gather(p->log2n, p->d_z, (long long)N * sizeof(float2), ns, n1, act, p->d_acc, p->d_tw_row, p->d_sum,
                                    p->avg_count > 0, nullptr, p->avg_count + n_frames, m.db_mode,
                                    m.db_mode == TDSA_DB_POW ? m.power_scale : 1.0f, m.log_floor, m.cal_offset_db,
                                    p->tare_active ? p->d_tare_base : nullptr, out_db_dev,
                                    (m.hold_flags & TDSA_HOLD_MAX) ? p->d_hold_max : nullptr,
                                    (m.hold_flags & TDSA_HOLD_MIN) ? p->d_hold_min : nullptr, p->held_max == 0, p->held_min == 0,
                                    p->d_bigq, p->bigq_tickets, p->bigq_rows, &used, p->big_fuse_gather >> 1, p->stream));
      p->bigq_tickets += used;
      break;
    }
    HIPCHK(launch_big_rows(p->d_z, (long long)N * sizeof(float2), ns, n1, act, p->d_acc, split_layout, s0 > 0, p->d_tw_row,
                                p->stream));
    if (p->profiling) {
      HIPCHK(hipEventRecord(p->prof_events[p->prof_used + 1], p->stream));
      p->prof_used += 2;
    }
  }
  const bool hmax = (m.hold_flags & TDSA_HOLD_MAX) != 0, hmin = (m.hold_flags & TDSA_HOLD_MIN) != 0;
  const float pscale = m.db_mode == TDSA_DB_POW ? m.power_scale : 1.0f;
  float* const tare = p->tare_active ? p->d_tare_base : nullptr;
  float* const hold_max = hmax ? p->d_hold_max : nullptr;
  float* const hold_min = hmin ? p->d_hold_min : nullptr;
  if (welch && fused_tail) {
    p->avg_count += n_frames;
    p->big_mean_in_sum = true;
  } else if (welch) {
    HIPCHK(launch_big_gather_finish(p->log2n, p->d_acc, split_layout, p->d_sum, p->avg_count > 0, nullptr, p->avg_count + n_frames,
                                    m.db_mode, pscale, m.log_floor, m.cal_offset_db, tare, out_db_dev, hold_max, hold_min,
                                    p->held_max == 0, p->held_min == 0, p->stream));
    p->avg_count += n_frames;
    p->big_mean_in_sum = true;
  } else if (averaging) {    // TraceAverager exp / capped lin, one frame (signal_processing.py:35-61)
    { const int rc_m = big_materialize_mean(p); if (rc_m != TDSA_OK) return rc_m; }
    HIPCHK(launch_big_gather(p->log2n, p->d_acc, split_layout, p->d_lin64, 0, p->stream));
    HIPCHK(launch_avg_host_frame(p->d_lin64, p->nfft, p->d_avg, p->avg_count, m.avg_mode, m.avg_n, p->stream));
    if (p->avg_count == 0) p->avg_count = 1;
    else if (m.avg_mode == TDSA_AVG_LIN && p->avg_count < m.avg_n) p->avg_count += 1;
    HIPCHK(launch_big_finish(p->d_avg, (long long)N, nullptr, 1, m.db_mode, pscale, m.log_floor, m.cal_offset_db, tare,
                             out_db_dev, hold_max, hold_min, p->held_max == 0, p->held_min == 0, p->stream));
  } else {
    HIPCHK(launch_big_gather_finish(p->log2n, p->d_acc, split_layout, p->d_lin64, 0, nullptr, 1, m.db_mode, pscale, m.log_floor,
                                    m.cal_offset_db, tare, out_db_dev, hold_max, hold_min, p->held_max == 0,
                                    p->held_min == 0, p->stream));
  }
  if (hmax) p->held_max += 1;
  if (hmin) p->held_min += 1;
  p->frames_seen += n_frames;
  return TDSA_OK;
}

// Chirp-z core of a plan whose frame length is not a power of two: frames at `in` (stride bytes apart) ->
// p->d_u0[f][k] = M * conj(convolution), k < nfft (tdsa_chirp.hip steps 1-3), on the main stream.
// the transforms can carry the element-wise passes: M <= 16384 and frames made of whole waves (the fused instantiations
// address their rows through wave-uniform descriptors)
// (M > 16384: the column passes of tdsa_big.hip carry them instead, BigChirpPre / BigChirpPost)
#ifdef TDSA_DEV
constexpr bool kChirpTwoLaunches = true;    // spectrum_kernel<L, true, 0, 1 | 2>: each transform carries one element-wise pass
#else
constexpr bool kChirpTwoLaunches = false;   // shipped: one launch, or (tdsa_debug_knob chirp_single 0) the passes as kernels of their own
#endif
static bool chirp_fusable(tdsa_plan p) {
  if (!p->chirp || p->log2m < 10) return false;
  return p->chirp_big ? p->chirp_fuse_big != 0 : (p->chirp_single != 0 || kChirpTwoLaunches);
}

// post (fusable plans only): what the second transform's stores turn the bins into - the dB / power rows and hold traces
// of tdsa_chirp.hip's step 4 - instead of leaving complex rows in d_u0 for chirp_post_kernel; null: complex rows
struct ChirpPost {
  int first_frame_index, db_mode;
  float pscale, log_floor, cal_db;
  const float* tare;
  float* out_db;
  float* out_lin;
  float* hold_max;
  float* hold_min;
};
int chirp_transform(tdsa_plan p, const void* in, int in_format, long long stride, int n_frames, const float2* dc_sub,
                    unsigned xor_mask, float in_off, const ChirpPost* post = nullptr) {
  const int N = p->nfft, M = p->m_fft;
  hipStream_t s = p->stream;
  const bool fused = chirp_fusable(p) && post != nullptr;  // both element-wise passes ride the transforms
  const bool bfused = fused && p->chirp_big;
  const int H = p->chirp_split;                            // > 0: two half-length rows per frame
  const size_t rows_max = size_t(p->max_frames) * (H ? 2 : 1);
  const int n_rows = n_frames * (H ? 2 : 1);
  if (!p->d_u0 && !fused) HIPCHK(hipMalloc(&p->d_u0, rows_max * M * sizeof(float2)));
  if (!p->d_u1 && !(fused && p->chirp_single && !p->chirp_big)) HIPCHK(hipMalloc(&p->d_u1, rows_max * M * sizeof(float2)));
  if (!fused)
    HIPCHK(launch_chirp_pre(in, in_format == TDSA_IN_C64, stride, N, M, n_frames, p->d_window[in_format], p->d_chirp_a, dc_sub,
                            xor_mask, in_off, p->d_u0, s, H));
  if (p->chirp_big) {
    // M = N1 x 16384: first transform as for a native long frame (column pass -> rows through the frame kernel, which
    // stores conj(X B) in its own [k1][k2] order); second transform transposed (rows first, then the per-column N1-point
    // DFT that leaves natural order): tdsa_big.hip
    const int n1 = M >> kMaxLog2N;
    const long long rowb = (long long)(1 << kMaxLog2N) * sizeof(float2), segb = (long long)M * sizeof(float2);
    if (!p->d_z) HIPCHK(hipMalloc(&p->d_z, rows_max * M * sizeof(float2)));
    BigWindow flat{};                   // the rows are windowed already (chirp_pre): one for every sample
    flat.mode = 2;
    flat.table = p->d_ones;
    flat.flat = 1.0f;
    if (bfused) {                       // ... or are never stored: the raw frames are unpacked by the column pass itself
      const BigChirpPre pre{p->d_chirp_aw[in_format], N, in_format == TDSA_IN_C64, H};
      HIPCHK(launch_big_cols(p->log2m, in, 1, stride, n_rows, flat, p->d_tw_seed, dc_sub, p->d_z, xor_mask, in_off, s, 0u, &pre));
    } else {
      HIPCHK(launch_big_cols(p->log2m, p->d_u0, 1, segb, n_rows, flat, p->d_tw_seed, nullptr, p->d_z, 0u, 0.0f, s,
                             unsigned(H ? H : N)));
    }
    // the last column pass turns the bins into the dB / power rows (fused plans); hold traces from the finished rows
    const BigChirpPost bpost = bfused ? BigChirpPost{N, H, 1.0f / float(M), post->db_mode, post->pscale, post->log_floor,
                                                     post->cal_db, post->tare, post->out_db, post->out_lin}
                                      : BigChirpPost{};
    const auto hold_rows = [&]() -> int {
      if (bfused && post->out_lin == nullptr && (post->hold_max || post->hold_min))
        HIPCHK(launch_chirp_hold(post->out_db, N, n_frames, post->first_frame_index, post->hold_max, post->hold_min, s));
      return TDSA_OK;
    };

    SpecParams sp{};
    sp.frame_stride = rowb;
    sp.n_frames = n_rows * n1;
    sp.first_frame_index = 1;
    sp.window = p->d_ones;
    sp.window_perm = p->d_ones;
    sp.tw = p->d_tw;
    sp.in_scale = 1.0f;
    sp.dc_mode = DC_NONE;
    sp.db_mode = TDSA_DB_POW;
    sp.pscale = 1.0f;
    const LaunchGeom g = spectrum_geometry(kMaxLog2N, sp.n_frames, p->num_cu);
    sp.in = p->d_z;
    sp.out_cplx = p->d_u1;
    sp.out_mul = H ? nullptr : p->d_chirp_b;       // [k1][k2] order, row k1 = frame mod N1
    sp.out_mul_rows = H ? 0 : n1;
    if (!H && p->chirp_single) {
      // the row pass of the first transform and the row pass of the transposed second one work on the SAME row k1 (its
      // bins k1 + N1 k2 over k2): one pass through the workgroup does both, with the filter multiply between them
      // (spectrum_kernel<14, true, 0, 4>) - the rows are written once and read once less
      sp.rows_twice = 1;
      { const int rc = launch_spectrum_profiled(p, 1, sp, g); if (rc != TDSA_OK) return rc; }
      HIPCHK(launch_big_cols_out(p->log2m, p->d_u1, segb, n_rows, p->d_tw_seed, p->d_u0, unsigned(N), s, bfused ? &bpost : nullptr));
      return hold_rows();
    }
    { const int rc = launch_spectrum_profiled(p, 1, sp, g); if (rc != TDSA_OK) return rc; }
    if (H)    // split plans: the two half-rows' spectra meet the three filter segments: conj(UA B0 + UB Bm), conj(UA Bp + UB B0)
      HIPCHK(launch_chirp_split_combine(p->d_u1, (long long)M, n_frames, p->d_chirp_b, p->d_chirp_bm, p->d_chirp_bp, s));
    sp.in = p->d_u1;
    sp.out_cplx = p->d_z;
    sp.out_mul = nullptr;
    sp.out_mul_rows = 0;
    { const int rc = launch_spectrum_profiled(p, 1, sp, g); if (rc != TDSA_OK) return rc; }

    HIPCHK(launch_big_cols_out(p->log2m, p->d_z, segb, n_rows, p->d_tw_seed, p->d_u0, unsigned(H ? H : N), s, bfused ? &bpost : nullptr));
    return hold_rows();
  }
  SpecParams sp{};
  sp.frame_stride = (long long)M * sizeof(float2);
  sp.n_frames = n_frames;
  sp.first_frame_index = 1;
  sp.window = p->d_ones;
  sp.window_perm = p->d_ones;
  sp.tw = p->d_tw;
  sp.in_scale = 1.0f;
  sp.dc_mode = DC_NONE;
  sp.db_mode = TDSA_DB_POW;
  sp.pscale = 1.0f;
  const LaunchGeom g = spectrum_geometry(p->log2m, n_frames, p->num_cu);
  sp.in = p->d_u0;
  sp.out_cplx = p->d_u1;
  sp.out_mul = p->d_chirp_b;          // the first transform stores conj(FFT_M(U) * B)
  sp.in_valid = N;                    // rows of U: N samples, the padding up to M is neither written nor read
  if (fused) {                        // ... and U itself is never stored: the raw frames are unpacked on load
    sp.in = in;
    sp.pre_raw = in;
    sp.pre_stride = stride;
    sp.pre_aw = p->d_chirp_aw[in_format];
    sp.pre_c64 = in_format == TDSA_IN_C64;
    sp.dc_sub = dc_sub;
    sp.pre_xor = xor_mask;
    sp.pre_off = in_off;
  }
  const auto set_post = [&] {         // what the second transform's stores turn the bins into
    sp.out_cplx = nullptr;
    sp.out_valid = 0;
    sp.post_n = N;
    sp.post_inv_m = 1.0f / float(M);
    sp.first_frame_index = post->first_frame_index;
    sp.db_mode = post->db_mode;
    sp.pscale = post->pscale;
    sp.log_floor = post->log_floor;
    sp.cal_db = post->cal_db;
    sp.tare = post->tare;
    sp.out_db = post->out_db;
    sp.out_lin = post->out_lin;
  };
  if (fused && p->chirp_single) {
    // the whole convolution of a frame in one pass through its workgroup (spectrum_kernel<L, true, 0, 3>): transform,
    // x B, conjugate, through LDS back into sample order, transform, dB rows - the complex64 intermediate never leaves the CU
    set_post();
    { const int rc = launch_spectrum_profiled(p, 1, sp, g); if (rc != TDSA_OK) return rc; }
    if (post->out_lin == nullptr && (post->hold_max || post->hold_min))
      HIPCHK(launch_chirp_hold(post->out_db, N, n_frames, post->first_frame_index, post->hold_max, post->hold_min, s));
    return TDSA_OK;
  }
  { const int rc = launch_spectrum_profiled(p, 1, sp, g); if (rc != TDSA_OK) return rc; }
  sp.in = p->d_u1;
  sp.pre_raw = nullptr;
  sp.dc_sub = nullptr;
  sp.out_cplx = p->d_u0;
  sp.out_mul = nullptr;
  sp.in_valid = 0;
  sp.out_valid = N;                   // only bins k < N of the convolution are needed
  if (fused) set_post();              // ... and leave as the dB / power rows themselves
  { const int rc = launch_spectrum_profiled(p, 1, sp, g); if (rc != TDSA_OK) return rc; }
  if (fused && post->out_lin == nullptr && (post->hold_max || post->hold_min))
    HIPCHK(launch_chirp_hold(post->out_db, N, n_frames, post->first_frame_index, post->hold_max, post->hold_min, s));
  return TDSA_OK;
}

// Plans whose frame length is not a power of two (tdsa_chirp.hip): same modes, same state, same outputs as the
// native sizes - every stage on the plan's main stream.
int process_chirp(tdsa_plan p, int in_format, const void* iq_dev, int hop, int n_frames, float* out_db_dev) {
  const tdsa_mode& m = p->mode;
  const bool averaging = avg_active(m);
  const int N = p->nfft, M = p->m_fft;
  const int in_c64 = in_format == TDSA_IN_C64;
  const unsigned xor_mask = in_format == TDSA_IN_I8 ? 0x80808080u : 0u;
  const float in_off = in_format == TDSA_IN_I8 ? 128.0f : (in_c64 ? 0.0f : 127.5f);
  const float in_scale = in_format == TDSA_IN_I8 ? 1.0f / 128.0f : (in_c64 ? 1.0f : 1.0f / 127.5f);
  const long long stride = (long long)hop * bytes_per_sample(in_format);
  hipStream_t s = p->stream;
  const float2* dc_sub = nullptr;
  const bool smooth = p->smooth && p->smooth_on;      // a transform of exactly N points instead of the convolution
  // ... whose kernel forms the frame means of byte samples itself when the call has few frames (a GUI tick has one: a launch
  // less, 34 -> 28 us per host call at N = 1000; in batches the frame-by-frame reductions cost more than the sums kernel)
  const bool dc_own = smooth && p->smooth_n1 == 0 && !in_c64 && m.dc_alpha >= 1.0f && n_frames <= 8;
  const int twice_zero = in_format == TDSA_IN_I8 ? 256 : (in_c64 ? 0 : 255);
  if (m.dc_alpha >= 0.0f && !dc_own) {
    // frame means as residuals (exact sums); 0 <= alpha < 1: the tracker of the native path fed with them
    // directly (n = 1, zero level 0)
    const bool tracked = m.dc_alpha < 1.0f;
    if (chirp_sum_chunks(N) > 1 && !p->d_sums64)      // long frames: several workgroups per frame leave partial sums here
      HIPCHK(hipMalloc(&p->d_sums64, size_t(p->max_frames) * chirp_sum_chunks(N) * 2 * sizeof(double)));
    HIPCHK(launch_chirp_sums(iq_dev, in_c64, xor_mask, stride, N, n_frames, twice_zero, p->d_sums,
                             tracked ? nullptr : p->d_dc_state, in_scale, s, p->d_sums64));
    dc_sub = p->d_sums;                   // dc_alpha >= 1: the frame's own mean
    if (tracked) {
      HIPCHK(launch_dc_track(p->d_sums, 1, n_frames, m.dc_alpha, 0.0f, in_scale, p->d_dc_state, p->d_dc_sub, s));
      dc_sub = p->d_dc_sub;
    }
  }
  const float pscale = m.db_mode == TDSA_DB_POW ? m.power_scale : 1.0f;
  float* const tare = p->tare_active ? p->d_tare_base : nullptr;
  const int first = p->frames_seen > 0 ? 1 : 0;
  if (averaging && !p->d_lin) HIPCHK(hipMalloc(&p->d_lin, size_t(p->max_frames) * N * sizeof(float)));
  // the power / dB rows leave the second transform directly (linear rows for the averager's scan, else dB rows + hold
  // traces); frames below 1024 points (and the A/B knobs): complex rows in d_u0, chirp_post below
  const bool fusable = chirp_fusable(p) || smooth;
  const bool holding = (m.hold_flags & (TDSA_HOLD_MAX | TDSA_HOLD_MIN)) != 0;
  float* rows = out_db_dev;
  if (fusable && !averaging && rows == nullptr && holding) {   // only the hold traces are wanted: the rows go to scratch
    if (!p->d_u0) HIPCHK(hipMalloc(&p->d_u0, size_t(p->max_frames) * (p->chirp_split ? 2 : 1) * M * sizeof(float2)));
    rows = reinterpret_cast<float*>(p->d_u0);
  }
  if (fusable && !averaging && rows == nullptr) {              // nothing to produce (no rows, no hold, no averaging)
    if (dc_own)                                                // (but the estimate the plan carries moves on)
      HIPCHK(launch_chirp_sums(iq_dev, in_c64, xor_mask, stride, N, n_frames, twice_zero, p->d_sums, p->d_dc_state, in_scale, s,
                               p->d_sums64));
    p->frames_seen += n_frames;
    return TDSA_OK;
  }
  const ChirpPost post{first, m.db_mode, pscale, m.log_floor, m.cal_offset_db, averaging ? nullptr : tare,
                       averaging ? nullptr : rows, averaging ? p->d_lin : nullptr,
                       (!averaging && (m.hold_flags & TDSA_HOLD_MAX)) ? p->d_hold_max : nullptr,
                       (!averaging && (m.hold_flags & TDSA_HOLD_MIN)) ? p->d_hold_min : nullptr};
  if (smooth) {
    SmoothParams sp{};
    sp.in = iq_dev;
    sp.in_c64 = in_c64;
    sp.frame_stride = stride;
    sp.n = N;
    sp.n_frames = n_frames;
    sp.n_stages = p->smooth_stages;
    for (int i = 0; i < p->smooth_stages; ++i) sp.radix[i] = p->smooth_radix[i];
    sp.tw = p->d_smooth_tw;
    sp.tw_step = 1;
    sp.window = p->d_window[in_format];
    sp.dc_sub = dc_sub;
    sp.dc_own = dc_own;
    sp.twice_zero = twice_zero;
    sp.in_scale = in_scale;
    sp.dc_state = p->d_dc_state;
    sp.xor_mask = xor_mask;
    sp.in_off = in_off;
    sp.db_mode = post.db_mode;
    sp.pscale = post.pscale;
    sp.log_floor = post.log_floor;
    sp.cal_db = post.cal_db;
    sp.tare = post.tare;
    sp.out_db = post.out_db;
    sp.out_lin = post.out_lin;
    if (p->smooth_n1 == 0) {
      HIPCHK(launch_smooth(sp, s));
    } else {
      // above the LDS limit: column pass (n1-point transforms of fpw adjacent columns, times W_N^(n2 k1)) into z, row pass
      // (n2-point transforms of adjacent rows k1) from z to the dB rows
      if (!p->d_smooth_z) HIPCHK(hipMalloc(&p->d_smooth_z, size_t(p->max_frames) * N * sizeof(float2)));
      sp.n_total = N;
      sp.n1 = p->smooth_n1;
      sp.n2 = p->smooth_n2;
      sp.z = p->d_smooth_z;
      sp.n = p->smooth_n1;
      sp.tw_step = p->smooth_n2;
      HIPCHK(launch_smooth(sp, s, 1));
      sp.n = p->smooth_n2;
      sp.tw_step = p->smooth_n1;
      sp.n_stages = p->smooth_stages2;
      for (int i = 0; i < p->smooth_stages2; ++i) sp.radix[i] = p->smooth_radix2[i];
      HIPCHK(launch_smooth(sp, s, 2));
    }
    if (post.out_lin == nullptr && (post.hold_max || post.hold_min))
      HIPCHK(launch_chirp_hold(post.out_db, N, n_frames, post.first_frame_index, post.hold_max, post.hold_min, s));
  } else
  { const int rc = chirp_transform(p, iq_dev, in_format, stride, n_frames, dc_sub, xor_mask, in_off, fusable ? &post : nullptr); if (rc != TDSA_OK) return rc; }
  if (averaging) {
    if (!p->d_carry && p->max_frames > 128) {
      HIPCHK(hipMalloc(&p->d_carry, size_t(avg_scan_chunks(p->max_frames)) * N * sizeof(double)));
      p->carry_chunks = size_t(avg_scan_chunks(p->max_frames));
    }
    if (!fusable)
      HIPCHK(launch_chirp_post(p->d_u0, N, M, n_frames, first, m.db_mode, pscale, m.log_floor,
                               m.cal_offset_db, nullptr, nullptr, p->d_lin, nullptr, nullptr, s, p->chirp_split));
    AvgParams ap{};
    ap.lin = p->d_lin;
    ap.n_frames = n_frames;
    ap.n = N;
    ap.state = p->d_avg;
    ap.count_in = p->avg_count;
    ap.mode = m.avg_mode;
    ap.avg_n = m.avg_n;
    ap.log_floor = m.log_floor;
    ap.cal_db = m.cal_offset_db;
    ap.tare = tare;
    ap.out_db = out_db_dev;
    ap.state_max = (m.hold_flags & TDSA_HOLD_MAX) ? p->d_hold_max : nullptr;
    ap.state_min = (m.hold_flags & TDSA_HOLD_MIN) ? p->d_hold_min : nullptr;
    { const int rc = avg_use_ranges(p, ap, n_frames, s); if (rc != TDSA_OK) return rc; }
    HIPCHK(launch_avg_scan(ap, s, p->d_carry));
    if (m.avg_mode == TDSA_AVG_LIN) {
      const long long c = (long long)p->avg_count + n_frames;
      p->avg_count = int(c < m.avg_n ? c : m.avg_n);
    } else {
      p->avg_count = 1;
    }
  } else if (!fusable) {
    HIPCHK(launch_chirp_post(p->d_u0, N, M, n_frames, first, m.db_mode, pscale, m.log_floor,
                             m.cal_offset_db, tare, out_db_dev, nullptr,
                             (m.hold_flags & TDSA_HOLD_MAX) ? p->d_hold_max : nullptr,
                             (m.hold_flags & TDSA_HOLD_MIN) ? p->d_hold_min : nullptr, s, p->chirp_split));
  }
  if (m.hold_flags & TDSA_HOLD_MAX) p->held_max += n_frames;
  if (m.hold_flags & TDSA_HOLD_MIN) p->held_min += n_frames;
  p->frames_seen += n_frames;
  return TDSA_OK;
}

}  // namespace

extern "C" {

const char* tdsa_last_error_string(void) { return g_err; }
int tdsa_version(void) { return TDSA_VERSION; }

int tdsa_device_count(int* count) {
  if (!count) return fail(TDSA_ERR_ARG, "count is null");
  HIPCHK(hipGetDeviceCount(count));
  return TDSA_OK;
}

static int plan_init(tdsa_plan p);

int tdsa_create(int device_id, int nfft, int max_frames, tdsa_plan* out) {
  if (!out) return fail(TDSA_ERR_ARG, "out is null");
  *out = nullptr;
  const bool native = nfft >= (1 << kMinLog2N) && nfft <= (1 << kBigMaxLog2N) && (nfft & (nfft - 1)) == 0;
  const bool chirp = !native && nfft >= 2 && nfft <= kChirpMaxN;
  if (!native && !chirp)
    return fail(TDSA_ERR_ARG, "nfft=%d: need a power of two in [%d, %d] or any size in [2, %d]", nfft, 1 << kMinLog2N,
                1 << kBigMaxLog2N, kChirpMaxN);
  const bool big = native && nfft > (1 << kMaxLog2N);
  if (max_frames < 1) return fail(TDSA_ERR_ARG, "max_frames=%d must be >= 1", max_frames);
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device_id < 0 || device_id >= ndev) return fail(TDSA_ERR_ARG, "device %d of %d", device_id, ndev);
  HIPCHK(hipSetDevice(device_id));
  tdsa_plan p = new (std::nothrow) tdsa_plan_s();
  if (!p) return fail(TDSA_ERR_NOMEM, "host allocation failed");
  p->device = device_id;
  p->nfft = nfft;
  p->log2n = ilog2i(nfft);
  p->max_frames = max_frames;
  p->big = big;
  p->chirp = chirp;
  if (chirp) {
    int m = 1 << kMinLog2N;
    if (nfft > (1 << 19)) {                  // 2^21 would be needed: split into half-length sub-convolutions of 2^20
      p->chirp_split = (nfft + 1) / 2;
      m = 1 << kBigMaxLog2N;
    } else {
      while (m < 2 * nfft - 1) m <<= 1;
    }
    p->m_fft = m;
    p->log2m = ilog2i(m);
    p->chirp_big = p->log2m > kMaxLog2N;
    p->log2n = p->chirp_big ? kMaxLog2N : p->log2m;      // what the frame kernel of this plan transforms
  }
  const int rc_init = plan_init(p);          // a failure half way leaves nothing behind
  if (rc_init != TDSA_OK) {
    (void)tdsa_destroy(p);
    return rc_init;
  }
  *out = p;
  return TDSA_OK;
}

// n = 2^a 3^b 5^c: the radices of its stages (4 while it divides, then 2, 3, 5); 0 stages: other factors
static int smooth_radices(int n, int* radix) {
  int r = n, st = 0;
  while (r % 4 == 0 && st < kSmoothMaxStages) { radix[st++] = 4; r /= 4; }
  for (const int f : {2, 3, 5})
    while (r % f == 0 && st < kSmoothMaxStages) { radix[st++] = f; r /= f; }
  return r == 1 ? st : 0;
}
static void smooth_plan(tdsa_plan p) {
  if (p->nfft <= kSmoothMaxN) {
    p->smooth_stages = smooth_radices(p->nfft, p->smooth_radix);
    return;
  }
  // two passes, both factors within the LDS limit; the column pass's length n1 near 128 measured best (N = 10^6: 309 us
  // per ten frames at 125 x 8000 against 389 at 1000 x 1000; N = 20 000: 185 at 125 x 160 against 254 at 2 x 10 000 -
  // short columns let a workgroup take sixteen adjacent ones, whose raw samples then sit side by side)
  int best = 0;
  double best_d = 1e30;
  for (int d = 2; d <= kSmoothMaxN && d <= p->nfft / 2; ++d)
    if (p->nfft % d == 0 && p->nfft / d <= kSmoothMaxN) {
      const double dist = std::fabs(std::log(double(d) / 128.0));
      if (dist < best_d) { best_d = dist; best = d; }
    }
  if (best == 0) return;
  p->smooth_n1 = best;
  p->smooth_n2 = p->nfft / best;
  p->smooth_stages = smooth_radices(p->smooth_n1, p->smooth_radix);
  p->smooth_stages2 = smooth_radices(p->smooth_n2, p->smooth_radix2);
  if (p->smooth_stages == 0 || p->smooth_stages2 == 0) p->smooth_stages = p->smooth_stages2 = p->smooth_n1 = p->smooth_n2 = 0;
}

static int plan_init(tdsa_plan p) {
  const int device_id = p->device, nfft = p->nfft, max_frames = p->max_frames;
  const bool big = p->big;
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device_id));
  p->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  // one thread per bin walking the frames beats the three launches of the chunked scan up to ~48 frames at N <= 4096
  // (N = 1024, 128 frames: 24.5 against 13.9 us) and up to ~128 at the larger sizes (N = 16384: 33.7 against 35.0 us)
  p->avg_wg_min = nfft <= 4096 ? 48 : 128;
  HIPCHK(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreate(&p->ev0));
  HIPCHK(hipEventCreate(&p->ev1));
  HIPCHK(hipEventCreateWithFlags(&p->ev_state, hipEventDisableTiming));
  const size_t nb = size_t(nfft) * sizeof(float);
  for (int f = 0; f < 3; ++f) HIPCHK(hipMalloc(&p->d_window[f], nb));
  if (!p->big && !p->chirp && p->log2n >= 11)     // only the 3-pass sizes read the permuted table (Cfg::WIN_LDS below)
    for (int f = 0; f < 3; ++f) HIPCHK(hipMalloc(&p->d_window_perm[f], nb));
  const int tw_n = p->chirp ? (p->chirp_big ? (1 << kMaxLog2N) : p->m_fft) : nfft;       // the size the frame kernel transforms
  HIPCHK(hipMalloc(&p->d_tw, size_t(tw_n) * sizeof(float2)));
  HIPCHK(hipMalloc(&p->d_hold_max, nb));
  HIPCHK(hipMalloc(&p->d_hold_min, nb));
  HIPCHK(hipMalloc(&p->d_avg, size_t(nfft) * sizeof(double)));
  HIPCHK(hipMalloc(&p->d_dc_state, sizeof(float2)));
  HIPCHK(hipMalloc(&p->d_sums, (size_t(max_frames) + 8) * sizeof(float2)));   // (+ 7: block sums of overlapping frames)
  HIPCHK(hipMalloc(&p->d_dc_sub, size_t(max_frames) * sizeof(float2)));
  HIPCHK(hipMalloc(&p->d_tare_base, nb));
  HIPCHK(hipMalloc(&p->d_tare_acc, nb));
  HIPCHK(hipMalloc(&p->d_trace_in, nb));
  HIPCHK(hipMalloc(&p->d_trace_live, nb));
  HIPCHK(hipMemsetAsync(p->d_dc_state, 0, sizeof(float2), p->stream));
  HIPCHK(hipMemsetAsync(p->d_avg, 0, size_t(nfft) * sizeof(double), p->stream));
  // twiddle table exp(-2 pi i m / N), evaluated in double, rounded once
  std::vector<float2> tw(tw_n);
  for (int m = 0; m < tw_n; ++m) {
    const double ang = -2.0 * M_PI * double(m) / double(tw_n);
    tw[m] = float2{float(std::cos(ang)), float(std::sin(ang))};
  }
  HIPCHK(hipMemcpy(p->d_tw, tw.data(), size_t(tw_n) * sizeof(float2), hipMemcpyHostToDevice));
  if (p->chirp) {
    // a[n] = exp(-i pi n^2 / N): the phase from n^2 mod 2N in integers, so that it is exact for every n;
    // B = FFT_M(b), b[n] = b[M - n] = conj(a[n]) for n < N, 0 elsewhere - in double (plain radix-2), rounded once
    const int M = p->m_fft;
    std::vector<double> ar(nfft), ai(nfft), br(M, 0.0), bi(M, 0.0);
    for (int n = 0; n < nfft; ++n) {
      const long long q = ((long long)n * n) % (2ll * nfft);
      const double ang = -M_PI * double(q) / double(nfft);
      ar[n] = std::cos(ang);
      ai[n] = std::sin(ang);
    }
    std::vector<double> twc(M / 2), tws(M / 2);      // exp(-2 pi i k / M), k < M / 2: one table for every stage
    for (int k = 0; k < M / 2; ++k) {
      const double ang = -2.0 * M_PI * double(k) / double(M);
      twc[k] = std::cos(ang);
      tws[k] = std::sin(ang);
    }
    // FFT_M, in double, of the filter segment  h[m mod M] = b[m + shift] = conj(a[|m + shift|])  for lo <= m <= hi, 0 elsewhere
    const auto filter_spectrum = [&](int shift, int lo, int hi) {
      std::fill(br.begin(), br.end(), 0.0);
      std::fill(bi.begin(), bi.end(), 0.0);
      for (int mm = lo; mm <= hi; ++mm) {
        const int idx = mm + shift < 0 ? -(mm + shift) : mm + shift;
        const int pos = mm < 0 ? M + mm : mm;
        br[pos] = ar[idx];
        bi[pos] = -ai[idx];
      }
      for (int i = 1, j = 0; i < M; ++i) {        // bit reversal
        int bit = M >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(br[i], br[j]); std::swap(bi[i], bi[j]); }
      }
      for (int len = 2; len <= M; len <<= 1) {
        const int step = M / len;
        for (int i = 0; i < M; i += len) {
          for (int k = 0; k < len / 2; ++k) {
            const double wr = twc[k * step], wi = tws[k * step];
            const double xr = br[i + k + len / 2] * wr - bi[i + k + len / 2] * wi;
            const double xi = br[i + k + len / 2] * wi + bi[i + k + len / 2] * wr;
            br[i + k + len / 2] = br[i + k] - xr;
            bi[i + k + len / 2] = bi[i + k] - xi;
            br[i + k] += xr;
            bi[i + k] += xi;
          }
        }
      }
    };
    // ... rounded once, in the order the first transform leaves its bins in ([k1][k2] on the long-frame kernels)
    const auto upload_spectrum = [&](float2** dst) -> int {
      std::vector<float2> b32(M);
      if (p->chirp_big) {
        const int n1 = M >> kMaxLog2N, n2 = 1 << kMaxLog2N;
        for (int k1 = 0; k1 < n1; ++k1)
          for (int k2 = 0; k2 < n2; ++k2) b32[size_t(k1) * n2 + k2] = float2{float(br[k1 + n1 * k2]), float(bi[k1 + n1 * k2])};
      } else {
        for (int k = 0; k < M; ++k) b32[k] = float2{float(br[k]), float(bi[k])};
      }
      HIPCHK(hipMalloc(dst, size_t(M) * sizeof(float2)));
      HIPCHK(hipMemcpy(*dst, b32.data(), size_t(M) * sizeof(float2), hipMemcpyHostToDevice));
      return TDSA_OK;
    };
    const int H = p->chirp_split;
    if (H == 0) {
      filter_spectrum(0, -(nfft - 1), nfft - 1);           // b[n] = b[M - n] = conj(a[n]) for n < N
      { const int rc = upload_spectrum(&p->d_chirp_b); if (rc != TDSA_OK) return rc; }
    } else {                                               // (tdsa_chirp.hip: the three segments of the split convolution)
      filter_spectrum(0, -(H - 1), H - 1);
      { const int rc = upload_spectrum(&p->d_chirp_b); if (rc != TDSA_OK) return rc; }
      filter_spectrum(-H, -(nfft - H - 1), H - 1);         // b[m - H], m = k - n' in (-(N - H), H)
      { const int rc = upload_spectrum(&p->d_chirp_bm); if (rc != TDSA_OK) return rc; }
      filter_spectrum(H, -(H - 1), nfft - H - 1);          // b[m + H], m = k' - n in (-H, N - H)
      { const int rc = upload_spectrum(&p->d_chirp_bp); if (rc != TDSA_OK) return rc; }
    }
    std::vector<float2> a32(nfft);
    for (int n = 0; n < nfft; ++n) a32[n] = float2{float(ar[n]), float(ai[n])};
    HIPCHK(hipMalloc(&p->d_chirp_a, size_t(nfft) * sizeof(float2)));
    HIPCHK(hipMalloc(&p->d_ones, size_t(M) * sizeof(float)));
    std::vector<float> ones(M, 1.0f);
    HIPCHK(hipMemcpy(p->d_chirp_a, a32.data(), size_t(nfft) * sizeof(float2), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p->d_ones, ones.data(), size_t(M) * sizeof(float), hipMemcpyHostToDevice));
    {   // 2^a 3^b 5^c up to 10 000 points: the stages of its mixed-radix transform and W_N^k
      int r = nfft;
      for (const int f : {2, 3, 5}) while (r % f == 0) r /= f;
      if (r == 1 && nfft >= 4) {
        smooth_plan(p);
        p->smooth = p->smooth_stages > 0;
      }
      if (p->smooth) {
        std::vector<float2> tw(nfft);
        for (int k = 0; k < nfft; ++k) {
          const double ang = -2.0 * M_PI * double(k) / double(nfft);
          tw[k] = float2{float(std::cos(ang)), float(std::sin(ang))};
        }
        HIPCHK(hipMalloc(&p->d_smooth_tw, size_t(nfft) * sizeof(float2)));
        HIPCHK(hipMemcpy(p->d_smooth_tw, tw.data(), size_t(nfft) * sizeof(float2), hipMemcpyHostToDevice));
      }
    }
    if (p->chirp_big) {   // the M-point transforms' column-pass seeds (as for a native long frame of M points)
      const int nrow = 1 << kMaxLog2N, n1 = M >> kMaxLog2N, na = n1 < 8 ? n1 : 8;
      const int rows = big_seed_rows(p->log2m);
      std::vector<float2> seed(size_t(rows > 0 ? rows : 1) * nrow);
      for (int r = 0; r < rows; ++r) {
        const long long mult = r < na - 1 ? (r + 1) : 8ll * (r - (na - 1) + 1);
        for (int c = 0; c < nrow; ++c) {
          const long long e = (mult * c) % M;
          const double ang = -2.0 * M_PI * double(e) / double(M);
          seed[size_t(r) * nrow + c] = float2{float(std::cos(ang)), float(std::sin(ang))};
        }
      }
      HIPCHK(hipMalloc(&p->d_tw_seed, seed.size() * sizeof(float2)));
      HIPCHK(hipMemcpy(p->d_tw_seed, seed.data(), seed.size() * sizeof(float2), hipMemcpyHostToDevice));
    }
  }
  if (big) {
    HIPCHK(hipMalloc(&p->d_sum, size_t(nfft) * sizeof(double)));
    HIPCHK(hipMalloc(&p->d_lin64, size_t(nfft) * sizeof(double)));
    {   // per-workgroup partial power sums of the row pass: [N1 * split][16384], split = workgroups per k1 row
      const int n1 = nfft >> 14;
      int split = p->num_cu / n1 > 1 ? p->num_cu / n1 : 1;
      const int gmax = p->max_frames < p->big_group ? p->max_frames : p->big_group;
      if (split > gmax) split = gmax;
      HIPCHK(hipMalloc(&p->d_acc, size_t(n1) * split * (size_t(1) << 14) * sizeof(float)));
    }
    const int nrow = 1 << kMaxLog2N;
    HIPCHK(hipMalloc(&p->d_tw_row, size_t(nrow) * sizeof(float2)));
    HIPCHK(hipMalloc(&p->d_ones, size_t(nrow) * sizeof(float)));
    std::vector<float2> t3(nrow);
    for (int m = 0; m < nrow; ++m) {
      const double a3 = -2.0 * M_PI * double(m) / double(nrow);
      t3[m] = float2{float(std::cos(a3)), float(std::sin(a3))};
    }
    {   // seeds of the column pass's twiddles W_N^(n2 k1), k1 = a + 8 b: per column n2 the factors W_N^(n2 a), a = 1 .. NA-1,
        // and W_N^(n2 8 b), b = 1 .. NB-1 - exponent reduced mod N in integers, angle and sin / cos in double, rounded once
      const int n1 = nfft >> kMaxLog2N, na = n1 < 8 ? n1 : 8, nb = n1 / na;
      const int rows = big_seed_rows(p->log2n);
      std::vector<float2> seed(size_t(rows > 0 ? rows : 1) * nrow);
      for (int r = 0; r < rows; ++r) {
        const long long mult = r < na - 1 ? (r + 1) : 8ll * (r - (na - 1) + 1);
        for (int n2 = 0; n2 < nrow; ++n2) {
          const long long e = (mult * n2) % nfft;
          const double ang = -2.0 * M_PI * double(e) / double(nfft);
          seed[size_t(r) * nrow + n2] = float2{float(std::cos(ang)), float(std::sin(ang))};
        }
      }
      (void)nb;
      HIPCHK(hipMalloc(&p->d_tw_seed, seed.size() * sizeof(float2)));
      HIPCHK(hipMemcpy(p->d_tw_seed, seed.data(), seed.size() * sizeof(float2), hipMemcpyHostToDevice));
    }
    std::vector<float> ones(nrow, 1.0f);
    HIPCHK(hipMemcpy(p->d_tw_row, t3.data(), size_t(nrow) * sizeof(float2), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p->d_ones, ones.data(), size_t(nrow) * sizeof(float), hipMemcpyHostToDevice));
  }
  int rc = reset_hold(p, true, true);
  if (rc != TDSA_OK) return rc;
  // defaults = HackRF plain branch (hackrf_samples.py:382-383)
  p->mode.db_mode = TDSA_DB_MAG;
  p->mode.power_scale = 1.0f;
  p->mode.log_floor = 1e-12f;
  p->mode.avg_mode = TDSA_AVG_OFF;
  p->mode.avg_n = 1;
  p->mode.dc_alpha = 1.0f;
  p->mode.cal_offset_db = 0.0f;
  p->mode.hold_flags = 0;
  HIPCHK(hipStreamSynchronize(p->stream));
  return TDSA_OK;
}

int tdsa_destroy(tdsa_plan p) {
  if (!p) return TDSA_OK;
  (void)hipSetDevice(p->device);
  for (hipStream_t a : p->aux)
    if (a) (void)hipStreamSynchronize(a);
  if (p->stream) (void)hipStreamSynchronize(p->stream);
  void* bufs[] = {p->d_window[0], p->d_window[1], p->d_window[2], p->d_window_perm[0], p->d_window_perm[1],
                  p->d_window_perm[2], p->d_tw, p->d_hold_max, p->d_hold_min,
                  p->d_avg, p->d_lin, p->d_carry, p->d_agg, p->d_agg_w, p->d_chunk_a, p->d_chunk_v, p->d_cplx, p->d_real, p->d_lin1, p->d_db1, p->d_dc_state, p->d_sums, p->d_dc_sub,
                  p->d_tare_base, p->d_tare_acc, p->d_in_stage, p->d_out_stage, p->d_trace_in,
                  p->d_trace_live, p->d_scratch, p->d_z, p->d_welch, p->d_clock, p->d_smooth_tw, p->d_smooth_z, p->d_chirp_aw[0], p->d_chirp_aw[1], p->d_chirp_aw[2], p->d_chirp_bm, p->d_chirp_bp, p->d_chirp_a, p->d_chirp_b, p->d_u0, p->d_u1, p->d_acc, p->d_sum, p->d_lin64, p->d_sums64, p->d_tw_seed, p->d_tw_row, p->d_ones,
                  p->d_dbg, p->d_bigq};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (p->h_in_pin) (void)hipHostFree(p->h_in_pin);
  if (p->h_out_pin) (void)hipHostFree(p->h_out_pin);
  for (auto& per_stream : p->fs)
    for (auto& sl : per_stream) {
      void* fb[] = {sl.d_part, sl.d_peak, sl.d_bin, sl.d_band};
      for (void* b : fb)
        if (b) (void)hipFree(b);
    }
  if (p->fs_stream) { (void)hipStreamSynchronize(p->fs_stream); (void)hipStreamDestroy(p->fs_stream); }
  for (hipEvent_t e : p->prof_events) (void)hipEventDestroy(e);
  if (p->ev0) (void)hipEventDestroy(p->ev0);
  if (p->ev1) (void)hipEventDestroy(p->ev1);
  if (p->ev_state) (void)hipEventDestroy(p->ev_state);
  for (hipEvent_t e : p->ev_aux)
    if (e) (void)hipEventDestroy(e);
  for (hipStream_t a : p->aux)
    if (a) (void)hipStreamDestroy(a);
  if (p->stream) (void)hipStreamDestroy(p->stream);
  delete p;
  return TDSA_OK;
}

int tdsa_get_info(tdsa_plan p, tdsa_info* out) {
  if (!p || !out) return fail(TDSA_ERR_ARG, "null argument");
  LaunchGeom g = p->big ? spectrum_geometry(kMaxLog2N, (p->nfft >> kMaxLog2N) * (p->max_frames < p->big_group ? p->max_frames : p->big_group), p->num_cu)
                        : spectrum_geometry(p->log2n, p->max_frames, p->num_cu);
  out->nfft = p->nfft;
  out->max_frames = p->max_frames;
  out->device_id = p->device;
  out->grid = g.grid;
  out->block = g.block;
  out->frames_per_block = g.fpw;
  out->lds_bytes = int(g.lds_bytes);
  out->num_cu = p->num_cu;
  out->frames_held_max = p->held_max;
  out->frames_held_min = p->held_min;
  out->avg_count = p->avg_count;
  out->version = TDSA_VERSION;
  return TDSA_OK;
}

// Long frames: the column pass fetches the window sample by sample (one 4-byte load each).  A window that is ONE value
// throughout - np.ones, rtl_samples.py:203-204 - travels as that value instead (BigWindow::flat); every other table
// is read.  (The cosine-sum windows the reference builds were also evaluated in the kernel in round 5 - two FMAs per
// sample from three scalar row constants: slower than the loads, see tdsa_big.hip; not kept.)
static int big_window_model(tdsa_plan p, const float* w, const float scale[3]) {
  bool flat = true;
  for (int i = 1; i < p->nfft && flat; ++i) flat = w[i] == w[0];
  for (int f = 0; f < 3; ++f) {
    p->big_win[f] = BigWindow{flat ? 2 : 0, p->d_window[f], w[0] * scale[f]};
  }
  return TDSA_OK;
}

int tdsa_set_window(tdsa_plan p, const float* w_host, int n) {
  if (!p || !w_host) return fail(TDSA_ERR_ARG, "null argument");
  if (n != p->nfft) return fail(TDSA_ERR_ARG, "window length %d != nfft %d", n, p->nfft);
  HIPCHK(hipSetDevice(p->device));
  JOIN(p);
  const float scale[3] = {1.0f / 128.0f, 1.0f / 127.5f, 1.0f};
  std::vector<float> tmp(n);
  HIPCHK(hipStreamSynchronize(p->stream));
  for (int f = 0; f < 3; ++f) {
    for (int i = 0; i < n; ++i) tmp[i] = w_host[i] * scale[f];
    HIPCHK(hipMemcpy(p->d_window[f], tmp.data(), size_t(n) * sizeof(float), hipMemcpyHostToDevice));
    if (p->d_window_perm[f]) HIPCHK(launch_window_perm(p->log2n, p->d_window[f], p->d_window_perm[f], p->stream));
  }
  if (p->big) {
    const int rc_w = big_window_model(p, w_host, scale);
    if (rc_w != TDSA_OK) return rc_w;
  }
  if (p->chirp && p->log2m >= 10) {
    // window x input scale x chirp a[n] = exp(-i pi n^2 / N) (phase from n^2 mod 2N in integers), product in double,
    // rounded once: the first transform multiplies the unpacked samples by it on load
    std::vector<float2> aw(n);
    std::vector<double> ca(n), sa(n);
    for (int i = 0; i < n; ++i) {
      const long long q = ((long long)i * i) % (2ll * n);
      const double ang = -M_PI * double(q) / double(n);
      ca[i] = std::cos(ang);
      sa[i] = std::sin(ang);
    }
    for (int f = 0; f < 3; ++f) {
      for (int i = 0; i < n; ++i) {
        const double wv = double(w_host[i]) * double(scale[f]);
        aw[i] = float2{float(wv * ca[i]), float(wv * sa[i])};
      }
      if (!p->d_chirp_aw[f]) HIPCHK(hipMalloc(&p->d_chirp_aw[f], aw.size() * sizeof(float2)));
      HIPCHK(hipMemcpy(p->d_chirp_aw[f], aw.data(), aw.size() * sizeof(float2), hipMemcpyHostToDevice));
    }
  }
  p->window_set = true;
  return TDSA_OK;
}

int tdsa_set_mode(tdsa_plan p, const tdsa_mode* m) {
  if (!p || !m) return fail(TDSA_ERR_ARG, "null argument");
  if (m->db_mode != TDSA_DB_MAG && m->db_mode != TDSA_DB_POW) return fail(TDSA_ERR_ARG, "db_mode %d", m->db_mode);
  if (m->avg_mode < TDSA_AVG_OFF || m->avg_mode > TDSA_AVG_LIN) return fail(TDSA_ERR_ARG, "avg_mode %d", m->avg_mode);
  if (m->dc_alpha > 1.0f) return fail(TDSA_ERR_ARG, "dc_alpha %g > 1", double(m->dc_alpha));
  if (!(m->log_floor >= 0.0f)) return fail(TDSA_ERR_ARG, "log_floor must be >= 0");
  tdsa_mode nm = *m;
  if (nm.avg_n < 1) nm.avg_n = 1;   // TraceAverager.set_mode: n = max(1, n)
  const bool avg_changed = nm.avg_mode != p->mode.avg_mode || nm.avg_n != p->mode.avg_n;
  p->mode = nm;
  if (avg_changed) p->avg_count = 0;   // set_mode() resets the buffer (signal_processing.py:26-28)
  return TDSA_OK;
}

int tdsa_set_overlap(tdsa_plan p, int n_streams) {
  if (!p) return fail(TDSA_ERR_ARG, "null plan");
  if (n_streams < 1 || n_streams > tdsa_plan_s::kMaxOverlap)
    return fail(TDSA_ERR_ARG, "n_streams=%d outside [1, %d]", n_streams, tdsa_plan_s::kMaxOverlap);
  HIPCHK(hipSetDevice(p->device));
  JOIN(p);
  for (int i = 0; i < n_streams - 1; ++i) {
    if (!p->aux[i]) {
      HIPCHK(hipStreamCreateWithFlags(&p->aux[i], hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&p->ev_aux[i], hipEventDisableTiming));
    }
  }
  p->n_overlap = n_streams;
  p->rr = 0;
  return TDSA_OK;
}

int tdsa_reset_state(tdsa_plan p, uint32_t what) {
  if (!p) return fail(TDSA_ERR_ARG, "null plan");
  HIPCHK(hipSetDevice(p->device));
  JOIN(p);
  if (what & TDSA_RESET_AVG) p->avg_count = 0;
  int rc = reset_hold(p, (what & TDSA_RESET_HOLD_MAX) != 0, (what & TDSA_RESET_HOLD_MIN) != 0);
  if (rc != TDSA_OK) return rc;
  if ((what & TDSA_RESET_HOLD_MAX) && (what & TDSA_RESET_HOLD_MIN)) p->frames_seen = 0;
  if (what & TDSA_RESET_DC) HIPCHK(hipMemsetAsync(p->d_dc_state, 0, sizeof(float2), p->stream));
  if (what & TDSA_RESET_TARE) {
    p->tare_active = false;
    p->tare_count = 0;
  }
  return TDSA_OK;
}

int tdsa_set_tare_baseline(tdsa_plan p, const float* baseline_db_host, int n) {
  if (!p) return fail(TDSA_ERR_ARG, "null plan");
  if (!baseline_db_host) {
    p->tare_active = false;
    return TDSA_OK;
  }
  if (n != p->nfft) return fail(TDSA_ERR_ARG, "baseline length %d != nfft %d", n, p->nfft);
  HIPCHK(hipSetDevice(p->device));
  JOIN(p);
  HIPCHK(hipStreamSynchronize(p->stream));
  HIPCHK(hipMemcpy(p->d_tare_base, baseline_db_host, size_t(n) * sizeof(float), hipMemcpyHostToDevice));
  p->tare_active = true;
  return TDSA_OK;
}

// ---- per-frame scalars of a call (tdsa_set_frame_stats) -------------------------------------------------------------
// The frame kernel's STATS instantiations exist for frames of whole waves (N >= 1024), dB rows without tare, hold none / max.
static bool frame_stats_fusable(tdsa_plan p, bool averaging) {
  return p->fs_on && !averaging && !p->chirp && !p->big && p->log2n >= 10 && !p->tare_active &&
         (p->mode.hold_flags & TDSA_HOLD_MIN) == 0;
}
// the slot the call at hand writes (the next one of its stream's), grown to n_frames
static int frame_stats_begin(tdsa_plan p, int n_frames, hipStream_t s, tdsa_plan_s::FsSlot** out) {
  int si = 0;
  for (int i = 0; i < tdsa_plan_s::kMaxOverlap - 1; ++i)
    if (s == p->aux[i] && s != p->stream) si = i + 1;
  tdsa_plan_s::FsSlot& sl = p->fs[si][p->fs_count[si]++ % tdsa_plan_s::kFsKeep];
  if (size_t(n_frames) > sl.cap) {
    HIPCHK(hipStreamSynchronize(s));                         // whatever last wrote the slot ran on this stream
    if (p->fs_stream) HIPCHK(hipStreamSynchronize(p->fs_stream));
    void* fb[] = {sl.d_part, sl.d_peak, sl.d_bin, sl.d_band};
    for (void* b : fb)
      if (b) HIPCHK(hipFree(b));
    sl.d_part = nullptr; sl.d_peak = nullptr; sl.d_bin = nullptr; sl.d_band = nullptr;
    sl.cap = 0;
    const size_t cap = size_t(n_frames) > size_t(p->max_frames) ? size_t(n_frames) : size_t(p->max_frames);
    HIPCHK(hipMalloc(&sl.d_part, cap * 16 * kStatsRecBytes));   // up to 16 waves per frame
    HIPCHK(hipMalloc(&sl.d_peak, cap * sizeof(float)));
    HIPCHK(hipMalloc(&sl.d_bin, cap * sizeof(int)));
    HIPCHK(hipMalloc(&sl.d_band, cap * sizeof(double)));
    sl.cap = cap;
  }
  sl.n_frames = n_frames;
  sl.state = 0;
  sl.pending = false;
  sl.stream = s;
  *out = &sl;
  return TDSA_OK;
}
static int frame_stats_end(tdsa_plan p, tdsa_plan_s::FsSlot* sl, int state) {
  sl->state = state;
  p->fs_hist[p->fs_seq++ % tdsa_plan_s::kFsKeep] = sl;
  return TDSA_OK;
}
// plans / modes without the fused epilogue: the same scalars from the rows the call wrote (rows_stats_kernel)
static int frame_stats_from_rows(tdsa_plan p, const float* rows, int n_frames, int n_seg, int frames_per_seg,
                                 long long seg_stride_elems, hipStream_t s) {
  if (!p->fs_on) return TDSA_OK;
  tdsa_plan_s::FsSlot* sl = nullptr;
  int rc = frame_stats_begin(p, n_frames, s, &sl);
  if (rc != TDSA_OK) return rc;
  if (rows == nullptr) return frame_stats_end(p, sl, 2);
  for (int g = 0; g < n_seg; ++g)
    HIPCHK(launch_rows_stats(rows + (long long)g * seg_stride_elems, frames_per_seg, p->nfft, p->fs_lo, p->fs_hi, -1.0,
                             sl->d_peak + (size_t)g * frames_per_seg, sl->d_bin + (size_t)g * frames_per_seg,
                             sl->d_band + (size_t)g * frames_per_seg, s));
  return frame_stats_end(p, sl, 1);
}

// several captures in one launch (tdsa_process_dev_batch): n_frames = n_seg * frames_per_seg frames in all
struct SegInfo {
  int n_seg = 1;
  int frames_per_seg = 0;
  long long in_stride_bytes = 0;
  long long out_stride_elems = 0;
  bool single_frames = false;   // one frame per capture: launched as ONE capture whose frames are in_stride_bytes apart
};

// before / after: optional events the device work waits for / signals (tdsa_pipe: H2D and D2H legs)
static int process_dev_impl(tdsa_plan p, int in_format, const void* iq_dev, size_t n_samples, int hop, int n_frames,
                            float* out_db_dev, hipEvent_t before, hipEvent_t after, const SegInfo* seg = nullptr) {
  if (!p) return fail(TDSA_ERR_ARG, "null plan");
  if (in_format < TDSA_IN_I8 || in_format > TDSA_IN_C64) return fail(TDSA_ERR_ARG, "in_format %d", in_format);
  if (n_frames == 0) return TDSA_OK;
  if (!iq_dev) return fail(TDSA_ERR_ARG, "iq pointer is null");
  const int frames_each = seg ? seg->frames_per_seg : n_frames;     // frames of ONE capture (n_samples is per capture)
  if (n_frames < 0 || frames_each > p->max_frames)
    return fail(TDSA_ERR_ARG, "n_frames=%d outside [0, max_frames=%d]", frames_each, p->max_frames);
  if (hop < 1) return fail(TDSA_ERR_ARG, "hop=%d must be >= 1", hop);
  if (n_samples < size_t(frames_each - 1) * size_t(hop) + size_t(p->nfft))
    return fail(TDSA_ERR_ARG, "n_samples=%zu too small for %d frames of %d at hop %d", n_samples, frames_each,
                p->nfft, hop);
  if (!p->window_set) return fail(TDSA_ERR_STATE, "tdsa_set_window has not been called");
  const int bps = bytes_per_sample(in_format);
  if ((reinterpret_cast<uintptr_t>(iq_dev) % (bps == 8 ? 8 : 2)) != 0)
    return fail(TDSA_ERR_ARG, "iq pointer must be aligned to one sample (%d bytes)", bps);
  HIPCHK(hipSetDevice(p->device));
  if (p->chirp) {
    JOIN(p);
    if (before) HIPCHK(hipStreamWaitEvent(p->stream, before, 0));
    int rc_chirp = process_chirp(p, in_format, iq_dev, hop, n_frames, out_db_dev);
    if (rc_chirp == TDSA_OK) rc_chirp = frame_stats_from_rows(p, out_db_dev, n_frames, 1, n_frames, 0, p->stream);
    if (rc_chirp == TDSA_OK && after) HIPCHK(hipEventRecord(after, p->stream));
    return rc_chirp;
  }
  if (p->big) {
    JOIN(p);
    if (before) HIPCHK(hipStreamWaitEvent(p->stream, before, 0));
    const int rc_big = process_big(p, in_format, iq_dev, hop, n_frames, out_db_dev);
    if (rc_big == TDSA_OK && after) HIPCHK(hipEventRecord(after, p->stream));
    return rc_big;
  }

  const tdsa_mode& m = p->mode;
  const bool averaging = avg_active(m);
  // calls whose result does not depend on the order they execute in may overlap (tdsa_set_overlap)
  const bool order_free = !averaging && (m.dc_alpha < 0.0f || m.dc_alpha >= 1.0f);
  if (seg && !order_free) return fail(TDSA_ERR_STATE, "internal: a batched launch needs an order-free mode");
  hipStream_t s = p->stream;
  // synchronous host entry points (process_host) wait for this very launch: nothing runs beside it, so it keeps
  // the main stream and the whole chip (a half-chip launch there only lost throughput)
  const bool overlap = order_free && !p->sync_call && !p->profiling;   // (profiled launches: main stream, events around)
  if (overlap) {
    int rc_s = pick_stream(p, &s);
    if (rc_s != TDSA_OK) return rc_s;
  } else {
    JOIN(p);
  }
  if (before) HIPCHK(hipStreamWaitEvent(s, before, 0));
  const bool hold = (m.hold_flags & 3u) != 0;
  const int in_c64 = in_format == TDSA_IN_C64;

  SpecParams sp{};
  sp.in = iq_dev;
  sp.frame_stride = (long long)hop * bps;
  sp.n_frames = n_frames;
  sp.first_frame_index = p->frames_seen > 0 ? 1 : 0;
  sp.window = p->d_window[in_format];
  sp.window_perm = p->d_window_perm[in_format];
  sp.tw = p->d_tw;
  sp.xor_mask = in_format == TDSA_IN_I8 ? 0x80808080u : 0u;
  sp.in_off = in_format == TDSA_IN_I8 ? 128.0f : (in_format == TDSA_IN_U8 ? 127.5f : 0.0f);
  sp.in_scale = in_format == TDSA_IN_I8 ? 1.0f / 128.0f : (in_format == TDSA_IN_U8 ? 1.0f / 127.5f : 1.0f);
  sp.db_mode = m.db_mode;
  sp.pscale = m.db_mode == TDSA_DB_POW ? m.power_scale : 1.0f;
  sp.log_floor = m.log_floor;
  sp.cal_db = m.cal_offset_db;
  sp.tare = p->tare_active ? p->d_tare_base : nullptr;
  sp.dbg = p->d_dbg;
  if (seg && seg->single_frames) {
    sp.frame_stride = seg->in_stride_bytes;      // seg_magic stays 0: frame f reads in + f * stride, row f follows row f - 1
  } else if (seg) {
    sp.seg_frames = unsigned(seg->frames_per_seg);
    sp.seg_magic = unsigned((0x100000000ull + sp.seg_frames - 1) / sp.seg_frames);   // ceil(2^32 / d)
    sp.seg_in_stride = seg->in_stride_bytes;
    sp.seg_out_stride = seg->out_stride_elems;
  }

  if (m.dc_alpha < 0.0f) {
    sp.dc_mode = DC_NONE;
  } else if (m.dc_alpha >= 1.0f) {
    sp.dc_mode = DC_FRAME_MEAN;
    sp.dc_state = p->d_dc_state;
  } else {
    sp.dc_mode = DC_TRACKED;
    // overlapping byte frames (the hop divides the frame): the sums of the hop-long blocks are formed once and a frame's
    // sum is `parts` of them - integer-valued floats, exact in any order - instead of every sample being summed parts times
    const int parts = (!in_c64 && hop >= 64 && hop % 4 == 0 && p->nfft % hop == 0 && p->nfft / hop >= 2 && p->nfft / hop <= 8)
                          ? p->nfft / hop : 1;
    HIPCHK(launch_frame_sums(iq_dev, in_c64, sp.xor_mask, sp.frame_stride, parts > 1 ? hop : p->nfft,
                             parts > 1 ? n_frames + parts - 1 : n_frames, p->d_sums, p->stream));
    HIPCHK(launch_dc_track(p->d_sums, p->nfft, n_frames, m.dc_alpha, sp.in_off, sp.in_scale, p->d_dc_state,
                           p->d_dc_sub, p->stream, parts));
    sp.dc_sub = p->d_dc_sub;
  }

  // With three or more streams the overlapped launches are sized for HALF the CUs: two of them run side by side, the
  // next one queues behind whichever ends first.  Each workgroup then carries twice the frames - its fixed costs (cold
  // first fetch, hold merge: ~5 us) count half - and the ragged end of one launch is filled by the next:
  // C3 76.4 -> 74.3 us per step, C2 26.7 -> 26.0 (profiles/r02_c3_experiments.txt).  More than three streams in flight
  // measured worse (4: 91 us), and strictly serial launches keep the whole chip.
  const LaunchGeom g = spectrum_geometry(p->log2n, n_frames,
                                         (overlap && p->n_overlap >= 3) ? (p->num_cu * p->overlap_share + 99) / 100 : p->num_cu);
  if (averaging) {
    if (!p->d_lin) HIPCHK(hipMalloc(&p->d_lin, size_t(p->max_frames) * p->nfft * sizeof(float)));
    // Long batches run as a chained scan over chunks of frames (tdsa_trace.hip).  The chunks ARE the frame ranges of the
    // frame kernel's workgroups, which form their chunk's aggregate themselves while the rows pass through their
    // registers - the scan's first pass over the rows disappears (below 4096 points a workgroup's slots take consecutive
    // runs of its range and add their partial sums through LDS).
    // Up to 256 chunks: grids of more workgroups (N = 8192: 512, N <= 4096: 1024) fold 2 / 4 consecutive ranges into one.
    const int wg_fold = (g.grid + 255) / 256;
    const int units = (n_frames + g.fpw - 1) / g.fpw;
    const bool wg_chunks = n_frames > p->avg_wg_min && g.grid <= kAvgMaxWgChunks &&
                           ((units + g.grid - 1) / g.grid) * g.fpw * wg_fold <= 160 && !p->avg_f64_chunks;
    const size_t agg_rows = size_t(spectrum_geometry(p->log2n, p->max_frames, p->num_cu).grid);
    const size_t need_chunks = wg_chunks ? (agg_rows < 256 ? agg_rows : size_t(256))
                                         : (p->max_frames > 128 ? size_t(avg_scan_chunks(p->max_frames)) : 0);
    if (need_chunks > p->carry_chunks) {
      if (p->d_carry) { HIPCHK(hipStreamSynchronize(p->stream)); HIPCHK(hipFree(p->d_carry)); p->d_carry = nullptr; p->carry_chunks = 0; }
      HIPCHK(hipMalloc(&p->d_carry, need_chunks * p->nfft * sizeof(double)));
      p->carry_chunks = need_chunks;
    }
    AvgParams ap{};
    ap.lin = p->d_lin;
    ap.n_frames = n_frames;
    ap.n = p->nfft;
    ap.state = p->d_avg;
    ap.count_in = p->avg_count;
    ap.mode = m.avg_mode;
    ap.avg_n = m.avg_n;
    ap.log_floor = m.log_floor;
    ap.cal_db = m.cal_offset_db;
    ap.tare = sp.tare;
    ap.out_db = out_db_dev;
    ap.state_max = (m.hold_flags & TDSA_HOLD_MAX) ? p->d_hold_max : nullptr;
    ap.state_min = (m.hold_flags & TDSA_HOLD_MIN) ? p->d_hold_min : nullptr;
    if (wg_chunks) {
      if (!p->d_agg) HIPCHK(hipMalloc(&p->d_agg, (agg_rows > 256 ? agg_rows : size_t(256)) * p->nfft * sizeof(float)));
      if (!p->d_agg_w) HIPCHK(hipMalloc(&p->d_agg_w, size_t(p->max_frames) * sizeof(float)));
      if (!p->d_chunk_a) HIPCHK(hipMalloc(&p->d_chunk_a, size_t(kAvgMaxWgChunks + 64) * sizeof(double)));
      if (!p->d_chunk_v) HIPCHK(hipMalloc(&p->d_chunk_v, size_t(kAvgMaxWgChunks + 64) * sizeof(float)));
      ap.chunk_a = p->d_chunk_a;
      ap.chunk_v = p->d_chunk_v;
      ap.wg_chunks = g.grid;
      ap.wg_fold = wg_fold;
      ap.wg_fpw = g.fpw;
      ap.agg = p->d_agg;
      // the weights depend on where the averager stands and on the chunking only: in steady state (exp mode, or lin
      // with its count at the cap) consecutive calls of one shape re-use them
      const long long key[7] = {p->avg_count, m.avg_mode, m.avg_n, n_frames, g.grid, g.fpw, 0};    // path 0: the frame kernel's workgroup ranges
      if (std::memcmp(key, p->agg_w_key, sizeof(key)) != 0) {
        HIPCHK(launch_avg_weights(ap, p->d_agg_w, p->d_chunk_a, p->d_chunk_v, p->stream));
        std::memcpy(p->agg_w_key, key, sizeof(key));
      }
      sp.agg_w = p->d_agg_w;
      sp.agg_out = p->d_agg;
      // only the averager's state is wanted (no dB rows, no hold: the averaged spectrum of a capture - Welch at a native
      // size): the linear rows are neither written nor re-scanned
      if (out_db_dev == nullptr && ap.state_max == nullptr && ap.state_min == nullptr) { sp.agg_only = 1; ap.state_only = 1; }
    } else {
      const int rc = avg_use_ranges(p, ap, n_frames, p->stream);
      if (rc != TDSA_OK) return rc;
    }
    sp.out_lin = p->d_lin;
    sp.hold_flags = 0;
    int rc_p = launch_spectrum_profiled(p, in_c64, sp, g);
    if (rc_p != TDSA_OK) return rc_p;
    HIPCHK(launch_avg_scan(ap, p->stream, p->d_carry));
    {
      const int rc_fs = frame_stats_from_rows(p, out_db_dev, n_frames, 1, n_frames, 0, p->stream);
      if (rc_fs != TDSA_OK) return rc_fs;
    }
    if (m.avg_mode == TDSA_AVG_LIN) {
      long long c = (long long)p->avg_count + n_frames;
      p->avg_count = int(c < m.avg_n ? c : m.avg_n);
    } else {
      p->avg_count = 1;
    }
  } else {
    sp.out_db = out_db_dev;
    sp.hold_flags = int(m.hold_flags & 3u);
    if (hold) {   // workgroups merge their register-resident traces into the plan's traces with atomics
      sp.part_max = p->d_hold_max;
      sp.part_min = p->d_hold_min;
    }
    tdsa_plan_s::FsSlot* fsl = nullptr;
    const bool fs_fused = frame_stats_fusable(p, false);
    if (fs_fused) {
      const int rc_b = frame_stats_begin(p, n_frames, s, &fsl);
      if (rc_b != TDSA_OK) return rc_b;
      sp.stats_part = fsl->d_part;
      sp.band_lohi = p->fs_lo <= p->fs_hi ? (unsigned(p->fs_lo) | (unsigned(p->fs_hi) << 16)) : 1u;
    }
    int rc_p = launch_spectrum_profiled(p, in_c64, sp, g, s);
    if (rc_p != TDSA_OK) return rc_p;
    if (fs_fused) {
      // the records hold the power before the calibration offset: 10^(dB / 10) = power x 10^(cal / 10).  They are folded
      // when someone asks (tdsa_get_frame_stats): nothing runs behind the frame kernel on its stream
      fsl->pending = true;
      fsl->wpf = spectrum_waves_per_frame(p->log2n);
      fsl->cal_lin = std::pow(10.0, double(m.cal_offset_db) / 10.0);
      const int rc_e = frame_stats_end(p, fsl, 1);
      if (rc_e != TDSA_OK) return rc_e;
    } else if (p->fs_on) {
      const bool one = !seg || seg->single_frames;
      const int rc_r = frame_stats_from_rows(p, out_db_dev, n_frames, one ? 1 : seg->n_seg, one ? n_frames : seg->frames_per_seg,
                                             one ? 0 : seg->out_stride_elems, s);
      if (rc_r != TDSA_OK) return rc_r;
    }
  }
  if (after) HIPCHK(hipEventRecord(after, s));
  if (m.hold_flags & TDSA_HOLD_MAX) p->held_max += n_frames;
  if (m.hold_flags & TDSA_HOLD_MIN) p->held_min += n_frames;
  p->frames_seen += n_frames;
  return TDSA_OK;
}

int tdsa_process_dev(tdsa_plan p, int in_format, const void* iq_dev, size_t n_samples, int hop, int n_frames,
                     float* out_db_dev) {
  return process_dev_impl(p, in_format, iq_dev, n_samples, hop, n_frames, out_db_dev, nullptr, nullptr);
}

// n_segments captures of one shape.  Where the plan's mode makes the captures independent of the order they are
// processed in (no averaging, no tracked DC remover) and the frames are LDS resident, ALL of them go out as ONE
// persistent launch: the per-launch costs (cold first fetch of window / twiddles, hold merge, the ragged last round
// of frames over the CUs) are paid once per call instead of once per capture.  Every other mode runs the captures
// one after the other through the same code as tdsa_process_dev - the results are the same either way.
int tdsa_process_dev_batch(tdsa_plan p, int in_format, const void* iq_dev, size_t seg_stride_bytes, int n_segments,
                           size_t n_samples_per_seg, int hop, int frames_per_seg, float* out_db_dev,
                           size_t out_seg_stride_floats) {
  if (!p) return fail(TDSA_ERR_ARG, "null plan");
  if (n_segments < 0) return fail(TDSA_ERR_ARG, "n_segments=%d", n_segments);
  if (n_segments == 0 || frames_per_seg == 0) return TDSA_OK;
  if (in_format < TDSA_IN_I8 || in_format > TDSA_IN_C64) return fail(TDSA_ERR_ARG, "in_format %d", in_format);
  const size_t bps = size_t(bytes_per_sample(in_format));
  if (n_segments > 1 && seg_stride_bytes % (bps == 8 ? 8 : 2) != 0)
    return fail(TDSA_ERR_ARG, "seg_stride_bytes=%zu must be a multiple of one sample", seg_stride_bytes);
  if (frames_per_seg < 0) return fail(TDSA_ERR_ARG, "frames_per_seg=%d", frames_per_seg);
  if (out_seg_stride_floats == 0) out_seg_stride_floats = size_t(frames_per_seg) * size_t(p->nfft);   // captures back to back
  // the captures' rows must not overlap: with a stride below one capture's rows the one-launch path would have
  // several workgroups write the same rows concurrently (which capture survives would not be deterministic)
  if (n_segments > 1 && out_db_dev != nullptr && !p->big &&
      out_seg_stride_floats < size_t(frames_per_seg) * size_t(p->nfft))
    return fail(TDSA_ERR_ARG, "out_seg_stride_floats=%zu is smaller than one capture's rows (%d x %d)",
                out_seg_stride_floats, frames_per_seg, p->nfft);
  const tdsa_mode& m = p->mode;
  const bool order_free = !avg_active(m) && (m.dc_alpha < 0.0f || m.dc_alpha >= 1.0f);
  const long long total = (long long)n_segments * frames_per_seg;
  // frames_per_seg == 1: ceil(2^32 / 1) does not fit SpecParams::seg_magic (it would read as 0 = "one capture").
  // One frame per capture IS one capture whose frames sit seg_stride apart, as long as its rows are contiguous;
  // with gapped rows the captures go out one by one.
  const bool single_frames = frames_per_seg == 1;
  const bool rows_contiguous = out_db_dev == nullptr || out_seg_stride_floats == size_t(p->nfft);
  const bool one_launch = n_segments > 1 && order_free && !p->chirp && !p->big && frames_per_seg > 0 &&
                          total * frames_per_seg < 0x100000000ll && total < 0x7fffffffll &&
                          (!single_frames || rows_contiguous);
  if (!one_launch) {
    for (int sg = 0; sg < n_segments; ++sg) {
      const int rc = process_dev_impl(p, in_format, static_cast<const unsigned char*>(iq_dev) + size_t(sg) * seg_stride_bytes,
                                      n_samples_per_seg, hop, frames_per_seg,
                                      out_db_dev ? out_db_dev + size_t(sg) * out_seg_stride_floats : nullptr, nullptr, nullptr);
      if (rc != TDSA_OK) return rc;
    }
    return TDSA_OK;
  }
  SegInfo seg;
  seg.n_seg = n_segments;
  seg.frames_per_seg = frames_per_seg;
  seg.in_stride_bytes = (long long)seg_stride_bytes;
  seg.out_stride_elems = (long long)out_seg_stride_floats;
  seg.single_frames = single_frames;
  return process_dev_impl(p, in_format, iq_dev, n_samples_per_seg, hop, int(total), out_db_dev, nullptr, nullptr, &seg);
}

// pinned, device-visible bounce buffers of the host entry points (grown on demand; every host call ends with a
// synchronize, so they are free when the next one starts)
static int ensure_pins(tdsa_plan p, size_t in_bytes, size_t out_bytes) {
  if (in_bytes > p->in_pin_bytes) {
    if (p->h_in_pin) { HIPCHK(hipStreamSynchronize(p->stream)); HIPCHK(hipHostFree(p->h_in_pin)); }
    p->h_in_pin = nullptr;
    p->in_pin_bytes = 0;
    HIPCHK(hipHostMalloc(&p->h_in_pin, in_bytes, hipHostMallocPortable | hipHostMallocMapped));
    p->in_pin_bytes = in_bytes;
  }
  if (out_bytes > p->out_pin_bytes) {
    if (p->h_out_pin) { HIPCHK(hipStreamSynchronize(p->stream)); HIPCHK(hipHostFree(p->h_out_pin)); }
    p->h_out_pin = nullptr;
    p->out_pin_bytes = 0;
    HIPCHK(hipHostMalloc(&p->h_out_pin, out_bytes, hipHostMallocPortable | hipHostMallocMapped));
    p->out_pin_bytes = out_bytes;
  }
  return TDSA_OK;
}

static int process_host(tdsa_plan p, int fmt, const void* iq_host, size_t n_samples, int hop, int n_frames,
                        float* out_db_host) {
  if (!p) return fail(TDSA_ERR_ARG, "null plan");
  if (n_frames == 0) return TDSA_OK;
  if (!iq_host) return fail(TDSA_ERR_ARG, "iq pointer is null");
  if (n_frames < 0 || n_frames > p->max_frames)
    return fail(TDSA_ERR_ARG, "n_frames=%d outside [0, max_frames=%d]", n_frames, p->max_frames);
  if (hop < 1) return fail(TDSA_ERR_ARG, "hop=%d must be >= 1", hop);
  const size_t need = size_t(n_frames - 1) * size_t(hop) + size_t(p->nfft);
  if (n_samples < need)
    return fail(TDSA_ERR_ARG, "n_samples=%zu too small for %d frames of %d at hop %d", n_samples, n_frames,
                p->nfft, hop);
  HIPCHK(hipSetDevice(p->device));
  JOIN(p);
  const size_t in_bytes = need * bytes_per_sample(fmt);
  if (in_bytes > p->in_stage_bytes) {
    HIPCHK(hipStreamSynchronize(p->stream));
    if (p->d_in_stage) HIPCHK(hipFree(p->d_in_stage));
    p->d_in_stage = nullptr;
    p->in_stage_bytes = 0;
    HIPCHK(hipMalloc(&p->d_in_stage, in_bytes));
    p->in_stage_bytes = in_bytes;
  }
  if (out_db_host && !p->d_out_stage)
    HIPCHK(hipMalloc(&p->d_out_stage, size_t(p->big ? 1 : p->max_frames) * p->nfft * sizeof(float)));
  const size_t out_bytes = out_db_host ? size_t(p->big ? 1 : n_frames) * p->nfft * sizeof(float) : 0;
  const bool bounce = in_bytes <= kPinnedBounceMax && out_bytes <= kPinnedBounceMax;
  if (bounce) {
    { const int rc_pin = ensure_pins(p, in_bytes, out_bytes); if (rc_pin != TDSA_OK) return rc_pin; }
    std::memcpy(p->h_in_pin, iq_host, in_bytes);      // the previous call ended with a synchronize: the buffer is free
  }
  // the smallest calls (one displayed frame) skip the two DMA operations as well: the kernels read the samples from
  // and write the row to the pinned buffers directly over the bus (hipHostMalloc memory is device-visible)
  const bool direct = bounce && in_bytes + out_bytes <= kZeroCopyMax && !p->big;
  if (direct) {
    p->sync_call = true;
    int rc_d = tdsa_process_dev(p, fmt, p->h_in_pin, need, hop, n_frames,
                                out_db_host ? static_cast<float*>(p->h_out_pin) : nullptr);
    p->sync_call = false;
    if (rc_d != TDSA_OK) return rc_d;
    JOIN(p);
    HIPCHK(hipStreamSynchronize(p->stream));
    if (out_db_host) std::memcpy(out_db_host, p->h_out_pin, out_bytes);
    return TDSA_OK;
  }
  HIPCHK(hipMemcpyAsync(p->d_in_stage, bounce ? p->h_in_pin : iq_host, in_bytes, hipMemcpyHostToDevice, p->stream));
  p->sync_call = true;
  int rc = tdsa_process_dev(p, fmt, p->d_in_stage, need, hop, n_frames, out_db_host ? p->d_out_stage : nullptr);
  p->sync_call = false;
  if (rc != TDSA_OK) return rc;
  // with tdsa_set_overlap(n > 1) the frame kernel may have gone to an auxiliary stream: order the main
  // stream (read-back + the synchronize below) behind it, so the host call stays sequentially consistent
  JOIN(p);
  if (out_db_host)
    HIPCHK(hipMemcpyAsync(bounce ? p->h_out_pin : static_cast<void*>(out_db_host), p->d_out_stage, out_bytes,
                          hipMemcpyDeviceToHost, p->stream));
  HIPCHK(hipStreamSynchronize(p->stream));
  if (bounce && out_db_host) std::memcpy(out_db_host, p->h_out_pin, out_bytes);
  return TDSA_OK;
}

int tdsa_process_i8(tdsa_plan p, const int8_t* iq_host, size_t n_samples, int hop, int n_frames,
                    float* out_db_host) {
  return process_host(p, TDSA_IN_I8, iq_host, n_samples, hop, n_frames, out_db_host);
}
int tdsa_process_u8(tdsa_plan p, const uint8_t* iq_host, size_t n_samples, int hop, int n_frames,
                    float* out_db_host) {
  return process_host(p, TDSA_IN_U8, iq_host, n_samples, hop, n_frames, out_db_host);
}
int tdsa_process_c64(tdsa_plan p, const float* iq_host, size_t n_samples, int hop, int n_frames,
                     float* out_db_host) {
  return process_host(p, TDSA_IN_C64, iq_host, n_samples, hop, n_frames, out_db_host);
}

int tdsa_real_input_supported(int nfft) {
  if (nfft < 2) return 0;
  const bool pow2 = (nfft & (nfft - 1)) == 0;
  if (pow2) return nfft <= (1 << kMaxLog2N) ? 1 : 0;        // the native long-frame plans have no real-input path
  return nfft <= (1 << 19) ? 1 : 0;                          // chirp-z plans; above 2^19 they split the frame in two
}

int tdsa_process_real2(tdsa_plan p, const float* lr_host, size_t n_samples, int hop, int n_frames, int channel,
                       float* out_db_host) {
  if (!p) return fail(TDSA_ERR_ARG, "null plan");
  if (p->big) return fail(TDSA_ERR_ARG, "real-input path needs an FFT size of at most 16384");
  if (p->chirp_split)      // (before any allocation or launch: the call leaves the plan as it found it)
    return fail(TDSA_ERR_ARG, "real-input path: frames above 2^19 points that are not a power of two have no plan (nfft=%d)", p->nfft);
  if (channel < TDSA_CH_MONO || channel > TDSA_CH_STEREO) return fail(TDSA_ERR_ARG, "channel %d", channel);
  if (n_frames == 0) return TDSA_OK;
  if (!lr_host || !out_db_host) return fail(TDSA_ERR_ARG, "null buffer");
  if (n_frames < 0 || n_frames > p->max_frames)
    return fail(TDSA_ERR_ARG, "n_frames=%d outside [0, max_frames=%d]", n_frames, p->max_frames);
  if (hop < 1) return fail(TDSA_ERR_ARG, "hop=%d must be >= 1", hop);
  const size_t need = size_t(n_frames - 1) * size_t(hop) + size_t(p->nfft);
  if (n_samples < need) return fail(TDSA_ERR_ARG, "n_samples=%zu too small", n_samples);
  if (!p->window_set) return fail(TDSA_ERR_STATE, "tdsa_set_window has not been called");
  const tdsa_mode& m = p->mode;
  if (m.db_mode != TDSA_DB_POW) return fail(TDSA_ERR_ARG, "real-input path computes power dB: set TDSA_DB_POW");
  if (avg_active(m) && channel == TDSA_CH_STEREO)
    return fail(TDSA_ERR_ARG, "stereo with averaging: process frame by frame (left is averaged, right is not)");
  HIPCHK(hipSetDevice(p->device));
  JOIN(p);
  const int n = p->nfft, nb = n / 2 + 1;
  const size_t in_bytes = need * sizeof(float2);
  if (in_bytes > p->in_stage_bytes) {
    HIPCHK(hipStreamSynchronize(p->stream));
    if (p->d_in_stage) HIPCHK(hipFree(p->d_in_stage));
    p->d_in_stage = nullptr;
    p->in_stage_bytes = 0;
    HIPCHK(hipMalloc(&p->d_in_stage, in_bytes));
    p->in_stage_bytes = in_bytes;
  }
  if (!p->d_cplx) HIPCHK(hipMalloc(&p->d_cplx, size_t(p->max_frames) * n * sizeof(float2)));
  if (!p->d_lin1) HIPCHK(hipMalloc(&p->d_lin1, size_t(p->max_frames) * 2 * nb * sizeof(float)));
  if (!p->d_db1) HIPCHK(hipMalloc(&p->d_db1, size_t(p->max_frames) * 2 * nb * sizeof(float)));
  const int n_sig = channel == TDSA_CH_STEREO ? 2 : 1;
  if (in_bytes * n_sig > p->real_bytes) {
    HIPCHK(hipStreamSynchronize(p->stream));
    if (p->d_real) HIPCHK(hipFree(p->d_real));
    p->d_real = nullptr;
    p->real_bytes = 0;
    HIPCHK(hipMalloc(&p->d_real, in_bytes * n_sig));
    p->real_bytes = in_bytes * n_sig;
  }
  // one tick of audio is small: samples and dB rows go through pinned, device-visible buffers that the kernels read
  // and write in place (no DMA operation either way); larger batches are copied as before
  const int rows_out = channel == TDSA_CH_STEREO ? 2 * n_frames : n_frames;
  const size_t out_bytes = size_t(rows_out) * nb * sizeof(float);
  const bool direct = in_bytes <= kZeroCopyMax && out_bytes <= kZeroCopyMax;
  const float2* lr_dev = static_cast<const float2*>(p->d_in_stage);
  float* db_dst = p->d_db1;
  if (direct) {
    { const int rc_pin = ensure_pins(p, in_bytes, out_bytes); if (rc_pin != TDSA_OK) return rc_pin; }
    std::memcpy(p->h_in_pin, lr_host, in_bytes);
    lr_dev = static_cast<const float2*>(p->h_in_pin);
    db_dst = static_cast<float*>(p->h_out_pin);
  } else {
    HIPCHK(hipMemcpyAsync(p->d_in_stage, lr_host, in_bytes, hipMemcpyHostToDevice, p->stream));
  }
  // one transform per real signal (no left / right packing: see real_select_kernel)
  float2* const za = p->d_real;
  float2* const zb = p->d_real + need;
  HIPCHK(launch_real_select(lr_dev, need, channel, za, zb, p->stream));
  for (int sig = 0; sig < n_sig; ++sig) {
    if (p->chirp) {
      // a size that is not a power of two: signal + 0i through the chirp-z core, mean removed (exact sums), and
      // the one-sided power straight from the full spectrum
      const float2* zin = sig == 0 ? za : zb;
      const long long stride = (long long)hop * sizeof(float2);
      if (chirp_sum_chunks(n) > 1 && !p->d_sums64)
        HIPCHK(hipMalloc(&p->d_sums64, size_t(p->max_frames) * chirp_sum_chunks(n) * 2 * sizeof(double)));
      HIPCHK(launch_chirp_sums(zin, 1, 0u, stride, n, n_frames, 0, p->d_sums, nullptr, 1.0f, p->stream, p->d_sums64));
      { const int rc = chirp_transform(p, zin, TDSA_IN_C64, stride, n_frames, p->d_sums, 0u, 0.0f); if (rc != TDSA_OK) return rc; }
      HIPCHK(launch_chirp_post_real(p->d_u0, n, p->m_fft, n_frames, n_sig, sig, m.power_scale, p->d_lin1, p->stream));
      continue;
    }
    SpecParams sp{};
    sp.in = sig == 0 ? za : zb;
    sp.frame_stride = (long long)hop * sizeof(float2);
    sp.n_frames = n_frames;
    sp.first_frame_index = 1;
    sp.window = p->d_window[TDSA_IN_C64];
    sp.window_perm = p->d_window_perm[TDSA_IN_C64];
    sp.tw = p->d_tw;
    sp.out_cplx = p->d_cplx;
    sp.in_scale = 1.0f;
    sp.dc_mode = DC_FRAME_MEAN;                     // signal - signal.mean()  (audio_samples.py:123)
    sp.db_mode = TDSA_DB_POW;
    sp.pscale = 1.0f;
    const LaunchGeom g = spectrum_geometry(p->log2n, n_frames, p->num_cu);
    HIPCHK(launch_spectrum(p->log2n, 1, sp, g, p->stream));
    HIPCHK(launch_real_fold(p->d_cplx, n, n_frames, n_sig, sig, m.power_scale, p->d_lin1, p->stream));
  }
  const int rows = channel == TDSA_CH_STEREO ? 2 * n_frames : n_frames;
  if (avg_active(m)) {
    AvgParams ap{};
    ap.lin = p->d_lin1;
    ap.n_frames = n_frames;
    ap.n = nb;
    ap.state = p->d_avg;
    ap.count_in = p->avg_count;
    ap.mode = m.avg_mode;
    ap.avg_n = m.avg_n;
    ap.log_floor = m.log_floor;
    ap.cal_db = m.cal_offset_db;
    ap.out_db = db_dst;
    // long batches: the chunked scan of the complex path (one thread per bin walking 2000 frames took 284 us)
    if (n_frames > 128) {
      const size_t need_chunks = size_t(avg_scan_chunks(p->max_frames));
      if (need_chunks > p->carry_chunks) {
        if (p->d_carry) { HIPCHK(hipStreamSynchronize(p->stream)); HIPCHK(hipFree(p->d_carry)); p->d_carry = nullptr; p->carry_chunks = 0; }
        HIPCHK(hipMalloc(&p->d_carry, need_chunks * p->nfft * sizeof(double)));
        p->carry_chunks = need_chunks;
      }
      const int rc = avg_use_ranges(p, ap, n_frames, p->stream);
      if (rc != TDSA_OK) return rc;
    }
    HIPCHK(launch_avg_scan(ap, p->stream, n_frames > 128 ? p->d_carry : nullptr));
    if (m.avg_mode == TDSA_AVG_LIN) {
      long long c = (long long)p->avg_count + n_frames;
      p->avg_count = int(c < m.avg_n ? c : m.avg_n);
    } else {
      p->avg_count = 1;
    }
  } else {
    HIPCHK(launch_lin_to_db(p->d_lin1, size_t(rows) * nb, m.log_floor, m.cal_offset_db, db_dst, p->stream));
  }
  if (!direct) HIPCHK(hipMemcpyAsync(out_db_host, p->d_db1, out_bytes, hipMemcpyDeviceToHost, p->stream));
  HIPCHK(hipStreamSynchronize(p->stream));
  if (direct) std::memcpy(out_db_host, p->h_out_pin, out_bytes);
  return TDSA_OK;
}

int tdsa_get_hold(tdsa_plan p, float* max_host, float* min_host, int64_t* frames_held) {
  if (!p) return fail(TDSA_ERR_ARG, "null plan");
  HIPCHK(hipSetDevice(p->device));
  JOIN(p);
  HIPCHK(hipStreamSynchronize(p->stream));
  const size_t nb = size_t(p->nfft) * sizeof(float);
  if (max_host && p->held_max > 0) HIPCHK(hipMemcpy(max_host, p->d_hold_max, nb, hipMemcpyDeviceToHost));
  if (min_host && p->held_min > 0) HIPCHK(hipMemcpy(min_host, p->d_hold_min, nb, hipMemcpyDeviceToHost));
  if (frames_held) *frames_held = p->held_max > p->held_min ? p->held_max : p->held_min;
  return TDSA_OK;
}

int tdsa_get_avg(tdsa_plan p, double* avg_linear_host, int* count) {
  if (!p) return fail(TDSA_ERR_ARG, "null plan");
  HIPCHK(hipSetDevice(p->device));
  JOIN(p);
  if (p->big) { const int rc_m = big_materialize_mean(p); if (rc_m != TDSA_OK) return rc_m; }
  HIPCHK(hipStreamSynchronize(p->stream));
  if (avg_linear_host && p->avg_count > 0)
    HIPCHK(hipMemcpy(avg_linear_host, p->d_avg, size_t(p->nfft) * sizeof(double), hipMemcpyDeviceToHost));
  if (count) *count = p->avg_count;
  return TDSA_OK;
}

static int welch_stage(tdsa_plan p, size_t bytes) {
  if (p->welch_bytes >= bytes) return TDSA_OK;
  if (p->d_welch) { HIPCHK(hipFree(p->d_welch)); p->d_welch = nullptr; p->welch_bytes = 0; }
  HIPCHK(hipMalloc(&p->d_welch, bytes));
  p->welch_bytes = bytes;
  return TDSA_OK;
}

int tdsa_host_register(void* host, size_t bytes) {
  if (!host || bytes == 0) return fail(TDSA_ERR_ARG, "null / empty host range");
  HIPCHK(hipHostRegister(host, bytes, hipHostRegisterPortable));
  return TDSA_OK;
}

int tdsa_host_unregister(void* host) {
  if (!host) return fail(TDSA_ERR_ARG, "null host pointer");
  HIPCHK(hipHostUnregister(host));
  return TDSA_OK;
}

int tdsa_welch_export(tdsa_plan p, void* mean_host, int as_f32, int* count) {
  if (!p || !mean_host) return fail(TDSA_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(p->device));
  JOIN(p);
  if (count) *count = p->avg_count;
  if (p->avg_count <= 0) return TDSA_OK;
  const size_t n = size_t(p->nfft), nb = n * (as_f32 ? sizeof(float) : sizeof(double));
  const bool from_sum = p->big && p->big_mean_in_sum;
  if (!as_f32 && !from_sum) {       // the state is the float64 mean already
    HIPCHK(hipMemcpyAsync(mean_host, p->d_avg, nb, hipMemcpyDeviceToHost, p->stream));
  } else {
    { const int rc = welch_stage(p, nb); if (rc != TDSA_OK) return rc; }
    HIPCHK(launch_welch_export(from_sum ? p->d_sum : p->d_avg, from_sum ? double(p->avg_count) : 1.0, p->d_welch, as_f32,
                               (long long)n, p->stream));
    HIPCHK(hipMemcpyAsync(mean_host, p->d_welch, nb, hipMemcpyDeviceToHost, p->stream));
  }
  HIPCHK(hipStreamSynchronize(p->stream));
  return TDSA_OK;
}

int tdsa_welch_export_dev(tdsa_plan p, void* mean_dev, int as_f32, int* count) {
  if (!p || !mean_dev) return fail(TDSA_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(p->device));
  JOIN(p);
  if (count) *count = p->avg_count;
  if (p->avg_count <= 0) return TDSA_OK;
  const bool from_sum = p->big && p->big_mean_in_sum;
  HIPCHK(launch_welch_export(from_sum ? p->d_sum : p->d_avg, from_sum ? double(p->avg_count) : 1.0, mean_dev, as_f32,
                             (long long)p->nfft, p->stream));
  HIPCHK(hipStreamSynchronize(p->stream));     // the caller tells the combining process next: the values must have landed
  return TDSA_OK;
}

// ---- device buffers another process of the node can read in place (HIP IPC; peer reads go over xGMI) ----
int tdsa_peer_alloc(int device_id, size_t bytes, void** dev_ptr, unsigned char* handle64) {
  if (!dev_ptr || !handle64 || bytes == 0) return fail(TDSA_ERR_ARG, "null / empty argument");
  static_assert(sizeof(hipIpcMemHandle_t) == TDSA_PEER_HANDLE_BYTES, "handle size");
  *dev_ptr = nullptr;
  HIPCHK(hipSetDevice(device_id));
  void* d = nullptr;
  HIPCHK(hipMalloc(&d, bytes));
  hipIpcMemHandle_t h;
  const hipError_t e = hipIpcGetMemHandle(&h, d);
  if (e != hipSuccess) {
    (void)hipFree(d);
    return fail(TDSA_ERR_HIP, "hipIpcGetMemHandle: %s", hipGetErrorString(e));
  }
  HIPCHK(hipMemset(d, 0, bytes));
  std::memcpy(handle64, &h, sizeof(h));
  *dev_ptr = d;
  return TDSA_OK;
}

int tdsa_peer_free(int device_id, void* dev_ptr) {
  if (!dev_ptr) return TDSA_OK;
  HIPCHK(hipSetDevice(device_id));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipFree(dev_ptr));
  return TDSA_OK;
}

int tdsa_peer_can_access(int device_id, int peer_device_id, int* can_access) {
  if (!can_access) return fail(TDSA_ERR_ARG, "null argument");
  *can_access = 0;
  if (device_id == peer_device_id) { *can_access = 1; return TDSA_OK; }
  HIPCHK(hipDeviceCanAccessPeer(can_access, device_id, peer_device_id));
  return TDSA_OK;
}

int tdsa_peer_open(int device_id, const unsigned char* handle64, int owner_device_id, void** dev_ptr) {
  if (!dev_ptr || !handle64) return fail(TDSA_ERR_ARG, "null argument");
  *dev_ptr = nullptr;
  HIPCHK(hipSetDevice(device_id));
  if (owner_device_id >= 0 && owner_device_id != device_id) {
    int can = 0;
    HIPCHK(hipDeviceCanAccessPeer(&can, device_id, owner_device_id));
    if (!can) return fail(TDSA_ERR_STATE, "device %d cannot read device %d's memory", device_id, owner_device_id);
  }
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle64, sizeof(h));
  void* d = nullptr;
  HIPCHK(hipIpcOpenMemHandle(&d, h, hipIpcMemLazyEnablePeerAccess));
  // probe: a mapping this device cannot read must show up here as an error code, not later as a fault inside a kernel
  unsigned probe = 0;
  const hipError_t e = hipMemcpy(&probe, d, sizeof(probe), hipMemcpyDeviceToHost);
  if (e != hipSuccess) {
    (void)hipIpcCloseMemHandle(d);
    return fail(TDSA_ERR_HIP, "mapped peer buffer is not readable: %s", hipGetErrorString(e));
  }
  *dev_ptr = d;
  return TDSA_OK;
}

int tdsa_peer_close(int device_id, void* dev_ptr) {
  if (!dev_ptr) return TDSA_OK;
  HIPCHK(hipSetDevice(device_id));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipIpcCloseMemHandle(dev_ptr));
  return TDSA_OK;
}

// parts_host != null: the partial means sit part_stride_bytes apart in host memory and are staged on this device first;
// else parts_dev[r] are device pointers this device can read (its own memory or tdsa_peer_open'ed buffers of other ranks)
static int welch_combine_impl(tdsa_plan p, const void* parts_host, size_t part_stride_bytes, const void* const* parts_dev,
                              const int32_t* counts, int n_parts, int as_f32, float* out_db_dev, float* out_db_host) {
  if (!p || !(parts_host || parts_dev) || !counts) return fail(TDSA_ERR_ARG, "null argument");
  if (n_parts < 1 || n_parts > kWelchMaxParts) return fail(TDSA_ERR_ARG, "n_parts=%d outside [1, %d]", n_parts, kWelchMaxParts);
  const tdsa_mode& m = p->mode;
  const size_t n = size_t(p->nfft), nb = n * (as_f32 ? sizeof(float) : sizeof(double));
  if (parts_host && part_stride_bytes < nb)
    return fail(TDSA_ERR_ARG, "part_stride_bytes=%zu < %zu bytes of one partial", part_stride_bytes, nb);
  long long total = 0;
  int cnt[kWelchMaxParts];
  for (int r = 0; r < n_parts; ++r) {
    if (counts[r] < 0) return fail(TDSA_ERR_ARG, "counts[%d]=%d", r, counts[r]);
    cnt[r] = counts[r];
    total += counts[r];
  }
  if (total == 0) return fail(TDSA_ERR_ARG, "no segments behind the partial means");
  if (!(avg_active(m) && m.avg_mode == TDSA_AVG_LIN && total <= m.avg_n))
    return fail(TDSA_ERR_STATE, "partial means combine only into an uncapped running mean: avg lin with avg_n >= %lld segments", total);
  if (p->chirp) return fail(TDSA_ERR_STATE, "not available for chirp-z plans");
  HIPCHK(hipSetDevice(p->device));
  JOIN(p);
  const size_t staged = parts_host ? nb * size_t(n_parts) : 0;
  { const int rc = welch_stage(p, staged + (out_db_host && !out_db_dev ? n * sizeof(float) : 0)); if (rc != TDSA_OK) return rc; }
  const void* part[kWelchMaxParts];
  if (parts_host) {
    // one strided copy: the parts may sit part_stride_bytes apart in the caller's (pinned, shared) slab
    HIPCHK(hipMemcpy2DAsync(p->d_welch, nb, parts_host, part_stride_bytes, nb, size_t(n_parts), hipMemcpyHostToDevice, p->stream));
    for (int r = 0; r < n_parts; ++r) part[r] = static_cast<const unsigned char*>(p->d_welch) + nb * size_t(r);
  } else {
    for (int r = 0; r < n_parts; ++r) {
      if (cnt[r] != 0 && !parts_dev[r]) return fail(TDSA_ERR_ARG, "parts_dev[%d] is null", r);
      part[r] = parts_dev[r];
    }
  }
  float* out_dev = out_db_dev ? out_db_dev
                              : (out_db_host ? reinterpret_cast<float*>(static_cast<unsigned char*>(p->d_welch) + staged) : nullptr);
  const bool hmax = (m.hold_flags & TDSA_HOLD_MAX) != 0, hmin = (m.hold_flags & TDSA_HOLD_MIN) != 0;
  const float pscale = (p->big && m.db_mode == TDSA_DB_POW) ? m.power_scale : 1.0f;
  HIPCHK(launch_welch_combine(part, cnt, n_parts, as_f32, (long long)n, p->big ? p->d_sum : nullptr,
                              p->big ? nullptr : p->d_avg, int(total), p->big ? 0 : 1, m.db_mode, pscale, m.log_floor,
                              m.cal_offset_db, p->tare_active ? p->d_tare_base : nullptr, out_dev,
                              hmax ? p->d_hold_max : nullptr, hmin ? p->d_hold_min : nullptr, p->held_max == 0,
                              p->held_min == 0, p->stream));
  p->avg_count = int(total);
  if (p->big) p->big_mean_in_sum = true;
  if (hmax) p->held_max += 1;
  if (hmin) p->held_min += 1;
  if (out_db_host) {
    HIPCHK(hipMemcpyAsync(out_db_host, out_dev, n * sizeof(float), hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
  }
  return TDSA_OK;
}

int tdsa_welch_combine(tdsa_plan p, const void* parts_host, size_t part_stride_bytes, const int32_t* counts, int n_parts,
                       int as_f32, float* out_db_dev, float* out_db_host) {
  if (!parts_host) return fail(TDSA_ERR_ARG, "null argument");
  return welch_combine_impl(p, parts_host, part_stride_bytes, nullptr, counts, n_parts, as_f32, out_db_dev, out_db_host);
}

int tdsa_welch_combine_dev(tdsa_plan p, const void* const* parts_dev, const int32_t* counts, int n_parts, int as_f32,
                           float* out_db_dev, float* out_db_host) {
  if (!parts_dev) return fail(TDSA_ERR_ARG, "null argument");
  return welch_combine_impl(p, nullptr, 0, parts_dev, counts, n_parts, as_f32, out_db_dev, out_db_host);
}

int tdsa_shader_clock(tdsa_plan p, float* shader_mhz, float* ns_per_valu) {
  if (!p) return fail(TDSA_ERR_ARG, "null plan");
  HIPCHK(hipSetDevice(p->device));
  JOIN(p);
  if (!p->d_clock) HIPCHK(hipMalloc(&p->d_clock, size_t(p->num_cu) * 4 * sizeof(float)));
  const int iters = 4000;                                    // 256 000 instructions per wave: about a millisecond
  HIPCHK(launch_valu_clock(p->d_clock, p->num_cu, 200, p->stream));
  HIPCHK(hipEventRecord(p->ev0, p->stream));
  HIPCHK(launch_valu_clock(p->d_clock, p->num_cu, iters, p->stream));
  HIPCHK(hipEventRecord(p->ev1, p->stream));
  HIPCHK(hipEventSynchronize(p->ev1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, p->ev0, p->ev1));
  // four waves per SIMD, each 64 * iters instructions: ns per wave-instruction per SIMD; a gfx950 SIMD retires one fp32
  // wave-instruction per 2 clocks (64 FLOP / clk / SIMD as FMAs: MI355X_MICROARCH.md)
  const double ns = double(ms) * 1e6 / (double(iters) * 64.0 * 4.0);
  if (ns_per_valu) *ns_per_valu = float(ns);
  if (shader_mhz) *shader_mhz = float(2.0e3 / ns);
  return TDSA_OK;
}

int tdsa_get_dc(tdsa_plan p, float* re, float* im) {
  if (!p) return fail(TDSA_ERR_ARG, "null plan");
  HIPCHK(hipSetDevice(p->device));
  JOIN(p);
  HIPCHK(hipStreamSynchronize(p->stream));
  float2 dc;
  HIPCHK(hipMemcpy(&dc, p->d_dc_state, sizeof(dc), hipMemcpyDeviceToHost));
  if (re) *re = dc.x;
  if (im) *im = dc.y;
  return TDSA_OK;
}

int tdsa_set_dc(tdsa_plan p, float re, float im) {
  if (!p) return fail(TDSA_ERR_ARG, "null plan");
  HIPCHK(hipSetDevice(p->device));
  JOIN(p);
  HIPCHK(hipStreamSynchronize(p->stream));
  const float2 dc{re, im};
  HIPCHK(hipMemcpy(p->d_dc_state, &dc, sizeof(dc), hipMemcpyHostToDevice));
  return TDSA_OK;
}

int tdsa_synchronize(tdsa_plan p) {
  if (!p) return fail(TDSA_ERR_ARG, "null plan");
  HIPCHK(hipSetDevice(p->device));
  JOIN(p);
  HIPCHK(hipStreamSynchronize(p->stream));
  return TDSA_OK;
}

// ---- trace objects ---------------------------------------------------------------------------------
struct tdsa_trace_s {
  int device = 0, n = 0;
  hipStream_t stream = nullptr;
  float* d_in = nullptr;
  float* d_live = nullptr;
  float* d_hold_max = nullptr;
  float* d_hold_min = nullptr;
  float* d_tare_base = nullptr;
  float* d_tare_acc = nullptr;
  double* d_avg = nullptr;
  double* d_avg_in = nullptr;
  float* h_pin = nullptr;       // pinned, device-visible: [4][n] row in, live / max / min out (one GUI tick, zero-copy)
  double* h_pin_avg = nullptr;  // pinned: [2][n] linear row in, averager state out (tdsa_trace_avg_process)
  long long held_max = 0, held_min = 0;
  bool tare_active = false;
  int tare_count = 0;
  int avg_mode = TDSA_AVG_OFF, avg_n = 1, avg_count = 0;
};

int tdsa_trace_create(int device_id, int n, tdsa_trace* out) {
  if (!out) return fail(TDSA_ERR_ARG, "out is null");
  *out = nullptr;
  if (n < 1) return fail(TDSA_ERR_ARG, "n=%d must be >= 1", n);
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device_id < 0 || device_id >= ndev) return fail(TDSA_ERR_ARG, "device %d of %d", device_id, ndev);
  HIPCHK(hipSetDevice(device_id));
  tdsa_trace t = new (std::nothrow) tdsa_trace_s();
  if (!t) return fail(TDSA_ERR_NOMEM, "host allocation failed");
  t->device = device_id;
  t->n = n;
  const size_t nb = size_t(n) * sizeof(float);
  hipError_t e = hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking);
  float** fbufs[] = {&t->d_in, &t->d_live, &t->d_hold_max, &t->d_hold_min, &t->d_tare_base, &t->d_tare_acc};
  for (float** b : fbufs)
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(b), nb);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&t->d_avg), size_t(n) * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&t->d_avg_in), size_t(n) * sizeof(double));
  if (e != hipSuccess) {                     // a failure half way leaves nothing behind
    (void)tdsa_trace_destroy(t);
    return fail(TDSA_ERR_HIP, "trace create: %s", hipGetErrorString(e));
  }
  *out = t;
  return TDSA_OK;
}

int tdsa_trace_destroy(tdsa_trace t) {
  if (!t) return TDSA_OK;
  (void)hipSetDevice(t->device);
  if (t->stream) (void)hipStreamSynchronize(t->stream);
  void* bufs[] = {t->d_in, t->d_live, t->d_hold_max, t->d_hold_min, t->d_tare_base, t->d_tare_acc, t->d_avg,
                  t->d_avg_in};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (t->h_pin) (void)hipHostFree(t->h_pin);
  if (t->h_pin_avg) (void)hipHostFree(t->h_pin_avg);
  if (t->stream) (void)hipStreamDestroy(t->stream);
  delete t;
  return TDSA_OK;
}

int tdsa_trace_reset(tdsa_trace t, uint32_t what) {
  if (!t) return fail(TDSA_ERR_ARG, "null trace");
  if (what & TDSA_RESET_AVG) t->avg_count = 0;
  if (what & TDSA_RESET_HOLD_MAX) t->held_max = 0;
  if (what & TDSA_RESET_HOLD_MIN) t->held_min = 0;
  if (what & TDSA_RESET_TARE) {
    t->tare_active = false;
    t->tare_count = 0;
  }
  return TDSA_OK;
}

int tdsa_trace_update(tdsa_trace t, const float* db_in_host, int n, float cal_offset_db, int tare_collect,
                      int tare_total, int tare_subtract, uint32_t hold_flags, float* live_out, float* max_out,
                      float* min_out, int* tare_done) {
  if (!t || !db_in_host) return fail(TDSA_ERR_ARG, "null argument");
  if (n != t->n) return fail(TDSA_ERR_ARG, "row length %d != trace length %d", n, t->n);
  if (tare_collect && tare_total < 1) return fail(TDSA_ERR_ARG, "tare_total=%d", tare_total);
  HIPCHK(hipSetDevice(t->device));
  const size_t nb = size_t(n) * sizeof(float);
  // one displayed frame: the kernel reads the row from and writes its results to pinned, device-visible memory of the
  // trace object - no DMA operation on the way in or out (each costs ~10 us from / to pageable memory)
  if (!t->h_pin) HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&t->h_pin), 4 * nb, hipHostMallocPortable | hipHostMallocMapped));
  float* const h_in = t->h_pin;
  float* const h_live = t->h_pin + n;
  float* const h_max = t->h_pin + 2 * size_t(n);
  float* const h_min = t->h_pin + 3 * size_t(n);
  std::memcpy(h_in, db_in_host, nb);
  TraceParams tp{};
  tp.db_in = h_in;
  tp.n = n;
  tp.cal_db = cal_offset_db;
  tp.tare_acc = t->d_tare_acc;
  tp.tare_base = t->d_tare_base;
  bool finish = false;
  if (tare_collect) {
    tp.tare_collect = 1;
    tp.tare_first = t->tare_count == 0;
    t->tare_count += 1;
    tp.tare_count = t->tare_count;
    finish = t->tare_count >= tare_total;
    tp.tare_finish = finish;
  }
  tp.tare_active = ((tare_subtract && t->tare_active) || finish) ? 1 : 0;
  tp.live = live_out ? h_live : nullptr;
  tp.state_max = (hold_flags & TDSA_HOLD_MAX) ? t->d_hold_max : nullptr;
  tp.state_min = (hold_flags & TDSA_HOLD_MIN) ? t->d_hold_min : nullptr;
  tp.max_copy = (max_out && tp.state_max) ? h_max : nullptr;
  tp.min_copy = (min_out && tp.state_min) ? h_min : nullptr;
  tp.max_first = t->held_max == 0;
  tp.min_first = t->held_min == 0;
  HIPCHK(launch_trace_update(tp, t->stream));
  if (finish) {
    t->tare_active = true;
    t->tare_count = 0;
  }
  if (tare_done) *tare_done = finish ? 1 : 0;
  if (hold_flags & TDSA_HOLD_MAX) t->held_max += 1;
  if (hold_flags & TDSA_HOLD_MIN) t->held_min += 1;
  HIPCHK(hipStreamSynchronize(t->stream));
  if (live_out) std::memcpy(live_out, h_live, nb);
  if (tp.max_copy) std::memcpy(max_out, h_max, nb);
  if (tp.min_copy) std::memcpy(min_out, h_min, nb);
  return TDSA_OK;
}

int tdsa_trace_get_tare_baseline(tdsa_trace t, float* baseline_db_host, int* active) {
  if (!t) return fail(TDSA_ERR_ARG, "null trace");
  HIPCHK(hipSetDevice(t->device));
  HIPCHK(hipStreamSynchronize(t->stream));
  if (baseline_db_host && t->tare_active)
    HIPCHK(hipMemcpy(baseline_db_host, t->d_tare_base, size_t(t->n) * sizeof(float), hipMemcpyDeviceToHost));
  if (active) *active = t->tare_active ? 1 : 0;
  return TDSA_OK;
}

int tdsa_trace_set_tare_baseline(tdsa_trace t, const float* baseline_db_host, int n) {
  if (!t) return fail(TDSA_ERR_ARG, "null trace");
  if (!baseline_db_host) {
    t->tare_active = false;
    return TDSA_OK;
  }
  if (n != t->n) return fail(TDSA_ERR_ARG, "baseline length %d != trace length %d", n, t->n);
  HIPCHK(hipSetDevice(t->device));
  HIPCHK(hipStreamSynchronize(t->stream));
  HIPCHK(hipMemcpy(t->d_tare_base, baseline_db_host, size_t(n) * sizeof(float), hipMemcpyHostToDevice));
  t->tare_active = true;
  return TDSA_OK;
}

int tdsa_trace_avg_set_mode(tdsa_trace t, int avg_mode, int avg_n) {
  if (!t) return fail(TDSA_ERR_ARG, "null trace");
  if (avg_mode < TDSA_AVG_OFF || avg_mode > TDSA_AVG_LIN) return fail(TDSA_ERR_ARG, "avg_mode %d", avg_mode);
  t->avg_mode = avg_mode;
  t->avg_n = avg_n < 1 ? 1 : avg_n;   // TraceAverager.set_mode: n = max(1, n), then reset()
  t->avg_count = 0;
  return TDSA_OK;
}

int tdsa_trace_avg_process(tdsa_trace t, const double* linear_in_host, int n, double* avg_out_host,
                           int* count_out) {
  if (!t || !linear_in_host) return fail(TDSA_ERR_ARG, "null argument");
  if (n != t->n) return fail(TDSA_ERR_ARG, "row length %d != trace length %d", n, t->n);
  if (t->avg_mode == TDSA_AVG_OFF || t->avg_n <= 1)
    return fail(TDSA_ERR_STATE, "averaging is off (pass-through is the caller's job)");
  HIPCHK(hipSetDevice(t->device));
  const size_t nb = size_t(n) * sizeof(double);
  // one row per call: in through pinned, device-visible memory the kernel reads in place, the state back through a
  // DMA copy into pinned memory (the float64 state itself stays on the device)
  if (!t->h_pin_avg)
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&t->h_pin_avg), 2 * nb, hipHostMallocPortable | hipHostMallocMapped));
  double* const h_in = t->h_pin_avg;
  double* const h_out = t->h_pin_avg + n;
  std::memcpy(h_in, linear_in_host, nb);
  HIPCHK(launch_avg_host_frame(h_in, n, t->d_avg, t->avg_count, t->avg_mode, t->avg_n, t->stream));
  if (t->avg_count == 0) t->avg_count = 1;
  else if (t->avg_mode == TDSA_AVG_LIN && t->avg_count < t->avg_n) t->avg_count += 1;
  if (avg_out_host) HIPCHK(hipMemcpyAsync(h_out, t->d_avg, nb, hipMemcpyDeviceToHost, t->stream));
  HIPCHK(hipStreamSynchronize(t->stream));
  if (avg_out_host) std::memcpy(avg_out_host, h_out, nb);
  if (count_out) *count_out = t->avg_count;
  return TDSA_OK;
}

int tdsa_dev_alloc(int device_id, size_t bytes, void** out_dev) {
  if (!out_dev) return fail(TDSA_ERR_ARG, "out is null");
  HIPCHK(hipSetDevice(device_id));
  HIPCHK(hipMalloc(out_dev, bytes));
  return TDSA_OK;
}
int tdsa_dev_free(int device_id, void* dev) {
  HIPCHK(hipSetDevice(device_id));
  if (dev) HIPCHK(hipFree(dev));
  return TDSA_OK;
}
int tdsa_memcpy_h2d(int device_id, void* dst_dev, const void* src_host, size_t bytes) {
  HIPCHK(hipSetDevice(device_id));
  HIPCHK(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
  return TDSA_OK;
}
int tdsa_memcpy_d2h(int device_id, void* dst_host, const void* src_dev, size_t bytes) {
  HIPCHK(hipSetDevice(device_id));
  HIPCHK(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
  return TDSA_OK;
}

// developer hook: plan parameters the tools and the tests move to reach code paths that otherwise need other hardware
// or very long batches (rounds 1-4 read them from the environment at plan creation: ADVICE r4)
int tdsa_debug_knob(tdsa_plan p, const char* name, int value) {
  if (!p || !name) return fail(TDSA_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(p->device));
  JOIN(p);
  HIPCHK(hipStreamSynchronize(p->stream));
  const std::string k(name);
  if (k == "num_cu") {                       // persistent grids sized for fewer CUs
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, p->device));
    (void)prop;
    if (value < 1 || value > p->num_cu) return fail(TDSA_ERR_ARG, "num_cu=%d outside [1, %d] (it can only shrink: buffers are sized for it)", value, p->num_cu);
    p->num_cu = value;
    p->agg_w_key[0] = -1;
  } else if (k == "avg_wg_min") {            // batches of more frames than this take the workgroup-chunk scan
    if (value < 1) return fail(TDSA_ERR_ARG, "avg_wg_min=%d", value);
    p->avg_wg_min = value;
  } else if (k == "avg_f64_chunks") {        // 1: always the scan over fixed 64-frame chunks with float64 aggregates
    p->avg_f64_chunks = value != 0;
    p->agg_w_key[0] = -1;
  } else if (k == "overlap_share") {         // percent of the CUs an overlapped launch is sized for
    if (value < 10 || value > 100) return fail(TDSA_ERR_ARG, "overlap_share=%d outside [10, 100]", value);
    p->overlap_share = value;
  } else if (k == "big_group") {             // long-frame plans: segments per column / row round (<= the 64 the plan was made for)
    if (!p->big || value < 1 || value > 64) return fail(TDSA_ERR_ARG, "big_group=%d (long-frame plans, 1 .. 64)", value);
    if (p->d_z && value > p->big_group) return fail(TDSA_ERR_STATE, "big_group can only shrink once the plan has run");
    p->big_group = value;
  } else if (k == "big_fuse_gather") {
    p->big_fuse_gather = value;
  } else if (k == "big_queue_gave_up") {       // reads (value ignored): how many workgroups of the fused launch ever gave up waiting - as the error text
    unsigned long long q[4] = {0, 0, 0, 0};
    if (p->d_bigq) { HIPCHK(hipStreamSynchronize(p->stream)); HIPCHK(hipMemcpy(q, p->d_bigq, sizeof(q), hipMemcpyDeviceToHost)); }
    return q[2] == 0 ? TDSA_OK : fail(TDSA_ERR_STATE, "%llu workgroups of a fused row + gather launch gave up waiting", q[2]);
  } else if (k == "smooth") {                // sizes 2^a 3^b 5^c (<= 10 000 in one pass, two passes above): 1 = mixed-radix transform of N points (default), 0 = chirp-z
    p->smooth_on = value != 0;
  } else if (k == "smooth_n1") {             // two-pass sizes: the column pass's transform length (a divisor; both factors <= 10 000)
    if (!p->smooth || p->smooth_n1 == 0) return fail(TDSA_ERR_STATE, "not a two-pass mixed-radix plan");
    if (value < 2 || p->nfft % value != 0 || value > kSmoothMaxN || p->nfft / value > kSmoothMaxN)
      return fail(TDSA_ERR_ARG, "smooth_n1=%d does not split %d into two factors <= %d", value, p->nfft, kSmoothMaxN);
    p->smooth_n1 = value;
    p->smooth_n2 = p->nfft / value;
    p->smooth_stages = smooth_radices(p->smooth_n1, p->smooth_radix);
    p->smooth_stages2 = smooth_radices(p->smooth_n2, p->smooth_radix2);
  } else if (k == "chirp_fuse_big") {        // long chirp-z frames: 1 = element-wise passes inside the column passes (default)
    p->chirp_fuse_big = value != 0;
  } else if (k == "chirp_single") {          // chirp-z plans: 1 = one launch per call (default), 0 = the separate passes
    p->chirp_single = value != 0;
#ifdef TDSA_DEV
  } else if (k == "cu_mask") {               // the plan's stream confined to a set of CUs (tools/c5_two_plans.py): 1 / 2 = mask words
    // 0-3 / 4-7 (four whole XCDs each on this chip), 3 / 4 = every second CU (even / odd bits of all eight words)
    if (value < 1 || value > 4) return fail(TDSA_ERR_ARG, "cu_mask=%d outside [1, 4]", value);
    uint32_t mask[8];
    for (int w = 0; w < 8; ++w)
      mask[w] = value == 1 ? (w < 4 ? 0xffffffffu : 0u) : value == 2 ? (w < 4 ? 0u : 0xffffffffu) : value == 3 ? 0x55555555u : 0xaaaaaaaau;
    HIPCHK(hipStreamSynchronize(p->stream));
    HIPCHK(hipStreamDestroy(p->stream));
    p->stream = nullptr;
    HIPCHK(hipExtStreamCreateWithCUMask(&p->stream, 8, mask));
  } else if (k == "big_pre_wgs") {           // long-frame plans: empty workgroups ahead of every column pass (XCD phase)
    if (value < 0 || value > 64) return fail(TDSA_ERR_ARG, "big_pre_wgs=%d outside [0, 64]", value);
    p->big_pre_wgs = value;
#endif
  } else {
    return fail(TDSA_ERR_ARG, "unknown knob '%s'", name);
  }
  return TDSA_OK;
}

// developer hook (developer section of include/tdsa_hip.h): phase timeline of workgroup 0, TDSA_TIMELINE builds
int tdsa_debug_timeline(tdsa_plan p, unsigned long long* host_out_2048) {
  if (!p) return fail(TDSA_ERR_ARG, "null plan");
  HIPCHK(hipSetDevice(p->device));
  JOIN(p);
  if (!p->d_dbg) {
    HIPCHK(hipMalloc(&p->d_dbg, 2048 * sizeof(unsigned long long)));
    HIPCHK(hipMemset(p->d_dbg, 0, 2048 * sizeof(unsigned long long)));
  }
  if (host_out_2048) {
    HIPCHK(hipStreamSynchronize(p->stream));
    HIPCHK(hipMemcpy(host_out_2048, p->d_dbg, 2048 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  }
  return TDSA_OK;
}

int tdsa_profile_enable(tdsa_plan p, int enable) {
  if (!p) return fail(TDSA_ERR_ARG, "null plan");
  p->profiling = enable != 0;
  return TDSA_OK;
}

int tdsa_profile_read(tdsa_plan p, int* launches, float* total_ms) {
  if (!p) return fail(TDSA_ERR_ARG, "null plan");
  HIPCHK(hipSetDevice(p->device));
  JOIN(p);
  HIPCHK(hipStreamSynchronize(p->stream));
  float total = 0.f;
  for (size_t i = 0; i + 1 < p->prof_used; i += 2) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, p->prof_events[i], p->prof_events[i + 1]));
    total += ms;
  }
  if (launches) *launches = int(p->prof_used / 2);
  if (total_ms) *total_ms = total;
  p->prof_used = 0;
  return TDSA_OK;
}

int tdsa_timer_begin(tdsa_plan p) {
  if (!p) return fail(TDSA_ERR_ARG, "null plan");
  HIPCHK(hipSetDevice(p->device));
  JOIN(p);
  HIPCHK(hipEventRecord(p->ev0, p->stream));
  return TDSA_OK;
}
int tdsa_timer_end(tdsa_plan p, float* elapsed_ms) {
  if (!p) return fail(TDSA_ERR_ARG, "null plan");
  HIPCHK(hipSetDevice(p->device));
  JOIN(p);
  HIPCHK(hipEventRecord(p->ev1, p->stream));
  HIPCHK(hipEventSynchronize(p->ev1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, p->ev0, p->ev1));
  if (elapsed_ms) *elapsed_ms = ms;
  return TDSA_OK;
}

}  // extern "C"

// ================================================================================================
// tdsa_pipe: pinned host ring + asynchronous H2D / frame kernel / D2H legs on separate streams.
// Counterpart of the reader-thread -> queue.Queue(4) -> get_power_levels() front end of
// HackrfSamplesDataSource (datasources/hackrf_samples.py:191-305) for batch users: the producer writes
// IQ bytes straight into a pinned slot, the copy of slot k+1 and the read-back of slot k-1 overlap
// the frame kernel of slot k.
// ================================================================================================
struct tdsa_pipe_s {
  tdsa_plan plan = nullptr;
  int fmt = TDSA_IN_I8;
  size_t slot_samples = 0;
  bool rows = false;        // dB rows are produced (kept per slot on the device)
  bool rows_host = false;   // ... and read back into pinned host memory
  bool rows_u8 = false;     // ... as bytes under the display's levels (1 B per bin over PCIe instead of 4)
  float lo_db = -120.0f, hi_db = 0.0f;
  struct Slot {
    void* h_in = nullptr;
    void* d_in = nullptr;
    float* h_out = nullptr;
    float* d_out = nullptr;
    unsigned char* h_u8 = nullptr;
    unsigned char* d_u8 = nullptr;
    hipEvent_t ev_h2d = nullptr, ev_done = nullptr, ev_d2h = nullptr;
    int n_frames = 0;
    bool acquired = false, in_flight = false;
  };
  std::vector<Slot> slots;
  hipStream_t s_in = nullptr, s_out = nullptr;
  size_t head = 0, tail = 0;   // next slot to acquire / to collect
  int pending = 0;
};

int tdsa_pipe_create(tdsa_plan p, int in_format, size_t slot_samples, int n_slots, int want_rows, tdsa_pipe* out) {
  if (!p || !out) return fail(TDSA_ERR_ARG, "null argument");
  if (in_format < TDSA_IN_I8 || in_format > TDSA_IN_C64) return fail(TDSA_ERR_ARG, "in_format %d", in_format);
  if (n_slots < 1 || n_slots > 16) return fail(TDSA_ERR_ARG, "n_slots=%d outside [1, 16]", n_slots);
  if (slot_samples < size_t(p->nfft)) return fail(TDSA_ERR_ARG, "slot_samples=%zu < nfft", slot_samples);
  if (want_rows < 0 || want_rows > 3)
    return fail(TDSA_ERR_ARG, "want_rows=%d (0 none, 1 host, 2 device, 3 host as uint8 levels)", want_rows);
  HIPCHK(hipSetDevice(p->device));
  tdsa_pipe q = new (std::nothrow) tdsa_pipe_s();
  if (!q) return fail(TDSA_ERR_NOMEM, "out of host memory");
  q->plan = p;
  q->fmt = in_format;
  q->slot_samples = slot_samples;
  q->rows = want_rows != 0;
  q->rows_host = want_rows == 1;
  q->rows_u8 = want_rows == 3;
  q->slots.resize(size_t(n_slots));
  const size_t in_bytes = slot_samples * size_t(bytes_per_sample(in_format));
  const size_t out_rows = p->big ? 1 : size_t(p->max_frames);
  const size_t out_bytes = out_rows * size_t(p->nfft) * sizeof(float);
  auto bail = [&](hipError_t e, const char* what) {
    (void)tdsa_pipe_destroy(q);
    return fail(TDSA_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
  };
  hipError_t e;
  if ((e = hipStreamCreateWithFlags(&q->s_in, hipStreamNonBlocking)) != hipSuccess) return bail(e, "stream");
  if ((e = hipStreamCreateWithFlags(&q->s_out, hipStreamNonBlocking)) != hipSuccess) return bail(e, "stream");
  for (auto& sl : q->slots) {
    if ((e = hipHostMalloc(&sl.h_in, in_bytes, hipHostMallocDefault)) != hipSuccess) return bail(e, "pinned input slot");
    if ((e = hipMalloc(&sl.d_in, in_bytes)) != hipSuccess) return bail(e, "device input slot");
    if (q->rows) {
      if (q->rows_host &&
          (e = hipHostMalloc(reinterpret_cast<void**>(&sl.h_out), out_bytes, hipHostMallocDefault)) != hipSuccess)
        return bail(e, "pinned output slot");
      if ((e = hipMalloc(reinterpret_cast<void**>(&sl.d_out), out_bytes)) != hipSuccess) return bail(e, "device output slot");
      if (q->rows_u8) {
        if ((e = hipMalloc(reinterpret_cast<void**>(&sl.d_u8), out_bytes / 4)) != hipSuccess) return bail(e, "device byte rows");
        if ((e = hipHostMalloc(reinterpret_cast<void**>(&sl.h_u8), out_bytes / 4, hipHostMallocDefault)) != hipSuccess)
          return bail(e, "pinned byte rows");
      }
    }
    if ((e = hipEventCreateWithFlags(&sl.ev_h2d, hipEventDisableTiming)) != hipSuccess) return bail(e, "event");
    if ((e = hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming)) != hipSuccess) return bail(e, "event");
    if ((e = hipEventCreateWithFlags(&sl.ev_d2h, hipEventDisableTiming)) != hipSuccess) return bail(e, "event");
  }
  *out = q;
  return TDSA_OK;
}

int tdsa_pipe_destroy(tdsa_pipe q) {
  if (!q) return TDSA_OK;
  if (q->plan) (void)hipSetDevice(q->plan->device);
  if (q->s_in) (void)hipStreamSynchronize(q->s_in);
  if (q->plan) (void)tdsa_synchronize(q->plan);
  if (q->s_out) (void)hipStreamSynchronize(q->s_out);
  for (auto& sl : q->slots) {
    if (sl.h_in) (void)hipHostFree(sl.h_in);
    if (sl.d_in) (void)hipFree(sl.d_in);
    if (sl.h_out) (void)hipHostFree(sl.h_out);
    if (sl.d_out) (void)hipFree(sl.d_out);
    if (sl.h_u8) (void)hipHostFree(sl.h_u8);
    if (sl.d_u8) (void)hipFree(sl.d_u8);
    if (sl.ev_h2d) (void)hipEventDestroy(sl.ev_h2d);
    if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
    if (sl.ev_d2h) (void)hipEventDestroy(sl.ev_d2h);
  }
  if (q->s_in) (void)hipStreamDestroy(q->s_in);
  if (q->s_out) (void)hipStreamDestroy(q->s_out);
  delete q;
  return TDSA_OK;
}

int tdsa_pipe_acquire(tdsa_pipe q, void** host_slot) {
  if (!q || !host_slot) return fail(TDSA_ERR_ARG, "null argument");
  auto& sl = q->slots[q->head % q->slots.size()];
  if (sl.acquired) return fail(TDSA_ERR_STATE, "slot already acquired: submit it first");
  if (sl.in_flight) return fail(TDSA_ERR_STATE, "all %zu slots in flight: collect one first", q->slots.size());
  sl.acquired = true;
  *host_slot = sl.h_in;
  return TDSA_OK;
}

int tdsa_pipe_submit(tdsa_pipe q, size_t n_samples, int hop, int n_frames) {
  if (!q) return fail(TDSA_ERR_ARG, "null pipe");
  auto& sl = q->slots[q->head % q->slots.size()];
  if (!sl.acquired) return fail(TDSA_ERR_STATE, "no acquired slot");
  if (n_samples > q->slot_samples) return fail(TDSA_ERR_ARG, "n_samples=%zu exceeds the slot (%zu)", n_samples, q->slot_samples);
  if (n_frames < 1) return fail(TDSA_ERR_ARG, "n_frames=%d", n_frames);
  tdsa_plan p = q->plan;
  HIPCHK(hipSetDevice(p->device));
  const size_t in_bytes = n_samples * size_t(bytes_per_sample(q->fmt));
  HIPCHK(hipMemcpyAsync(sl.d_in, sl.h_in, in_bytes, hipMemcpyHostToDevice, q->s_in));
  HIPCHK(hipEventRecord(sl.ev_h2d, q->s_in));
  const int rc = process_dev_impl(p, q->fmt, sl.d_in, n_samples, hop, n_frames, q->rows ? sl.d_out : nullptr,
                                  sl.ev_h2d, sl.ev_done);
  if (rc != TDSA_OK) {
    sl.acquired = false;
    return rc;
  }
  sl.n_frames = p->big ? 1 : n_frames;
  if (q->rows_host) {
    HIPCHK(hipStreamWaitEvent(q->s_out, sl.ev_done, 0));
    HIPCHK(hipMemcpyAsync(sl.h_out, sl.d_out, size_t(sl.n_frames) * p->nfft * sizeof(float), hipMemcpyDeviceToHost,
                          q->s_out));
    HIPCHK(hipEventRecord(sl.ev_d2h, q->s_out));
  }
  if (q->rows_u8) {   // the read-back leg carries what setImage(img, levels) makes of the rows: a quarter of the bytes
    const size_t cnt = size_t(sl.n_frames) * p->nfft;
    HIPCHK(hipStreamWaitEvent(q->s_out, sl.ev_done, 0));
    HIPCHK(launch_quantize_u8(sl.d_out, sl.d_u8, cnt, q->lo_db, q->hi_db, q->s_out));
    HIPCHK(hipMemcpyAsync(sl.h_u8, sl.d_u8, cnt, hipMemcpyDeviceToHost, q->s_out));
    HIPCHK(hipEventRecord(sl.ev_d2h, q->s_out));
  }
  sl.acquired = false;
  sl.in_flight = true;
  ++q->head;
  ++q->pending;
  return TDSA_OK;
}

static int pipe_collect(tdsa_pipe q, const float** rows_host, const float** rows_dev, int* n_frames,
                        const uint8_t** rows_u8 = nullptr) {
  if (!q) return fail(TDSA_ERR_ARG, "null pipe");
  if (q->pending == 0) return fail(TDSA_ERR_STATE, "nothing submitted");
  auto& sl = q->slots[q->tail % q->slots.size()];
  HIPCHK(hipSetDevice(q->plan->device));
  HIPCHK(hipEventSynchronize(q->rows_host || q->rows_u8 ? sl.ev_d2h : sl.ev_done));
  sl.in_flight = false;
  if (rows_u8) *rows_u8 = sl.h_u8;
  if (rows_host) *rows_host = q->rows_host ? sl.h_out : nullptr;
  if (rows_dev) *rows_dev = q->rows ? sl.d_out : nullptr;
  if (n_frames) *n_frames = sl.n_frames;
  ++q->tail;
  --q->pending;
  return TDSA_OK;
}

int tdsa_pipe_collect(tdsa_pipe q, const float** rows_host, int* n_frames) {
  return pipe_collect(q, rows_host, nullptr, n_frames);
}

int tdsa_pipe_collect_dev(tdsa_pipe q, const float** rows_dev, int* n_frames) {
  if (q && !q->rows) return fail(TDSA_ERR_STATE, "this pipe keeps no dB rows (want_rows = 0)");
  return pipe_collect(q, nullptr, rows_dev, n_frames);
}

int tdsa_pipe_collect_u8(tdsa_pipe q, const uint8_t** rows_host, int* n_frames) {
  if (q && !q->rows_u8) return fail(TDSA_ERR_STATE, "this pipe reads no byte rows back (want_rows != 3)");
  return pipe_collect(q, nullptr, nullptr, n_frames, rows_host);
}

int tdsa_pipe_set_levels(tdsa_pipe q, float min_db, float max_db) {
  if (!q) return fail(TDSA_ERR_ARG, "null pipe");
  if (!(max_db > min_db)) return fail(TDSA_ERR_ARG, "levels (%g, %g): need max > min", double(min_db), double(max_db));
  q->lo_db = min_db;      // slots submitted from now on
  q->hi_db = max_db;
  return TDSA_OK;
}

int tdsa_pipe_pending(tdsa_pipe q, int* pending) {
  if (!q || !pending) return fail(TDSA_ERR_ARG, "null argument");
  *pending = q->pending;
  return TDSA_OK;
}

// ================================================================================================
// Trace analytics and display accumulators on device-resident dB rows
// ================================================================================================
namespace {

// scratch that grows on demand, owned by the plan (freed by tdsa_destroy; one plan = one thread at a time)
int plan_scratch(tdsa_plan p, size_t need) {
  if (need <= p->scratch_bytes) return TDSA_OK;
  HIPCHK(hipStreamSynchronize(p->stream));
  if (p->d_scratch) HIPCHK(hipFree(p->d_scratch));
  p->d_scratch = nullptr;
  p->scratch_bytes = 0;
  HIPCHK(hipMalloc(&p->d_scratch, need));
  p->scratch_bytes = need;
  return TDSA_OK;
}

// Where the result arrays of a rows_* call are formed and how they reach the caller's (pageable) arrays: every copy
// into pageable memory costs ~10 us in the runtime's own staging, so the arrays of a call sit back to back in ONE
// region - the plan's pinned, device-visible buffer itself when they are a few KB (one displayed row per GUI tick:
// the kernel stores over the bus, the host only waits), else device scratch and one DMA into the pinned buffer -
// and leave it by memcpy.  Results too large for the bounce buffer go piece by piece as before.
constexpr size_t kResultsDirectMax = 4096;
struct RowsResults {
  char* base = nullptr;      // where the kernel writes (device-visible)
  const char* host = nullptr;   // where the host reads after fetch(): the pinned buffer, or null = piece by piece
  size_t total = 0;
};
int rows_results_begin(tdsa_plan p, size_t total, RowsResults* r) {
  r->total = total;
  if (total <= kPinnedBounceMax) {
    const int rc = ensure_pins(p, 0, total);
    if (rc != TDSA_OK) return rc;
    r->host = static_cast<const char*>(p->h_out_pin);
  }
  if (total <= kResultsDirectMax) {
    r->base = static_cast<char*>(p->h_out_pin);
    return TDSA_OK;
  }
  const int rc = plan_scratch(p, total);
  if (rc != TDSA_OK) return rc;
  r->base = static_cast<char*>(p->d_scratch);
  return TDSA_OK;
}
// after the launch: wait; afterwards piece(off) is readable on the host (r.host != null)
int rows_results_fetch(tdsa_plan p, const RowsResults& r) {
  if (r.host && r.base != r.host)
    HIPCHK(hipMemcpyAsync(p->h_out_pin, r.base, r.total, hipMemcpyDeviceToHost, p->stream));
  if (r.host) HIPCHK(hipStreamSynchronize(p->stream));
  return TDSA_OK;
}
int rows_results_piece(tdsa_plan p, const RowsResults& r, void* dst_host, size_t off, size_t bytes) {
  if (!dst_host || bytes == 0) return TDSA_OK;
  if (r.host) std::memcpy(dst_host, r.host + off, bytes);
  else HIPCHK(hipMemcpyAsync(dst_host, r.base + off, bytes, hipMemcpyDeviceToHost, p->stream));
  return TDSA_OK;
}

}  // namespace

int tdsa_set_frame_stats(tdsa_plan p, int enable, int band_lo, int band_hi) {
  if (!p) return fail(TDSA_ERR_ARG, "null plan");
  if (enable && p->big) return fail(TDSA_ERR_STATE, "long-frame plans return one row per call: take tdsa_rows_stats of it");
  if (enable && band_lo <= band_hi && (band_lo < 0 || band_hi >= p->nfft))
    return fail(TDSA_ERR_ARG, "band [%d, %d] outside [0, %d)", band_lo, band_hi, p->nfft);
  HIPCHK(hipSetDevice(p->device));
  JOIN(p);
  p->fs_on = enable != 0;
  p->fs_lo = band_lo;
  p->fs_hi = band_hi;
  HIPCHK(hipStreamSynchronize(p->stream));                   // (joined: everything in flight is behind the main stream)
  for (auto& per_stream : p->fs)
    for (auto& sl : per_stream) sl.state = 0;
  for (auto& h : p->fs_hist) h = nullptr;
  p->fs_seq = 0;
  return TDSA_OK;
}

int tdsa_get_frame_stats(tdsa_plan p, int calls_back, int capacity, int* n_frames, float* peak_db_host,
                         int32_t* peak_bin_host, double* band_lin_host) {
  if (!p) return fail(TDSA_ERR_ARG, "null plan");
  if (!p->fs_on) return fail(TDSA_ERR_STATE, "tdsa_set_frame_stats has not enabled the per-frame scalars");
  if (calls_back < 0 || calls_back >= tdsa_plan_s::kFsKeep || (unsigned long long)calls_back >= p->fs_seq)
    return fail(TDSA_ERR_ARG, "calls_back=%d: the results of the last %d calls are kept, %llu made", calls_back,
                tdsa_plan_s::kFsKeep, p->fs_seq);
  tdsa_plan_s::FsSlot& sl = *p->fs_hist[(p->fs_seq - 1 - calls_back) % tdsa_plan_s::kFsKeep];
  if (sl.state == 2) return fail(TDSA_ERR_STATE, "that call wrote no dB rows and its plan / mode has no fused statistics");
  if (sl.state != 1) return fail(TDSA_ERR_STATE, "no statistics in that slot");
  if (n_frames) *n_frames = sl.n_frames;
  if (capacity < sl.n_frames && (peak_db_host || peak_bin_host || band_lin_host))
    return fail(TDSA_ERR_ARG, "capacity %d < %d frames", capacity, sl.n_frames);
  HIPCHK(hipSetDevice(p->device));
  HIPCHK(hipStreamSynchronize(sl.stream));                   // the call's own stream (later calls on other streams stay in flight)
  if (!p->fs_stream) HIPCHK(hipStreamCreateWithFlags(&p->fs_stream, hipStreamNonBlocking));
  if (sl.pending) {
    HIPCHK(launch_frame_stats_finish(sl.d_part, sl.n_frames, sl.wpf, sl.cal_lin, sl.d_peak, sl.d_bin, sl.d_band,
                                     p->fs_stream));
    sl.pending = false;
  }
  const size_t nf = size_t(sl.n_frames);
  if (nf * 16 <= kPinnedBounceMax) {       // DMA into the pinned bounce buffer, one wait, memcpy out (pageable targets cost ~10 us each)
    const int rc = ensure_pins(p, 0, nf * 16);
    if (rc != TDSA_OK) return rc;
    char* pin = static_cast<char*>(p->h_out_pin);
    if (band_lin_host) HIPCHK(hipMemcpyAsync(pin, sl.d_band, nf * sizeof(double), hipMemcpyDeviceToHost, p->fs_stream));
    if (peak_db_host) HIPCHK(hipMemcpyAsync(pin + nf * 8, sl.d_peak, nf * sizeof(float), hipMemcpyDeviceToHost, p->fs_stream));
    if (peak_bin_host) HIPCHK(hipMemcpyAsync(pin + nf * 12, sl.d_bin, nf * sizeof(int), hipMemcpyDeviceToHost, p->fs_stream));
    HIPCHK(hipStreamSynchronize(p->fs_stream));
    if (band_lin_host) std::memcpy(band_lin_host, pin, nf * sizeof(double));
    if (peak_db_host) std::memcpy(peak_db_host, pin + nf * 8, nf * sizeof(float));
    if (peak_bin_host) std::memcpy(peak_bin_host, pin + nf * 12, nf * sizeof(int));
    return TDSA_OK;
  }
  if (peak_db_host) HIPCHK(hipMemcpyAsync(peak_db_host, sl.d_peak, nf * sizeof(float), hipMemcpyDeviceToHost, p->fs_stream));
  if (peak_bin_host) HIPCHK(hipMemcpyAsync(peak_bin_host, sl.d_bin, nf * sizeof(int), hipMemcpyDeviceToHost, p->fs_stream));
  if (band_lin_host) HIPCHK(hipMemcpyAsync(band_lin_host, sl.d_band, nf * sizeof(double), hipMemcpyDeviceToHost, p->fs_stream));
  HIPCHK(hipStreamSynchronize(p->fs_stream));
  return TDSA_OK;
}

int tdsa_rows_stats(tdsa_plan p, const float* rows_dev, int n_rows, int n_bins, int band_lo, int band_hi,
                    double bin_width, float* peak_db_host, int32_t* peak_bin_host, double* band_db_host) {
  if (!p) return fail(TDSA_ERR_ARG, "null plan");
  if (n_rows == 0) return TDSA_OK;
  if (!rows_dev || n_rows < 0 || n_bins < 1) return fail(TDSA_ERR_ARG, "bad rows (%p, %d x %d)", (const void*)rows_dev, n_rows, n_bins);
  if (band_db_host && (band_lo < 0 || band_hi >= n_bins) && band_lo <= band_hi)
    return fail(TDSA_ERR_ARG, "band [%d, %d] outside [0, %d)", band_lo, band_hi, n_bins);
  HIPCHK(hipSetDevice(p->device));
  JOIN(p);
  const size_t nr = size_t(n_rows), o_peak = nr * sizeof(double), o_bin = o_peak + nr * sizeof(float);
  RowsResults res;
  int rc = rows_results_begin(p, o_bin + nr * sizeof(int), &res);
  if (rc != TDSA_OK) return rc;
  double* d_band = reinterpret_cast<double*>(res.base);
  float* d_peak = reinterpret_cast<float*>(res.base + o_peak);
  int* d_bin = reinterpret_cast<int*>(res.base + o_bin);
  HIPCHK(launch_rows_stats(rows_dev, n_rows, n_bins, band_lo, band_hi, bin_width, d_peak, d_bin,
                           band_db_host ? d_band : nullptr, p->stream));
  if ((rc = rows_results_fetch(p, res)) != TDSA_OK) return rc;
  if ((rc = rows_results_piece(p, res, peak_db_host, o_peak, nr * sizeof(float))) != TDSA_OK) return rc;
  if ((rc = rows_results_piece(p, res, peak_bin_host, o_bin, nr * sizeof(int))) != TDSA_OK) return rc;
  if ((rc = rows_results_piece(p, res, band_db_host, 0, nr * sizeof(double))) != TDSA_OK) return rc;
  if (!res.host) HIPCHK(hipStreamSynchronize(p->stream));
  return TDSA_OK;
}

int tdsa_rows_top_peaks(tdsa_plan p, const float* rows_dev, int n_rows, int n_bins, int n_peaks, int min_sep_bins,
                        float min_excursion_db, int32_t* peak_bins_host, float* peak_db_host) {
  if (!p) return fail(TDSA_ERR_ARG, "null plan");
  if (n_rows == 0) return TDSA_OK;
  if (!rows_dev || !peak_bins_host || n_rows < 0) return fail(TDSA_ERR_ARG, "null / negative argument");
  if (n_peaks < 1 || n_peaks > 8) return fail(TDSA_ERR_ARG, "n_peaks=%d outside [1, 8]", n_peaks);
  if (n_bins < 1 || n_bins > 16384) return fail(TDSA_ERR_ARG, "n_bins=%d outside [1, 16384] (row must fit the LDS)", n_bins);
  HIPCHK(hipSetDevice(p->device));
  JOIN(p);
  const size_t cnt = size_t(n_rows) * n_peaks;
  RowsResults res;
  int rc = rows_results_begin(p, cnt * (sizeof(int) + sizeof(float)), &res);
  if (rc != TDSA_OK) return rc;
  int* d_bins = reinterpret_cast<int*>(res.base);
  float* d_db = reinterpret_cast<float*>(res.base + cnt * sizeof(int));
  HIPCHK(launch_top_peaks(rows_dev, n_rows, n_bins, n_peaks, min_sep_bins, min_excursion_db, d_bins, d_db, p->stream));
  if ((rc = rows_results_fetch(p, res)) != TDSA_OK) return rc;
  if ((rc = rows_results_piece(p, res, peak_bins_host, 0, cnt * sizeof(int))) != TDSA_OK) return rc;
  if ((rc = rows_results_piece(p, res, peak_db_host, cnt * sizeof(int), cnt * sizeof(float))) != TDSA_OK) return rc;
  if (!res.host) HIPCHK(hipStreamSynchronize(p->stream));
  return TDSA_OK;
}

int tdsa_rows_marker_peaks(tdsa_plan p, const float* rows_dev, int n_rows, int n_bins, double height, double prominence,
                           int distance, int current_idx, int max_list, int32_t* n_peaks_host, int32_t* snap_bin_host,
                           int32_t* next_bin_host, int32_t* peak_bins_host, double* peak_prom_host) {
  if (!p) return fail(TDSA_ERR_ARG, "null plan");
  if (n_rows == 0) return TDSA_OK;
  if (!rows_dev || n_rows < 0) return fail(TDSA_ERR_ARG, "null / negative argument");
  if (n_bins < 1 || n_bins > 16384) return fail(TDSA_ERR_ARG, "n_bins=%d outside [1, 16384] (row must fit the LDS)", n_bins);
  if (distance < 1) return fail(TDSA_ERR_ARG, "distance=%d (scipy: `distance` must be greater or equal to 1)", distance);
  if (max_list < 0 || (max_list > 0 && !peak_bins_host)) return fail(TDSA_ERR_ARG, "max_list=%d without a list buffer", max_list);
  if (height != height || prominence != prominence) return fail(TDSA_ERR_ARG, "NaN height / prominence");
  HIPCHK(hipSetDevice(p->device));
  JOIN(p);
  const size_t cnt = size_t(n_rows) * max_list, rb = size_t(n_rows) * sizeof(int);
  const size_t o_bins = cnt * sizeof(double), o_count = o_bins + cnt * sizeof(int), o_snap = o_count + rb, o_next = o_snap + rb;
  RowsResults res;
  int rc = rows_results_begin(p, o_next + rb, &res);
  if (rc != TDSA_OK) return rc;
  double* d_prom = reinterpret_cast<double*>(res.base);
  int* d_bins = reinterpret_cast<int*>(res.base + o_bins);
  int* d_count = reinterpret_cast<int*>(res.base + o_count);
  int* d_snap = reinterpret_cast<int*>(res.base + o_snap);
  int* d_next = reinterpret_cast<int*>(res.base + o_next);
  HIPCHK(launch_marker_peaks(rows_dev, n_rows, n_bins, height, prominence, distance, current_idx, max_list, d_count, d_snap,
                             d_next, max_list > 0 ? d_bins : nullptr, max_list > 0 && peak_prom_host ? d_prom : nullptr,
                             p->stream));
  if ((rc = rows_results_fetch(p, res)) != TDSA_OK) return rc;
  if ((rc = rows_results_piece(p, res, n_peaks_host, o_count, rb)) != TDSA_OK) return rc;
  if ((rc = rows_results_piece(p, res, snap_bin_host, o_snap, rb)) != TDSA_OK) return rc;
  if ((rc = rows_results_piece(p, res, next_bin_host, o_next, rb)) != TDSA_OK) return rc;
  if (max_list > 0) {
    if ((rc = rows_results_piece(p, res, peak_bins_host, o_bins, cnt * sizeof(int))) != TDSA_OK) return rc;
    if ((rc = rows_results_piece(p, res, peak_prom_host, 0, cnt * sizeof(double))) != TDSA_OK) return rc;
  }
  if (!res.host) HIPCHK(hipStreamSynchronize(p->stream));
  return TDSA_OK;
}

// ---- density histogram --------------------------------------------------------------------------
struct tdsa_density_s {
  int device = 0, n = 0;
  float decay = 0.96f;
  float* d_hist = nullptr;     // [n][512]
  float* d_img = nullptr;      // log1p image scratch
  unsigned char* d_u8 = nullptr;   // the image as bytes (+ 8 bytes: its min / max)
  float* h_row[2] = {nullptr, nullptr};   // pinned, device-visible staging of host rows (the kernel reads them in place)
  hipEvent_t ev_row[2] = {nullptr, nullptr};   // ... free again when the update that read them has run
  unsigned tick = 0;
  hipStream_t stream = nullptr;
};

int tdsa_density_create(int device_id, int n_bins, float decay, tdsa_density* out) {
  if (!out) return fail(TDSA_ERR_ARG, "null out");
  if (n_bins < 1) return fail(TDSA_ERR_ARG, "n_bins=%d", n_bins);
  HIPCHK(hipSetDevice(device_id));
  tdsa_density d = new (std::nothrow) tdsa_density_s();
  if (!d) return fail(TDSA_ERR_NOMEM, "out of host memory");
  d->device = device_id;
  d->n = n_bins;
  d->decay = decay;
  const size_t hb = size_t(n_bins) * 512 * sizeof(float);
  hipError_t e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc(&d->d_hist, hb);
  for (int k = 0; k < 2; ++k) {
    if (e == hipSuccess)
      e = hipHostMalloc(reinterpret_cast<void**>(&d->h_row[k]), size_t(n_bins) * sizeof(float),
                        hipHostMallocPortable | hipHostMallocMapped);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&d->ev_row[k], hipEventDisableTiming);
  }
  if (e == hipSuccess) e = hipMemsetAsync(d->d_hist, 0, hb, d->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
  if (e != hipSuccess) {
    (void)tdsa_density_destroy(d);
    return fail(TDSA_ERR_HIP, "density create: %s", hipGetErrorString(e));
  }
  *out = d;
  return TDSA_OK;
}

int tdsa_density_destroy(tdsa_density d) {
  if (!d) return TDSA_OK;
  (void)hipSetDevice(d->device);
  if (d->stream) (void)hipStreamSynchronize(d->stream);
  if (d->d_hist) (void)hipFree(d->d_hist);
  if (d->d_img) (void)hipFree(d->d_img);
  if (d->d_u8) (void)hipFree(d->d_u8);
  for (int k = 0; k < 2; ++k) {
    if (d->h_row[k]) (void)hipHostFree(d->h_row[k]);
    if (d->ev_row[k]) (void)hipEventDestroy(d->ev_row[k]);
  }
  if (d->stream) (void)hipStreamDestroy(d->stream);
  delete d;
  return TDSA_OK;
}

int tdsa_density_set_decay(tdsa_density d, float decay) {
  if (!d) return fail(TDSA_ERR_ARG, "null density");
  d->decay = decay;
  return TDSA_OK;
}

int tdsa_density_reset(tdsa_density d) {
  if (!d) return fail(TDSA_ERR_ARG, "null density");
  HIPCHK(hipSetDevice(d->device));
  HIPCHK(hipMemsetAsync(d->d_hist, 0, size_t(d->n) * 512 * sizeof(float), d->stream));
  HIPCHK(hipStreamSynchronize(d->stream));
  return TDSA_OK;
}

// rows produced on plan p's stream (p may be NULL when the rows are otherwise known to be complete)
int tdsa_density_update_dev(tdsa_density d, tdsa_plan p, const float* rows_dev, int n_rows) {
  if (!d) return fail(TDSA_ERR_ARG, "null density");
  if (n_rows == 0) return TDSA_OK;
  if (!rows_dev || n_rows < 0) return fail(TDSA_ERR_ARG, "bad rows");
  if (p && (p->device != d->device)) return fail(TDSA_ERR_ARG, "plan and histogram live on different devices");
  HIPCHK(hipSetDevice(d->device));
  if (p) {   // order after the producer: the plan's stream signals, ours waits
    JOIN(p);
    HIPCHK(hipEventRecord(p->ev_state, p->stream));
    HIPCHK(hipStreamWaitEvent(d->stream, p->ev_state, 0));
  }
  HIPCHK(launch_density(rows_dev, n_rows, d->n, d->decay, d->d_hist, d->stream));
  HIPCHK(hipStreamSynchronize(d->stream));
  return TDSA_OK;
}

int tdsa_density_update(tdsa_density d, const float* row_host, int n) {
  if (!d || !row_host) return fail(TDSA_ERR_ARG, "null argument");
  if (n != d->n) return fail(TDSA_ERR_ARG, "row of %d bins, histogram has %d (re-create it: _ensure_hist)", n, d->n);
  HIPCHK(hipSetDevice(d->device));
  // the per-tick call returns when the row is staged and its update queued (every other entry point is ordered behind
  // it on the histogram's stream; tdsa_density_read waits): two pinned rows the kernel reads in place, each free again
  // once the update that read it has run
  const unsigned k = d->tick++ & 1u;
  HIPCHK(hipEventSynchronize(d->ev_row[k]));
  std::memcpy(d->h_row[k], row_host, size_t(n) * sizeof(float));
  HIPCHK(launch_density(d->h_row[k], 1, d->n, d->decay, d->d_hist, d->stream));
  HIPCHK(hipEventRecord(d->ev_row[k], d->stream));
  return TDSA_OK;
}

int tdsa_density_read(tdsa_density d, float* hist_host, int as_log1p) {
  if (!d || !hist_host) return fail(TDSA_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(d->device));
  const size_t cnt = size_t(d->n) * 512;
  const float* src = d->d_hist;
  if (as_log1p) {
    if (!d->d_img) HIPCHK(hipMalloc(&d->d_img, cnt * sizeof(float)));
    HIPCHK(launch_log1p(d->d_hist, d->d_img, cnt, d->stream));
    src = d->d_img;
  }
  HIPCHK(hipMemcpyAsync(hist_host, src, cnt * sizeof(float), hipMemcpyDeviceToHost, d->stream));
  HIPCHK(hipStreamSynchronize(d->stream));
  return TDSA_OK;
}

int tdsa_density_read_u8(tdsa_density d, uint8_t* img_host, float* levels2) {
  if (!d || !img_host) return fail(TDSA_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(d->device));
  const size_t cnt = size_t(d->n) * 512;
  if (!d->d_img) HIPCHK(hipMalloc(&d->d_img, cnt * sizeof(float)));
  if (!d->d_u8) HIPCHK(hipMalloc(&d->d_u8, cnt + 16));
  unsigned* d_mm = reinterpret_cast<unsigned*>(d->d_u8 + ((cnt + 7) & ~size_t(7)));
  HIPCHK(launch_log1p(d->d_hist, d->d_img, cnt, d->stream));
  HIPCHK(launch_minmax_pos(d->d_img, cnt, d_mm, d->stream));
  float mm[2];
  HIPCHK(hipMemcpyAsync(mm, d_mm, sizeof(mm), hipMemcpyDeviceToHost, d->stream));
  HIPCHK(hipStreamSynchronize(d->stream));
  if (levels2) { levels2[0] = mm[0]; levels2[1] = mm[1]; }
  if (mm[1] > mm[0]) {
    HIPCHK(launch_quantize_u8(d->d_img, d->d_u8, cnt, mm[0], mm[1], d->stream));
  } else {
    HIPCHK(hipMemsetAsync(d->d_u8, 0, cnt, d->stream));      // a flat image (an empty histogram): every pixel at the lower level
  }
  HIPCHK(hipMemcpyAsync(img_host, d->d_u8, cnt, hipMemcpyDeviceToHost, d->stream));
  HIPCHK(hipStreamSynchronize(d->stream));
  return TDSA_OK;
}

// ---- waterfall ring -----------------------------------------------------------------------------
struct tdsa_waterfall_s {
  int device = 0, n = 0, history = 0;
  int ptr = 0;
  bool have_last = false;
  float* d_ring = nullptr;     // [history][n]: every line once, the view is two copies
  float* d_last = nullptr;     // [n] Waterfall._last_row
  unsigned char* d_u8 = nullptr;   // [history][n] the view as bytes (tdsa_waterfall_view_u8)
  float* h_row = nullptr;      // pinned staging of a host row: small rows are read in place by the kernels,
  float* d_row = nullptr;      // larger ones take one DMA into d_row first (the scatter reads every bin)
  int* d_flags = nullptr;      // [2][cap] differs, destination line per pushed row
  int* d_info = nullptr;       // {new rows, last new row} of the push in flight, for the scatter
  int* h_info = nullptr;       // the same two words, pinned: what the host waits for
  int cap = 0;
  hipStream_t stream = nullptr;
};

int tdsa_waterfall_create(int device_id, int history_lines, int n_bins, float min_db, tdsa_waterfall* out) {
  if (!out) return fail(TDSA_ERR_ARG, "null out");
  if (history_lines < 1 || n_bins < 1) return fail(TDSA_ERR_ARG, "history=%d n_bins=%d", history_lines, n_bins);
  HIPCHK(hipSetDevice(device_id));
  tdsa_waterfall w = new (std::nothrow) tdsa_waterfall_s();
  if (!w) return fail(TDSA_ERR_NOMEM, "out of host memory");
  w->device = device_id;
  w->n = n_bins;
  w->history = history_lines;
  const size_t cnt = size_t(history_lines) * n_bins;
  hipError_t e = hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc(&w->d_ring, cnt * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&w->d_info, 2 * sizeof(int));
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&w->h_info), 2 * sizeof(int), hipHostMallocDefault);
  if (e == hipSuccess) e = hipMalloc(&w->d_last, size_t(n_bins) * sizeof(float));
  if (e == hipSuccess)
    e = hipHostMalloc(reinterpret_cast<void**>(&w->h_row), size_t(n_bins) * sizeof(float),
                      hipHostMallocPortable | hipHostMallocMapped);
  if (e == hipSuccess) e = hipMalloc(&w->d_row, size_t(n_bins) * sizeof(float));
  if (e == hipSuccess) e = launch_fill(w->d_ring, cnt, min_db, w->stream);    // np.full((2H, W), wf_min_db)
  if (e == hipSuccess) e = hipStreamSynchronize(w->stream);
  if (e != hipSuccess) {
    (void)tdsa_waterfall_destroy(w);
    return fail(TDSA_ERR_HIP, "waterfall create: %s", hipGetErrorString(e));
  }
  *out = w;
  return TDSA_OK;
}

int tdsa_waterfall_destroy(tdsa_waterfall w) {
  if (!w) return TDSA_OK;
  (void)hipSetDevice(w->device);
  if (w->stream) (void)hipStreamSynchronize(w->stream);
  if (w->d_ring) (void)hipFree(w->d_ring);
  if (w->d_last) (void)hipFree(w->d_last);
  if (w->d_u8) (void)hipFree(w->d_u8);
  if (w->h_row) (void)hipHostFree(w->h_row);
  if (w->d_row) (void)hipFree(w->d_row);
  if (w->d_flags) (void)hipFree(w->d_flags);
  if (w->d_info) (void)hipFree(w->d_info);
  if (w->h_info) (void)hipHostFree(w->h_info);
  if (w->stream) (void)hipStreamDestroy(w->stream);
  delete w;
  return TDSA_OK;
}

static int waterfall_push_rows(tdsa_waterfall w, const float* rows_dev, int n_rows, int* n_new) {
  if (n_rows > w->cap) {
    if (w->d_flags) HIPCHK(hipFree(w->d_flags));
    w->d_flags = nullptr;
    w->cap = 0;
    HIPCHK(hipMalloc(&w->d_flags, size_t(2) * n_rows * sizeof(int)));
    w->cap = n_rows;
  }
  // new-row flags, the pointer walk of _add_row as a scan over them, the scatter: three launches, one wait
  w->h_info[0] = 0;
  w->h_info[1] = -1;
  HIPCHK(launch_waterfall_push(rows_dev, n_rows, w->n, w->have_last ? 1 : 0, w->ptr, w->history, w->d_flags,
                               w->d_flags + w->cap, w->d_info, w->h_info, w->d_ring, w->d_last, w->stream));
  HIPCHK(hipStreamSynchronize(w->stream));
  const int fresh = w->h_info[0];
  if (fresh > 0) {
    w->ptr = ((w->ptr - fresh % w->history) % w->history + w->history) % w->history;
    w->have_last = true;
  }
  if (n_new) *n_new = fresh;
  return TDSA_OK;
}

int tdsa_waterfall_push_dev(tdsa_waterfall w, tdsa_plan p, const float* rows_dev, int n_rows, int* n_new) {
  if (!w) return fail(TDSA_ERR_ARG, "null waterfall");
  if (n_new) *n_new = 0;
  if (n_rows == 0) return TDSA_OK;
  if (!rows_dev || n_rows < 0) return fail(TDSA_ERR_ARG, "bad rows");
  if (p && p->device != w->device) return fail(TDSA_ERR_ARG, "plan and waterfall live on different devices");
  HIPCHK(hipSetDevice(w->device));
  if (p) {
    JOIN(p);
    HIPCHK(hipEventRecord(p->ev_state, p->stream));
    HIPCHK(hipStreamWaitEvent(w->stream, p->ev_state, 0));
  }
  return waterfall_push_rows(w, rows_dev, n_rows, n_new);
}

int tdsa_waterfall_push(tdsa_waterfall w, const float* row_host, int n, int* is_new) {
  if (!w || !row_host) return fail(TDSA_ERR_ARG, "null argument");
  if (n != w->n) return fail(TDSA_ERR_ARG, "row of %d bins, ring has %d", n, w->n);
  HIPCHK(hipSetDevice(w->device));
  std::memcpy(w->h_row, row_host, size_t(n) * sizeof(float));     // (every push ends with a wait: the row is free)
  if (n <= 4096) return waterfall_push_rows(w, w->h_row, 1, is_new);
  HIPCHK(hipMemcpyAsync(w->d_row, w->h_row, size_t(n) * sizeof(float), hipMemcpyHostToDevice, w->stream));
  return waterfall_push_rows(w, w->d_row, 1, is_new);
}

int tdsa_waterfall_view(tdsa_waterfall w, float* view_host, int* ptr) {
  if (!w) return fail(TDSA_ERR_ARG, "null waterfall");
  HIPCHK(hipSetDevice(w->device));
  if (view_host) {   // _display_view: buf[ptr : ptr + H], newest row first
    const size_t head = size_t(w->history - w->ptr) * w->n;      // lines ptr ... H-1, then 0 ... ptr-1
    HIPCHK(hipMemcpyAsync(view_host, w->d_ring + size_t(w->ptr) * w->n, head * sizeof(float), hipMemcpyDeviceToHost,
                          w->stream));
    if (w->ptr > 0)
      HIPCHK(hipMemcpyAsync(view_host + head, w->d_ring, size_t(w->ptr) * w->n * sizeof(float), hipMemcpyDeviceToHost,
                            w->stream));
    HIPCHK(hipStreamSynchronize(w->stream));
  }
  if (ptr) *ptr = w->ptr;
  return TDSA_OK;
}

int tdsa_waterfall_view_u8(tdsa_waterfall w, float min_db, float max_db, uint8_t* view_host) {
  if (!w || !view_host) return fail(TDSA_ERR_ARG, "null argument");
  if (!(max_db > min_db)) return fail(TDSA_ERR_ARG, "levels (%g, %g): need max > min", double(min_db), double(max_db));
  HIPCHK(hipSetDevice(w->device));
  const size_t cnt = size_t(w->history) * w->n;
  if (!w->d_u8) HIPCHK(hipMalloc(&w->d_u8, cnt));
  const size_t head = size_t(w->history - w->ptr) * w->n;
  HIPCHK(launch_quantize_u8(w->d_ring + size_t(w->ptr) * w->n, w->d_u8, head, min_db, max_db, w->stream));
  HIPCHK(launch_quantize_u8(w->d_ring, w->d_u8 + head, cnt - head, min_db, max_db, w->stream));
  HIPCHK(hipMemcpyAsync(view_host, w->d_u8, cnt, hipMemcpyDeviceToHost, w->stream));
  HIPCHK(hipStreamSynchronize(w->stream));
  return TDSA_OK;
}
