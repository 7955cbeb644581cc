/*
 * tdsa_hip.h - C-ABI of libtdsa_hip.so, the MI355X (gfx950) IQ -> spectrum engine.
 *
 * This is the drop-in boundary for the reference's "sample method" hot path
 * (CWNE88/topdogspectrumanalyser).  The reference is pure Python and reaches native code only
 * through numpy.fft / scipy.fft; a maintainer binds this library with ctypes.
 * Every entry point names the reference code it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every function returns int: 0 = TDSA_OK, negative = error; tdsa_last_error_string() gives text
 *   - no C++ exception and no torch/numpy type crosses this ABI: plain pointers and sizes only
 *   - the caller owns every host buffer; a plan owns its device buffers, HIP stream and events
 *   - one plan is used by one host thread at a time (the reference serialises the path under
 *     HackrfSamplesDataSource._lock, datasources/hackrf_samples.py:50,341)
 *   - "samples" always counts complex samples (one I,Q pair)
 *   - spectra are returned fftshift-ed (DC at index N/2) exactly as
 *     np.fft.fftshift(np.fft.fft(x)) orders them (datasources/hackrf_samples.py:370)
 */
#ifndef TDSA_HIP_H
#define TDSA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDSA_OK 0
#define TDSA_ERR_ARG (-1)      /* bad argument (size, mode, null pointer) */
#define TDSA_ERR_HIP (-2)      /* a HIP runtime call failed               */
#define TDSA_ERR_STATE (-3)    /* call sequence error (no window set ...) */
#define TDSA_ERR_NOMEM (-4)

#define TDSA_VERSION 100

typedef struct tdsa_plan_s* tdsa_plan;

/* input sample formats (the unpack pyhackrf / pyrtlsdr do on the host) */
#define TDSA_IN_I8 0   /* interleaved int8  I,Q ; x = (I + jQ)/128          (HackRF / build contract) */
#define TDSA_IN_U8 1   /* interleaved uint8 I,Q ; x = (u/127.5 - 1) pairs   (pyrtlsdr convention)     */
#define TDSA_IN_C64 2  /* interleaved float32 re,im (numpy complex64)                               */

/* dB conversion (utils/constants.py:152-155 floors are passed in tdsa_mode.log_floor) */
#define TDSA_DB_MAG 0  /* 20*log10(|X| + floor)            datasources/hackrf_samples.py:382-383 */
#define TDSA_DB_POW 1  /* 10*log10(|X|^2 * scale + floor)  hackrf_samples.py:374-381, rtl_samples.py:175-184 */

/* trace averaging (utils/signal_processing.py:5-73) */
#define TDSA_AVG_OFF 0
#define TDSA_AVG_EXP 1
#define TDSA_AVG_LIN 2

/* hold_flags / reset bits */
#define TDSA_HOLD_MAX 1u
#define TDSA_HOLD_MIN 2u
#define TDSA_RESET_AVG 1u
#define TDSA_RESET_HOLD_MAX 2u
#define TDSA_RESET_HOLD_MIN 4u
#define TDSA_RESET_DC 8u
#define TDSA_RESET_TARE 16u
#define TDSA_RESET_ALL 31u

typedef struct tdsa_mode {
  int32_t db_mode;       /* TDSA_DB_MAG | TDSA_DB_POW                                                */
  float power_scale;     /* POW only: 1 or 1/(fs*N) for PSD (hackrf_samples.py:375, rtl_samples.py:177) */
  float log_floor;       /* 1e-12 (LOG_FLOOR) or 1e-10 (POWER_LOG_FLOOR)                               */
  int32_t avg_mode;      /* TDSA_AVG_*  ; a CHANGE of avg_mode / avg_n restarts the average (see below) */
  int32_t avg_n;         /* clamped to >= 1; n <= 1 means pass-through                                 */
  float dc_alpha;        /* < 0: no DC removal (RTL branch); 1: per-frame mean removal; (0,1): tracker
                            dc <- (1-a)*dc + a*mean(x)  (hackrf_samples.py:360-365, :32, :654-657)     */
  float cal_offset_db;   /* added to every dB value (core/display_data_processor.py:317-327)           */
  uint32_t hold_flags;   /* TDSA_HOLD_MAX | TDSA_HOLD_MIN (display_data_processor.py:371-395)          */
} tdsa_mode;

typedef struct tdsa_info {
  int32_t nfft, max_frames, device_id;
  int32_t grid, block, frames_per_block, lds_bytes; /* launch geometry of the frame kernel          */
  int32_t num_cu;
  int64_t frames_held_max, frames_held_min;          /* frames folded into the hold traces           */
  int32_t avg_count;                                 /* TraceAverager._count                          */
  int32_t version;
} tdsa_info;

/* ---- library / device --------------------------------------------------------------------- */
const char* tdsa_last_error_string(void);
int tdsa_version(void);
int tdsa_device_count(int* count);

/* ---- plan lifetime ------------------------------------------------------------------------ */
/* One plan = (device, FFT size, batch capacity).  Replaces _allocate_fft_resources
 * (datasources/hackrf_samples.py:311-324) + the per-source TraceAverager (datasources/base.py:59)
 * + the hold buffers mw.max_power_levels / mw.min_power_levels (main.py:70-105).
 * nfft: ANY size from 2 to 2^20 (HackrfSamplesDataSource.set_num_samples
 * is unbounded and np.fft.fft takes any N, hackrf_samples.py:392-405, :370): powers of two 64 .. 16384 run as ONE
 * LDS-resident kernel; 2^15 .. 2^20 as N1 x 16384 in two passes (in-register column DFT kernel + a 16384-point
 * row pass); sizes 2^a 3^b 5^c as a mixed-radix transform of exactly nfft points (tdsa_smooth.hip: in LDS up to 10 000 points,
 * two passes above);
 * every other size as a chirp-z convolution on the power-of-two kernels, M = 2^ceil(log2(2 nfft - 1))
 * (tdsa_chirp.hip: same modes, state and outputs, one row per frame; M <= 16384 - sizes up to 8192 - as ONE kernel per
 * call, about 5x the time of a native size; larger M through the long-frame kernels; sizes above 2^19 as four half-length
 * sub-convolutions of 2^20 points; fftshift by nfft / 2 as np.fft.fftshift does for odd sizes).  A plan of
 * 2^15 .. 2^20 points (a power of two: "long-frame plan") takes one
 * frame per call, or - with avg_mode lin and avg_n >= the frames seen since the last reset - a batch of K
 * segments whose Welch average comes back as ONE dB row. */
int tdsa_create(int device_id, int nfft, int max_frames, tdsa_plan* out);
int tdsa_destroy(tdsa_plan p);
int tdsa_get_info(tdsa_plan p, tdsa_info* out);

/* Window table, n == nfft float32 values built by the host exactly as the reference builds them
 * (np.hanning(N).astype(f32) / sqrt(mean(w^2)) for HackRF, raw np.hanning/np.hamming/np.ones for RTL;
 * hackrf_samples.py:314-316, rtl_samples.py:199-206).  Applied as x*w before the FFT (:368 / :169). */
int tdsa_set_window(tdsa_plan p, const float* w_host, int n);

/* Replaces set_psd_mode / set_averaging / set_dc_alpha / CalibrationManager.get_offset / hold toggles
 * (datasources/base.py:148-165, hackrf_samples.py:654-657, core/calibration_manager.py:29-31).
 * Changing avg_mode/avg_n resets the averager (TraceAverager.set_mode :19-28); setting the same values again does
 * not (the whole mode is one struct: a new calibration offset must not restart an average) - callers that want the
 * reference's unconditional restart follow up with tdsa_reset_state(TDSA_RESET_AVG), as the Python sources do. */
int tdsa_set_mode(tdsa_plan p, const tdsa_mode* m);

/* reset_averaging (base.py:167), hold clears (core/display_manager.py:139-185), _flush_buffers DC
 * reset (hackrf_samples.py:444-445), _clear_tare. */
int tdsa_reset_state(tdsa_plan p, uint32_t what);

/* Tare baseline in dB (core/display_data_processor.py:329-369): NULL disables.  While set, every
 * dB value has baseline[k] subtracted after the calibration offset and before the hold update. */
int tdsa_set_tare_baseline(tdsa_plan p, const float* baseline_db_host, int n);

/* ---- the hot path -------------------------------------------------------------------------- */
/* Batched get_power_levels(): frame k = iq[k*hop : k*hop + nfft],
 * n_frames frames, each: unpack -> DC removal -> window -> FFT -> fftshift -> |X| / |X|^2 ->
 * (PSD scale) -> (TraceAverager) -> dB(+floor) -> +cal offset -> -tare -> max/min hold.
 * Replaces hackrf_samples.py:357-386 / rtl_samples.py:167-188 per frame and
 * display_data_processor.py:177-181 per frame.
 * Host-pointer variants copy in/out synchronously (out_db_host may be NULL: hold/avg only).  They may be called
 * with ordinary (pageable) memory: calls of up to 256 KiB in + out - one displayed frame - are staged through pinned,
 * device-visible buffers of the plan that the kernels read and write in place (no DMA operation: 20 us per call at
 * 1024 points, 35 us at 16384), calls of up to 1 MiB each way bounce through the same buffers with DMA copies.
 * n_samples >= (n_frames-1)*hop + nfft. */
int tdsa_process_i8(tdsa_plan p, const int8_t* iq_host, size_t n_samples, int hop, int n_frames,
                    float* out_db_host);
int tdsa_process_u8(tdsa_plan p, const uint8_t* iq_host, size_t n_samples, int hop, int n_frames,
                    float* out_db_host);
int tdsa_process_c64(tdsa_plan p, const float* iq_host, size_t n_samples, int hop, int n_frames,
                     float* out_db_host);

/* Device-resident variant: iq_dev / out_db_dev are device pointers on the plan's device; the work
 * is enqueued on the plan's stream and the call returns immediately (tdsa_synchronize to wait).
 * This is what bench.py times (inputs resident in HBM).  out_db_dev may be NULL: the plan's state (hold traces,
 * averager) is still updated; with averaging on and both holds off the call then costs the transforms and a short
 * chain only (the averaged spectrum of a capture - Welch at a native size - without its rows: tdsa_get_avg). */
int tdsa_process_dev(tdsa_plan p, int in_format, const void* iq_dev, size_t n_samples, int hop,
                     int n_frames, float* out_db_dev);

/* Several captures ("segments") of one shape in ONE call - what a host loop over queued chunks of the reader
 * thread (datasources/hackrf_samples.py:254-305 drains its queue chunk by chunk, one get_power_levels() each,
 * :339-386) or over recorded seconds does with tdsa_process_dev, handed over at once: capture s starts at
 * iq_dev + s*seg_stride_bytes (n_samples_per_seg samples, frames_per_seg frames at `hop`, framed on its own -
 * no frame straddles two captures) and its dB rows go to out_db_dev + s*out_seg_stride_floats.
 * Results and plan state (hold traces, DC, averager) are exactly those of n_segments consecutive
 * tdsa_process_dev calls.  Where the captures do not depend on each other (no averaging, no tracked DC
 * remover, LDS-resident frame length) all of them leave as ONE persistent launch, which pays the per-launch
 * costs - cold fetch of window and twiddles, hold merge, the ragged last round of frames over the CUs - once
 * per call instead of once per capture; every other mode runs the captures one after the other inside the
 * call.  frames_per_seg <= max_frames; out_db_dev may be NULL; out_seg_stride_floats = 0 means "captures back to back"
 * (frames_per_seg * nfft), a smaller non-zero stride - rows of different captures would overlap - is TDSA_ERR_ARG. */
int tdsa_process_dev_batch(tdsa_plan p, int in_format, const void* iq_dev, size_t seg_stride_bytes,
                           int n_segments, size_t n_samples_per_seg, int hop, int frames_per_seg,
                           float* out_db_dev, size_t out_seg_stride_floats);

/* Real-input path of MicrophoneSamplesDataSource (datasources/audio_samples.py:121-184): frames of
 * stereo float32 samples [L0,R0,L1,R1,...]; every real signal of a frame (the mono mix, left, right - both for
 * stereo) is transformed on its own as signal + 0i, so a loud channel leaves nothing in a quiet one, as in the
 * reference.  Per frame: mean removal, window, N-point FFT, one-sided power with the
 * non-DC / non-Nyquist bins doubled (:131), PSD scale, TraceAverager, 10*log10(. + floor).
 * channel: TDSA_CH_MONO ((L+R)/2), _LEFT, _RIGHT -> out [n_frames][N/2+1];
 *          TDSA_CH_STEREO -> out [n_frames][2][N/2+1] (left averaged, right not, as :158-171).
 * Uses the plan's window, power_scale, log_floor and averager settings (db_mode must be TDSA_DB_POW).
 * Frame lengths: any size up to 16384, and sizes that are not a power of two up to 2^19 = 524288 points; the
 * native long-frame plans (2^15 ... 2^20) and the split chirp-z plans above 2^19 return TDSA_ERR_ARG before any work
 * is queued (tdsa_real_input_supported tells beforehand). */
#define TDSA_CH_MONO 0
#define TDSA_CH_LEFT 1
#define TDSA_CH_RIGHT 2
#define TDSA_CH_STEREO 3
int tdsa_real_input_supported(int nfft);     /* 1 / 0: has tdsa_process_real2 a path for frames of nfft points */
int tdsa_process_real2(tdsa_plan p, const float* lr_host, size_t n_samples, int hop, int n_frames,
                       int channel, float* out_db_host);

/* Hold traces (mw.max_power_levels / mw.min_power_levels); either pointer may be NULL.
 * *frames_held = number of frames folded in (0: trace is empty, buffer left untouched). */
int tdsa_get_hold(tdsa_plan p, float* max_host, float* min_host, int64_t* frames_held);

/* TraceAverager._buffer (float64 linear power, fftshift-ed) and ._count; used by the host to
 * combine per-GPU Welch partials. */
int tdsa_get_avg(tdsa_plan p, double* avg_linear_host, int* count);

/* ---- Welch partials across GPUs ----------------------------------------------------------------
 * One capture whose K segments are averaged on W GPUs (one plan each, avg lin with avg_n >= K): every plan hands out
 * its running mean of linear power - TraceAverager._buffer after `count` frames, utils/signal_processing.py:56-59 -
 * and ONE plan reassembles the overall mean on its device: mean = sum_r count_r mean_r / sum_r count_r in float64, in
 * rank order (reproducible), which becomes that plan's averager state (count = the total) exactly as if it had seen
 * every segment itself, and the dB row 10*log10(mean*scale + floor) + cal offset - tare (+ hold traces) that
 * get_power_levels + _apply_cal_offset would return (core/display_data_processor.py:317-327).  No collective: the
 * partials travel through host memory the caller owns (bench.py: a POSIX shared-memory slab every rank maps). */
/* Pin a range of caller memory (e.g. a shared-memory mapping) so that copies from / to it are plain DMA. */
int tdsa_host_register(void* host, size_t bytes);
int tdsa_host_unregister(void* host);
/* mean_host receives nfft values, float32 (as_f32 != 0: 4 MiB instead of 8 at 2^20 points; the combined dB row moves by
 * < 1e-6 dB) or float64 (exact); *count = frames behind it (0: nothing averaged, buffer untouched).  Synchronous. */
int tdsa_welch_export(tdsa_plan p, void* mean_host, int as_f32, int* count);
/* parts_host: n_parts (<= 64) partial means as tdsa_welch_export wrote them, part_stride_bytes apart; counts[r] = 0
 * skips part r.  The plan must be in avg lin mode with avg_n >= the total.  The dB row goes to out_db_dev (device
 * pointer, asynchronous) and / or out_db_host (then the call synchronises); either may be NULL. */
int tdsa_welch_combine(tdsa_plan p, const void* parts_host, size_t part_stride_bytes, const int32_t* counts, int n_parts,
                       int as_f32, float* out_db_dev, float* out_db_host);
/* The same exchange without the detour through host memory, for ranks on GPUs of one node: every rank keeps its partial
 * mean in a device buffer the other processes can map (HIP IPC: tdsa_peer_alloc hands out the 64-byte handle, which
 * travels through any host channel; tdsa_peer_open maps it on the combining rank's device, refusing devices without peer
 * access), tdsa_welch_export_dev writes the running mean there (synchronous: the values have landed when it returns) and
 * tdsa_welch_combine_dev reads all partials IN PLACE - its own from HBM, the others' over xGMI, each over its own link -
 * with the arithmetic, state and outputs of tdsa_welch_combine.  Still no collective and no RCCL: one kernel on one
 * device loading through mapped peer pointers.  owner_device_id < 0 skips the peer-access check. */
#define TDSA_PEER_HANDLE_BYTES 64
int tdsa_peer_alloc(int device_id, size_t bytes, void** dev_ptr, unsigned char* handle64);
int tdsa_peer_free(int device_id, void* dev_ptr);
/* hipDeviceCanAccessPeer(device_id -> peer_device_id): what tdsa_peer_open checks, for a launcher that wants to report the
 * node's peer matrix before it picks the exchange */
int tdsa_peer_can_access(int device_id, int peer_device_id, int* can_access);
int tdsa_peer_open(int device_id, const unsigned char* handle64, int owner_device_id, void** dev_ptr);
int tdsa_peer_close(int device_id, void* dev_ptr);
int tdsa_welch_export_dev(tdsa_plan p, void* mean_dev, int as_f32, int* count);
int tdsa_welch_combine_dev(tdsa_plan p, const void* const* parts_dev, const int32_t* counts, int n_parts, int as_f32,
                           float* out_db_dev, float* out_db_host);

/* Saturated VALU rate of the SIMDs right now - independent v_add_f32 chains, four waves per SIMD on every CU, about a
 * millisecond, timed with events on the plan's stream: *ns_per_valu = ns per wave-instruction per SIMD,
 * *shader_mhz = 2 clocks / that (a gfx950 SIMD retires one fp32 wave-instruction per 2 clocks).  bench.py reports it
 * next to the frame kernel's time: the kernel is VALU-issue bound, so kernel time x clock is what compares across boxes. */
int tdsa_shader_clock(tdsa_plan p, float* shader_mhz, float* ns_per_valu);

/* HackrfSamplesDataSource._dc_estimate (complex, units of x). */
int tdsa_get_dc(tdsa_plan p, float* re, float* im);
/* ... and its assignment: the estimate belongs to the source, not to an FFT size - the reference keeps it across
 * set_num_samples (datasources/hackrf_samples.py:392-405), which is a new plan here. */
int tdsa_set_dc(tdsa_plan p, float re, float im);

int tdsa_synchronize(tdsa_plan p);

/* Let consecutive tdsa_process_dev calls overlap on up to n_streams (1..4) HIP streams owned by the plan.
 * A persistent launch ends ragged (2440 frames over 256 CUs = 9 or 10 frames per workgroup); with more
 * than one stream the next call's workgroups start on the CUs that finish first and the inter-kernel
 * gap disappears.  With n_streams >= 3 the overlapped launches are sized for half the CUs, so that two run side by
 * side and the third queues behind them: a workgroup's fixed costs are spread over twice the frames (C3: 76.4 ->
 * 74.3 us per step; strictly serial launches keep the whole chip).  Only calls whose results do not depend on
 * execution order rotate over the extra streams: no averaging, dc_alpha < 0 or >= 1 (hold traces are
 * merged with atomics and stay exact).  Every other entry point first orders the plan's main stream
 * after the work in flight, so the API stays sequentially consistent; with overlap on, tdsa_get_dc
 * reports the last frame of whichever call finished last.  n_streams = 1 (default) restores strictly
 * serial execution.  No counterpart in the reference (its path is one frame per 20 ms timer tick,
 * core/ui_setup.py:60-61); this is the batch front end of SURVEY.md 8(a) a2. */
int tdsa_set_overlap(tdsa_plan p, int n_streams);

/* ---- trace objects: DataProcessor / TraceAverager arithmetic on host-provided rows ------------- */
/* -------- trace analytics on device-resident rows --------------------------
 * rows_dev: [n_rows][n_bins] float32 dB rows on the plan's device, e.g. what tdsa_process_dev wrote;
 * the calls are ordered after everything the plan has in flight and return with the results on the host.
 *
 * tdsa_rows_stats      per row: np.max (DutyCycleAnalyser.update_from_power, core/duty_cycle.py:36),
 *                      np.argmax (marker snap fallback, core/marker_manager.py:97; first index of equal
 *                      maxima, NaN wins like numpy) and MarkerManager._band_power over the inclusive bin
 *                      range [band_lo, band_hi] (core/marker_manager.py:308-319: 10 log10(max(sum(10^(dB/10))
 *                      * bin_width, 1e-30)); NaN when band_lo > band_hi = "no bin in the band" -> None).
 *                      Any output pointer may be NULL.
 * tdsa_rows_top_peaks  DataProcessor._find_top_peaks (core/display_data_processor.py:432-471): up to
 *                      n_peaks (<= 8) strongest strict local maxima that are >= min_sep_bins apart and
 *                      separated by a valley min_excursion_db below both; bins [n_rows][n_peaks] padded
 *                      with -1 (dB padded with NaN).  n_bins <= 16384 (the row lives in LDS).  Equal-valued
 *                      candidates are visited larger index first (the reference's order for ties is that
 *                      of np.argsort's unstable sort).
 * tdsa_rows_marker_peaks  MarkerManager.snap_to_peak / snap_to_next_peak (core/marker_manager.py:74-127): per row
 *                      scipy.signal.find_peaks(levels, height=, prominence=, distance=) - local maxima (a flat top
 *                      counts once, at its middle; never the first / last sample), height >= `height`, from the
 *                      highest peak down every peak closer than `distance` bins to a kept one is removed, then
 *                      prominence >= `prominence` (float64, the walk scipy does with wlen = None) - and what the
 *                      two methods take from it: snap_bin = the highest peak (first of equals), or np.argmax of
 *                      the row when no peak qualifies (:93-97); next_bin = the first peak right of current_idx
 *                      (= np.searchsorted(bins, marker position)), wrapping to the first peak, -1 when there is
 *                      none (:120-126: the marker stays).  The reference calls it with height = peak_threshold
 *                      (default -200), prominence = peak_excursion (default 6), distance = 3.  Optionally the
 *                      first max_list peaks in bin order (padded with -1) and their prominences (NaN).  Any
 *                      output pointer may be NULL.  n_bins <= 16384.  Two EQUAL peaks closer than `distance`:
 *                      the larger bin is kept (scipy orders them by np.argsort, whose default sort is not stable:
 *                      the reference's choice between them depends on the CPU's sorting network). */
int tdsa_rows_stats(tdsa_plan p, const float* rows_dev, int n_rows, int n_bins, int band_lo, int band_hi,
                    double bin_width, float* peak_db_host, int32_t* peak_bin_host, double* band_db_host);
int tdsa_rows_top_peaks(tdsa_plan p, const float* rows_dev, int n_rows, int n_bins, int n_peaks, int min_sep_bins,
                        float min_excursion_db, int32_t* peak_bins_host, float* peak_db_host);
/* Per-frame scalars as a by-product of the spectra ("fused as optional epilogues so only scalars
 * return to the host").  With enable != 0 every tdsa_process_* call also leaves, per frame, what tdsa_rows_stats would
 * report for its dB row: peak_db = np.max (DutyCycleAnalyser.update_from_power, core/duty_cycle.py:36), peak_bin =
 * np.argmax (marker snap fallback, core/marker_manager.py:97; first of equals, a NaN first) and band_lin = the sum of
 * 10^(dB / 10) over the inclusive display-bin range [band_lo, band_hi] (MarkerManager._band_power,
 * core/marker_manager.py:308-319, before `* bin_width` and the log; band_lo > band_hi: no band, zeros).
 * Frames of 1024 ... 16384 points in the plain dB modes (no averaging, no tare, hold none / max): the frame kernel's
 * waves form them from the bins in their registers: a wave leaves its maximum, its band sum and the sixteen dB values of
 * the one lane that holds the maximum (80 bytes per wave and frame; ties between lanes and NaNs are resolved in the
 * kernel); tdsa_get_frame_stats folds a frame's waves when it is called.  No second pass over the rows, which need not
 * even be written (out_db_dev = NULL).  band_lin is summed from the linear power the kernel holds (float32 per wave,
 * float64 across waves; within 1e-5 relative of the sum over the rounded dB values, one
 * float32 unit of a dB value near -110 being 2e-6 of the power).  Every other plan / mode: rows_stats_kernel runs on
 * the rows the call wrote (a call without rows then has no statistics).  Not for plans above 16384 x 2^k points that
 * return one row per call.  The results of the last four calls are kept: calls_back = 0 is the latest call, 1 the one
 * before ... - tdsa_get_frame_stats waits for that call only, so overlapped calls (tdsa_set_overlap) stay in flight.
 * Any output pointer may be NULL; *n_frames = frames of that call (batched captures: all of them, capture after
 * capture). */
int tdsa_set_frame_stats(tdsa_plan p, int enable, int band_lo, int band_hi);
int tdsa_get_frame_stats(tdsa_plan p, int calls_back, int capacity, int* n_frames, float* peak_db_host,
                         int32_t* peak_bin_host, double* band_lin_host);
int tdsa_rows_marker_peaks(tdsa_plan p, const float* rows_dev, int n_rows, int n_bins, double height, double prominence,
                           int distance, int current_idx, int max_list, int32_t* n_peaks_host, int32_t* snap_bin_host,
                           int32_t* next_bin_host, int32_t* peak_bins_host, double* peak_prom_host);

/* -------- display accumulators kept on the device ---------------------------
 * tdsa_density: DensityDisplay._hist (displays/density_display.py:300-320): float32 [n_bins][512]
 * amplitude histogram over -200..+100 dB; every row first multiplies the histogram by `decay` (when
 * decay < 1) and then adds 1 at int32((dB + 200) / 300 * 512) (truncation toward zero; NaN and out of
 * range dropped), rows applied in order with float32 arithmetic.  _update_dev takes rows already on the
 * device (p = the plan that produced them, or NULL), _update one host row (the per-tick call: returns
 * when the row is staged in pinned memory and its update queued - every other entry point is ordered
 * behind it and _read waits), _read copies the histogram or log1p(histogram) (what setImage receives).
 * tdsa_waterfall: Waterfall._buf (displays/waterfall.py:163-180), a circular row buffer initialised to
 * min_db; a row that equals the previous pushed row (np.array_equal) is skipped (:330-336), a new one
 * is written at ptr = (ptr - 1) % H; _view returns what the reference's buf[ptr : ptr + H] holds (newest
 * first).  The reference doubles the buffer ([2H][n_bins], every row at ptr and ptr + H) so that the view
 * is one slice; the device ring keeps every line once and _view copies its two runs.  These keep C4's 2 GiB of rows on the GPU: only the image leaves. */
typedef struct tdsa_density_s* tdsa_density;
int tdsa_density_create(int device_id, int n_bins, float decay, tdsa_density* out);
int tdsa_density_destroy(tdsa_density d);
int tdsa_density_set_decay(tdsa_density d, float decay);
int tdsa_density_reset(tdsa_density d);
int tdsa_density_update_dev(tdsa_density d, tdsa_plan p, const float* rows_dev, int n_rows);
int tdsa_density_update(tdsa_density d, const float* row_host, int n);
int tdsa_density_read(tdsa_density d, float* hist_host, int as_log1p);
/* The image as the display takes it - what ImageItem.setImage(np.log1p(hist), autoLevels=True) feeds its colour table
 * (displays/density_display.py:318): uint8 [n_bins][512], float32 (v - lo) / (hi - lo) * 255 clipped and truncated with
 * lo / hi = the image's minimum / maximum (returned in levels2, may be NULL; pyqtgraph samples a large image for its auto
 * levels, here they are those of the whole image).  One byte per pixel over PCIe instead of four. */
int tdsa_density_read_u8(tdsa_density d, uint8_t* img_host, float* levels2);

typedef struct tdsa_waterfall_s* tdsa_waterfall;
int tdsa_waterfall_create(int device_id, int history_lines, int n_bins, float min_db, tdsa_waterfall* out);
int tdsa_waterfall_destroy(tdsa_waterfall w);
int tdsa_waterfall_push_dev(tdsa_waterfall w, tdsa_plan p, const float* rows_dev, int n_rows, int* n_new);
int tdsa_waterfall_push(tdsa_waterfall w, const float* row_host, int n, int* is_new);
int tdsa_waterfall_view(tdsa_waterfall w, float* view_host, int* ptr);
/* _display_view() as ImageItem.setImage(img, autoLevels=False, levels=(wf_min_db, wf_max_db)) quantises it
 * (displays/waterfall.py:353-356): uint8 [history][n_bins], np.clip((view - min_db) / (max_db - min_db) * 255, 0, 255)
 * truncated, float32 arithmetic; a NaN pixel -> 0. */
int tdsa_waterfall_view_u8(tdsa_waterfall w, float min_db, float max_db, uint8_t* view_host);

/* -------- host pipeline: pinned ring + asynchronous copy / compute / read-back legs -------------
 * Batch counterpart of the reader-thread -> queue.Queue(4) -> get_power_levels() front end
 * (datasources/hackrf_samples.py:54,102-107,191-305).  The producer writes IQ
 * straight into a pinned slot (tdsa_pipe_acquire), tdsa_pipe_submit enqueues H2D -> frame kernel ->
 * D2H on three streams and returns at once, tdsa_pipe_collect waits for the OLDEST submitted slot and
 * hands back its dB rows (pinned, valid until that slot is acquired again).  Slots complete in
 * submission order; plan state (hold traces, averager, DC tracker) advances exactly as if
 * tdsa_process_i8 had been called per slot.  want_rows: 0 keeps only the plan state (hold / Welch
 * traces); 1 reads the dB rows back into pinned host memory (tdsa_pipe_collect); 2 keeps them in the
 * slot's device buffer for the analytics / accumulators below (tdsa_pipe_collect_dev hands out the device
 * pointer, valid until that slot is acquired again) and skips the read-back leg; 3 reads them back as
 * uint8 [n_frames][nfft] - what ImageItem.setImage(rows, levels=(min_db, max_db)) makes of them before the
 * colour table (displays/waterfall.py:353-356: float32 (v - lo) / (hi - lo) * 255, clipped, truncated; NaN -> 0;
 * levels from tdsa_pipe_set_levels, (-120, 0) until it is called, applied to slots submitted afterwards):
 * a quarter of the bytes over PCIe, which is what bounds a pipe that returns rows (tdsa_pipe_collect_u8; the
 * float32 rows of the slot stay on the device, tdsa_pipe_collect_dev hands them out as well).  One thread
 * drives a pipe; destroy it before its plan. */
typedef struct tdsa_pipe_s* tdsa_pipe;
int tdsa_pipe_create(tdsa_plan p, int in_format, size_t slot_samples, int n_slots, int want_rows, tdsa_pipe* out);
int tdsa_pipe_destroy(tdsa_pipe q);
int tdsa_pipe_acquire(tdsa_pipe q, void** host_slot);
int tdsa_pipe_submit(tdsa_pipe q, size_t n_samples, int hop, int n_frames);
int tdsa_pipe_collect(tdsa_pipe q, const float** rows_host, int* n_frames);
int tdsa_pipe_collect_dev(tdsa_pipe q, const float** rows_dev, int* n_frames);
int tdsa_pipe_collect_u8(tdsa_pipe q, const uint8_t** rows_host, int* n_frames);
int tdsa_pipe_set_levels(tdsa_pipe q, float min_db, float max_db);
int tdsa_pipe_pending(tdsa_pipe q, int* pending);

/* A trace object owns the per-bin state the reference keeps in numpy arrays on MainWindow /
 * DisplayManager / TraceAverager for ONE displayed trace of n bins (any n >= 1, not tied to an FFT
 * plan): hold traces (mw.max_power_levels / mw.min_power_levels, main.py:70-105), the tare
 * accumulator + baseline (core/tare_state.py:9-13, mw.baseline_power_levels) and a TraceAverager
 * buffer (utils/signal_processing.py:16-17). */
typedef struct tdsa_trace_s* tdsa_trace;
int tdsa_trace_create(int device_id, int n, tdsa_trace* out);
int tdsa_trace_destroy(tdsa_trace t);
int tdsa_trace_reset(tdsa_trace t, uint32_t what); /* TDSA_RESET_* bits */

/* One displayed frame: db_in (n float32 dB values from ANY SampleDataSource) -> +cal offset ->
 * tare (collect / subtract) -> live; max/min hold updated.  Replaces _apply_cal_offset (:317-327),
 * _apply_tare (:329-369), _update_max_hold (:371-382), _update_min_hold (:384-395) of
 * core/display_data_processor.py in one launch.
 * tare_collect != 0: this frame is accumulated into the tare buffer (10^(dB/10)); when
 * tare_total frames have been collected the baseline becomes active (reported in *tare_done) and
 * is subtracted from this very frame, as the reference does.  tare_subtract != 0: subtract the
 * active baseline.  live_out/max_out/min_out may be NULL. */
int tdsa_trace_update(tdsa_trace t, const float* db_in_host, int n, float cal_offset_db,
                      int tare_collect, int tare_total, int tare_subtract, uint32_t hold_flags,
                      float* live_out, float* max_out, float* min_out, int* tare_done);
int tdsa_trace_get_tare_baseline(tdsa_trace t, float* baseline_db_host, int* active);
int tdsa_trace_set_tare_baseline(tdsa_trace t, const float* baseline_db_host, int n);

/* TraceAverager on host rows of linear power (utils/signal_processing.py:19-61); also used for the
 * sweep averager DataProcessor owns (display_data_processor.py:41,217-221).  set_mode resets. */
int tdsa_trace_avg_set_mode(tdsa_trace t, int avg_mode, int avg_n);
int tdsa_trace_avg_process(tdsa_trace t, const double* linear_in_host, int n, double* avg_out_host,
                           int* count_out);

/* ---- helpers for callers without their own device allocator -------------------------------- */
int tdsa_dev_alloc(int device_id, size_t bytes, void** out_dev);
int tdsa_dev_free(int device_id, void* dev);
int tdsa_memcpy_h2d(int device_id, void* dst_dev, const void* src_host, size_t bytes);
int tdsa_memcpy_d2h(int device_id, void* dst_host, const void* src_dev, size_t bytes);

/* HIP-event timer on the plan's stream (bench.py: kernel time on the stream the kernels run on). */
int tdsa_timer_begin(tdsa_plan p);
int tdsa_timer_end(tdsa_plan p, float* elapsed_ms); /* records, synchronises, returns elapsed */

/* Per-launch HIP-event bracketing of the frame kernel alone (the dominant kernel bench.py prices
 * against the HBM roofline).  enable != 0 starts collecting; read synchronises the stream and returns
 * the number of frame-kernel launches and the sum of their durations since the last read. */
int tdsa_profile_enable(tdsa_plan p, int enable);
int tdsa_profile_read(tdsa_plan p, int* launches, float* total_ms);

/* ---- developer section (no reference counterpart; used by tools/ and tests/ only) ---------------------- */
/* Plan parameters the tools and tests move to reach code paths that otherwise need other hardware or very long
 * batches (the library reads nothing from the environment): "num_cu" (persistent grids sized for fewer CUs),
 * "avg_wg_min" (batches of more frames take the workgroup-chunk averager scan), "avg_f64_chunks" (1: always the scan
 * over fixed 64-frame chunks with float64 aggregates), "overlap_share" (percent of the CUs an overlapped launch is
 * sized for), "big_group" (long-frame plans: segments per column / row round, 1 .. 64), "chirp_single" (chirp-z plans:
 * 0 = the passes of the convolution as separate kernels instead of one launch), "smooth" (frame lengths 2^a 3^b 5^c: 0 = as a chirp-z convolution like every other size that is not a
 * power of two, instead of the mixed-radix transform), "smooth_n1" (such sizes above 10 000 points: the column pass's length of the two-pass transform, a divisor with both factors
 * <= 10 000), "chirp_fuse_big" (chirp-z plans with
 * M > 16384: 0 = the unpack / window / chirp and the power / dB passes as kernels of their own), "big_fuse_gather"
 * (long-frame Welch captures of one round: 1 = row pass + gather + finish as ONE launch with a dependency-counted ticket
 * queue instead of two launches - same bits, measured slower, profiles/r06_c5_fused_gather.txt; default 0) and
 * "big_queue_gave_up" (reads: an error if a workgroup of such a launch ever gave up waiting); a library built with
 * -DTDSA_DEV also knows "big_pre_wgs" (empty workgroups ahead of every column pass: tools/c5_xcd_phase.py) and "cu_mask"
 * (the plan's stream confined to a set of CUs: tools/c5_two_plans.py).
 * Unknown names are an error. */
int tdsa_debug_knob(tdsa_plan p, const char* name, int value);
/* Phase timeline of workgroup 0 of the frame kernel: allocates the plan's stamp buffer on first call;
 * host_out_2048 != NULL copies 2048 s_memtime stamps back.  Stamps are only written by a library built
 * with -DTDSA_TIMELINE (tools/timeline.py); a production build leaves them zero. */
int tdsa_debug_timeline(tdsa_plan p, unsigned long long* host_out_2048);

#ifdef __cplusplus
}
#endif
#endif /* TDSA_HIP_H */
