"""SpectrumEngine - thin Python owner of one libtdsa_hip plan (device, FFT size, batch capacity).

All arithmetic happens in the HIP kernels behind the C-ABI (include/tdsa_hip.h); this class only
marshals numpy buffers / raw device pointers and mirrors the knobs the reference exposes on its
sources (set_psd_mode, set_averaging, set_dc_alpha, window, cal offset, hold toggles).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _native as nat

_AVG = {"off": nat.AVG_OFF, "exp": nat.AVG_EXP, "lin": nat.AVG_LIN}


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class SpectrumEngine:
    def __init__(self, nfft: int, max_frames: int = 1, device: int = 0):
        self.nfft = int(nfft)
        self.max_frames = int(max_frames)
        self.device = int(device)
        self._h = C.c_void_p()
        nat.check(nat.lib.tdsa_create(self.device, self.nfft, self.max_frames, C.byref(self._h)))
        self._mode = nat.Mode(nat.DB_MAG, 1.0, 1e-12, nat.AVG_OFF, 1, 1.0, 0.0, 0)
        self._pipes = []                  # live HostPipe objects: they hold slots the plan's streams write to

    # ------------------------------------------------------------------ lifetime
    def close(self) -> None:
        for q in list(getattr(self, "_pipes", [])):
            q.close()                     # a pipe must go before its plan (tdsa_pipe_destroy syncs the plan)
        if getattr(self, "_h", None) is not None and self._h:
            nat.lib.tdsa_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------ configuration
    def set_window(self, window: np.ndarray) -> None:
        w = np.ascontiguousarray(window, dtype=np.float32)
        nat.check(nat.lib.tdsa_set_window(self._h, _ptr(w), int(w.size)))

    def configure(self, *, db_mode: Optional[str] = None, power_scale: Optional[float] = None,
                  log_floor: Optional[float] = None, avg: Optional[Tuple[str, int]] = None,
                  dc_alpha: Optional[float] = None, cal_offset_db: Optional[float] = None,
                  hold_max: Optional[bool] = None, hold_min: Optional[bool] = None) -> None:
        m = self._mode
        if db_mode is not None:
            m.db_mode = {"mag": nat.DB_MAG, "pow": nat.DB_POW}[db_mode]
        if power_scale is not None:
            m.power_scale = float(power_scale)
        if log_floor is not None:
            m.log_floor = float(log_floor)
        if avg is not None:
            m.avg_mode, m.avg_n = _AVG[avg[0]], int(avg[1])
        if dc_alpha is not None:
            m.dc_alpha = float(dc_alpha)
        if cal_offset_db is not None:
            m.cal_offset_db = float(cal_offset_db)
        flags = m.hold_flags
        if hold_max is not None:
            flags = (flags | nat.HOLD_MAX) if hold_max else (flags & ~nat.HOLD_MAX)
        if hold_min is not None:
            flags = (flags | nat.HOLD_MIN) if hold_min else (flags & ~nat.HOLD_MIN)
        m.hold_flags = flags
        nat.check(nat.lib.tdsa_set_mode(self._h, C.byref(m)))

    def reset(self, what: int = nat.RESET_ALL) -> None:
        nat.check(nat.lib.tdsa_reset_state(self._h, what))

    def set_tare_baseline(self, baseline_db: Optional[np.ndarray]) -> None:
        if baseline_db is None:
            nat.check(nat.lib.tdsa_set_tare_baseline(self._h, None, 0))
        else:
            b = np.ascontiguousarray(baseline_db, dtype=np.float32)
            nat.check(nat.lib.tdsa_set_tare_baseline(self._h, _ptr(b), int(b.size)))

    def set_frame_stats(self, enable: bool = True, band_bins: Optional[Tuple[int, int]] = None) -> None:
        """Per-frame scalars as a by-product of every process call (tdsa_set_frame_stats): np.max / np.argmax of each
        dB row (core/duty_cycle.py:36, core/marker_manager.py:97) and the linear power of the inclusive display-bin
        range `band_bins` (MarkerManager._band_power, core/marker_manager.py:308-319; analytics.band_bin_range maps a
        frequency band onto bins).  Read them with frame_stats()."""
        lo, hi = (1, 0) if band_bins is None else (int(band_bins[0]), int(band_bins[1]))
        nat.check(nat.lib.tdsa_set_frame_stats(self._h, int(bool(enable)), lo, hi))

    def frame_stats(self, calls_back: int = 0, bin_width: Optional[float] = None):
        """(peak_db[frames] f32, peak_bin[frames] i32, band) of the latest call (calls_back = 1: the one before, ... the
        last four are kept).  band = the band's linear sums (float64), or with bin_width (Hz per bin: `(bins[-1] -
        bins[0]) / max(len(bins) - 1, 1)`) the reference's 10 log10(max(sum * bin_width, 1e-30))."""
        n = C.c_int()
        nat.check(nat.lib.tdsa_get_frame_stats(self._h, int(calls_back), 0, C.byref(n), None, None, None))
        peak = np.empty(n.value, dtype=np.float32)
        pbin = np.empty(n.value, dtype=np.int32)
        band = np.empty(n.value, dtype=np.float64)
        nat.check(nat.lib.tdsa_get_frame_stats(self._h, int(calls_back), n.value, None, _ptr(peak), _ptr(pbin), _ptr(band)))
        if bin_width is not None:
            total = band * float(bin_width)
            band = 10.0 * np.log10(np.where(total < 1e-30, 1e-30, total))      # max(total, 1e-30) keeps a NaN
        return peak, pbin, band

    # ------------------------------------------------------------------ hot path
    def process(self, iq: np.ndarray, hop: Optional[int] = None, n_frames: Optional[int] = None,
                want_db: bool = True) -> Optional[np.ndarray]:
        """Host arrays in, [frames, nfft] float32 dB out.

        iq: int8/uint8 interleaved I,Q (length 2*samples) or complex64 (length samples)."""
        iq = np.ascontiguousarray(iq)
        hop = self.nfft if hop is None else int(hop)
        if iq.dtype == np.complex64:
            n_samples, fn = iq.size, nat.lib.tdsa_process_c64
        elif iq.dtype == np.int8:
            n_samples, fn = iq.size // 2, nat.lib.tdsa_process_i8
        elif iq.dtype == np.uint8:
            n_samples, fn = iq.size // 2, nat.lib.tdsa_process_u8
        else:
            raise TypeError(f"unsupported IQ dtype {iq.dtype}; need int8, uint8 or complex64")
        if n_frames is None:
            n_frames = 0 if n_samples < self.nfft else (n_samples - self.nfft) // hop + 1
        # native long-frame plans (2^15 .. 2^20 points) return ONE row (Welch average / one frame per call); every other
        # plan - the chirp-z plans of long frames that are not a power of two included - one row per frame
        long_native = self.nfft > 16384 and (self.nfft & (self.nfft - 1)) == 0
        rows = min(n_frames, 1) if long_native else n_frames
        out = np.empty((rows, self.nfft), dtype=np.float32) if want_db else None
        nat.check(fn(self._h, _ptr(iq), n_samples, hop, n_frames, _ptr(out)))
        return out

    def process_real2(self, stereo: np.ndarray, channel: str = "mono", hop: Optional[int] = None,
                      n_frames: Optional[int] = None) -> np.ndarray:
        """Real-input path: stereo float32 samples [samples, 2] -> one-sided dB rows
        [frames, N/2+1] (or [frames, 2, N/2+1] for channel="stereo")."""
        x = np.ascontiguousarray(stereo, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != 2:
            raise ValueError("stereo samples must have shape [samples, 2]")
        ch = {"mono": nat.CH_MONO, "left": nat.CH_LEFT, "right": nat.CH_RIGHT, "stereo": nat.CH_STEREO}[channel]
        hop = self.nfft if hop is None else int(hop)
        ns = x.shape[0]
        if n_frames is None:
            n_frames = 0 if ns < self.nfft else (ns - self.nfft) // hop + 1
        nb = self.nfft // 2 + 1
        shape = (n_frames, 2, nb) if ch == nat.CH_STEREO else (n_frames, nb)
        out = np.empty(shape, dtype=np.float32)
        if n_frames:
            nat.check(nat.lib.tdsa_process_real2(self._h, _ptr(x), ns, hop, n_frames, ch, _ptr(out)))
        return out

    def process_device(self, in_format: int, iq_dev: int, n_samples: int, hop: int, n_frames: int,
                       out_db_dev: Optional[int]) -> None:
        """Raw device pointers, asynchronous on the plan's stream (bench path)."""
        nat.check(nat.lib.tdsa_process_dev(self._h, in_format, C.c_void_p(iq_dev), n_samples, hop,
                                           n_frames, C.c_void_p(out_db_dev) if out_db_dev else None))

    def process_device_batch(self, in_format: int, iq_dev: int, seg_stride_bytes: int, n_segments: int,
                             n_samples_per_seg: int, hop: int, frames_per_seg: int, out_db_dev: Optional[int],
                             out_seg_stride_floats: Optional[int] = None) -> None:
        """n_segments captures of one shape (device pointers, asynchronous): the same results and state as
        n_segments process_device() calls, as ONE persistent launch where the mode allows (tdsa_process_dev_batch).
        out_seg_stride_floats defaults to one capture's rows (frames_per_seg * nfft: captures back to back); a
        smaller stride is refused by the library."""
        if out_seg_stride_floats is None:
            out_seg_stride_floats = int(frames_per_seg) * self.nfft
        nat.check(nat.lib.tdsa_process_dev_batch(self._h, in_format, C.c_void_p(iq_dev), seg_stride_bytes, n_segments,
                                                 n_samples_per_seg, hop, frames_per_seg,
                                                 C.c_void_p(out_db_dev) if out_db_dev else None,
                                                 out_seg_stride_floats))

    def synchronize(self) -> None:
        nat.check(nat.lib.tdsa_synchronize(self._h))

    def pipe(self, slot_samples: int, n_slots: int = 3, rows=True, in_format: int = nat.IN_I8,
             levels: Optional[Tuple[float, float]] = None) -> "HostPipe":
        """Pinned host ring with asynchronous copy / compute / read-back legs (tdsa_pipe_*).
        rows: True / "host" = dB rows come back to pinned host memory, "device" = they stay on the GPU
        (collect_device), "u8" = they come back as bytes under `levels` = (min_db, max_db), what
        ImageItem.setImage(rows, levels=...) feeds its colour table (collect_u8; a quarter of the bytes over PCIe),
        False / None = plan state only."""
        q = HostPipe(self, slot_samples, n_slots, rows, in_format)
        if levels is not None:
            q.set_levels(*levels)
        return q

    def set_overlap(self, n_streams: int) -> None:
        """Let consecutive order-independent process_device() calls overlap on n_streams HIP streams
        (tdsa_set_overlap); 1 = strictly serial (default)."""
        nat.check(nat.lib.tdsa_set_overlap(self._h, int(n_streams)))

    # ------------------------------------------------------------------ state read-back
    def hold(self) -> Tuple[Optional[np.ndarray], Optional[np.ndarray]]:
        info = self.info()
        mx = np.empty(self.nfft, dtype=np.float32) if info.frames_held_max > 0 else None
        mn = np.empty(self.nfft, dtype=np.float32) if info.frames_held_min > 0 else None
        held = C.c_int64()
        nat.check(nat.lib.tdsa_get_hold(self._h, _ptr(mx), _ptr(mn), C.byref(held)))
        return mx, mn

    def averaged(self) -> Tuple[Optional[np.ndarray], int]:
        cnt = C.c_int()
        buf = np.empty(self.nfft, dtype=np.float64)
        nat.check(nat.lib.tdsa_get_avg(self._h, _ptr(buf), C.byref(cnt)))
        return (buf if cnt.value > 0 else None), cnt.value

    def welch_export(self, dst: np.ndarray) -> int:
        """The running mean of linear power (TraceAverager._buffer) into `dst` - float32 or float64 [nfft], e.g. a view
        of a shared-memory slab - and the number of frames behind it (tdsa_welch_export; synchronous)."""
        if dst.dtype not in (np.float32, np.float64) or dst.size != self.nfft or not dst.flags.c_contiguous:
            raise ValueError("dst must be a contiguous float32 / float64 array of nfft values")
        cnt = C.c_int()
        nat.check(nat.lib.tdsa_welch_export(self._h, _ptr(dst), int(dst.dtype == np.float32), C.byref(cnt)))
        return cnt.value

    def welch_combine(self, parts: np.ndarray, counts, out_db_dev: Optional[int] = None,
                      want_host: bool = False) -> Optional[np.ndarray]:
        """Partial means [n_parts, nfft] (float32 / float64, rows may be strided) + frame counts -> this plan's averager
        state and the dB row (tdsa_welch_combine): written to the device pointer and / or returned as a host array."""
        if parts.ndim != 2 or parts.shape[1] != self.nfft or parts.strides[1] != parts.itemsize:
            raise ValueError("parts must be [n_parts, nfft] with contiguous rows")
        if parts.dtype not in (np.float32, np.float64):
            raise TypeError("parts must be float32 or float64")
        cnt = np.ascontiguousarray(counts, dtype=np.int32)
        if cnt.size != parts.shape[0]:
            raise ValueError("one count per partial mean")
        out = np.empty(self.nfft, dtype=np.float32) if want_host else None
        nat.check(nat.lib.tdsa_welch_combine(self._h, _ptr(parts), int(parts.strides[0]),
                                             cnt.ctypes.data_as(C.POINTER(C.c_int32)), int(cnt.size),
                                             int(parts.dtype == np.float32),
                                             C.c_void_p(out_db_dev) if out_db_dev else None, _ptr(out)))
        return out

    def welch_export_dev(self, dst_dev: int, as_f32: bool = True) -> int:
        """The running mean into a DEVICE buffer (e.g. a slot of a sharding.WelchPeerSlab other ranks read in place);
        returns the number of frames behind it (tdsa_welch_export_dev; synchronous)."""
        cnt = C.c_int()
        nat.check(nat.lib.tdsa_welch_export_dev(self._h, C.c_void_p(int(dst_dev)), int(bool(as_f32)), C.byref(cnt)))
        return cnt.value

    def welch_combine_dev(self, parts_dev, counts, as_f32: bool = True, out_db_dev: Optional[int] = None,
                          want_host: bool = False) -> Optional[np.ndarray]:
        """welch_combine on partial means that sit in device memory this plan's GPU can read - its own or other ranks'
        buffers mapped with tdsa_peer_open (read over xGMI, no staging): tdsa_welch_combine_dev."""
        cnt = np.ascontiguousarray(counts, dtype=np.int32)
        if cnt.size != len(parts_dev):
            raise ValueError("one count per partial mean")
        ptrs = (C.c_void_p * len(parts_dev))(*[C.c_void_p(int(p)) if p else None for p in parts_dev])
        out = np.empty(self.nfft, dtype=np.float32) if want_host else None
        nat.check(nat.lib.tdsa_welch_combine_dev(self._h, ptrs, cnt.ctypes.data_as(C.POINTER(C.c_int32)), int(cnt.size),
                                                 int(bool(as_f32)), C.c_void_p(out_db_dev) if out_db_dev else None,
                                                 _ptr(out)))
        return out

    def shader_clock(self) -> Tuple[float, float]:
        """(shader MHz, ns per VALU wave-instruction per SIMD) from a millisecond of saturated v_add_f32 (tdsa_shader_clock)."""
        mhz, ns = C.c_float(), C.c_float()
        nat.check(nat.lib.tdsa_shader_clock(self._h, C.byref(mhz), C.byref(ns)))
        return float(mhz.value), float(ns.value)

    @property
    def dc_estimate(self) -> complex:
        re, im = C.c_float(), C.c_float()
        nat.check(nat.lib.tdsa_get_dc(self._h, C.byref(re), C.byref(im)))
        return complex(re.value, im.value)

    @dc_estimate.setter
    def dc_estimate(self, value: complex) -> None:
        value = complex(value)
        nat.check(nat.lib.tdsa_set_dc(self._h, float(value.real), float(value.imag)))

    def info(self) -> nat.Info:
        inf = nat.Info()
        nat.check(nat.lib.tdsa_get_info(self._h, C.byref(inf)))
        return inf

    def debug_knob(self, name: str, value: int) -> None:
        """developer hook (tdsa_debug_knob): num_cu, avg_wg_min, avg_f64_chunks, overlap_share, big_group"""
        nat.check(nat.lib.tdsa_debug_knob(self._h, name.encode(), int(value)))

    def profile_enable(self, on: bool = True) -> None:
        nat.check(nat.lib.tdsa_profile_enable(self._h, int(bool(on))))

    def profile_read(self):
        """(launches, total_ms) of the frame kernel alone since the last read (HIP events)."""
        n, ms = C.c_int(), C.c_float()
        nat.check(nat.lib.tdsa_profile_read(self._h, C.byref(n), C.byref(ms)))
        return n.value, float(ms.value)

    # ------------------------------------------------------------------ timing (HIP events, plan stream)
    def timer_begin(self) -> None:
        nat.check(nat.lib.tdsa_timer_begin(self._h))

    def timer_end(self) -> float:
        ms = C.c_float()
        nat.check(nat.lib.tdsa_timer_end(self._h, C.byref(ms)))
        return float(ms.value)


class TraceState:
    """Device-side state of ONE displayed trace of n bins (any n): hold traces, tare accumulator and
    baseline, TraceAverager buffer.  The arithmetic of DataProcessor._apply_cal_offset / _apply_tare /
    _update_max_hold / _update_min_hold and TraceAverager.process runs in HIP kernels."""

    def __init__(self, n: int, device: int = 0):
        self.n = int(n)
        self.device = int(device)
        self._h = C.c_void_p()
        nat.check(nat.lib.tdsa_trace_create(self.device, self.n, C.byref(self._h)))

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            nat.lib.tdsa_trace_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, what: int = nat.RESET_ALL) -> None:
        nat.check(nat.lib.tdsa_trace_reset(self._h, what))

    def update(self, db_in: np.ndarray, cal_offset_db: float = 0.0, tare_collect: bool = False,
               tare_total: int = 32, tare_subtract: bool = False, hold_max: bool = False,
               hold_min: bool = False):
        """-> (live, max or None, min or None, tare_done)"""
        x = np.ascontiguousarray(db_in, dtype=np.float32)
        live = np.empty(self.n, dtype=np.float32)
        mx = np.empty(self.n, dtype=np.float32) if hold_max else None
        mn = np.empty(self.n, dtype=np.float32) if hold_min else None
        done = C.c_int()
        flags = (nat.HOLD_MAX if hold_max else 0) | (nat.HOLD_MIN if hold_min else 0)
        nat.check(nat.lib.tdsa_trace_update(self._h, _ptr(x), int(x.size), float(cal_offset_db),
                                            int(bool(tare_collect)), int(tare_total), int(bool(tare_subtract)),
                                            flags, _ptr(live), _ptr(mx), _ptr(mn), C.byref(done)))
        return live, mx, mn, bool(done.value)

    def tare_baseline(self) -> Optional[np.ndarray]:
        act = C.c_int()
        b = np.empty(self.n, dtype=np.float32)
        nat.check(nat.lib.tdsa_trace_get_tare_baseline(self._h, _ptr(b), C.byref(act)))
        return b if act.value else None

    def tare_is_active(self) -> bool:
        act = C.c_int()
        nat.check(nat.lib.tdsa_trace_get_tare_baseline(self._h, None, C.byref(act)))
        return bool(act.value)

    def set_tare_baseline(self, baseline_db: Optional[np.ndarray]) -> None:
        if baseline_db is None:
            nat.check(nat.lib.tdsa_trace_set_tare_baseline(self._h, None, 0))
        else:
            b = np.ascontiguousarray(baseline_db, dtype=np.float32)
            nat.check(nat.lib.tdsa_trace_set_tare_baseline(self._h, _ptr(b), int(b.size)))

    def avg_set_mode(self, mode: str, n: int) -> None:
        nat.check(nat.lib.tdsa_trace_avg_set_mode(self._h, _AVG[mode], int(n)))

    def avg_process(self, linear_power: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(linear_power, dtype=np.float64)
        out = np.empty(self.n, dtype=np.float64)
        cnt = C.c_int()
        nat.check(nat.lib.tdsa_trace_avg_process(self._h, _ptr(x), int(x.size), _ptr(out), C.byref(cnt)))
        return out


class HostPipe:
    """Batch front end over `tdsa_pipe_*`: the producer fills pinned slots, H2D / frame kernel / D2H of
    neighbouring slots overlap.  Counterpart of the reader thread + queue of HackrfSamplesDataSource
    (datasources/hackrf_samples.py:191-305 of the reference) for recorders and offline analysis.

        with eng.pipe(slot_samples) as q:
            q.acquire()[: 2 * n] = iq_chunk           # int8 view of the pinned slot
            q.submit(n, hop, n_frames)                # returns immediately
            rows = q.collect()                        # oldest slot's dB rows (view, valid until reuse)
    """

    _DTYPES = {nat.IN_I8: (np.int8, 2), nat.IN_U8: (np.uint8, 2), nat.IN_C64: (np.complex64, 1)}

    def __init__(self, engine: SpectrumEngine, slot_samples: int, n_slots: int, rows: bool, in_format: int):
        self._eng = engine
        self.slot_samples = int(slot_samples)
        self.rows_mode = {True: 1, "host": 1, "device": 2, "u8": 3, False: 0, None: 0}[rows]
        self.rows = self.rows_mode == 1
        self.in_format = int(in_format)
        self._q = C.c_void_p()
        nat.check(nat.lib.tdsa_pipe_create(engine._h, self.in_format, self.slot_samples, int(n_slots),
                                            self.rows_mode, C.byref(self._q)))
        engine._pipes.append(self)

    def close(self) -> None:
        if getattr(self, "_q", None) is not None and self._q:
            if self._eng._h:              # (the engine closes its pipes before it destroys the plan)
                nat.lib.tdsa_pipe_destroy(self._q)
            self._q = C.c_void_p()
            if self in self._eng._pipes:
                self._eng._pipes.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def acquire(self) -> np.ndarray:
        """Next free pinned input slot as a numpy view (int8/uint8: interleaved I,Q; complex64: samples)."""
        ptr = C.c_void_p()
        nat.check(nat.lib.tdsa_pipe_acquire(self._q, C.byref(ptr)))
        dt, per = self._DTYPES[self.in_format]
        n = self.slot_samples * per
        buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=dt, count=n)

    def submit(self, n_samples: int, hop: int, n_frames: int) -> None:
        nat.check(nat.lib.tdsa_pipe_submit(self._q, int(n_samples), int(hop), int(n_frames)))

    def collect(self) -> Optional[np.ndarray]:
        """Wait for the oldest submitted slot; its dB rows [n_frames, N] (None for a rows=False pipe)."""
        rows = C.POINTER(C.c_float)()
        nf = C.c_int()
        nat.check(nat.lib.tdsa_pipe_collect(self._q, C.byref(rows), C.byref(nf)))
        if not self.rows:
            return None
        return np.ctypeslib.as_array(rows, shape=(nf.value, self._eng.nfft))

    def set_levels(self, min_db: float, max_db: float) -> None:
        """levels of a rows="u8" pipe (displays/waterfall.py:353-356), for the slots submitted from now on"""
        nat.check(nat.lib.tdsa_pipe_set_levels(self._q, float(min_db), float(max_db)))

    def collect_u8(self) -> np.ndarray:
        """Wait for the oldest submitted slot of a rows="u8" pipe: its rows as uint8 [n_frames, N] (view of the
        pinned slot, valid until that slot is acquired again)."""
        rows = C.POINTER(C.c_ubyte)()
        nf = C.c_int()
        nat.check(nat.lib.tdsa_pipe_collect_u8(self._q, C.byref(rows), C.byref(nf)))
        return np.ctypeslib.as_array(rows, shape=(nf.value, self._eng.nfft))

    def collect_device(self) -> Tuple[int, int]:
        """Wait for the oldest submitted slot of a rows="device" pipe: (device pointer of its dB rows,
        n_frames); the rows stay valid until that slot is acquired again."""
        rows = C.POINTER(C.c_float)()
        nf = C.c_int()
        nat.check(nat.lib.tdsa_pipe_collect_dev(self._q, C.byref(rows), C.byref(nf)))
        return C.cast(rows, C.c_void_p).value, nf.value

    @property
    def pending(self) -> int:
        n = C.c_int()
        nat.check(nat.lib.tdsa_pipe_pending(self._q, C.byref(n)))
        return n.value
