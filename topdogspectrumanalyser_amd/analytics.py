"""Device-side trace analytics and display accumulators.

Everything here consumes dB rows that already sit in HBM (the output of SpectrumEngine.process_device /
HostPipe) and returns scalars or a small image, so a batch user never reads the [frames, N] block back:

  rows_stats        np.max / np.argmax per row (DutyCycleAnalyser.update_from_power, core/duty_cycle.py:36;
                    marker snap fallback core/marker_manager.py:97) + MarkerManager._band_power (:308-319)
  rows_top_peaks    DataProcessor._find_top_peaks (core/display_data_processor.py:432-471)
  rows_marker_peaks MarkerManager.snap_to_peak / snap_to_next_peak (core/marker_manager.py:74-127): scipy's
                    find_peaks(height, prominence, distance) per row + the bin each method would move the marker to
  DutyCycle         DutyCycleAnalyser (core/duty_cycle.py) fed with device-computed per-frame peaks
  DensityHistogram  DensityDisplay._hist (displays/density_display.py:300-320)
  WaterfallRing     Waterfall._buf / _add_row / _display_view (displays/waterfall.py:163-180, 330-336)
"""
import ctypes as C
from collections import deque
from typing import List, Optional, Tuple

import numpy as np

from . import _native as nat
from .engine import SpectrumEngine

AMP_BINS = 512
AMP_MIN = -200.0
AMP_RNG = 300.0


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def band_bin_range(freq_bins: np.ndarray, f_start: float, f_stop: float) -> Tuple[int, int]:
    """Inclusive bin range selected by `(bins >= lo) & (bins <= hi)` on an increasing axis; (0, -1) if empty."""
    lo, hi = min(f_start, f_stop), max(f_start, f_stop)
    a = int(np.searchsorted(freq_bins, lo, side="left"))
    b = int(np.searchsorted(freq_bins, hi, side="right")) - 1
    return (a, b) if a <= b else (0, -1)


def rows_stats(engine: SpectrumEngine, rows_dev: int, n_rows: int, n_bins: Optional[int] = None,
               freq_bins: Optional[np.ndarray] = None, band: Optional[Tuple[float, float]] = None):
    """(peak_db[n_rows] f32, peak_bin[n_rows] i32, band_db[n_rows] f64 or None).  `band` = (f_start, f_stop) on
    `freq_bins`; rows whose band holds no bin report NaN (the reference returns None)."""
    n = int(n_bins or engine.nfft)
    peak = np.empty(n_rows, dtype=np.float32)
    pbin = np.empty(n_rows, dtype=np.int32)
    bdb = None
    lo, hi, width = 0, -1, 0.0
    if band is not None:
        if freq_bins is None:
            raise ValueError("band power needs the frequency axis")
        lo, hi = band_bin_range(freq_bins, *band)
        width = float((freq_bins[-1] - freq_bins[0]) / max(len(freq_bins) - 1, 1))
        bdb = np.empty(n_rows, dtype=np.float64)
    nat.check(nat.lib.tdsa_rows_stats(engine._h, C.c_void_p(rows_dev), int(n_rows), n, lo, hi, width,
                                      _p(peak), _p(pbin), _p(bdb) if bdb is not None else None))
    return peak, pbin, bdb


def rows_top_peaks(engine: SpectrumEngine, rows_dev: int, n_rows: int, n_bins: Optional[int] = None, n: int = 5,
                   min_sep_bins: Optional[int] = None, min_excursion_db: float = 10.0):
    """(bins[n_rows, n] i32 padded with -1, db[n_rows, n] f32 padded with NaN); min_sep_bins defaults to the
    reference's max(10, n_bins // 50) (display_data_processor.py:416)."""
    nb = int(n_bins or engine.nfft)
    sep = max(10, nb // 50) if min_sep_bins is None else int(min_sep_bins)
    bins = np.empty((n_rows, n), dtype=np.int32)
    db = np.empty((n_rows, n), dtype=np.float32)
    nat.check(nat.lib.tdsa_rows_top_peaks(engine._h, C.c_void_p(rows_dev), int(n_rows), nb, int(n), sep,
                                          float(min_excursion_db), _p(bins), _p(db)))
    return bins, db


def rows_marker_peaks(engine: SpectrumEngine, rows_dev: int, n_rows: int, n_bins: Optional[int] = None,
                      peak_threshold: float = -200.0, peak_excursion: float = 6.0, distance: int = 3,
                      current_idx: int = -1, max_list: int = 0):
    """The marker peak search of core/marker_manager.py:74-127 on device rows.  Defaults are the reference's
    (getattr(main_window, 'peak_threshold', -200.0), 'peak_excursion' 6.0, distance=3).  Returns a dict:
      n_peaks[n_rows]   how many peaks find_peaks(levels, height, prominence, distance) reports
      snap_bin[n_rows]  where snap_to_peak puts the marker: the highest peak, or np.argmax(levels) without one
      next_bin[n_rows]  where snap_to_next_peak puts it from bin `current_idx` (= np.searchsorted(bins, position)):
                        next peak to the right, wrapping; -1 = no peak, the marker stays
      peaks[n_rows, max_list] / prominences   (max_list > 0) the first peaks in bin order, padded with -1 / NaN"""
    nb = int(n_bins or engine.nfft)
    cnt = np.empty(n_rows, dtype=np.int32)
    snap = np.empty(n_rows, dtype=np.int32)
    nxt = np.empty(n_rows, dtype=np.int32)
    bins = np.empty((n_rows, max_list), dtype=np.int32) if max_list > 0 else None
    prom = np.empty((n_rows, max_list), dtype=np.float64) if max_list > 0 else None
    nat.check(nat.lib.tdsa_rows_marker_peaks(engine._h, C.c_void_p(rows_dev), int(n_rows), nb, float(peak_threshold),
                                             float(peak_excursion), int(distance), int(current_idx), int(max_list),
                                             _p(cnt), _p(snap), _p(nxt), _p(bins) if bins is not None else None,
                                             _p(prom) if prom is not None else None))
    out = {"n_peaks": cnt, "snap_bin": snap, "next_bin": nxt}
    if max_list > 0:
        out["peaks"], out["prominences"] = bins, prom
    return out


def peaks_as_reference(freq_bins: np.ndarray, bins_row: np.ndarray, db_row: np.ndarray) -> List[Tuple[float, float]]:
    """One row of rows_top_peaks in the reference's return shape: [(freq, power), ...]."""
    return [(float(freq_bins[b]), float(p)) for b, p in zip(bins_row, db_row) if b >= 0]


from .core.duty_cycle import DutyCycleAnalyser  # noqa: E402


class DutyCycle(DutyCycleAnalyser):
    """The analyser of core/duty_cycle.py under the name earlier callers of this module use."""


class _Handle:
    _destroy = None

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            type(self)._destroy(self._h)
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


class DensityHistogram(_Handle):
    """DensityDisplay._hist on the device: [n_bins, 512] float32."""
    _destroy = staticmethod(lambda h: nat.lib.tdsa_density_destroy(h))

    def __init__(self, n_bins: int, decay: float = 0.96, device: int = 0):
        self.n_bins = int(n_bins)
        self._h = C.c_void_p()
        nat.check(nat.lib.tdsa_density_create(int(device), self.n_bins, float(decay), C.byref(self._h)))

    def set_decay(self, decay: float) -> None:
        nat.check(nat.lib.tdsa_density_set_decay(self._h, float(decay)))

    def reset(self) -> None:
        nat.check(nat.lib.tdsa_density_reset(self._h))

    def update(self, live_db: np.ndarray) -> None:
        row = np.ascontiguousarray(live_db, dtype=np.float32)
        nat.check(nat.lib.tdsa_density_update(self._h, _p(row), int(row.size)))

    def update_rows(self, engine: Optional[SpectrumEngine], rows_dev: int, n_rows: int) -> None:
        nat.check(nat.lib.tdsa_density_update_dev(self._h, engine._h if engine is not None else None,
                                                  C.c_void_p(rows_dev), int(n_rows)))

    def hist(self) -> np.ndarray:
        out = np.empty((self.n_bins, AMP_BINS), dtype=np.float32)
        nat.check(nat.lib.tdsa_density_read(self._h, _p(out), 0))
        return out

    def image(self) -> np.ndarray:
        """np.log1p(hist): what the reference hands to setImage (density_display.py:320)."""
        out = np.empty((self.n_bins, AMP_BINS), dtype=np.float32)
        nat.check(nat.lib.tdsa_density_read(self._h, _p(out), 1))
        return out


    def image_u8(self):
        """(uint8 [n_bins, 512], (lo, hi)): np.log1p(hist) as setImage(..., autoLevels=True) quantises it for its colour
        table (density_display.py:318), levels = the image's minimum / maximum; a quarter of the bytes of image()."""
        out = np.empty((self.n_bins, AMP_BINS), dtype=np.uint8)
        lv = np.empty(2, dtype=np.float32)
        nat.check(nat.lib.tdsa_density_read_u8(self._h, _p(out), _p(lv)))
        return out, (float(lv[0]), float(lv[1]))


class WaterfallRing(_Handle):
    """Waterfall._buf on the device ([H, n_bins] float32, every line once; the reference doubles it to make the view one
    slice) with the reference's pointer walk and dedup."""
    _destroy = staticmethod(lambda h: nat.lib.tdsa_waterfall_destroy(h))

    def __init__(self, history_lines: int, n_bins: int, min_db: float, device: int = 0):
        self.history_lines = int(history_lines)
        self.n_bins = int(n_bins)
        self._h = C.c_void_p()
        nat.check(nat.lib.tdsa_waterfall_create(int(device), self.history_lines, self.n_bins, float(min_db),
                                                C.byref(self._h)))

    def push(self, live_power_levels) -> bool:
        row = np.ascontiguousarray(live_power_levels, dtype=np.float32)
        new = C.c_int()
        nat.check(nat.lib.tdsa_waterfall_push(self._h, _p(row), int(row.size), C.byref(new)))
        return bool(new.value)

    def push_rows(self, engine: Optional[SpectrumEngine], rows_dev: int, n_rows: int) -> int:
        new = C.c_int()
        nat.check(nat.lib.tdsa_waterfall_push_dev(self._h, engine._h if engine is not None else None,
                                                  C.c_void_p(rows_dev), int(n_rows), C.byref(new)))
        return new.value

    @property
    def ptr(self) -> int:
        v = C.c_int()
        nat.check(nat.lib.tdsa_waterfall_view(self._h, None, C.byref(v)))
        return v.value

    def view(self) -> np.ndarray:
        out = np.empty((self.history_lines, self.n_bins), dtype=np.float32)
        nat.check(nat.lib.tdsa_waterfall_view(self._h, _p(out), None))
        return out

    def view_u8(self, min_db: float, max_db: float) -> np.ndarray:
        """uint8 [history_lines, n_bins]: the view as setImage(img, autoLevels=False, levels=(wf_min_db, wf_max_db))
        quantises it (displays/waterfall.py:353-356)."""
        out = np.empty((self.history_lines, self.n_bins), dtype=np.uint8)
        nat.check(nat.lib.tdsa_waterfall_view_u8(self._h, float(min_db), float(max_db), _p(out)))
        return out
