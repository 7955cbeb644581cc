"""CPU restatement of the trace analytics / accumulation step that follows the IQ -> spectrum path
(SURVEY.md 8(f) f-3, f-4).  TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing else); the product
never routes through it.

Pinned against the imported reference (tests/golden/analytics.npz, generator tests/golden/make_golden_analytics.py):
  find_top_peaks          <- DataProcessor._find_top_peaks        (core/display_data_processor.py:432-471)
  DutyCycleOracle         <- DutyCycleAnalyser.update_from_power   (core/duty_cycle.py:30-50)
  band_power_db           <- MarkerManager._band_power             (core/marker_manager.py:308-319)
  frame_peak              <- np.max / np.argmax as used by core/duty_cycle.py:36 and marker snap (marker_manager.py:97)
Pinned against the imported display classes themselves (tests/golden/displays.npz, generator
tests/golden/make_golden_displays.py: displays/density_display.py and displays/waterfall.py imported against
stub PyQt6 / pyqtgraph modules made of plain do-nothing classes, real objects constructed, their numpy runs):
  DensityOracle           <- DensityDisplay._ensure_hist/_update_hist (displays/density_display.py:300-320)
  WaterfallOracle         <- Waterfall._init_buffer/_add_row/_display_view + dedup (displays/waterfall.py:163-180, 330-336)
"""
from collections import deque
from typing import List, Optional, Tuple

import numpy as np

AMP_BINS = 512          # displays/density_display.py:12
AMP_MIN = -200.0        # :13
AMP_RNG = 300.0         # :14
DUTY_BUFFER_FRAMES = 100  # core/duty_cycle.py:7


def _accepts(power: np.ndarray, cand: int, chosen: List[int], min_sep_bins: int, excursion: float) -> bool:
    """A candidate joins the list unless some already chosen peak is nearer than min_sep_bins or the deepest
    point between the two is less than `excursion` below either of them."""
    for other in chosen:
        if abs(cand - other) < min_sep_bins:
            return False
        a, b = (cand, other) if cand < other else (other, cand)
        floor_between = float(power[a:b + 1].min())
        if power[cand] - floor_between < excursion or float(power[other]) - floor_between < excursion:
            return False
    return True


def find_top_peaks(freq_bins: np.ndarray, power: np.ndarray, n: int = 5, min_sep_bins: int = 10,
                   min_excursion_db: float = 10.0) -> List[Tuple[float, float]]:
    """display_data_processor.py:432-471: strict interior local maxima visited from the strongest down
    (order = reversed np.argsort, as there), greedily accepted by `_accepts`, at most n of them."""
    if len(power) < 3:
        return []
    mid = power[1:-1]
    cands = np.flatnonzero((mid > power[:-2]) & (mid > power[2:])) + 1
    if cands.size == 0:
        return []
    order = cands[np.argsort(power[cands])[::-1]]
    chosen: List[int] = []
    for c in order:
        if len(chosen) >= n:
            break
        if _accepts(power, int(c), chosen, min_sep_bins, min_excursion_db):
            chosen.append(int(c))
    return [(float(freq_bins[i]), float(power[i])) for i in chosen]


def find_top_peak_bins(power: np.ndarray, n: int = 5, min_sep_bins: int = 10,
                       min_excursion_db: float = 10.0) -> List[int]:
    """Same selection, returning bin indices (what the device kernel reports)."""
    bins = np.arange(len(power), dtype=np.float64)
    return [int(f) for f, _ in find_top_peaks(bins, power, n, min_sep_bins, min_excursion_db)]


def peak_list_params(n_bins: int, peak_excursion: float = 10.0) -> Tuple[int, float]:
    """display_data_processor.py:416-417: min_sep = max(10, len(freq_bins)//50), excursion from the window."""
    return max(10, n_bins // 50), float(peak_excursion)


def frame_peak(power_db: np.ndarray) -> Tuple[float, int]:
    """np.max / np.argmax of one trace (duty_cycle.py:36; marker_manager.py:97 fallback)."""
    return float(np.max(power_db)), int(np.argmax(power_db))


class DutyCycleOracle:
    """core/duty_cycle.py:10-50 restricted to update_from_power/_recompute/reset."""

    def __init__(self):
        self._envelope = deque(maxlen=DUTY_BUFFER_FRAMES)
        self.duty_pct = 0.0
        self.on_power_dbm: Optional[float] = None
        self.off_power_dbm: Optional[float] = None
        self.threshold_dbm = -60.0

    def update_from_power(self, power_levels_db, threshold_dbm=None) -> None:
        if power_levels_db is None or len(power_levels_db) == 0:
            return
        if threshold_dbm is not None:
            self.threshold_dbm = threshold_dbm
        self.push_peak(float(np.max(power_levels_db)))

    def push_peak(self, peak: float) -> None:
        self._envelope.append(peak)
        arr = np.array(self._envelope)
        on_mask = arr >= self.threshold_dbm
        on_count = int(np.sum(on_mask))
        self.duty_pct = 100.0 * on_count / len(arr)
        self.on_power_dbm = float(np.mean(arr[on_mask])) if on_count > 0 else None
        off_mask = ~on_mask
        self.off_power_dbm = float(np.mean(arr[off_mask])) if np.any(off_mask) else None

    def reset(self) -> None:
        self._envelope.clear()
        self.duty_pct = 0.0
        self.on_power_dbm = None
        self.off_power_dbm = None


def band_power_db(bins: np.ndarray, levels: np.ndarray, f_start: float, f_stop: float) -> Optional[float]:
    """core/marker_manager.py:308-319."""
    lo, hi = min(f_start, f_stop), max(f_start, f_stop)
    mask = (bins >= lo) & (bins <= hi)
    if not np.any(mask):
        return None
    bin_width = (bins[-1] - bins[0]) / max(len(bins) - 1, 1)
    total = np.sum(10.0 ** (levels[mask] / 10.0)) * bin_width
    return 10.0 * np.log10(max(total, 1e-30))


def band_bin_range(bins: np.ndarray, f_start: float, f_stop: float) -> Tuple[int, int]:
    """Inclusive bin range the mask of band_power_db selects on a monotonically increasing axis
    ((0, -1) when empty)."""
    lo, hi = min(f_start, f_stop), max(f_start, f_stop)
    idx = np.where((bins >= lo) & (bins <= hi))[0]
    if idx.size == 0:
        return 0, -1
    return int(idx[0]), int(idx[-1])


class DensityOracle:
    """displays/density_display.py:300-320: per-frequency amplitude histogram with exponential decay."""

    def __init__(self, decay: float = 0.96):
        self.decay = float(decay)
        self.hist: Optional[np.ndarray] = None

    def update(self, live_db: np.ndarray) -> np.ndarray:
        n = len(live_db)
        if self.hist is None or self.hist.shape[0] != n:
            self.hist = np.zeros((n, AMP_BINS), dtype=np.float32)
        if self.decay < 1.0:
            self.hist *= self.decay
        valid = ~np.isnan(live_db)
        raw_idx = np.full(n, -1, dtype=np.int32)
        raw_idx[valid] = ((live_db[valid] - AMP_MIN) / AMP_RNG * AMP_BINS).astype(np.int32)
        in_range = (raw_idx >= 0) & (raw_idx < AMP_BINS)
        fi = np.where(in_range)[0]
        if len(fi):
            self.hist[fi, raw_idx[fi]] += 1.0
        return self.hist

    def image(self) -> np.ndarray:
        return np.log1p(self.hist)            # density_display.py:320


class WaterfallOracle:
    """displays/waterfall.py:163-180 (double-height circular buffer) + the new-row test of :330-336."""

    def __init__(self, history_lines: int, n_bins: int, min_db: float):
        self.H = int(history_lines)
        self.W = int(n_bins)
        self.buf = np.full((2 * self.H, self.W), min_db, dtype=np.float32)
        self.ptr = 0
        self.last_row: Optional[np.ndarray] = None

    def update(self, live_power_levels) -> bool:
        data = np.asarray(live_power_levels, dtype=np.float32)
        is_new = self.last_row is None or not np.array_equal(data, self.last_row)
        if is_new:
            self.last_row = data.copy()
            self.ptr = (self.ptr - 1) % self.H
            self.buf[self.ptr] = data
            self.buf[self.ptr + self.H] = data
        return is_new

    def view(self) -> np.ndarray:
        return self.buf[self.ptr:self.ptr + self.H]


# ---- marker peak search (core/marker_manager.py:74-127) -----------------------------------------------------------
# The arithmetic lives in scipy.signal.find_peaks (third party: scipy == 1.16.3 pinned in the reference's
# requirements.txt:99, 1.15.3 in this image; scipy/signal/_peak_finding.py + _peak_finding_utils.pyx, unchanged between
# the two), called as find_peaks(levels, height=threshold, prominence=excursion, distance=3).  Restated here from its
# published algorithm, pinned by tests/golden/markers.npz (captured from the imported MarkerManager running the real
# scipy, generator tests/golden/make_golden_markers.py).

def _local_maxima(x: np.ndarray) -> np.ndarray:
    """scipy _local_maxima_1d: samples (or the middle `(left + right) // 2` of flat tops) that rise strictly before and
    fall strictly after; the first and last sample never count, and a plateau that reaches the last sample is no peak."""
    n = len(x)
    out = []
    i, i_max = 1, n - 1
    while i < i_max:
        if x[i - 1] < x[i]:
            ahead = i + 1
            while ahead < i_max and x[ahead] == x[i]:
                ahead += 1
            if x[ahead] < x[i]:
                out.append((i + ahead - 1) // 2)
                i = ahead
        i += 1
    return np.array(out, dtype=np.intp)


def _select_by_distance(peaks: np.ndarray, priority: np.ndarray, distance: float) -> np.ndarray:
    """scipy _select_by_peak_distance: from the highest peak down, a kept peak removes every peak closer than
    ceil(distance) samples.  Equal heights: scipy walks np.argsort(priority) backwards, and numpy's default argsort is
    not a stable sort, so the reference's order between EQUAL peaks closer than `distance` is not defined; here (and on
    the device) the larger index goes first, which is what a stable sort gives."""
    d = int(np.ceil(distance))
    keep = np.ones(len(peaks), dtype=bool)
    order = np.argsort(priority, kind="stable")
    for j in order[::-1]:
        if not keep[j]:
            continue
        k = j - 1
        while k >= 0 and peaks[j] - peaks[k] < d:
            keep[k] = False
            k -= 1
        k = j + 1
        while k < len(peaks) and peaks[k] - peaks[j] < d:
            keep[k] = False
            k += 1
    return keep


def _prominences(x: np.ndarray, peaks: np.ndarray) -> np.ndarray:
    """scipy _peak_prominences (wlen = None): walk outwards from the peak while the samples are <= the peak, the lowest
    sample met on each side is that side's base, prominence = peak - the higher base."""
    n = len(x)
    out = np.empty(len(peaks), dtype=np.float64)
    for m, p in enumerate(peaks):
        i, left_min = p, x[p]
        while i >= 0 and x[i] <= x[p]:
            if x[i] < left_min:
                left_min = x[i]
            i -= 1
        i, right_min = p, x[p]
        while i < n and x[i] <= x[p]:
            if x[i] < right_min:
                right_min = x[i]
            i += 1
        out[m] = x[p] - max(left_min, right_min)
    return out


def marker_find_peaks(levels: np.ndarray, height: float = -200.0, prominence: float = 6.0, distance: float = 3):
    """find_peaks(levels, height=, prominence=, distance=) as marker_manager.py:90-91 / 117-118 calls it: conditions in
    scipy's order - local maxima, height, distance, prominence - on the trace as float64.
    Returns (peaks, peak_heights, prominences)."""
    x = np.asarray(levels, dtype=np.float64)
    peaks = _local_maxima(x)
    peaks = peaks[x[peaks] >= height]
    peaks = peaks[_select_by_distance(peaks, x[peaks], distance)]
    prom = _prominences(x, peaks)
    ok = prom >= prominence
    return peaks[ok], x[peaks[ok]], prom[ok]


def snap_to_peak_bin(levels: np.ndarray, height: float = -200.0, prominence: float = 6.0, distance: float = 3) -> int:
    """marker_manager.py:93-97: the highest qualifying peak (first of equals), else np.argmax of the trace."""
    peaks, heights, _ = marker_find_peaks(levels, height, prominence, distance)
    if len(peaks) > 0:
        return int(peaks[int(np.argmax(heights))])
    return int(np.argmax(levels))


def snap_to_next_peak_bin(levels: np.ndarray, current_idx: int, height: float = -200.0, prominence: float = 6.0,
                          distance: float = 3) -> int:
    """marker_manager.py:120-126: first qualifying peak right of current_idx (= np.searchsorted(bins, position)),
    wrapping to the first peak; -1 when there is no peak (the reference then leaves the marker where it is)."""
    peaks, _, _ = marker_find_peaks(levels, height, prominence, distance)
    if len(peaks) == 0:
        return -1
    right = peaks[peaks > current_idx]
    return int(right[0]) if len(right) > 0 else int(peaks[0])
