"""DataProcessor: what happens to a trace between the data source and the screen, on the GPU.

Drop-in for the reference's core/display_data_processor.py (constructor `(main_window, display_manager)`,
timer entry point `update_data()`, and the private method set its test_smoke.py:222-236 asserts).  The
arithmetic that belongs to the hot path - calibration offset (:317-327 there), tare collect / subtract
(:329-369), max / min hold (:371-395) - runs in `trace_update_kernel` through a `TraceState`
(`tdsa_trace_update`, include/tdsa_hip.h); the attributes other code reads on the window object
(`live_power_levels`, `max_power_levels`, `min_power_levels`, `baseline_power_levels`) are refreshed
from the device after every frame.

Scope notes
  * The two hold traces are independent buffers.  The reference lets them alias ONE ndarray when both
    are switched on for the same first frame (SURVEY.md 8(a) quirk ii), after which both just follow
    the live trace; that accident is pinned in the oracle (golden max_hold_both / min_hold_both) and is
    reproduced only on request: DataProcessor(..., reference_hold_alias=True).
  * The GUI feeds around the spectrum path (sweep traces, constellation + EVM read-out, zero-span with
    its rise / fall trigger, peak-list read-out) behave as the reference's do, on the host: they are
    per-tick scalar work with no GPU side; batch users go through `analytics.py` instead.
"""
import logging
import time
from typing import Callable, List, Optional, Tuple

import numpy as np

from .. import _native as nat
from ..datasources.base import SampleDataSource, SweepDataSource
from ..engine import TraceState
from ..utils.constants import DisplayMode, FrequencyPresets, UIConstants, format_hz
from ..utils.signal_processing import TraceAverager
from .tare_state import TareState

log = logging.getLogger(__name__)

STALE_AFTER_S = 3.0                       # a sample source that has been silent this long gets a status note
_REDRAW_ON_TIMER = (DisplayMode.TWO_D, DisplayMode.THREE_D, DisplayMode.SURFACE, DisplayMode.RIBBON,
                    DisplayMode.DENSITY)
_IQ_VIEWS = {DisplayMode.CONSTELLATION_2D: "constellation_2d_widget",
             DisplayMode.CONSTELLATION_3D: "constellation_3d_widget"}


def _plural(n: int, word: str) -> str:
    return f"{n} {word}" + ("" if n == 1 else "s")


class DataProcessor:
    def __init__(self, main_window, display_manager, gpu_device: int = 0, reference_hold_alias: bool = False):
        self.mw = main_window
        self.dm = display_manager
        self._device = gpu_device
        # True: hold traces kept the reference's way, object identity included (see _hold); default: two
        # independent running fmax / fmin traces
        self.reference_hold_alias = bool(reference_hold_alias)
        self._state: Optional[TraceState] = None          # device-side trace state, sized on first use
        self._sweep_averager = TraceAverager(device=gpu_device)
        self._sweeps_since_axis_refresh = 0
        self._fused: Optional[dict] = None                # results of the current frame's fused device call

    # ================================================================== public
    def reset_sweep_averager(self) -> None:
        self._sweep_averager.reset()

    def update_data(self) -> None:
        """One display-timer tick: fetch a frame from the current source, process it, show it."""
        mw = self.mw
        source = mw.current_source
        if source is None or mw.paused:
            return
        self._check_stale_data()
        handler = self._select_handler(source)
        if handler is None:
            return
        try:
            handler()
        except Exception as exc:                        # never let a bad frame kill the timer
            self._say(f"Error updating data: {exc}")
            log.error("update_data failed: %s", exc)

    # ================================================================== routing (GUI plumbing)
    def _select_handler(self, source) -> Optional[Callable[[], None]]:
        mw, dm = self.mw, self.dm
        sampled = isinstance(source, SampleDataSource)
        if sampled and getattr(dm, "zero_span_active", False):
            return self._process_zero_span_data
        if sampled and getattr(mw, "analysis_mode", None) == "constellation":
            return self._constellation_tick
        if self._get_active_widget() is None:
            return None
        if sampled:
            return lambda: self._spectrum_tick(self._process_sample_data)
        if isinstance(source, SweepDataSource):
            return lambda: self._spectrum_tick(self._process_sweep_data)
        self._say(f"Invalid source type: {type(source)}")
        return None

    def _constellation_tick(self) -> None:
        self._process_constellation_data()
        self.mw.marker_manager.update()

    def _spectrum_tick(self, produce: Callable[[], None]) -> None:
        produce()
        mw = self.mw
        if mw.current_stacked_index == DisplayMode.WATERFALL:
            seconds_per_row = self.dm._calc_time_per_row()
            if seconds_per_row > 0:
                mw.waterfall_widget.set_time_per_row(seconds_per_row)
            self._dispatch_widget_data(self._get_active_widget())
        else:
            self._refresh_display()

    def _get_active_widget(self):
        lookup = self.dm.DISPLAY_WIDGETS_MAP.get(self.mw.current_stacked_index)
        return None if lookup is None else lookup(self.mw)

    def _refresh_display(self) -> None:
        if self.mw.current_stacked_index in _REDRAW_ON_TIMER:
            widget = self._get_active_widget()
            if widget is not None:
                self._dispatch_widget_data(widget)

    def _dispatch_widget_data(self, widget) -> None:
        mw = self.mw
        if mw.live_power_levels is None or mw.frequency_bins is None:
            return
        clone = getattr(mw, "popout_clone_widget", None) if getattr(mw, "is_popped_out", False) else None
        target = clone if clone else (widget if widget.isVisible() else None)
        try:
            if target is not None:
                target.update_widget_data(mw.live_power_levels, mw.max_power_levels, mw.frequency_bins,
                                          mw.min_power_levels)
            mw.marker_manager.update()
        except Exception as exc:
            self._say(f"Error updating display: {exc}")
            log.error("widget update failed: %s", exc)

    def _check_stale_data(self) -> None:
        mw = self.mw
        src = mw.current_source
        if mw.live_power_levels is None or not isinstance(src, SampleDataSource) or src.last_data_time <= 0:
            return
        silent_for = time.monotonic() - src.last_data_time
        if silent_for > STALE_AFTER_S:
            self._say(f"No data for {silent_for:.1f}s — source may have stopped")

    def _say(self, text: str) -> None:
        self.mw.status_label.setText(text)

    # ================================================================== the sample path
    def _process_sample_data(self) -> None:
        mw = self.mw
        levels, axis = mw.current_source.get_power_levels()
        if isinstance(levels, tuple):                     # stereo microphone: (left, right) dB traces
            self._stereo_frame(levels, axis)
            return
        if levels is None or len(levels) == 0:
            return
        mw.frequency_bins = axis
        # cal offset + tare + both holds in ONE tdsa_trace_update launch; the four methods the reference
        # calls next (kept for API parity, test_smoke.py:222-236) are served from this frame's results
        self._fused = self._fuse_frame(levels)
        try:
            trace = self._apply_tare(self._apply_cal_offset(levels))
            mw.live_power_levels = trace
            self._update_max_hold(trace)
            self._update_min_hold(trace)
        finally:
            self._fused = None
        self._update_duty_cycle(trace)
        self._update_peak_list(axis, trace)

    def _stereo_frame(self, pair: Tuple[np.ndarray, np.ndarray], axis) -> None:
        left, right = pair
        if left is None or len(left) == 0:
            return
        left, right = self._apply_cal_offset(left), self._apply_cal_offset(right)
        self.mw.frequency_bins = axis
        self.mw.live_power_levels = (left, right)
        self._update_max_hold(left)                       # the hold traces follow the left channel
        self._update_min_hold(left)

    # ------------------------------------------------------------------ device-side trace state
    def _trace_for(self, n_bins: int) -> TraceState:
        if self._state is not None and self._state.n != n_bins:
            self._state.close()
            self._state = None
        if self._state is None:
            self._state = TraceState(n_bins, device=self._device)
        return self._state

    def _cal_offset_value(self) -> float:
        cal = getattr(self.mw, "calibration_manager", None)
        kind = self.mw.source_manager.last_source_type if cal is not None else None
        return float(cal.get_offset(kind)) if kind else 0.0

    def _hold_plan(self, n_bins: int, attr: str, enabled: bool):
        """(wanted, must_reset) for one hold trace whose host copy is `attr` on the window object."""
        held = getattr(self.mw, attr)
        fits = held is not None and held.shape == (n_bins,)
        return bool(enabled), bool(enabled) and not fits

    def _fuse_frame(self, levels: np.ndarray) -> Optional[dict]:
        """Everything _process_sample_data does to one frame between the source and the window object
        (display_data_processor.py:177-181 of the reference: + cal offset, tare collect / subtract, fmax /
        fmin hold) as ONE device call.  Returns None when the frame needs the step-by-step route (a tare
        baseline of another length has to be cleared first)."""
        mw, dm = self.mw, self.dm
        n = len(levels)
        run = dm.tare_state
        collecting = bool(run.collecting)
        baseline = mw.baseline_power_levels if mw.tare_active else None
        if baseline is not None and np.shape(baseline) != np.shape(levels):
            return None
        want_max, want_min = bool(dm.max_peak_search_enabled), bool(mw.min_hold_enabled)
        if self.reference_hold_alias:                     # that mode keeps its hold traces on the host (_hold)
            want_max = want_min = False
        if not (collecting or baseline is not None or want_max or want_min) and self._cal_offset_value() == 0.0:
            return None                                   # nothing to do to this frame: no device call at all,
                                                          # the frame passes through untouched as in the reference
        gpu = self._trace_for(n)
        if collecting and run.count == 0:
            gpu.reset(nat.RESET_TARE)                     # a new run starts from an empty accumulator
        subtract = baseline is not None
        if subtract and not gpu.tare_is_active():
            gpu.set_tare_baseline(baseline)               # baseline restored from a preset, not collected here
        want_max, reset_max = self._hold_plan(n, "max_power_levels", dm.max_peak_search_enabled)
        want_min, reset_min = self._hold_plan(n, "min_power_levels", mw.min_hold_enabled)
        if self.reference_hold_alias:                     # that mode keeps its hold traces on the host (_hold)
            want_max = want_min = reset_max = reset_min = False
        if reset_max or reset_min:                        # first frame is adopted (NaN -> -500 / +500)
            gpu.reset((nat.RESET_HOLD_MAX if reset_max else 0) | (nat.RESET_HOLD_MIN if reset_min else 0))
        offset_db = self._cal_offset_value()
        live, mx, mn, complete = gpu.update(levels, cal_offset_db=offset_db, tare_collect=collecting,
                                            tare_total=UIConstants.TARE_NUM_SAMPLES, tare_subtract=subtract,
                                            hold_max=want_max, hold_min=want_min)
        touched = offset_db != 0.0 or collecting or subtract
        return dict(raw=levels, live=live if touched else levels, max=mx, min=mn, complete=complete,
                    collecting=collecting, subtract=subtract, offset=offset_db)

    def _apply_cal_offset(self, power_levels: np.ndarray) -> np.ndarray:
        fused = getattr(self, "_fused", None)
        if fused is not None and power_levels is fused["raw"]:
            return power_levels                           # the offset is inside the fused result (_apply_tare)
        offset_db = self._cal_offset_value()
        if offset_db == 0.0:
            return power_levels                           # untouched object, as the reference returns it
        return self._trace_for(len(power_levels)).update(power_levels, cal_offset_db=offset_db)[0]

    def _tare_bookkeeping(self, gpu: TraceState, complete: bool) -> None:
        mw, dm = self.mw, self.dm
        run = dm.tare_state
        run.count += 1
        left = UIConstants.TARE_NUM_SAMPLES - run.count
        self._say(f"Collecting normalisation baseline... {_plural(left, 'frame')} remaining")
        if complete:
            mw.baseline_power_levels = gpu.tare_baseline()
            mw.tare_active = True
            dm.tare_state = TareState()
            dm._update_tare_button_label("Clear\nNormalisation")
            self._say("Tare active — baseline captured")

    def _apply_tare(self, power_levels: np.ndarray) -> np.ndarray:
        fused = getattr(self, "_fused", None)
        if fused is not None and power_levels is fused["raw"]:
            if fused["collecting"]:
                self._tare_bookkeeping(self._state, fused["complete"])
            return fused["live"]
        mw, dm = self.mw, self.dm
        run = dm.tare_state
        collecting = bool(run.collecting)
        baseline = mw.baseline_power_levels if mw.tare_active else None
        if not collecting and baseline is None:
            return power_levels                           # nothing to do: the frame object passes through, no device call
        gpu = self._trace_for(len(power_levels))
        if collecting and run.count == 0:
            gpu.reset(nat.RESET_TARE)                     # a new run starts from an empty accumulator
        if baseline is not None and baseline.shape != power_levels.shape:
            dm._clear_tare()                              # the span changed under an active baseline
            gpu.reset(nat.RESET_TARE)
            self._say("Tare cleared — frequency range changed")
            baseline = mw.baseline_power_levels if mw.tare_active else None
            if not collecting:
                return power_levels
        subtract = baseline is not None
        if not (collecting or subtract):
            return power_levels
        if subtract and not gpu.tare_is_active():
            gpu.set_tare_baseline(baseline)               # baseline restored from a preset, not collected here
        live, _, _, complete = gpu.update(power_levels, tare_collect=collecting,
                                          tare_total=UIConstants.TARE_NUM_SAMPLES, tare_subtract=subtract)
        if collecting:
            self._tare_bookkeeping(gpu, complete)
        return live

    def _hold(self, trace: np.ndarray, *, attr: str, enabled: bool, reset_bit: int, which: str) -> None:
        """Common part of the two hold traces: `attr` on the window is the host copy of the device trace."""
        mw = self.mw
        held = getattr(mw, attr)
        fits = held is not None and held.shape == trace.shape
        if not enabled:
            if held is not None and not fits:
                setattr(mw, attr, None)                   # a stale trace of another length is dropped
            return
        if self.reference_hold_alias:
            # the reference's own bookkeeping, object identity included: a clean first frame is adopted as the very
            # array (so two holds adopting on one frame share it, and it is the live trace of that frame), later
            # frames are folded in place (display_data_processor.py:371-395 + _nan_safe :473-480, SURVEY quirk ii)
            if not fits:
                setattr(mw, attr, self._nan_safe(trace, -500.0 if which == "hold_max" else 500.0))
            else:
                (np.fmax if which == "hold_max" else np.fmin)(held, trace, out=held)
            return
        fused = getattr(self, "_fused", None)
        if fused is not None and trace is fused["live"]:
            setattr(mw, attr, fused["max" if which == "hold_max" else "min"])
            return
        gpu = self._trace_for(len(trace))
        if not fits:
            gpu.reset(reset_bit)                          # first frame is adopted (NaN -> -500 / +500)
        _, mx, mn, _ = gpu.update(trace, **{which: True})
        setattr(mw, attr, mx if which == "hold_max" else mn)

    def _update_max_hold(self, power_levels: np.ndarray) -> None:
        self._hold(power_levels, attr="max_power_levels", enabled=bool(self.dm.max_peak_search_enabled),
                   reset_bit=nat.RESET_HOLD_MAX, which="hold_max")

    def _update_min_hold(self, power_levels: np.ndarray) -> None:
        self._hold(power_levels, attr="min_power_levels", enabled=bool(self.mw.min_hold_enabled),
                   reset_bit=nat.RESET_HOLD_MIN, which="hold_min")

    # ================================================================== per-tick scalars (host side)
    def _update_duty_cycle(self, power_levels: np.ndarray) -> None:
        if getattr(self.dm, "duty_cycle_enabled", False):
            analyser = self.dm.duty_cycle_analyser
            analyser.update_from_power(power_levels)
            label = getattr(self.mw, "marker_readout_label", None)
            if label is not None:
                label.setText(analyser.get_readout())

    def _update_peak_list(self, freq_bins: np.ndarray, power_levels: np.ndarray) -> None:
        if not getattr(self.dm, "peak_list_enabled", False):
            return
        plot = self.mw.two_d_widget
        if not hasattr(plot, "set_peak_list"):
            return
        found = self._find_top_peaks(freq_bins, power_levels, n=5, min_sep_bins=max(10, len(freq_bins) // 50),
                                     min_excursion_db=getattr(self.mw, "peak_excursion", 10.0))
        plot.set_peak_list(found)
        label = getattr(self.mw, "marker_readout_label", None)
        if label is not None:                             # numbered table under the plot, strongest first
            label.setText("\n".join(f"{rank}: Freq: {format_hz(f)}, Power: {p:.1f} dBm"
                                    for rank, (f, p) in enumerate(found, start=1)))

    @staticmethod
    def _find_top_peaks(freq_bins, power, n: int = 5, min_sep_bins: int = 10,
                        min_excursion_db: float = 10.0) -> List[Tuple[float, float]]:
        """Up to n (frequency, level) pairs, strongest first.  A strict local maximum is taken unless an
        already taken one is nearer than min_sep_bins or the lowest point between the two is less than
        min_excursion_db below either (device version for whole batches: analytics.rows_top_peaks)."""
        level = np.asarray(power)
        if level.size < 3:
            return []
        inner = level[1:-1]
        candidates = 1 + np.flatnonzero((inner > level[:-2]) & (inner > level[2:]))
        taken: List[int] = []
        for c in candidates[np.argsort(level[candidates])[::-1]]:
            if len(taken) == n:
                break
            for t in taken:
                a, b = sorted((int(c), t))
                floor = float(level[a:b + 1].min())
                if b - a < min_sep_bins or level[c] - floor < min_excursion_db or level[t] - floor < min_excursion_db:
                    break
            else:
                taken.append(int(c))
        return [(float(freq_bins[i]), float(level[i])) for i in taken]

    @staticmethod
    def _nan_safe(arr: np.ndarray, fill: float) -> np.ndarray:
        """NaN -> fill; an array without NaN comes back as the same object."""
        bad = np.isnan(arr)
        if not bad.any():
            return arr
        return np.where(bad, fill, arr).astype(arr.dtype, copy=False)

    # ================================================================== feeds outside the spectrum path
    def _process_sweep_data(self) -> None:
        """Sweep sources hand over finished dB traces (external tools; no FFT on our side)."""
        mw = self.mw
        trace = mw.current_source.get_data()
        if trace is None or len(trace) == 0:
            return
        span = mw.frequency
        if span.start is None or span.stop is None:      # lost range: fall back to the HackRF default span
            log.warning("frequency range unset, restoring the default span")
            span.set_start_stop(FrequencyPresets.HACKRF_DEFAULT_START, FrequencyPresets.HACKRF_DEFAULT_STOP)
        mw.frequency_bins = np.linspace(span.start, span.stop, len(trace))
        trace = self._apply_cal_offset(trace)
        if np.isnan(trace).all():
            return
        if self._sweep_averager.is_active:               # averaging is done on linear power
            mean_lin = self._sweep_averager.process(np.power(10.0, np.asarray(trace, dtype=np.float64) / 10.0))
            trace = 10.0 * np.log10(np.maximum(mean_lin, 1e-30))
        mw.live_power_levels = trace
        self._update_max_hold(trace)
        self._update_min_hold(trace)
        self._update_peak_list(mw.frequency_bins, trace)
        self._sweeps_since_axis_refresh += 1
        if self._sweeps_since_axis_refresh >= UIConstants.SWEEP_RATE_UPDATE_INTERVAL:
            self._sweeps_since_axis_refresh = 0
            mw.frequency_manager.update_frequency_values()

    def _process_constellation_data(self) -> None:
        mw = self.mw
        iq = mw.current_source.read_samples_only()
        if iq is None or len(iq) == 0:
            return
        attr = _IQ_VIEWS.get(mw.current_stacked_index)
        if attr is None:                                  # constellation mode but another page is up: switch to it
            self.dm.set_display(mw._resolve_display_index(), UIConstants.BUTTON_ACTIVE_STYLE, None)
            return
        view = getattr(mw, attr, None)
        if view is None:
            return
        view.update_iq_data(iq)
        label = getattr(mw, "marker_readout_label", None)
        if label is None:
            return
        evm = getattr(view, "last_evm_rms", None)         # error vector magnitude the widget just measured
        if evm is None or evm <= 0:
            label.setText("")
            return
        scheme = getattr(self.dm, "constellation_modulation", "").upper()
        label.setText(f"EVM  {scheme}\n{evm * 100.0:.1f}%  ({20.0 * np.log10(evm):+.1f} dB)")

    @staticmethod
    def _trigger_start(history: np.ndarray, shown: int, mode: str, level: float) -> Optional[int]:
        """Index where a triggered zero-span trace starts: the LAST crossing of `level` (upward for "rise",
        downward otherwise) inside the 8 display windows that precede the newest one; None = no crossing."""
        last_start = len(history) - shown
        first = max(0, last_start - 8 * shown)
        if last_start <= first:
            return None
        before, after = history[first:last_start - 1], history[first + 1:last_start]
        hit = (before < level) & (after >= level) if mode == "rise" else (before >= level) & (after < level)
        where = np.flatnonzero(hit)
        return None if where.size == 0 else first + int(where[-1]) + 1

    def _process_zero_span_data(self) -> None:
        """Time-domain view: keep the last two seconds of (real) samples, show the configured window,
        free running or aligned to the most recent trigger crossing."""
        mw, dm = self.mw, self.dm
        block = mw.current_source.read_samples_only()
        if block is None or len(block) == 0:
            return
        if block.ndim == 2:
            block = block.mean(axis=1)                    # stereo -> mono
        block = np.real(block).ravel().astype(np.float32)
        rate = float(getattr(mw.current_source, "sample_rate", 44100))
        history = block if dm.zero_span_buffer is None else np.concatenate((dm.zero_span_buffer, block))
        dm.zero_span_buffer = history = history[-int(2.0 * rate):]
        n_shown = max(int(dm.zero_span_time_window * rate), 4)
        mode = getattr(dm, "zero_span_trigger_mode", "free_run")
        start = None
        if len(history) >= n_shown and mode != "free_run":
            start = self._trigger_start(history, n_shown, mode, getattr(dm, "zero_span_trigger_level", 0.0))
        shown = history[-n_shown:] if start is None else history[start:start + n_shown]
        mw.zero_span_widget.update_zero_span_data(np.arange(shown.size, dtype=np.float32) / rate, shown)
