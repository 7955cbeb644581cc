"""Host-side GUI feeds of DataProcessor (no GPU): zero-span trigger, peak-list and EVM read-outs, sweep
range recovery - against vectors captured from the imported reference (tests/golden/gui_feeds.npz)."""
import os
import types

import numpy as np
import pytest

from topdogspectrumanalyser_amd.core.display_data_processor import DataProcessor
from topdogspectrumanalyser_amd.utils.constants import DisplayMode, FrequencyPresets, UIConstants


class Label:
    text = None

    def setText(self, s):
        self.text = s


def _bare(mw, dm):
    dp = DataProcessor.__new__(DataProcessor)         # no device objects: these paths never touch the GPU
    dp.mw, dp.dm = mw, dm
    dp._fused = None
    dp._sweeps_since_axis_refresh = 0
    dp.reference_hold_alias = False
    return dp


def test_zero_span_trigger_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "gui_feeds.npz"))

    class W:
        def update_zero_span_data(self, t, y):
            self.t, self.y = np.array(t), np.array(y)

    src = types.SimpleNamespace(sample_rate=float(g["zs_rate"]), block=None)
    src.read_samples_only = lambda: src.block
    mw = types.SimpleNamespace(current_source=src, zero_span_widget=W())
    dm = types.SimpleNamespace(zero_span_buffer=None, zero_span_time_window=float(g["zs_window"]),
                               zero_span_trigger_mode="free_run", zero_span_trigger_level=0.0)
    dp = _bare(mw, dm)
    for i, mode in enumerate(g["zs_modes"]):
        dm.zero_span_trigger_mode, dm.zero_span_trigger_level = str(mode), float(g["zs_levels"][i])
        src.block = g[f"zs_block_{i}"]
        dp._process_zero_span_data()
        want = g[f"zs_shown_{i}"]
        assert np.array_equal(mw.zero_span_widget.y, want), (i, mode)
        assert np.array_equal(mw.zero_span_widget.t, np.arange(len(want), dtype=np.float32) / float(g["zs_rate"]))


def test_peak_list_readout_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "gui_feeds.npz"))

    class TwoD:
        def set_peak_list(self, peaks):
            self.peaks = peaks

    mw = types.SimpleNamespace(two_d_widget=TwoD(), marker_readout_label=Label(), peak_excursion=8.0)
    dp = _bare(mw, types.SimpleNamespace(peak_list_enabled=True))
    dp._update_peak_list(g["pl_bins"], g["pl_trace"])
    assert mw.marker_readout_label.text == str(g["pl_text"])
    assert np.array_equal(np.array(mw.two_d_widget.peaks, dtype=np.float64), g["pl_peaks"])
    dp.dm.peak_list_enabled = False
    mw.marker_readout_label.text = "untouched"
    dp._update_peak_list(g["pl_bins"], g["pl_trace"])
    assert mw.marker_readout_label.text == "untouched"


def test_constellation_evm_readout_and_fallback(golden_dir):
    g = np.load(os.path.join(golden_dir, "gui_feeds.npz"))
    for evm, want in zip(g["evm_values"], g["evm_texts"]):
        view = types.SimpleNamespace(last_evm_rms=None if np.isnan(evm) else float(evm), got=None)
        view.update_iq_data = lambda s, v=view: setattr(v, "got", s)
        lab = Label()
        mw = types.SimpleNamespace(current_source=types.SimpleNamespace(read_samples_only=lambda: np.ones(8, np.complex64)),
                                   current_stacked_index=DisplayMode.CONSTELLATION_2D, constellation_2d_widget=view,
                                   marker_readout_label=lab)
        dp = _bare(mw, types.SimpleNamespace(constellation_modulation="qpsk"))
        dp._process_constellation_data()
        assert view.got is not None and lab.text == str(want)
    # constellation mode while another page is showing: the display is switched, nothing is drawn
    calls = []
    mw = types.SimpleNamespace(current_source=types.SimpleNamespace(read_samples_only=lambda: np.ones(8, np.complex64)),
                               current_stacked_index=DisplayMode.TWO_D, _resolve_display_index=lambda: 7)
    dm = types.SimpleNamespace(set_display=lambda *a: calls.append(a))
    _bare(mw, dm)._process_constellation_data()
    assert calls == [(7, UIConstants.BUTTON_ACTIVE_STYLE, None)]


def test_sweep_path_recovers_a_lost_frequency_range():
    """display_data_processor.py:193-201 of the reference: start/stop None -> default span, no exception."""
    class Span:
        start = stop = None

        def set_start_stop(self, a, b):
            self.start, self.stop = a, b

    trace = np.linspace(-90.0, -40.0, 64)
    mw = types.SimpleNamespace(current_source=types.SimpleNamespace(get_data=lambda: trace), frequency=Span(),
                               calibration_manager=None, source_manager=types.SimpleNamespace(last_source_type=None),
                               live_power_levels=None, max_power_levels=None, min_power_levels=None, frequency_bins=None,
                               min_hold_enabled=False, frequency_manager=types.SimpleNamespace(update_frequency_values=lambda: None))
    dm = types.SimpleNamespace(max_peak_search_enabled=False, peak_list_enabled=False)
    dp = _bare(mw, dm)
    dp._sweep_averager = types.SimpleNamespace(is_active=False)
    dp._process_sweep_data()
    assert mw.frequency.start == FrequencyPresets.HACKRF_DEFAULT_START
    assert mw.frequency.stop == FrequencyPresets.HACKRF_DEFAULT_STOP
    assert mw.frequency_bins[0] == 2400e6 and mw.frequency_bins[-1] == 2500e6 and len(mw.frequency_bins) == 64
    assert mw.live_power_levels is trace


def test_sharding_rejects_order_dependent_modes():
    from topdogspectrumanalyser_amd.sharding import process_sharded
    iq = np.zeros(4096, dtype=np.int8)
    w = np.ones(1024, dtype=np.float32)
    with pytest.raises(ValueError, match="averaging"):
        process_sharded(iq, 1024, 1024, [0, 0], w, avg=("exp", 4))
    with pytest.raises(ValueError, match="DC remover"):
        process_sharded(iq, 1024, 1024, [0, 0], w, dc_alpha=0.25)


def test_plain_display_frame_makes_no_device_call():
    """with no calibration offset, no tare run / baseline and no hold wanted, a displayed frame passes through
    _process_sample_data untouched, as in the reference (display_data_processor.py:153-183) - and WITHOUT a device call:
    on this GPU-less box any tdsa_* call would raise, so completing at all proves it; the frame object itself becomes
    the live trace."""
    from topdogspectrumanalyser_amd.core.tare_state import TareState
    trace = np.linspace(-90.0, -20.0, 512).astype(np.float32)
    axis = np.linspace(88e6, 108e6, 512)
    src = types.SimpleNamespace(get_power_levels=lambda: (trace, axis))
    cal = types.SimpleNamespace(get_offset=lambda kind: 0.0)
    mw = types.SimpleNamespace(current_source=src, live_power_levels=None, max_power_levels=None, min_power_levels=None,
                               frequency_bins=None, min_hold_enabled=False, tare_active=False, baseline_power_levels=None,
                               calibration_manager=cal, source_manager=types.SimpleNamespace(last_source_type="hackrf_samples"),
                               status_label=Label())
    dm = types.SimpleNamespace(tare_state=TareState(), max_peak_search_enabled=False, duty_cycle_enabled=False,
                               peak_list_enabled=False)
    dp = _bare(mw, dm)
    dp._state, dp._device = None, 0
    for _ in range(3):
        dp._process_sample_data()
    assert mw.live_power_levels is trace and mw.frequency_bins is axis
    assert mw.max_power_levels is None and mw.min_power_levels is None and dp._state is None
