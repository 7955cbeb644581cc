"""Host-side drop-in surface (no GPU needed): the classes, methods and error conventions the reference
exposes on this path, mirroring what the reference's own script tests assert
(test_smoke.py:137-175,222-309, test_fft_size_changes.py:27-73, test_rbw_calculation.py:54-76)."""
import inspect

import numpy as np
import pytest

import topdogspectrumanalyser_amd as pkg
from topdogspectrumanalyser_amd.core.display_data_processor import DataProcessor
from topdogspectrumanalyser_amd.datasources.base import SampleDataSource, SweepDataSource
from topdogspectrumanalyser_amd.utils.constants import DSPConstants, FFTSize, UIConstants
from topdogspectrumanalyser_amd.utils.synthetic import synth_iq_int8


def test_data_processor_method_set():
    expected = ["_process_sample_data", "_process_sweep_data", "_process_constellation_data",
                "_process_zero_span_data", "_find_top_peaks", "_nan_safe", "_apply_tare", "_apply_cal_offset",
                "_update_max_hold", "_update_min_hold", "_update_duty_cycle", "_update_peak_list",
                "_dispatch_widget_data", "_refresh_display", "update_data", "reset_sweep_averager"]
    for m in expected:
        assert callable(getattr(DataProcessor, m)), m
    assert list(inspect.signature(DataProcessor.__init__).parameters)[:3] == ["self", "main_window",
                                                                              "display_manager"]


def test_sample_data_source_abc():
    abstract = SampleDataSource.__abstractmethods__
    assert {"start", "stop", "get_power_levels", "sample_count", "update_frequency",
            "update_centre_frequency"} == set(abstract)
    for m in ("get_raw_samples", "read_samples_only", "_store_raw", "set_psd_mode", "set_averaging",
              "reset_averaging"):
        assert callable(getattr(SampleDataSource, m))
    assert {"start", "stop", "get_data"} == set(SweepDataSource.__abstractmethods__)
    with pytest.raises(TypeError):
        SampleDataSource()


def test_plugin_table():
    assert set(pkg.SOURCE_CLASSES) == {"rtl_samples", "hackrf_samples", "microphone_samples"}
    for cls in pkg.SOURCE_CLASSES.values():
        assert issubclass(cls, SampleDataSource)


@pytest.mark.parametrize("size", [512, 1024, 2048, 4096])
def test_fft_size_changes(size):
    h = pkg.HackrfSamplesDataSource(sample_rate=20_000_000, centre_freq=100_000_000)
    h.sample_count = size
    assert h.num_samples == size and h.sample_count == size
    h._allocate_fft_resources()
    assert len(h._window) == size and h._window.dtype == np.float32
    assert abs(float(np.mean(h._window.astype(np.float64) ** 2)) - 1.0) < 1e-6     # unit mean power
    r = pkg.RtlSamplesDataSource(sample_rate=2_400_000, centre_freq=100_000_000)
    r.set_window_type("hamming")
    r.sample_count = size
    assert r.fft_size == size and len(r.window) == size
    if size != 1024:
        assert np.array_equal(r.window, np.hanning(size))     # reference quirk: size change -> Hann again
    a = pkg.MicrophoneSamplesDataSource()
    a.sample_count = size
    assert a.fft_size == size and len(a.window) == size and len(a.get_power_levels()[1]) == size // 2 + 1


def test_rbw():
    for fs, n, rbw in ((20e6, 1024, 19531.25), (2e6, 1024, 1953.125), (44100, 1024, 43.06640625)):
        h = pkg.HackrfSamplesDataSource(sample_rate=int(fs), centre_freq=0)
        h.num_samples = n
        h._allocate_fft_resources()
        assert abs((h._freq_bins[1] - h._freq_bins[0]) - rbw) < 1e-9


def test_error_conventions_without_hardware():
    h = pkg.HackrfSamplesDataSource(sample_rate=20_000_000, centre_freq=100_000_000)
    with pytest.raises(RuntimeError):
        h.start()
    p, f = h.get_power_levels()                 # never raises: zeros + axis (hackrf_samples.py:342-347)
    assert p.shape == (1024,) and not p.any() and f.shape == (1024,)
    with pytest.raises(ValueError):
        h.set_num_samples(0)
    with pytest.raises(ValueError):
        h.set_gains(lna_gain=41)
    with pytest.raises(ValueError):
        h.set_gains(vga_gain=63)
    r = pkg.RtlSamplesDataSource(sample_rate=2_000_000, centre_freq=100_000_000)
    with pytest.raises(RuntimeError):
        r.start()
    p, f = r.get_power_levels()
    assert p.shape == (1024,) and not p.any()
    assert f[0] == 100_000_000 - 1_000_000 and f[-1] == 100_000_000 + 1_000_000
    a = pkg.MicrophoneSamplesDataSource()
    with pytest.raises(RuntimeError):
        a.start(None)                            # no sounddevice / PortAudio here
    p, f = a.get_power_levels()
    assert np.all(p == -120.0) and len(p) == 513


def test_trace_averager_host_config():
    av = pkg.TraceAverager()
    assert av.mode == "off" and av.n == 1 and not av.is_active
    x = np.ones(8, dtype=np.float32)
    assert av.process(x) is x                    # pass-through returns the very same object
    av.set_mode("exp", 0)
    assert av.n == 1 and not av.is_active        # n = max(1, n)
    assert av.process(x) is x
    av.set_mode("lin", 16)
    assert av.is_active and av.mode == "lin" and av.n == 16
    with pytest.raises(ValueError):
        av.set_mode("median", 3)


def test_tare_state_and_constants():
    ts = pkg.TareState()
    assert ts.collecting is False and ts.buffer is None and ts.count == 0
    assert DSPConstants.LOG_FLOOR == 1e-12 and DSPConstants.POWER_LOG_FLOOR == 1e-10
    assert UIConstants.TARE_NUM_SAMPLES == 32
    assert FFTSize.get_min() == 512 and FFTSize.get_max() == 8192 and FFTSize.is_valid(4096)


def test_find_top_peaks_and_nan_safe():
    f = np.linspace(0, 1e6, 1000)
    p = np.full(1000, -100.0)
    for centre, amp in ((200, -20.0), (500, -10.0), (800, -30.0)):
        p = np.maximum(p, amp - 0.05 * (np.arange(1000) - centre) ** 2)
    peaks = DataProcessor._find_top_peaks(f, p, n=5, min_sep_bins=20, min_excursion_db=10.0)
    assert [round(pw) for _, pw in peaks] == [-10, -20, -30]
    assert [int(round(fr / (1e6 / 999))) for fr, _ in peaks] == [500, 200, 800]
    assert DataProcessor._find_top_peaks(f[:2], p[:2]) == []
    a = np.arange(4.0)
    assert DataProcessor._nan_safe(a, -500.0) is a
    b = np.array([1.0, np.nan])
    out = DataProcessor._nan_safe(b, -500.0)
    assert out is not b and out[1] == -500.0 and np.isnan(b[1])


def test_synthetic_generator_matches_oracle_copy():
    from oracle import spectrum_oracle as so
    assert np.array_equal(synth_iq_int8(5000, 1024, 7), so.synth_iq_int8(5000, 1024, 7))


def test_replay_devices_unpack_convention():
    from topdogspectrumanalyser_amd.datasources.replay import ReplayHackRF, ReplayRtlSdr
    from oracle import spectrum_oracle as so
    iq = synth_iq_int8(4096, 1024, 1)
    d = ReplayHackRF(iq)
    x = d.read_samples(1024)
    assert x.dtype == np.complex64 and np.array_equal(x, so.unpack_iq_int8(iq)[:1024])
    assert np.array_equal(d.read_samples(1024), so.unpack_iq_int8(iq)[1024:2048])
    r = ReplayRtlSdr(iq, sample_rate=2e6, center_freq=1e8)
    assert r.get_sample_rate() == 2e6 and r.get_center_freq() == 1e8 and len(r.read_samples(512)) == 512


def test_unsupported_fft_size_never_raises():
    """set_num_samples() is unbounded in the reference (hackrf_samples.py:392-405) and get_power_levels()
    never raises (:342-355): a size the device library has no plan for yields zeros + the frequency axis."""
    from topdogspectrumanalyser_amd.datasources.replay import ReplayHackRF
    iq = synth_iq_int8(65536, 1024, 3)
    h = pkg.HackrfSamplesDataSource(sample_rate=20_000_000, centre_freq=100_000_000,
                                    device_factory=lambda: ReplayHackRF(iq))
    h.start()
    try:
        h.set_num_samples(1500000)               # above 2^20: the library has no plan for it
        p, f = h.get_power_levels()
        assert p.shape == (1500000,) and not p.any() and f.shape == (1500000,)
        assert abs((f[1] - f[0]) - 20e6 / 1500000) < 1e-6
        p2, _ = h.get_power_levels()             # and again (the failure is remembered, not re-raised)
        assert p2.shape == (1500000,) and not p2.any()
    finally:
        h.stop()
