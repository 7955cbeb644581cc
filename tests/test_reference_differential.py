"""Live differential tests against the IMPORTED reference, for the host-side pieces of DataProcessor that carry no GPU
work (peak search, NaN clean-up, zero-span view with its trigger, peak-list read-out).  They run where
/root/reference exists (the build container) and are skipped elsewhere (the GPU box has no reference); the committed
fixtures of tests/golden/ pin the same functions there.  Nothing is copied: both implementations are driven with
the same seeded random inputs through plain stub objects and must agree exactly."""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import pytest

REF = os.environ.get("TDSA_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "core")), reason="reference tree not present")


def _forget_reference_modules(before, stubs=()):
    """Drop what an import of the reference left in sys.modules: its own modules (anything whose file lies under REF)
    and the stand-in modules named in `stubs`.  Third-party modules first imported on the way (scipy.fft, logging
    handlers ...) STAY cached - dropping C-extension submodules made later imports order dependent."""
    ref_root = os.path.realpath(REF)
    for name in set(sys.modules) - set(before):
        mod = sys.modules.get(name)
        origin = getattr(mod, "__file__", None) or ""
        paths = [q for q in (getattr(mod, "__path__", None) or []) if isinstance(q, str)]
        inside = [os.path.realpath(q) for q in ([origin] if origin else []) + paths]
        if any(q == ref_root or q.startswith(ref_root + os.sep) for q in inside) or name in stubs or \
                name.split(".")[0] in stubs:
            del sys.modules[name]
    for name in stubs:
        sys.modules.pop(name, None)



@pytest.fixture(scope="module")
def ref_dp():
    sys.dont_write_bytecode = True                       # never leave __pycache__ in the reference tree
    before = set(sys.modules)
    mocked = [m for m in ("hackrf", "rtlsdr", "sounddevice") if m not in sys.modules]
    for m in mocked:
        sys.modules[m] = MagicMock()
    sys.path.insert(0, REF)
    try:
        from core.display_data_processor import DataProcessor as RefDP
    finally:
        sys.path.remove(REF)
        _forget_reference_modules(before, mocked)
    return RefDP


@pytest.fixture(scope="module")
def our_dp():
    from topdogspectrumanalyser_amd.core.display_data_processor import DataProcessor
    return DataProcessor


class _Label:
    text = None

    def setText(self, s):
        self.text = s


def _bare(cls, mw, dm):
    dp = cls.__new__(cls)
    dp.mw, dp.dm = mw, dm
    for k, v in dict(_fused=None, _sweeps_since_axis_refresh=0, reference_hold_alias=False).items():
        setattr(dp, k, v)
    return dp


def test_find_top_peaks_random(ref_dp, our_dp):
    rng = np.random.default_rng(11)
    for case in range(400):
        n = int(rng.integers(3, 3000))
        p = rng.exponential(1.0, n) * 10.0 ** rng.uniform(-12, -6)
        k = np.arange(n)
        for _ in range(int(rng.integers(0, 6))):
            p += 10.0 ** rng.uniform(-9, -2) * np.sinc((k - rng.integers(0, n)) / rng.uniform(0.6, 4.0)) ** 2
        tr = (10 * np.log10(p + 1e-15)).astype(np.float32 if rng.integers(0, 2) else np.float64)
        bins = np.linspace(88e6, 108e6, n)
        kw = dict(n=int(rng.integers(1, 9)), min_sep_bins=int(rng.integers(1, 60)), min_excursion_db=float(rng.choice([3.0, 6.0, 10.0])))
        want = ref_dp._find_top_peaks(bins, tr, **kw)
        got = our_dp._find_top_peaks(bins, tr, **kw)
        assert got == want, (case, kw)


def test_nan_safe_random(ref_dp, our_dp):
    rng = np.random.default_rng(12)
    for case in range(200):
        a = rng.normal(-70, 10, int(rng.integers(1, 500))).astype(np.float32 if rng.integers(0, 2) else np.float64)
        if rng.integers(0, 2):
            a[rng.integers(0, a.size, int(rng.integers(1, 4)))] = np.nan
        fill = float(rng.choice([-500.0, 500.0]))
        want, got = ref_dp._nan_safe(a, fill), our_dp._nan_safe(a, fill)
        assert np.array_equal(got, want) and got.dtype == want.dtype
        assert (got is a) == (want is a)                 # a clean array comes back as the same object in both


def test_zero_span_random_histories(ref_dp, our_dp):
    class W:
        t = y = None

        def update_zero_span_data(self, t, y):
            self.t, self.y = np.array(t), np.array(y)

    rng = np.random.default_rng(13)
    for trial in range(30):
        rate = float(rng.choice([8000.0, 44100.0, 2_000_000.0]))
        pair = []
        for cls in (ref_dp, our_dp):
            src = types.SimpleNamespace(sample_rate=rate, block=None)
            src.read_samples_only = (lambda s=src: s.block)
            mw = types.SimpleNamespace(current_source=src, zero_span_widget=W())
            dm = types.SimpleNamespace(zero_span_buffer=None, zero_span_time_window=0.01,
                                       zero_span_trigger_mode="free_run", zero_span_trigger_level=0.0)
            pair.append((_bare(cls, mw, dm), src, mw, dm))
        for step in range(60):
            ev = rng.random()
            if ev < 0.2:
                mode, level = str(rng.choice(["free_run", "rise", "fall"])), float(rng.uniform(-0.8, 0.8))
                for _, _, _, dm in pair:
                    dm.zero_span_trigger_mode, dm.zero_span_trigger_level = mode, level
            elif ev < 0.3:
                win = float(rng.choice([0.001, 0.01, 0.05, 0.5]))
                for _, _, _, dm in pair:
                    dm.zero_span_time_window = win
            kind = int(rng.integers(0, 6))
            m = int(rng.integers(1, 5000))
            if kind == 0:
                block = None
            elif kind == 1:
                block = np.zeros(0, dtype=np.complex64)
            elif kind == 2:
                block = (rng.standard_normal((m, 2)) * 0.5).astype(np.float32)          # stereo audio block
            else:
                t = np.arange(m)
                block = (np.sin(2 * np.pi * rng.uniform(0.001, 0.2) * t + rng.uniform(0, 6)) * rng.uniform(0.1, 1.0)
                         + 0.05 * rng.standard_normal(m) + 1j * rng.standard_normal(m)).astype(np.complex64)
            for dp, src, _, _ in pair:
                src.block = None if block is None else block.copy()
                dp._process_zero_span_data()
            (rw, ow) = (pair[0][2].zero_span_widget, pair[1][2].zero_span_widget)
            assert (rw.y is None) == (ow.y is None), (trial, step)
            if rw.y is not None:
                assert np.array_equal(ow.y, rw.y) and np.array_equal(ow.t, rw.t), (trial, step)
            rb, ob = pair[0][3].zero_span_buffer, pair[1][3].zero_span_buffer
            assert (rb is None) == (ob is None) and (rb is None or np.array_equal(ob, rb)), (trial, step)


def test_peak_list_readout_random(ref_dp, our_dp):
    class TwoD:
        peaks = None

        def set_peak_list(self, peaks):
            self.peaks = peaks

    rng = np.random.default_rng(14)
    for case in range(100):
        n = int(rng.integers(16, 4096))
        tr = (rng.normal(-90, 4, n) + 40 * (rng.random(n) < 0.01)).astype(np.float32)
        bins = np.linspace(2.4e9, 2.5e9, n)
        exc = float(rng.choice([3.0, 6.0, 8.0, 10.0]))
        out = []
        for cls in (ref_dp, our_dp):
            mw = types.SimpleNamespace(two_d_widget=TwoD(), marker_readout_label=_Label(), peak_excursion=exc)
            dp = _bare(cls, mw, types.SimpleNamespace(peak_list_enabled=True))
            dp._update_peak_list(bins, tr)
            out.append((mw.marker_readout_label.text, mw.two_d_widget.peaks))
        assert out[0][0] == out[1][0], case
        assert out[0][1] == out[1][1], case


# ----------------------------------------------------------------------------------------------------------------
# the ORACLE against the imported reference on random inputs: oracle/spectrum_oracle.py (precision="ref") has to be
# np.array_equal to what the reference's own source classes return, frame by frame, through random knob histories -
# the committed fixtures pin the same thing at fixed inputs, this pins it at thousands more
# ----------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ref_sources():
    sys.dont_write_bytecode = True
    before = set(sys.modules)
    mocked = [m for m in ("hackrf", "rtlsdr", "sounddevice") if m not in sys.modules]
    for m in mocked:
        sys.modules[m] = MagicMock()
    sys.path.insert(0, REF)
    try:
        from datasources.hackrf_samples import HackrfSamplesDataSource
        from datasources.rtl_samples import RtlSamplesDataSource
        from datasources.audio_samples import MicrophoneSamplesDataSource
        from utils.signal_processing import TraceAverager
    finally:
        sys.path.remove(REF)
        _forget_reference_modules(before, mocked)
    return types.SimpleNamespace(hackrf=HackrfSamplesDataSource, rtl=RtlSamplesDataSource,
                                 audio=MicrophoneSamplesDataSource, averager=TraceAverager)


def test_oracle_hackrf_branch_random_histories(ref_sources):
    from oracle import spectrum_oracle as so
    rng = np.random.default_rng(21)
    for trial in range(12):
        n = int(2 ** rng.integers(6, 13))
        fs = 20_000_000
        src = ref_sources.hackrf(sample_rate=fs, centre_freq=2_450_000_000)
        src.num_samples = n
        src.running = True
        src._allocate_fft_resources()
        br = so.HackrfBranchOracle(n, float(fs), precision="ref")
        x = so.unpack_iq_int8(so.synth_iq_int8(n * 40, n, seed=int(rng.integers(1, 1 << 30))))
        for k in range(40):
            ev = rng.random()
            if ev < 0.12:
                mode = [("off", 1), ("exp", int(rng.integers(1, 9))), ("lin", int(rng.integers(1, 12)))][int(rng.integers(0, 3))]
                src.set_averaging(*mode)
                br.averager.set_mode(*mode)
            elif ev < 0.18:
                src.reset_averaging()
                br.averager.reset()
            elif ev < 0.26:
                psd = bool(rng.integers(0, 2))
                src.set_psd_mode(psd)
                br.use_psd = psd
            elif ev < 0.34:
                a = float(rng.choice([1.0, 0.25, 0.05, 0.0]))
                src.set_dc_alpha(a)
                br.dc_alpha = a
            fr = x[k * n:(k + 1) * n]
            src._reservoir = np.array(fr, dtype=np.complex64, copy=True)
            want, _ = src.get_power_levels()
            got = br.power_levels(np.array(fr, copy=True))
            assert np.asarray(got).dtype == np.asarray(want).dtype, (trial, k)
            assert np.array_equal(got, want), (trial, k, n, br.averager.mode, br.use_psd, br.dc_alpha)
        assert complex(br.dc_estimate) == complex(src._dc_estimate)


def test_oracle_rtl_branch_random_histories(ref_sources):
    from oracle import spectrum_oracle as so

    class Sdr:
        def __init__(self, x, fs, fc):
            self.x, self.pos, self.fs, self.fc = x, 0, fs, fc

        def read_samples(self, n):
            out = self.x[self.pos: self.pos + n]
            self.pos += n
            return np.array(out, copy=True)

        def get_sample_rate(self): return self.fs
        def get_center_freq(self): return self.fc

    rng = np.random.default_rng(22)
    for trial in range(12):
        n = int(2 ** rng.integers(6, 13))
        fs, fc = 2_000_000.0, 100_300_000.0
        x = so.unpack_iq_int8(so.synth_iq_int8(n * 40, n, seed=int(rng.integers(1, 1 << 30)))).astype(np.complex128)
        src = ref_sources.rtl(sample_rate=int(fs), centre_freq=int(fc))
        src.set_fft_size(n)
        src.sdr = Sdr(x, fs, fc)
        src.running = True
        br = so.RtlBranchOracle(n, fs, "hanning", precision="ref")
        for k in range(40):
            ev = rng.random()
            if ev < 0.12:
                w = str(rng.choice(["hanning", "hamming", "rectangle"]))
                src.set_window_type(w)
                br.window = so.rtl_window(w, n)
            elif ev < 0.24:
                mode = [("off", 1), ("exp", int(rng.integers(1, 9))), ("lin", int(rng.integers(1, 12)))][int(rng.integers(0, 3))]
                src.set_averaging(*mode)
                br.averager.set_mode(*mode)
            elif ev < 0.32:
                psd = bool(rng.integers(0, 2))
                src.set_psd_mode(psd)
                br.use_psd = psd
            want, _ = src.get_power_levels()
            got = br.power_levels(x[k * n:(k + 1) * n])
            assert np.array_equal(got, want), (trial, k, n, br.averager.mode, br.use_psd)


def test_oracle_trace_averager_random_histories(ref_sources):
    from oracle import spectrum_oracle as so
    rng = np.random.default_rng(23)
    for trial in range(40):
        ref, ora = ref_sources.averager(), so.TraceAveragerOracle()
        shape = (64,)
        for step in range(60):
            ev = rng.random()
            if ev < 0.12:
                mode = [("off", 1), ("exp", int(rng.integers(1, 10))), ("lin", int(rng.integers(1, 12)))][int(rng.integers(0, 3))]
                ref.set_mode(*mode)
                ora.set_mode(*mode)
            elif ev < 0.18:
                ref.reset()
                ora.reset()
            elif ev < 0.24:
                shape = [(64,), (100,), (2, 32)][int(rng.integers(0, 3))]
            xin = (10.0 ** rng.uniform(-12, 3, size=shape)).astype(np.float32 if rng.integers(0, 2) else np.float64)
            want = np.array(ref.process(xin.copy()), copy=True)
            got = np.array(ora.process(xin.copy()), copy=True)
            assert got.dtype == want.dtype and np.array_equal(got, want), (trial, step)
            assert ora.is_active == ref.is_active


def test_processor_model_against_reference(ref_dp):
    """The float64 model the GPU suite checks DataProcessor against (tests/test_gpu_parity.py::ProcessorModel) is
    itself checked here against the reference's DataProcessor._process_sample_data, through the same seeded random
    GUI histories (holds switched, tare runs, offset and trace-length changes, NaN bins)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_gpu_parity as T
    sys.path.insert(0, REF)
    try:
        from core.tare_state import TareState
    finally:
        sys.path.remove(REF)
        for m in [k for k in sys.modules if k == "core" or k.startswith("core.")]:
            del sys.modules[m]
    for seed in range(40):
        rng = np.random.default_rng(8000 + seed)
        cal = {"v": 0.0}
        src = types.SimpleNamespace(trace=None, axis=None, last_data_time=0.0)
        src.get_power_levels = lambda s=src: (s.trace, s.axis)
        mw = types.SimpleNamespace(current_source=src, status_label=_Label(), tare_active=False, baseline_power_levels=None,
                                   live_power_levels=None, max_power_levels=None, min_power_levels=None,
                                   frequency_bins=None, min_hold_enabled=bool(rng.integers(0, 2)))
        mw.calibration_manager = types.SimpleNamespace(get_offset=lambda source_type: cal["v"])
        mw.source_manager = types.SimpleNamespace(last_source_type="hackrf_samples")
        dm = types.SimpleNamespace(tare_state=TareState(), max_peak_search_enabled=bool(rng.integers(0, 2)),
                                   duty_cycle_enabled=False, peak_list_enabled=False,
                                   _update_tare_button_label=lambda s: None)

        def _clear():
            mw.tare_active, mw.baseline_power_levels = False, None
            dm.tare_state = TareState()
        dm._clear_tare = _clear
        dp = ref_dp(mw, dm)
        model = T.ProcessorModel(alias_quirk=True)           # the reference's own behaviour incl. quirk ii
        for tick, (event, x) in enumerate(T.processor_history(rng)):
            if event == ("max",):
                dm.max_peak_search_enabled = not dm.max_peak_search_enabled
            elif event == ("min",):
                mw.min_hold_enabled = not mw.min_hold_enabled
            elif event == ("tare",):
                dm.tare_state = TareState(collecting=True)
                model.start_tare()
            elif event == ("clear",):
                dm._clear_tare()
                model.clear_tare()
            elif event is not None:
                cal["v"] = event[1]
            src.trace, src.axis = x.copy(), np.arange(len(x), dtype=np.float64)
            dp._process_sample_data()
            lv = model.frame(x, cal["v"], dm.max_peak_search_enabled, mw.min_hold_enabled)
            what = (seed, tick)
            assert mw.tare_active == model.active, what
            assert np.allclose(mw.live_power_levels, lv, rtol=0, atol=2e-4, equal_nan=True), what
            for got, want in ((mw.max_power_levels, model.max), (mw.min_power_levels, model.min)):
                assert (got is None) == (want is None), what
                if want is not None:
                    assert np.allclose(got, want, rtol=0, atol=2e-4, equal_nan=True), what


# ----------------------------------------------------------------------------------------------------------------
# the display accumulators of the reference (density histogram, waterfall ring) against oracle/analytics_oracle.py
# on random streams; PyQt6 / pyqtgraph are absent, so the reference classes are imported against stub modules made of
# plain do-nothing classes (as tests/golden/make_golden_displays.py does) and run their own numpy code
# ----------------------------------------------------------------------------------------------------------------
class _Stub:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return lambda *a, **k: _Stub()

    def __call__(self, *a, **k):
        return _Stub()


@pytest.fixture(scope="module")
def ref_displays():
    sys.dont_write_bytecode = True
    before = set(sys.modules)

    def module(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class Qt:
        class PenStyle:
            DashLine, SolidLine = 2, 1

        class AlignmentFlag:
            AlignCenter = 0

    qtcore = module("PyQt6.QtCore", Qt=Qt, QRectF=type("QRectF", (_Stub,), {}), QTimer=_Stub, pyqtSignal=_Stub)
    qtwidgets = module("PyQt6.QtWidgets", QWidget=type("QWidget", (_Stub,), {}),
                       QVBoxLayout=type("QVBoxLayout", (_Stub,), {}), QLabel=_Stub)
    qtgui = module("PyQt6.QtGui", QColor=_Stub, QFont=_Stub)
    module("PyQt6", QtCore=qtcore, QtWidgets=qtwidgets, QtGui=qtgui)
    module("pyqtgraph", AxisItem=type("AxisItem", (_Stub,), {}), PlotWidget=_Stub, ImageItem=_Stub, PlotCurveItem=_Stub,
           InfiniteLine=_Stub, GraphicsLayoutWidget=_Stub, ColorMap=_Stub, colormap=_Stub(), mkPen=_Stub(),
           mkBrush=_Stub(), TextItem=_Stub, ScatterPlotItem=_Stub)
    sys.path.insert(0, REF)
    try:
        from displays.density_display import DensityDisplay
        from displays.waterfall import Waterfall
    finally:
        sys.path.remove(REF)
        _forget_reference_modules(before, ("PyQt6", "pyqtgraph"))
    return types.SimpleNamespace(density=DensityDisplay, waterfall=Waterfall)


@pytest.mark.filterwarnings("ignore:invalid value encountered in cast")
def test_density_oracle_random_streams(ref_displays):
    from oracle import analytics_oracle as ao
    rng = np.random.default_rng(31)
    for trial in range(30):
        n = int(rng.choice([16, 100, 256, 1000]))
        mode = str(rng.choice(["medium", "fast", "off"]))
        d = ref_displays.density()
        d.set_decay(mode)
        ora = ao.DensityOracle(float(d._decay))
        fb = np.linspace(99e6, 101e6, n)
        for r in range(int(rng.integers(1, 80))):
            row = rng.normal(rng.uniform(-150, 50), float(rng.choice([0.3, 5.0, 40.0, 150.0])), n).astype(np.float32)
            for _ in range(int(rng.integers(0, 4))):
                row[rng.integers(0, n)] = rng.choice(np.array([np.nan, np.inf, -np.inf, -200.0, 100.0, -200.3, 99.99], dtype=np.float32))
            with np.errstate(invalid="ignore"):
                d._update_hist(row, fb)
            ora.update(row)
            assert np.array_equal(ora.hist, d._hist), (trial, r, mode)


def test_waterfall_oracle_random_streams(ref_displays):
    from oracle import analytics_oracle as ao
    rng = np.random.default_rng(32)
    for trial in range(30):
        w = ref_displays.waterfall()
        w.wf_time_span, w.seconds_per_row = float(rng.uniform(0.3, 3.0)), 0.1
        nb = int(rng.choice([16, 64, 300]))
        fb = np.linspace(2.40e9, 2.48e9, nb)
        pool = rng.normal(-90, 5, size=(10, nb)).astype(np.float32)
        ora = None
        for step in range(int(rng.integers(5, 80))):
            row = pool[int(rng.integers(0, len(pool)))] if rng.integers(0, 3) else pool[0]
            before = None if w._last_row is None else w._last_row.copy()
            w.update_widget_data(row, None, fb)
            if ora is None:
                ora = ao.WaterfallOracle(int(w.history_lines), nb, float(w.wf_min_db))
            new = ora.update(row)
            assert new == (before is None or not np.array_equal(before, row)), (trial, step)
            assert ora.ptr == w._ptr, (trial, step)
            assert np.array_equal(ora.view(), w._display_view()), (trial, step)


@pytest.fixture(scope="module")
def ref_analytics():
    sys.dont_write_bytecode = True
    before = set(sys.modules)
    mocked = [m for m in ("hackrf", "rtlsdr", "sounddevice") if m not in sys.modules]
    for m in mocked:
        sys.modules[m] = MagicMock()
    sys.path.insert(0, REF)
    try:
        from core.duty_cycle import DutyCycleAnalyser
        from core.marker_manager import MarkerManager
        from datasources.audio_samples import MicrophoneSamplesDataSource
    finally:
        sys.path.remove(REF)
        _forget_reference_modules(before, mocked)
    return types.SimpleNamespace(duty=DutyCycleAnalyser, markers=MarkerManager, audio=MicrophoneSamplesDataSource)


def test_band_power_oracle_random(ref_analytics):
    from oracle import analytics_oracle as ao
    rng = np.random.default_rng(41)
    for case in range(300):
        n = int(rng.integers(8, 5000))
        fb = np.linspace(99e6, 101e6, n)
        tr = rng.normal(-80, 10, n).astype(np.float32 if rng.integers(0, 2) else np.float64)
        if rng.integers(0, 4) == 0:
            tr[rng.integers(0, n)] = np.nan
        a, b = rng.uniform(98e6, 102e6, 2)
        want = ref_analytics.markers._band_power(types.SimpleNamespace(_data=lambda: (fb, tr)), a, b)
        got = ao.band_power_db(fb, tr, a, b)
        assert (want is None) == (got is None), case
        if want is not None:
            assert (np.isnan(want) and np.isnan(got)) or want == got, (case, want, got)


def test_marker_peak_search_oracle_random(ref_analytics):
    """oracle.marker_find_peaks / snap_to_peak_bin / snap_to_next_peak_bin against the imported MarkerManager (which
    runs the real scipy.signal.find_peaks) through random traces, thresholds, excursions and marker positions.
    Pairs of EQUAL maxima closer than 3 bins are broken up first: between those the reference follows numpy's
    unstable argsort (see oracle/analytics_oracle.py::_select_by_distance)."""
    from oracle import analytics_oracle as ao
    rng = np.random.default_rng(2718)
    moved = stayed = fallback = 0
    for case in range(120):
        n = int(rng.integers(3, 4000))
        k = np.arange(n)
        p = rng.exponential(1.0, n) * 10.0 ** rng.uniform(-12, -6)
        for _ in range(int(rng.integers(0, 6))):
            p += 10.0 ** rng.uniform(-9, -2) * np.sinc((k - rng.integers(0, n)) / rng.uniform(0.6, 4.0)) ** 2
        tr = 10 * np.log10(p + 1e-15)
        mode = int(rng.integers(0, 4))
        if mode == 1:
            tr = np.round(tr * 2) / 2
        elif mode == 2:
            tr = np.minimum(tr, np.percentile(tr, 95))
        elif mode == 3:
            tr = -100 + 10.0 * k / n                                       # nothing qualifies
        tr = tr.astype(np.float32)
        for _ in range(1000):                                               # no equal maxima closer than 3 bins
            lm = ao._local_maxima(tr.astype(np.float64))
            hit = [b for a, b in zip(lm, lm[1:]) if b - a < 3 and tr[a] == tr[b]]
            if not hit:
                break
            tr[hit] += np.float32(0.5)
        fb = np.linspace(88e6, 108e6, n)
        mw = types.SimpleNamespace(frequency_bins=fb, live_power_levels=tr)
        thr, exc = -200.0, 6.0
        if rng.integers(0, 3):
            thr, exc = float(rng.choice([-200.0, -90.0, -70.0])), float(rng.choice([0.5, 3.0, 6.0, 10.0]))
            mw.peak_threshold, mw.peak_excursion = thr, exc
        mm = ref_analytics.markers(mw)
        mm._sync_display = lambda name: None
        mm._refresh_status = lambda: None
        mm.active_marker = "F1"
        f1 = mm.markers["F1"]
        f1.enabled, f1.position = True, float(fb[0])
        mm.snap_to_peak()
        assert f1.position == fb[ao.snap_to_peak_bin(tr, thr, exc, 3)], case
        fallback += len(ao.marker_find_peaks(tr, thr, exc, 3)[0]) == 0
        # the marker sits anywhere on the axis (also between bins and outside it)
        f1.position = float(rng.uniform(fb[0] - 1e5, fb[-1] + 1e5))
        for _ in range(5):
            before = f1.position
            cur = int(np.searchsorted(fb, before))
            mm.snap_to_next_peak()
            nxt = ao.snap_to_next_peak_bin(tr, cur, thr, exc, 3)
            assert f1.position == (before if nxt < 0 else fb[nxt]), case
            moved += nxt >= 0
            stayed += nxt < 0
    assert moved and stayed and fallback


def test_duty_cycle_oracle_random(ref_analytics):
    from oracle import analytics_oracle as ao
    rng = np.random.default_rng(42)
    for trial in range(20):
        ref, ora = ref_analytics.duty(), ao.DutyCycleOracle()
        for i in range(int(rng.integers(5, 300))):
            fr = rng.normal(-70, 6, 256).astype(np.float32) + (0.0 if rng.integers(0, 3) else 40.0)
            thr = float(rng.choice([-60.0, -45.0, -30.0]))
            ref.update_from_power(fr, threshold_dbm=thr)
            ora.update_from_power(fr, threshold_dbm=thr)
            assert ora.duty_pct == ref.duty_pct, (trial, i)
            for name in ("on_power_dbm", "off_power_dbm"):
                a, b = getattr(ref, name), getattr(ora, name)
                assert (a is None) == (b is None) and (a is None or a == b), (trial, i, name)


def test_duty_cycle_analyser_of_the_product_random(ref_analytics):
    """The product's DutyCycleAnalyser (host class; its device feed is checked in the GPU suite) against the
    reference's through random histories of all three entry points: raw complex / real samples, dB spectra with
    and without a threshold, resets - every attribute and the readout string identical."""
    from topdogspectrumanalyser_amd.core.duty_cycle import DutyCycleAnalyser
    rng = np.random.default_rng(420)
    for trial in range(10):
        ref, got = ref_analytics.duty(), DutyCycleAnalyser()
        assert ref.get_readout() == got.get_readout() == ""
        for i in range(int(rng.integers(50, 400))):
            k = int(rng.integers(0, 10))
            thr = float(rng.uniform(-80, -20))
            if k < 3:
                x = ((rng.normal(size=64) + 1j * rng.normal(size=64)) * 10 ** rng.uniform(-6, 0)).astype(np.complex64)
                ref.update(x, thr), got.update(x, thr)
            elif k < 5:
                x = (rng.normal(size=(32, 2)) * 10 ** rng.uniform(-6, 0)).astype(np.float32)
                ref.update(x, thr), got.update(x, thr)
            elif k < 9:
                d = rng.uniform(-120, 0, size=128)
                t = None if rng.random() < 0.5 else thr
                ref.update_from_power(d, t), got.update_from_power(d, t)
            elif rng.random() < 0.3:
                ref.reset(), got.reset()
            else:
                ref.update(None, thr), got.update(None, thr)
                ref.update_from_power(np.empty(0)), got.update_from_power(np.empty(0))
            for name in ("duty_pct", "on_power_dbm", "off_power_dbm", "threshold_dbm"):
                assert getattr(ref, name) == getattr(got, name), (trial, i, name)
            assert ref.get_readout() == got.get_readout(), (trial, i)
            assert list(ref._envelope) == list(got._envelope)


def test_audio_oracle_random(ref_analytics):
    from oracle import spectrum_oracle as so

    class Stream:
        def __init__(self, data):
            self.d, self.pos = data, 0

        def read(self, n):
            out = self.d[self.pos: self.pos + n]
            self.pos += n
            return np.array(out, copy=True), False

    rng = np.random.default_rng(43)
    for trial in range(16):
        n = int(2 ** rng.integers(6, 13))
        fs = 44100
        nf = 12
        data = (0.3 * rng.standard_normal((n * nf, 2))).astype(np.float32)
        data[:, 0] += (0.5 * np.sin(2 * np.pi * 997.0 * np.arange(n * nf) / fs)).astype(np.float32)
        chan, psd = str(rng.choice(["mono", "left", "right"])), bool(rng.integers(0, 2))
        src = ref_analytics.audio(sample_rate=fs, centre_freq=0)
        src.set_fft_size(n)
        src.set_channel_mode(chan)
        src.set_psd_mode(psd)
        src.stream = Stream(data)
        src.running = True
        src._audio_block = n
        win = np.array(src.window, copy=True)
        for k in range(nf):
            want, _ = src.get_power_levels()
            blk = data[k * n:(k + 1) * n]
            sig = {"mono": (blk[:, 0] + blk[:, 1]) * 0.5, "left": blk[:, 0], "right": blk[:, 1]}[chan]
            got = so.audio_db(so.audio_compute_power(np.array(sig, copy=True), win, n, fs, psd, precision="ref"), psd)
            assert np.array_equal(got, want), (trial, k, chan, psd)


# ----------------------------------------------------------------------------------------------------------------
# the product's source classes against the reference's, on everything that needs no samples: constructor state,
# property round trips, setter validation (exception types), what a source that is not running answers
# ----------------------------------------------------------------------------------------------------------------
def _outcome(fn):
    try:
        return ("ok", fn())
    except Exception as exc:                                        # compare the exception TYPE, as callers do
        return ("raised", type(exc).__name__)


def _same(a, b):
    if isinstance(a, tuple) and isinstance(b, tuple) and len(a) == len(b):
        return all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        return isinstance(a, np.ndarray) and isinstance(b, np.ndarray) and a.shape == b.shape and \
            np.array_equal(a, b, equal_nan=True)
    return a == b


def test_sources_idle_behaviour_matches_reference(ref_sources, ref_analytics):
    import topdogspectrumanalyser_amd as pkg
    pairs = [("hackrf", ref_sources.hackrf, pkg.HackrfSamplesDataSource, dict(sample_rate=20_000_000, centre_freq=2_450_000_000)),
             ("rtl", ref_sources.rtl, pkg.RtlSamplesDataSource, dict(sample_rate=2_048_000, centre_freq=100_000_000)),
             ("audio", ref_analytics.audio, pkg.MicrophoneSamplesDataSource, dict(sample_rate=44100, centre_freq=0))]
    for name, RefCls, OurCls, kw in pairs:
        ref, our = RefCls(**kw), OurCls(**kw)
        probes = [
            ("sample_rate", lambda s: s.sample_rate), ("centre_freq", lambda s: s.centre_freq),
            ("sample_count", lambda s: s.sample_count), ("last_data_time", lambda s: s.last_data_time),
            ("idle frame", lambda s: s.get_power_levels()),
            ("raw samples", lambda s: s.get_raw_samples()),
            ("read_samples_only", lambda s: s.read_samples_only()),
            ("averaging is_active", lambda s: (s.set_averaging("exp", 4), s._averager.is_active)[1]),
            ("averaging n<1", lambda s: (s.set_averaging("lin", 0), s._averager.is_active)[1]),
            ("reset_averaging", lambda s: s.reset_averaging()),
            ("psd mode", lambda s: s.set_psd_mode(True)),
            ("idle frame after psd", lambda s: s.get_power_levels()),
            ("sample_count = 2048", lambda s: (setattr(s, "sample_count", 2048), s.sample_count)[1]),
            ("idle frame at 2048", lambda s: s.get_power_levels()),
            ("stop when idle", lambda s: s.stop()),
        ]
        if name == "hackrf":
            probes += [
                ("set_num_samples(0)", lambda s: s.set_num_samples(0)),
                ("set_num_samples(-5)", lambda s: s.set_num_samples(-5)),
                ("set_gains ok", lambda s: (s.set_gains(lna_gain=16, vga_gain=20), s.lna_gain, s.vga_gain)[1:]),
                ("set_gains lna too high", lambda s: s.set_gains(lna_gain=100)),
                ("set_gains vga negative", lambda s: s.set_gains(vga_gain=-2)),
                ("set_dc_alpha clamps", lambda s: (s.set_dc_alpha(7.0), s._DC_ALPHA, s.set_dc_alpha(-1.0), s._DC_ALPHA)[1::2]),
                ("stats keys", lambda s: sorted(s.get_stats().keys())),
                ("update_centre_frequency idle", lambda s: (s.update_centre_frequency(2_400_000_000), s.centre_freq)[1]),
            ]
        if name == "rtl":
            probes += [
                ("window type", lambda s: (s.set_window_type("hamming"), np.array(s.window))[1]),
                ("unknown window type", lambda s: (s.set_window_type("kaiser"), np.array(s.window))[1]),
                ("fft size resets window", lambda s: (s.set_fft_size(512), np.array(s.window))[1]),
                ("idle frame at 512", lambda s: s.get_power_levels()),
            ]
        if name == "audio":
            probes += [
                ("channel mode", lambda s: (s.set_channel_mode("left"), s.channel_mode)[1]),
                ("bad channel mode ignored", lambda s: (s.set_channel_mode("quad"), s.channel_mode)[1]),
                ("fft size", lambda s: (s.set_fft_size(4096), s.sample_count, np.array(s.window))[1:]),
                ("idle frame at 4096", lambda s: s.get_power_levels()),
            ]
        for what, probe in probes:
            want, got = _outcome(lambda: probe(ref)), _outcome(lambda: probe(our))
            assert want[0] == got[0], (name, what, want, got)
            assert _same(want[1], got[1]), (name, what, want[1], got[1])

